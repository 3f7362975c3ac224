// mtmfft_dec_kernel instances for 3 x a power of two: N = 768, 1536, 3072, 6144 (16 values per thread, radix-3 decimation
// in front; see mtmfft_dec_launch.h)
#include "mtmfft_dec_launch.h"

namespace spyfft {
int dec_launch_i(hipStream_t stream, const MtmArgs& a, int nfft, int nquads, int outk, bool mean) {
    switch (nfft) {
        case 768: return dec_launch_mode<CfgD<16, 16, 1, 1, 4, 3>>(stream, a, nquads, outk, mean);
        case 1536: return dec_launch_mode<CfgD<16, 16, 2, 1, 2, 3>>(stream, a, nquads, outk, mean);
        case 3072: return dec_launch_mode<CfgD<16, 16, 4, 1, 1, 3>>(stream, a, nquads, outk, mean);
        case 6144: return dec_launch_mode<CfgD<16, 16, 8, 1, 1, 3>>(stream, a, nquads, outk, mean);
        default: return -100;
    }
}
}  // namespace spyfft
