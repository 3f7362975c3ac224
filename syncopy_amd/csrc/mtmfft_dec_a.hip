// mtmfft_dec_kernel instances for N = 100, 200, 500 (see mtmfft_dec_launch.h)
#include "mtmfft_dec_launch.h"

namespace spyfft {
int dec_launch_a(hipStream_t stream, const MtmArgs& a, int nfft, int nquads, int outk, bool mean) {
    switch (nfft) {
        case 100: return dec_launch_mode<CfgD<10, 10, 1, 1, 16>>(stream, a, nquads, outk, mean);
        case 200: return dec_launch_mode<CfgD<10, 10, 2, 1, 8>>(stream, a, nquads, outk, mean);
        case 500: return dec_launch_mode<CfgD<10, 10, 5, 1, 4>>(stream, a, nquads, outk, mean);
        default: return -100;
    }
}
}  // namespace spyfft
