// mtmfft_dec_kernel instances with 20 values per thread: N = 3200, 8000 (see mtmfft_dec_launch.h)
#include "mtmfft_dec_launch.h"

namespace spyfft {
int dec_launch_k(hipStream_t stream, const MtmArgs& a, int nfft, int nquads, int outk, bool mean) {
    switch (nfft) {
        case 3200: return dec_launch_mode<CfgD<20, 20, 4, 2, 1>>(stream, a, nquads, outk, mean);
        case 8000: return dec_launch_mode<CfgD<20, 20, 20, 1, 1>>(stream, a, nquads, outk, mean);
        default: return -100;
    }
}
}  // namespace spyfft
