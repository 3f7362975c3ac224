// mtmfft_dec_kernel instances with 20 values per thread: N = 400, 800, 1600 (see mtmfft_dec_launch.h)
#include "mtmfft_dec_launch.h"

namespace spyfft {
int dec_launch_j(hipStream_t stream, const MtmArgs& a, int nfft, int nquads, int outk, bool mean) {
    switch (nfft) {
        case 400: return dec_launch_mode<CfgD<20, 20, 1, 1, 8>>(stream, a, nquads, outk, mean);
        case 800: return dec_launch_mode<CfgD<20, 20, 2, 1, 4>>(stream, a, nquads, outk, mean);
        case 1600: return dec_launch_mode<CfgD<20, 20, 4, 1, 2>>(stream, a, nquads, outk, mean);
        default: return -100;
    }
}
}  // namespace spyfft
