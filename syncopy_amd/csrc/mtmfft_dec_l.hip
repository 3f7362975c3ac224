// mtmfft_dec_kernel instances with 20 values per thread: N = 1200, 2400, 4800 (see mtmfft_dec_launch.h)
#include "mtmfft_dec_launch.h"

namespace spyfft {
int dec_launch_l(hipStream_t stream, const MtmArgs& a, int nfft, int nquads, int outk, bool mean) {
    switch (nfft) {
        case 1200: return dec_launch_mode<CfgD<20, 20, 1, 1, 2, 3>>(stream, a, nquads, outk, mean);
        case 2400: return dec_launch_mode<CfgD<20, 20, 2, 1, 1, 3>>(stream, a, nquads, outk, mean);
        case 4800: return dec_launch_mode<CfgD<20, 20, 4, 1, 1, 3>>(stream, a, nquads, outk, mean);
        default: return -100;
    }
}
}  // namespace spyfft
