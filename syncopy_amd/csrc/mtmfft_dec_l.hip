// mtmfft_dec_kernel instances for 3 x (400, 800, 1600): N = 1200, 2400, 4800 (see mtmfft_dec_launch.h)
#include "mtmfft_dec_launch.h"

namespace spyfft {
int dec_launch_l(hipStream_t stream, const MtmArgs& a, int nfft, int nquads, int outk, bool mean) {
    switch (nfft) {
        // (1200 = 3 x (10 x 10 x 2 x 2): 39 vs 55 ms for 102400 Hann windows x 128 channels against 3 x (20 x 20))
        case 1200: return dec_launch_mode<CfgD<10, 10, 2, 2, 2, 3>>(stream, a, nquads, outk, mean);
        case 2400: return dec_launch_mode<CfgD<20, 20, 2, 1, 1, 3>>(stream, a, nquads, outk, mean);
        case 4800: return dec_launch_mode<CfgD<20, 20, 4, 1, 1, 3>>(stream, a, nquads, outk, mean);
        default: return -100;
    }
}
}  // namespace spyfft
