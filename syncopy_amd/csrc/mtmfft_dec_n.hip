// mtmfft_dec_kernel instances in HALF form (real transforms of 2 N samples through the length-N schedule, channel pairs):
// nfft = 16000, 20000 (see mtmfft_dec_launch.h; 16384 takes the power-of-two engine in the same form, mtmfft_quad_half.hip)
#include "mtmfft_dec_launch.h"

namespace spyfft {
int dec_launch_half_b(hipStream_t stream, const MtmArgs& a, int nfft, int npairs, int outk, bool mean) {
    switch (nfft) {
        case 16000: return dec_launch_mode<CfgD<20, 20, 20, 1, 1, 1, false, true>>(stream, a, npairs, outk, mean);
        case 20000: return dec_launch_mode<CfgD<20, 20, 5, 5, 1, 1, true, true>>(stream, a, npairs, outk, mean);
        default: return -100;
    }
}
}  // namespace spyfft
