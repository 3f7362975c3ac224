// mtmfft_dec_kernel instances in HALF form (real transforms of 2 N samples through the length-N schedule, channel pairs):
// nfft = 12000, 12288, 15000 (see mtmfft_dec_launch.h)
#include "mtmfft_dec_launch.h"

namespace spyfft {
int dec_launch_half_a(hipStream_t stream, const MtmArgs& a, int nfft, int npairs, int outk, bool mean) {
    switch (nfft) {
        case 12000: return dec_launch_mode<CfgD<10, 10, 10, 2, 1, 3, false, true>>(stream, a, npairs, outk, mean);
        case 12288: return dec_launch_mode<CfgD<16, 16, 8, 1, 1, 3, false, true>>(stream, a, npairs, outk, mean);
        case 15000: return dec_launch_mode<CfgD<10, 10, 5, 5, 1, 3, false, true>>(stream, a, npairs, outk, mean);
        default: return -100;
    }
}
}  // namespace spyfft
