// mtmfft_dec_kernel instances in HALF form for lengths that also fit in quad form and measured faster on channel pairs:
// nfft = 5000, 10000 (see mtmfft_dec_launch.h, tools/half_probe.py)
#include "mtmfft_dec_launch.h"

namespace spyfft {
int dec_launch_half_c(hipStream_t stream, const MtmArgs& a, int nfft, int npairs, int outk, bool mean) {
    switch (nfft) {
        case 10000: return dec_launch_mode<CfgD<10, 10, 10, 5, 1, 1, false, true>>(stream, a, npairs, outk, mean);
        case 5000: return dec_launch_mode<CfgD<10, 10, 5, 5, 1, 1, false, true>>(stream, a, npairs, outk, mean);
        default: return -100;
    }
}
}  // namespace spyfft
