// mtmfft_dec_kernel instances for N = 2000 (see mtmfft_dec_launch.h)
#include "mtmfft_dec_launch.h"

namespace spyfft {
int dec_launch_c(hipStream_t stream, const MtmArgs& a, int nfft, int nquads, int outk, bool mean) {
    switch (nfft) {
        case 2000: return dec_launch_mode<CfgD<10, 10, 10, 2, 1>>(stream, a, nquads, outk, mean);
        default: return -100;
    }
}
// complex spectra of every taper at N = 2000 are store-bound: 20 x 10 x 10 with two quads per workgroup writes 64
// contiguous bytes per bin row (7.3 vs 8.5 us/trial at 256 channels)
int dec_launch_c2(hipStream_t stream, const MtmArgs& a, int nquads) {
    return dec_launch_one<CfgD<20, 10, 10, 1, 2>, 2, false>(stream, a, nquads);
}
}  // namespace spyfft
