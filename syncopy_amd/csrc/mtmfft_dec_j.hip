// mtmfft_dec_kernel instances with 20 values per thread: N = 400, 800, 1600 (see mtmfft_dec_launch.h)
#include "mtmfft_dec_launch.h"

namespace spyfft {
int dec_launch_j(hipStream_t stream, const MtmArgs& a, int nfft, int nquads, int outk, bool mean) {
    switch (nfft) {
        // (400 = 10 x 10 x 2 x 2: four passes at four waves per SIMD beat 20 x 20 at one - 16.4 vs 22.0 ms for 102400 Hann
        // windows x 128 channels, 53 vs 63 with seven tapers)
        case 400: return dec_launch_mode<CfgD<10, 10, 2, 2, 8>>(stream, a, nquads, outk, mean);
        case 800: return dec_launch_mode<CfgD<20, 20, 2, 1, 4>>(stream, a, nquads, outk, mean);
        case 1600: return dec_launch_mode<CfgD<20, 20, 4, 1, 2>>(stream, a, nquads, outk, mean);
        default: return -100;
    }
}
}  // namespace spyfft
