// mtmfft_dec64_kernel instances for N = 1600, 3200, 8000 (see mtmfft_dec64_launch.h)
#include "mtmfft_dec64_launch.h"

namespace spyfft {
int dec64_launch_l(hipStream_t stream, const F64Args& a, int nfft, int npairs, int outk, bool mean) {
    switch (nfft) {
        case 1600: return dec64_launch_mode<D64_1600>(stream, a, npairs, outk, mean);
        case 3200: return dec64_launch_mode<D64_3200>(stream, a, npairs, outk, mean);
        case 8000: return dec64_launch_mode<D64_8000>(stream, a, npairs, outk, mean);
        default: return -100;
    }
}
}  // namespace spyfft
