// mtmfft_dec64_kernel instances for N = 600, 1500, 7500: 3 x a scheduled length (see mtmfft_dec64_launch.h)
#include "mtmfft_dec64_launch.h"

namespace spyfft {
int dec64_launch_i(hipStream_t stream, const F64Args& a, int nfft, int npairs, int outk, bool mean) {
    switch (nfft) {
        case 600: return dec64_launch_mode<D64_600>(stream, a, npairs, outk, mean);
        case 1500: return dec64_launch_mode<D64_1500>(stream, a, npairs, outk, mean);
        case 7500: return dec64_launch_mode<D64_7500>(stream, a, npairs, outk, mean);
        default: return -100;
    }
}
}  // namespace spyfft
