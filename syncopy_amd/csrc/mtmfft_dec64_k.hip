// mtmfft_dec64_kernel instances for N = 100, 400, 800 (see mtmfft_dec64_launch.h)
#include "mtmfft_dec64_launch.h"

namespace spyfft {
int dec64_launch_k(hipStream_t stream, const F64Args& a, int nfft, int npairs, int outk, bool mean) {
    switch (nfft) {
        case 100: return dec64_launch_mode<D64_100>(stream, a, npairs, outk, mean);
        case 400: return dec64_launch_mode<D64_400>(stream, a, npairs, outk, mean);
        case 800: return dec64_launch_mode<D64_800>(stream, a, npairs, outk, mean);
        default: return -100;
    }
}
}  // namespace spyfft
