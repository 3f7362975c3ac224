// mtmfft_dec64_kernel instances for N = 300, 1200 (see mtmfft_dec64_launch.h)
#include "mtmfft_dec64_launch.h"

namespace spyfft {
int dec64_launch_m(hipStream_t stream, const F64Args& a, int nfft, int npairs, int outk, bool mean) {
    switch (nfft) {
        case 300: return dec64_launch_mode<D64_300>(stream, a, npairs, outk, mean);
        case 1200: return dec64_launch_mode<D64_1200>(stream, a, npairs, outk, mean);
        default: return -100;
    }
}
}  // namespace spyfft
