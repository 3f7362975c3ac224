// mtmfft_dec64_kernel instances for N = 2048 4096 (see mtmfft_dec64_launch.h)
#include "mtmfft_dec64_launch.h"

namespace spyfft {
int dec64_launch_b(hipStream_t stream, const F64Args& a, int nfft, int npairs, int outk, bool mean) {
    switch (nfft) {
        case 2048: return dec64_launch_mode<D64_2048>(stream, a, npairs, outk, mean);
        case 4096: return dec64_launch_mode<D64_4096>(stream, a, npairs, outk, mean);
        default: return -100;
    }
}
}  // namespace spyfft
