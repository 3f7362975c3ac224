// mtmfft_dec64_kernel instances for N = 8192 16384 (see mtmfft_dec64_launch.h)
#include "mtmfft_dec64_launch.h"

namespace spyfft {
int dec64_launch_c(hipStream_t stream, const F64Args& a, int nfft, int npairs, int outk, bool mean) {
    switch (nfft) {
        case 8192: return dec64_launch_mode<D64_8192>(stream, a, npairs, outk, mean);
        case 16384: return dec64_launch_mode<D64_16384>(stream, a, npairs, outk, mean);
        default: return -100;
    }
}
}  // namespace spyfft
