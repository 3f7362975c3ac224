// mtmfft_dec64_kernel instances for N = 256 512 1024 (see mtmfft_dec64_launch.h)
#include "mtmfft_dec64_launch.h"

namespace spyfft {
int dec64_launch_a(hipStream_t stream, const F64Args& a, int nfft, int npairs, int outk, bool mean) {
    switch (nfft) {
        case 256: return dec64_launch_mode<D64_256>(stream, a, npairs, outk, mean);
        case 512: return dec64_launch_mode<D64_512>(stream, a, npairs, outk, mean);
        case 1024: return dec64_launch_mode<D64_1024>(stream, a, npairs, outk, mean);
        default: return -100;
    }
}
}  // namespace spyfft
