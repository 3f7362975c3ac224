// mtmfft_dec64_kernel instances for N = 4000 (see mtmfft_dec64_launch.h)
#include "mtmfft_dec64_launch.h"

namespace spyfft {
int dec64_launch_g(hipStream_t stream, const F64Args& a, int nfft, int npairs, int outk, bool mean) {
    switch (nfft) {
        case 4000: return dec64_launch_mode<D64_4000>(stream, a, npairs, outk, mean);
        default: return -100;
    }
}
}  // namespace spyfft
