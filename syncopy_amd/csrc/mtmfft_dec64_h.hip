// mtmfft_dec64_kernel instances for N = 3000, 6000: 3 x a scheduled length (see mtmfft_dec64_launch.h)
#include "mtmfft_dec64_launch.h"

namespace spyfft {
int dec64_launch_h(hipStream_t stream, const F64Args& a, int nfft, int npairs, int outk, bool mean) {
    switch (nfft) {
        case 3000: return dec64_launch_mode<D64_3000>(stream, a, npairs, outk, mean);
        case 6000: return dec64_launch_mode<D64_6000>(stream, a, npairs, outk, mean);
        default: return -100;
    }
}
}  // namespace spyfft
