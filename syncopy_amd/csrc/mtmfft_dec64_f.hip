// mtmfft_dec64_kernel instances for N = 5000 10000 (see mtmfft_dec64_launch.h)
#include "mtmfft_dec64_launch.h"

namespace spyfft {
int dec64_launch_f(hipStream_t stream, const F64Args& a, int nfft, int npairs, int outk, bool mean) {
    switch (nfft) {
        case 5000: return dec64_launch_mode<D64_5000>(stream, a, npairs, outk, mean);
        case 10000: return dec64_launch_mode<D64_10000>(stream, a, npairs, outk, mean);
        default: return -100;
    }
}
}  // namespace spyfft
