// mtmfft_dec64_kernel instances for N = 2000 2500 (see mtmfft_dec64_launch.h)
#include "mtmfft_dec64_launch.h"

namespace spyfft {
int dec64_launch_e(hipStream_t stream, const F64Args& a, int nfft, int npairs, int outk, bool mean) {
    switch (nfft) {
        case 2000: return dec64_launch_mode<D64_2000>(stream, a, npairs, outk, mean);
        case 2500: return dec64_launch_mode<D64_2500>(stream, a, npairs, outk, mean);
        default: return -100;
    }
}
}  // namespace spyfft
