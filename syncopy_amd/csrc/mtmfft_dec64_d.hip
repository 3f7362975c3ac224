// mtmfft_dec64_kernel instances for N = 200 500 1000 (see mtmfft_dec64_launch.h)
#include "mtmfft_dec64_launch.h"

namespace spyfft {
int dec64_launch_d(hipStream_t stream, const F64Args& a, int nfft, int npairs, int outk, bool mean) {
    switch (nfft) {
        case 200: return dec64_launch_mode<D64_200>(stream, a, npairs, outk, mean);
        case 500: return dec64_launch_mode<D64_500>(stream, a, npairs, outk, mean);
        case 1000: return dec64_launch_mode<D64_1000>(stream, a, npairs, outk, mean);
        default: return -100;
    }
}
}  // namespace spyfft
