// mtmfft_dec64_kernel instances for N = 2400, 4800 (see mtmfft_dec64_launch.h)
#include "mtmfft_dec64_launch.h"

namespace spyfft {
int dec64_launch_n(hipStream_t stream, const F64Args& a, int nfft, int npairs, int outk, bool mean) {
    switch (nfft) {
        case 2400: return dec64_launch_mode<D64_2400>(stream, a, npairs, outk, mean);
        case 4800: return dec64_launch_mode<D64_4800>(stream, a, npairs, outk, mean);
        default: return -100;
    }
}
}  // namespace spyfft
