// mtmfft_dec64_kernel instances for N = 768, 1536, 3072, 6144: 3 x a scheduled length (see mtmfft_dec64_launch.h)
#include "mtmfft_dec64_launch.h"

namespace spyfft {
int dec64_launch_j(hipStream_t stream, const F64Args& a, int nfft, int npairs, int outk, bool mean) {
    switch (nfft) {
        case 768: return dec64_launch_mode<D64_768>(stream, a, npairs, outk, mean);
        case 1536: return dec64_launch_mode<D64_1536>(stream, a, npairs, outk, mean);
        case 3072: return dec64_launch_mode<D64_3072>(stream, a, npairs, outk, mean);
        case 6144: return dec64_launch_mode<D64_6144>(stream, a, npairs, outk, mean);
        default: return -100;
    }
}
}  // namespace spyfft
