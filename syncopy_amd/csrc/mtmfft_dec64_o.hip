// mtmfft_dec64_kernel instances in HALF form: nfft = 12000, 12288, 15000 (mtmfft_dec64_cfg.h)
#include "mtmfft_dec64_launch.h"

namespace spyfft {
int dec64_launch_half_a(hipStream_t stream, const F64Args& a, int nfft, int nchan, int outk, bool mean) {
    switch (nfft) {
        case 12000: return dec64_launch_mode<D64H_12000>(stream, a, nchan, outk, mean);
        case 12288: return dec64_launch_mode<D64H_12288>(stream, a, nchan, outk, mean);
        case 15000: return dec64_launch_mode<D64H_15000>(stream, a, nchan, outk, mean);
        default: return -100;
    }
}
}  // namespace spyfft
