// mtmfft_dec64_kernel instances in HALF form: nfft = 16000, 16384, 20000 (mtmfft_dec64_cfg.h)
#include "mtmfft_dec64_launch.h"

namespace spyfft {
int dec64_launch_half_b(hipStream_t stream, const F64Args& a, int nfft, int nchan, int outk, bool mean) {
    switch (nfft) {
        case 16000: return dec64_launch_mode<D64H_16000>(stream, a, nchan, outk, mean);
        case 16384: return dec64_launch_mode<D64H_16384>(stream, a, nchan, outk, mean);
        case 20000: return dec64_launch_mode<D64H_20000>(stream, a, nchan, outk, mean);
        default: return -100;
    }
}
}  // namespace spyfft
