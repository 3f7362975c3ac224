// mtmfft_dec_kernel instances for N = 1000 (see mtmfft_dec_launch.h)
#include "mtmfft_dec_launch.h"

namespace spyfft {
int dec_launch_b(hipStream_t stream, const MtmArgs& a, int nfft, int nquads, int outk, bool mean) {
    switch (nfft) {
        case 1000: return dec_launch_mode<CfgD<10, 10, 10, 1, 2>>(stream, a, nquads, outk, mean);
        default: return -100;
    }
}
}  // namespace spyfft
