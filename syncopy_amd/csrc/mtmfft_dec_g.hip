// mtmfft_dec_kernel instances for 3 x a scheduled length: N = 6000, 7500 (see mtmfft_dec_launch.h)
#include "mtmfft_dec_launch.h"

namespace spyfft {
int dec_launch_g(hipStream_t stream, const MtmArgs& a, int nfft, int nquads, int outk, bool mean) {
    switch (nfft) {
        case 6000: return dec_launch_mode<CfgD<10, 10, 10, 2, 1, 3>>(stream, a, nquads, outk, mean);
        case 7500: return dec_launch_mode<CfgD<10, 10, 5, 5, 1, 3>>(stream, a, nquads, outk, mean);
        default: return -100;
    }
}
}  // namespace spyfft
