// mtmfft_dec_kernel instances for N = 2500, 5000 (see mtmfft_dec_launch.h)
#include "mtmfft_dec_launch.h"

namespace spyfft {
int dec_launch_d(hipStream_t stream, const MtmArgs& a, int nfft, int nquads, int outk, bool mean) {
    switch (nfft) {
        case 2500: return dec_launch_mode<CfgD<10, 10, 5, 5, 1>>(stream, a, nquads, outk, mean);
        case 5000: return dec_launch_mode<CfgD<10, 10, 10, 5, 1>>(stream, a, nquads, outk, mean);
        default: return -100;
    }
}
}  // namespace spyfft
