// mtmfft_dec_kernel instances for 3 x a scheduled length: N = 300, 600, 1500, 3000 (see mtmfft_dec_launch.h)
#include "mtmfft_dec_launch.h"

namespace spyfft {
int dec_launch_f(hipStream_t stream, const MtmArgs& a, int nfft, int nquads, int outk, bool mean) {
    switch (nfft) {
        case 300: return dec_launch_mode<CfgD<10, 10, 1, 1, 8, 3>>(stream, a, nquads, outk, mean);
        case 600: return dec_launch_mode<CfgD<10, 10, 2, 1, 4, 3>>(stream, a, nquads, outk, mean);
        case 1500: return dec_launch_mode<CfgD<10, 10, 5, 1, 2, 3>>(stream, a, nquads, outk, mean);
        case 3000: return dec_launch_mode<CfgD<10, 10, 10, 1, 1, 3>>(stream, a, nquads, outk, mean);
        default: return -100;
    }
}
}  // namespace spyfft
