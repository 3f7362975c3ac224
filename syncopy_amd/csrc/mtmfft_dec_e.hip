// mtmfft_dec_kernel instances for N = 4000 (see mtmfft_dec_launch.h)
#include "mtmfft_dec_launch.h"

namespace spyfft {
int dec_launch_e(hipStream_t stream, const MtmArgs& a, int nfft, int nquads, int outk, bool mean) {
    switch (nfft) {
        // 1 kHz x 4 s: 20 values per thread, 20 x 20 x 10 (10 values per thread cannot end on a radix that divides 4000 / 1000)
        case 4000: return dec_launch_mode<CfgD<20, 20, 10, 1, 1>>(stream, a, nquads, outk, mean);
        default: return -100;
    }
}
}  // namespace spyfft
