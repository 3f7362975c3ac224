// mtmfft_dec_kernel instance with split exchanges (real parts, then imaginary parts through one 8-byte plane): N = 10000
// = 20 x 20 x 5 x 5, 500 threads x 20 values (176 KB of float4 would not fit; 1000 threads x 10 values are held to 128
// registers and spill ~460 bytes per lane: 65.4 vs 52.4 us/trial; the mixed-radix engine: 69.8).  The same split on
// N = 5000 (two workgroups per CU instead of one) measured 19.6 vs 17.9 us/trial: not used.
#include "mtmfft_dec_launch.h"

namespace spyfft {
int dec_launch_h(hipStream_t stream, const MtmArgs& a, int nfft, int nquads, int outk, bool mean) {
    switch (nfft) {
        case 10000: return dec_launch_mode<CfgD<20, 20, 5, 5, 1, 1, true>>(stream, a, nquads, outk, mean);
        default: return -100;
    }
}
}  // namespace spyfft
