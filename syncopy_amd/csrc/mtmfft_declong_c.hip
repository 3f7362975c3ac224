// declong_post_kernel instances (mtmfft_declong.h): radix P = 2, 3, 4, 5, 6, 8 x output modes
#include "spy_common.h"
#include "mtmfft_declong.h"

namespace spyfft {

template <int P, int OUTK, bool MEAN>
static int declong_post(hipStream_t stream, const LongArgs& a, int M) {
    const long long tot = (long long)a.nsegc * a.nquad * (M / 2 + 1);
    const long long nb = (tot + 255) / 256;
    if (nb > 0x7fffffffLL) { spy::set_error("fft_exec: grid too large (%lld blocks)", nb); return -1; }
    hipLaunchKernelGGL((declong_post_kernel<P, OUTK, MEAN>), dim3((unsigned)nb), dim3(256), 0, stream, a, M);
    SPY_HIP_CHECK(hipGetLastError());
    return 0;
}

template <int P>
static int declong_post_mode(hipStream_t stream, const LongArgs& a, int M, int outk, bool mean) {
    switch (outk * 2 + (mean ? 1 : 0)) {
        case 0: return declong_post<P, 0, false>(stream, a, M);
        case 1: return declong_post<P, 0, true>(stream, a, M);
        case 2: return declong_post<P, 1, false>(stream, a, M);
        case 3: return declong_post<P, 1, true>(stream, a, M);
        case 4: return declong_post<P, 2, false>(stream, a, M);
        default: return declong_post<P, 2, true>(stream, a, M);
    }
}

int declong_launch_post(hipStream_t stream, const LongArgs& a, int P, int M, int outk, bool mean) {
    switch (P) {
        case 2: return declong_post_mode<2>(stream, a, M, outk, mean);
        case 3: return declong_post_mode<3>(stream, a, M, outk, mean);
        case 4: return declong_post_mode<4>(stream, a, M, outk, mean);
        case 5: return declong_post_mode<5>(stream, a, M, outk, mean);
        case 6: return declong_post_mode<6>(stream, a, M, outk, mean);
        case 8: return declong_post_mode<8>(stream, a, M, outk, mean);
        default: spy::set_error("fft_exec: no radix-%d step", P); return -1;
    }
}

}  // namespace spyfft
