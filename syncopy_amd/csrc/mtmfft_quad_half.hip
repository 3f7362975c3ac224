// mtmfft_quad_kernel<13, 1, ..., HALF>: 2^14 samples as channel pairs through the 8192-point radix-16 engine (mtmfft2_kernel.h)
#include "spy_common.h"
#include "mtmfft2_kernel.h"

namespace spyfft {

template <int OUTK, bool MEAN>
static int quad_half_launch_one(hipStream_t stream, MtmArgs a, int npairs) {
    using C = Cfg2<13, 1>;
    a.npg = npairs;
    int S = 16; if (S > a.npg) S = a.npg;                 // workgroups sharing 128-byte rows (XCD cluster)
    a.S = S;
    a.ncl = (a.npg + S - 1) / S;
    const long long nclusters = (long long)a.nseg * a.ncl;
    const long long grid = ((nclusters + 7) / 8) * S * 8;
    if (grid > 0x7fffffffLL) { spy::set_error("fft_exec: grid too large (%lld blocks)", grid); return -1; }
    auto kern = mtmfft_quad_kernel<13, 1, OUTK, MEAN, true>;
    SPY_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)C::LDS_BYTES));
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(C::NTHREADS), C::LDS_BYTES, stream, a);
    SPY_HIP_CHECK(hipGetLastError());
    return 0;
}

// `a.tapers` = the windows times scale / 2 (the plan's pre-scaled table), `a.tw` = exp(-2 pi i m / 8192), `a.twh` the half-step table
int quad_half_launch(hipStream_t stream, const MtmArgs& a, int npairs, int outk, bool mean) {
    switch (outk * 2 + (mean ? 1 : 0)) {
        case 0: return quad_half_launch_one<0, false>(stream, a, npairs);
        case 1: return quad_half_launch_one<0, true>(stream, a, npairs);
        case 2: return quad_half_launch_one<1, false>(stream, a, npairs);
        case 3: return quad_half_launch_one<1, true>(stream, a, npairs);
        case 4: return quad_half_launch_one<2, false>(stream, a, npairs);
        default: return quad_half_launch_one<2, true>(stream, a, npairs);
    }
}

}  // namespace spyfft
