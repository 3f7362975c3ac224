// Launch plumbing shared by the translation units of the reference-precision compile-time-schedule kernel
// (mtmfft_dec64_kernel.h)
#pragma once
#include "spy_common.h"
#include "mtmfft_dec64_cfg.h"

namespace spyfft {

template <class Cf, int OUTK, bool MEAN>
int dec64_launch_one(hipStream_t stream, F64Args fa, int npairs) {      // (Cf::HALF: single channels)
    constexpr int G = Cf::G;
    MtmArgs& a = fa.m;
    a.npg = (npairs + G - 1) / G;
    int S = (Cf::HALF ? 32 : 16) / G; if (S < 1) S = 1; if (S > a.npg) S = a.npg;    // workgroups sharing 128-byte rows (XCD cluster)
    a.S = S;
    a.ncl = (a.npg + S - 1) / S;
    const long long nclusters = (long long)a.nseg * a.ncl;
    const long long grid = ((nclusters + 7) / 8) * S * 8;
    if (grid > 0x7fffffffLL) { spy::set_error("fft_exec: grid too large (%lld blocks)", grid); return -1; }
    auto kern = mtmfft_dec64_kernel<Cf, OUTK, MEAN>;
    SPY_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)Cf::LDS_BYTES));
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(Cf::NTHREADS), Cf::LDS_BYTES, stream, fa);
    SPY_HIP_CHECK(hipGetLastError());
    return 0;
}

template <class Cf>
int dec64_launch_mode(hipStream_t stream, const F64Args& a, int npairs, int outk, bool mean) {
    switch (outk * 2 + (mean ? 1 : 0)) {
        case 0: return dec64_launch_one<Cf, 0, false>(stream, a, npairs);
        case 1: return dec64_launch_one<Cf, 0, true>(stream, a, npairs);
        case 2: return dec64_launch_one<Cf, 1, false>(stream, a, npairs);
        case 3: return dec64_launch_one<Cf, 1, true>(stream, a, npairs);
        case 4: return dec64_launch_one<Cf, 2, false>(stream, a, npairs);
        default: return dec64_launch_one<Cf, 2, true>(stream, a, npairs);
    }
}

}  // namespace spyfft
