// Launchers of the reference-precision tapered FFT (mtmfft_f64_kernel.h); its own translation unit.
#include <algorithm>

#include "spy_common.h"
#include "mtmfft_f64_kernel.h"

namespace spyfft {

template <int OUTK, bool MEAN>
static int f64_any_launch_one(hipStream_t stream, const F64Args& a, unsigned grid) {
    auto kern = mtmfft_f64_any_kernel<OUTK, MEAN>;
    const size_t lds = a.work ? 0 : (size_t)2 * a.plan.L * sizeof(double2);      // (plan.L: nfft or the Bluestein length)
    if (lds) SPY_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(256), lds, stream, a);
    SPY_HIP_CHECK(hipGetLastError());
    return 0;
}

// `work` holds 2 * nfft complex128 per workgroup for `chunk` workgroups: the grid goes in launches of that many
int f64_any_launch(hipStream_t stream, F64Args a, long long grid, long long chunk, int outk, bool mean) {
    for (long long w0 = 0; w0 < grid; w0 += chunk) {
        a.wg0 = w0;
        const unsigned g = (unsigned)std::min(chunk, grid - w0);
        int rc;
        switch (outk * 2 + (mean ? 1 : 0)) {
            case 0: rc = f64_any_launch_one<0, false>(stream, a, g); break;
            case 1: rc = f64_any_launch_one<0, true>(stream, a, g); break;
            case 2: rc = f64_any_launch_one<1, false>(stream, a, g); break;
            case 3: rc = f64_any_launch_one<1, true>(stream, a, g); break;
            case 4: rc = f64_any_launch_one<2, false>(stream, a, g); break;
            default: rc = f64_any_launch_one<2, true>(stream, a, g); break;
        }
        if (rc) return rc;
    }
    return 0;
}

}  // namespace spyfft
