// K1m launcher: the packed mixed-radix mtmfft kernels live in their own translation unit (they compile for a minute).
#include <hip/hip_runtime.h>

#include "spy_common.h"
#include "mtmfft_mixed.h"

namespace {
template <int OUTK, bool MEAN, int LB>
int launch(hipStream_t stream, const spyfft::MtmArgs& a, const spyfft::MixPlan& g, int threads, size_t lds, unsigned grid) {
    auto kern = spyfft::mtmfft_mixed_kernel<OUTK, MEAN, LB>;
    static size_t attr = 0;
    if (lds > attr) {
        SPY_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        attr = lds;
    }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(threads), lds, stream, a, g);
    SPY_HIP_CHECK(hipGetLastError());
    return 0;
}
template <int LB>
int launch_mode(hipStream_t stream, const spyfft::MtmArgs& a, const spyfft::MixPlan& g, int threads, size_t lds, unsigned grid,
                int outk, bool mean) {
    switch (outk * 2 + (mean ? 1 : 0)) {
        case 0: return launch<0, false, LB>(stream, a, g, threads, lds, grid);
        case 1: return launch<0, true, LB>(stream, a, g, threads, lds, grid);
        case 2: return launch<1, false, LB>(stream, a, g, threads, lds, grid);
        case 3: return launch<1, true, LB>(stream, a, g, threads, lds, grid);
        case 4: return launch<2, false, LB>(stream, a, g, threads, lds, grid);
        default: return launch<2, true, LB>(stream, a, g, threads, lds, grid);
    }
}
}  // namespace

namespace spyfft {
int mixed_launch(hipStream_t stream, const MtmArgs& a, const MixPlan& g, int threads, size_t lds, unsigned grid, int outk,
                 bool mean) {
    return threads <= 512 ? launch_mode<512>(stream, a, g, threads, lds, grid, outk, mean)
                          : launch_mode<1024>(stream, a, g, threads, lds, grid, outk, mean);
}
}  // namespace spyfft
