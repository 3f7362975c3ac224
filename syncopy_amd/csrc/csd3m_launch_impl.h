// Body shared by the csd3m_*.hip translation units: each defines SPY_M3_LIST (its channel counts) and a part name.
#pragma once
#include "spy_common.h"
#include "csd3m_launch.h"
#include "csd3m_kernel.h"

namespace spycsd {

template <int CH, bool EXACT = true, bool RECT = false, bool M4 = false>
int m3_launch_one(hipStream_t stream, CsdArgs a, long long nprow) {
    if (nprow <= 0) return 0;
    constexpr int NP = M3Tab<CH, RECT>::NP;            // workgroups per packed row (> 1 above 256 channels)
    a.item_base = 0;
    a.item_end = nprow * M3_TILES_PER_F;
    auto kern = csd3m_kernel<CH, 8, EXACT, RECT, M4>;
    // (the attribute is per device and the call is cheap: no process-wide "already set" flag)
    SPY_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      M3_LDS_BYTES));
    // XCD-aware groups of 8 frequencies (NP > 1) / of 32 for the channel-quad-blocked layout (the kernel permutes them)
    const long long grid = NP > 1 ? ((nprow + 7) / 8) * 8 * NP : (a.blocked ? ((nprow + 31) / 32) * 32 : nprow);
    if (grid > 0x7fffffffLL) { spy::set_error("csd_accumulate: grid too large"); return -1; }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(512), M3_LDS_BYTES, stream, a);
    SPY_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace spycsd
