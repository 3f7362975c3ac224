// Host side of K4h (csdh_kernel.h): launches of the half-precision cross-spectral kernel, of its float32 stand-in on the
// frequencies it declined, and of the per-channel range pass.
#include "spy_common.h"
#include "csd_args.h"
#include "csdh_kernel.h"
#include "csdh_launch.h"

namespace spycsd {

int csdh_run(hipStream_t stream, const float2* spec, long long nrows, int F, float2* acc, const float* absmax, int* flags, int f0,
             int nf, bool phase_exact) {
    if (nf <= 0 || nrows <= 0) return 0;
    if (nrows > 0x7fffffffLL) { spy::set_error("csd_accumulate: more than 2^31 - 1 rows in one call"); return -1; }
    CsdhArgs a{};
    a.spec = spec; a.nrows = nrows; a.F = F; a.acc = acc; a.absmax = absmax; a.flags = flags; a.f0 = f0; a.nf = nf;
    a.rs = (long long)F * 256; a.fs = 256;
    SPY_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(csdh_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      CSDH_LDS_BYTES));
    hipLaunchKernelGGL(csdh_kernel, dim3((unsigned)nf), dim3(512), CSDH_LDS_BYTES, stream, a);
    SPY_HIP_CHECK(hipGetLastError());
    // the frequencies it flagged (dynamic range beyond the fp16 pair, non-finite values): float32, all others return at once
    CsdArgs b{};
    b.spec = spec; b.nrows = nrows; b.F = F; b.C = 256; b.acc = acc;
    b.nt = 8; b.ntiles = M3_TILES_PER_F; b.nitems = (long long)F * M3_TILES_PER_F; b.cpad = 256;
    b.item_base = (long long)f0 * M3_TILES_PER_F; b.item_end = (long long)(f0 + nf) * M3_TILES_PER_F;
    b.only_flagged = flags;
    if (phase_exact) {
        auto k = csd3m_kernel<256, 8, true, false, true>;
        SPY_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, M3_LDS_BYTES));
        hipLaunchKernelGGL(k, dim3((unsigned)nf), dim3(512), M3_LDS_BYTES, stream, b);
    } else {
        auto k = csd3m_kernel<256, 8, true, false, false>;
        SPY_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(k), hipFuncAttributeMaxDynamicSharedMemorySize, M3_LDS_BYTES));
        hipLaunchKernelGGL(k, dim3((unsigned)nf), dim3(512), M3_LDS_BYTES, stream, b);
    }
    SPY_HIP_CHECK(hipGetLastError());
    return 0;
}

int csdh_absmax(hipStream_t stream, const float2* spec, long long nvalues, int nchan, float* absmax) {
    if (nvalues <= 0) return 0;
    if (nchan & 1 || nchan > 512 || 512 % nchan != 0) { spy::set_error("csd range pass: %d channels", nchan); return -1; }
    // a thread stays on one channel pair: the grid stride (blocks * 256 float4s) is a multiple of nchan / 2
    hipLaunchKernelGGL(csdh_absmax_kernel, dim3(2048), dim3(256), 0, stream, reinterpret_cast<const float4*>(spec), nvalues / 2, nchan,
                       reinterpret_cast<unsigned*>(absmax));
    SPY_HIP_CHECK(hipGetLastError());
    return 0;
}

}  // namespace spycsd
