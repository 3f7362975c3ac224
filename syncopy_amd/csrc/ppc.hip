// Host side of K7: pairwise phase consistency (spyhip_ppc_accumulate / spyhip_ppc_accumulate_csd /
// spyhip_ppc_finalize).
#include "spy_common.h"
#include "ppc_kernel.h"

extern "C" int spyhip_ppc_accumulate(spyhip_ctx* ctx, const void* spec_d, int ntrials, int ntaper, int nfreq,
                                     int nchan, void* acc_d) {
    if (!ctx || !spec_d || !acc_d) { spy::set_error("ppc_accumulate: null argument"); return -1; }
    if (ntrials < 0 || ntaper < 1 || nfreq < 1 || nchan < 1) { spy::set_error("ppc_accumulate: bad shape"); return -1; }
    if (ntrials == 0) return 0;
    SPY_HIP_CHECK(hipSetDevice(ctx->device));
    spyppc::PpcArgs a{};
    a.spec = reinterpret_cast<const float2*>(spec_d);
    a.ntrials = ntrials; a.ntaper = ntaper; a.F = nfreq; a.C = nchan;
    a.acc = reinterpret_cast<float2*>(acc_d);
    const long long nt = (nchan + 31) / 32, blocks = 8LL * ((nfreq + 7) / 8) * (nt * (nt + 1) / 2);   // 8 XCDs x frequencies each
    if (blocks > 0x7fffffffLL) { spy::set_error("ppc_accumulate: grid too large"); return -1; }
    const size_t lds = 2 * (size_t)2 * ntaper * 32 * sizeof(float2);
    if (lds > ctx->lds_per_block) { spy::set_error("ppc_accumulate: %d tapers do not fit the LDS staging buffer", ntaper); return -3; }
    SPY_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(spyppc::ppc_accum_kernel),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(spyppc::ppc_accum_kernel, dim3((unsigned)blocks), dim3(256), lds, ctx->stream, a);
    SPY_HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" int spyhip_ppc_accumulate_csd(spyhip_ctx* ctx, const void* csd_d, int ntrials, int64_t nelem, void* acc_d) {
    if (!ctx || !csd_d || !acc_d) { spy::set_error("ppc_accumulate_csd: null argument"); return -1; }
    if (ntrials < 0 || nelem < 1) { spy::set_error("ppc_accumulate_csd: bad shape"); return -1; }
    if (ntrials == 0) return 0;
    SPY_HIP_CHECK(hipSetDevice(ctx->device));
    const long long blocks = (nelem + 255) / 256;
    if (blocks > 0x7fffffffLL) { spy::set_error("ppc_accumulate_csd: grid too large"); return -1; }
    hipLaunchKernelGGL(spyppc::ppc_accum_csd_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream,
                       reinterpret_cast<const float2*>(csd_d), (long long)nelem, ntrials,
                       reinterpret_cast<float2*>(acc_d));
    SPY_HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" int spyhip_ppc_finalize(spyhip_ctx* ctx, const void* acc_d, int nfreq, int ni, int nj, int lower_only,
                                   int64_t ntrials, void* out_d) {
    if (!ctx || !acc_d || !out_d) { spy::set_error("ppc_finalize: null argument"); return -1; }
    if (nfreq < 1 || ni < 1 || nj < 1) { spy::set_error("ppc_finalize: bad shape"); return -1; }
    if (ntrials < 2) { spy::set_error("ppc_finalize: at least two trials are needed"); return -1; }
    if (lower_only && ni != nj) { spy::set_error("ppc_finalize: a lower-triangle accumulator is square"); return -1; }
    SPY_HIP_CHECK(hipSetDevice(ctx->device));
    const long long blocks = ((long long)nfreq * ni * nj + 255) / 256;
    if (blocks > 0x7fffffffLL) { spy::set_error("ppc_finalize: grid too large"); return -1; }
    hipLaunchKernelGGL(spyppc::ppc_finalize_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream,
                       reinterpret_cast<const float2*>(acc_d), nfreq, ni, nj, lower_only, (double)ntrials,
                       reinterpret_cast<float*>(out_d));
    SPY_HIP_CHECK(hipGetLastError());
    return 0;
}
