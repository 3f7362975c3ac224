// Host side of K8: cross-covariance lags from accumulated cross spectra (spyhip_ccov_from_accumulator).
#include <algorithm>
#include <cmath>
#include <mutex>
#include <vector>

#include "spy_common.h"
#include "ccov_kernel.h"
#include "../../include/spyhip.h"

namespace {
const double PI = 3.14159265358979323846264338327950288;

// twiddle tables exp(-2 pi i m / L) per (device, log2 L), built once per process
const float2* twiddles(spyhip_ctx* ctx, int log2n) {
    static std::mutex mtx;
    static spy::DevBuf<float2>* cache[64][16] = {};
    std::lock_guard<std::mutex> lock(mtx);
    if (ctx->device < 0 || ctx->device >= 64 || log2n < 0 || log2n >= 16) return nullptr;
    auto*& slot = cache[ctx->device][log2n];
    if (!slot) {
        const int L = 1 << log2n;
        std::vector<float2> tw(L);
        for (int m = 0; m < L; ++m) {
            const double ang = -2.0 * PI * m / L;
            tw[m] = make_float2((float)std::cos(ang), (float)std::sin(ang));
        }
        auto* buf = new spy::DevBuf<float2>();
        if (buf->upload(tw, ctx->stream)) { delete buf; return nullptr; }
        slot = buf;
    }
    return slot->p;
}

template <int LOG2N, int G>
int launch_lags(spyhip_ctx* ctx, const spyfft::CcovArgs& a) {
    using C = spyfft::Cfg2<LOG2N, G>;
    auto kern = spyfft::ccov_lags_kernel<LOG2N, G>;
    SPY_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES));
    const long long grid = 8 * (((a.npairs + 2 * G - 1) / (2 * G) + 7) / 8);     // 8 XCDs x pair blocks each
    if (grid > 0x7fffffffLL) { spy::set_error("ccov: grid too large"); return -1; }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(C::NTHREADS), C::LDS_BYTES, ctx->stream, a);
    SPY_HIP_CHECK(hipGetLastError());
    return 0;
}
int normalize(spyhip_ctx* ctx, float* out, const float2* acc, int nlag, int nchan, int mode, float dc) {
    const size_t need = (size_t)nchan * sizeof(float);
    if (need > ctx->scratch_bytes) {
        if (ctx->scratch) { (void)hipFree(ctx->scratch); ctx->scratch = nullptr; ctx->scratch_bytes = 0; }
        SPY_HIP_CHECK(hipMalloc(&ctx->scratch, need));
        ctx->scratch_bytes = need;
    }
    float* d = reinterpret_cast<float*>(ctx->scratch);
    hipLaunchKernelGGL(spyfft::ccov_diag_kernel, dim3((nchan + 255) / 256), dim3(256), 0, ctx->stream,
                       out, acc, nchan, mode, dc, d);
    const long long n = (long long)nlag * nchan * nchan;
    if ((n + 255) / 256 > 0x7fffffffLL) { spy::set_error("ccov: grid too large"); return -1; }
    hipLaunchKernelGGL(spyfft::ccov_normalize_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, ctx->stream,
                       out, n, nchan, d);
    SPY_HIP_CHECK(hipGetLastError());
    return 0;
}

// lags for L > 8192 through the forward real transform (ccov_kernel.h): chunks of channel pairs, ~1 GiB of scratch each way
int long_lags(spyhip_ctx* ctx, const float2* acc, int L, int nchan, int nsamples, double scale, float* out) {
    const long long npairs = (long long)nchan * (nchan + 1) / 2;
    long long cap = ((long long)1 << 27) / L;                      // pairs per chunk: L x 2 cap floats <= 1 GiB
    if (cap > npairs) cap = npairs;
    cap = ((cap + 1) / 2) * 2;                                     // 2 cap channels: a multiple of 4 (16-byte row reads)
    if (cap < 2) cap = 2;
    const int F = L / 2 + 1;
    spy::DevBuf<float> pq;
    spy::DevBuf<float2> spec;
    spy::DevBuf<long long> seg;
    if (pq.alloc((size_t)L * 2 * cap) || spec.alloc((size_t)F * 2 * cap) || seg.alloc(3)) {
        spy::set_error("ccov_from_accumulator: out of device memory for the %d-point lag transform", L);
        return -2;
    }
    const long long hseg[3] = {0, 0, L};
    SPY_HIP_CHECK(hipMemcpyAsync(seg.p, hseg, sizeof hseg, hipMemcpyHostToDevice, ctx->stream));
    std::vector<double> ones((size_t)L, 1.0);
    spyhip_fft_plan* plan = nullptr;
    int rc = spyhip_fft_plan_create(ctx, L, L, (int)(2 * cap), 1, ones.data(), 1.0, -1, 0, nullptr, 0, SPYHIP_OUT_FOURIER, 1, &plan);
    if (rc) return rc;
    const int nlag = nsamples / 2 + (nsamples & 1), q = (nsamples & 1) ? 0 : 1;
    const float sc = (float)(scale / (double)L);
    for (long long p0 = 0; p0 < npairs && !rc; p0 += cap) {
        const int npc = (int)std::min<long long>(cap, npairs - p0);
        const unsigned g1 = (unsigned)std::min<long long>(((long long)L * cap + 255) / 256, 1 << 20);
        hipLaunchKernelGGL(spyfft::ccov_extend_kernel, dim3(g1), dim3(256), 0, ctx->stream, acc, nchan, L, p0, npc, (int)cap, pq.p);
        rc = spyhip_fft_exec(plan, pq.p, 2 * cap, nullptr, reinterpret_cast<const int64_t*>(seg.p),
                             reinterpret_cast<const int64_t*>(seg.p + 1), reinterpret_cast<const int64_t*>(seg.p + 2), 1, spec.p);
        if (rc) break;
        const unsigned g2 = (unsigned)std::min<long long>(((long long)F * cap + 255) / 256, 1 << 20);
        hipLaunchKernelGGL(spyfft::ccov_combine_kernel, dim3(g2), dim3(256), 0, ctx->stream, spec.p, L, p0, npc, (int)cap, nchan,
                           nsamples, nlag, q, sc, out);
        if (hipGetLastError() != hipSuccess) { spy::set_error("ccov: lag kernels failed to launch"); rc = -1; }
    }
    if (hipStreamSynchronize(ctx->stream) != hipSuccess && !rc) { spy::set_error("ccov: lag transform failed"); rc = -1; }
    spyhip_fft_plan_destroy(plan);
    return rc;
}
}  // namespace

extern "C" int spyhip_ccov_nfft(int nsamples) {
    if (nsamples < 1) return -1;
    const int nlag = nsamples / 2 + (nsamples & 1);
    long long L = 1024;
    while (L < (long long)nsamples + nlag) L <<= 1;        // no circular wrap for |lag| <= nlag
    return L <= (1 << 19) ? (int)L : -1;
}

extern "C" int spyhip_ccov_from_accumulator(spyhip_ctx* ctx, const void* acc_d, int nfft, int nchan, int nsamples,
                                            double scale, int norm, void* out_d) {
    if (!ctx || !acc_d || !out_d) { spy::set_error("ccov_from_accumulator: null argument"); return -1; }
    if (nchan < 1 || nsamples < 1) { spy::set_error("ccov_from_accumulator: bad shape"); return -1; }
    if (norm < 0 || norm > 2) { spy::set_error("ccov_from_accumulator: norm must be 0, 1 or 2"); return -1; }
    if (spyhip_ccov_nfft(nsamples) < 0 || nfft != spyhip_ccov_nfft(nsamples)) {
        spy::set_error("ccov_from_accumulator: %d samples need a transform of %d points (got %d); trials longer "
                       "than 349525 samples are not supported", nsamples, spyhip_ccov_nfft(nsamples), nfft);
        return -1;
    }
    if (nfft > 8192) {
        SPY_HIP_CHECK(hipSetDevice(ctx->device));
        const int rc = long_lags(ctx, reinterpret_cast<const float2*>(acc_d), nfft, nchan, nsamples, scale,
                                 reinterpret_cast<float*>(out_d));
        if (rc) return rc;
        if (norm) {
            const float dc = (float)(scale / ((double)nsamples * (double)nsamples));
            return normalize(ctx, reinterpret_cast<float*>(out_d), reinterpret_cast<const float2*>(acc_d),
                             nsamples / 2 + (nsamples & 1), nchan, norm, dc);
        }
        return 0;
    }
    SPY_HIP_CHECK(hipSetDevice(ctx->device));
    const int log2n = spy::ilog2((unsigned)nfft);
    spyfft::CcovArgs a{};
    a.acc = reinterpret_cast<const float2*>(acc_d);
    a.tw = twiddles(ctx, log2n);
    if (!a.tw) { spy::set_error("ccov_from_accumulator: twiddle table allocation failed"); return -2; }
    a.C = nchan; a.nsamples = nsamples;
    a.nlag = nsamples / 2 + (nsamples & 1);
    a.q = (nsamples & 1) ? 0 : 1;
    a.npairs = (long long)nchan * (nchan + 1) / 2;
    a.scale = (float)(scale / (double)nfft);
    a.out = reinterpret_cast<float*>(out_d);
    int rc;
    switch (log2n) {
        case 10: rc = launch_lags<10, 4>(ctx, a); break;
        case 11: rc = launch_lags<11, 2>(ctx, a); break;
        case 12: rc = launch_lags<12, 1>(ctx, a); break;
        case 13: rc = launch_lags<13, 1>(ctx, a); break;
        default: spy::set_error("ccov_from_accumulator: unsupported transform length %d", nfft); return -1;
    }
    if (rc) return rc;
    if (norm) {
        // mode 2: mean_a^2 = |X_a(0)|^2 / N^2 in the units of out: acc[0,a,a] * scale / N^2
        const float dc = (float)(scale / ((double)nsamples * (double)nsamples));
        return normalize(ctx, a.out, a.acc, a.nlag, nchan, norm, dc);
    }
    return 0;
}

extern "C" int spyhip_ccov_normalize(spyhip_ctx* ctx, void* cc_d, int nlag, int nchan) {
    if (!ctx || !cc_d) { spy::set_error("ccov_normalize: null argument"); return -1; }
    if (nlag < 1 || nchan < 1) { spy::set_error("ccov_normalize: bad shape"); return -1; }
    SPY_HIP_CHECK(hipSetDevice(ctx->device));
    return normalize(ctx, reinterpret_cast<float*>(cc_d), nullptr, nlag, nchan, 1, 0.f);
}
