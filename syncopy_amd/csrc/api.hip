// Context, error reporting and small utility kernels of libspyhip.
#include "spy_common.h"

namespace spy {
static thread_local std::string g_last_error;
void set_error(const char* fmt, ...) {
    char buf[1024];
    va_list ap;
    va_start(ap, fmt);
    std::vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_last_error = buf;
}
}  // namespace spy

extern "C" int spyhip_version(void) { return 200; }

extern "C" const char* spyhip_last_error(void) { return spy::g_last_error.c_str(); }

extern "C" int spyhip_ctx_create(int device, spyhip_ctx** out) {
    if (!out) { spy::set_error("ctx_create: null out pointer"); return -1; }
    int ndev = 0;
    SPY_HIP_CHECK(hipGetDeviceCount(&ndev));
    if (device < 0 || device >= ndev) {
        spy::set_error("ctx_create: device %d not available (%d HIP devices visible)", device, ndev);
        return -1;
    }
    hipDeviceProp_t prop;
    SPY_HIP_CHECK(hipGetDeviceProperties(&prop, device));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        spy::set_error("ctx_create: device %d is %s; libspyhip is built for gfx950 (MI355X) only", device,
                       prop.gcnArchName);
        return -4;
    }
    auto* c = new spyhip_ctx();
    c->device = device;
    c->num_cu = prop.multiProcessorCount;
    c->lds_per_block = prop.sharedMemPerBlockOptin ? prop.sharedMemPerBlockOptin : prop.sharedMemPerBlock;
    if (c->lds_per_block < 64 * 1024) c->lds_per_block = 64 * 1024;
    *out = c;
    return 0;
}

extern "C" int spyhip_ctx_destroy(spyhip_ctx* ctx) {
    if (ctx) {
        if (ctx->comm) (void)spyhip_comm_destroy(ctx);
        if (ctx->scratch) (void)hipFree(ctx->scratch);
        if (ctx->comm_buf) (void)hipFree(ctx->comm_buf);
        if (ctx->arena) (void)hipFree(ctx->arena);
        if (ctx->k4h_buf) (void)hipFree(ctx->k4h_buf);
        if (ctx->k4h_done) (void)hipEventDestroy(ctx->k4h_done);
    }
    delete ctx;
    return 0;
}

extern "C" int spyhip_ctx_trim(spyhip_ctx* ctx) {
    if (!ctx) { spy::set_error("ctx_trim: null ctx"); return -1; }
    SPY_HIP_CHECK(hipSetDevice(ctx->device));
    SPY_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (ctx->scratch) { (void)hipFree(ctx->scratch); ctx->scratch = nullptr; ctx->scratch_bytes = 0; }
    if (ctx->arena) { (void)hipFree(ctx->arena); ctx->arena = nullptr; ctx->arena_bytes = 0; }
    return 0;
}

extern "C" int spyhip_ctx_set_stream(spyhip_ctx* ctx, void* stream) {
    if (!ctx) { spy::set_error("ctx_set_stream: null ctx"); return -1; }
    ctx->stream = reinterpret_cast<hipStream_t>(stream);
    return 0;
}

extern "C" int spyhip_ctx_synchronize(spyhip_ctx* ctx) {
    if (!ctx) { spy::set_error("ctx_synchronize: null ctx"); return -1; }
    SPY_HIP_CHECK(hipSetDevice(ctx->device));
    SPY_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return 0;
}

// ---- device memory for hosts without a tensor library ----------------------------------------------------------
extern "C" int spyhip_alloc(spyhip_ctx* ctx, size_t bytes, void** ptr_d) {
    if (!ctx || !ptr_d) { spy::set_error("alloc: null argument"); return -1; }
    *ptr_d = nullptr;
    if (bytes == 0) return 0;
    SPY_HIP_CHECK(hipSetDevice(ctx->device));
    SPY_HIP_CHECK(hipMalloc(ptr_d, bytes));
    return 0;
}

extern "C" int spyhip_free(spyhip_ctx* ctx, void* ptr_d) {
    if (!ctx) { spy::set_error("free: null ctx"); return -1; }
    if (!ptr_d) return 0;
    SPY_HIP_CHECK(hipSetDevice(ctx->device));
    SPY_HIP_CHECK(hipStreamSynchronize(ctx->stream));      // nothing enqueued may still use the block
    SPY_HIP_CHECK(hipFree(ptr_d));
    return 0;
}

extern "C" int spyhip_memset(spyhip_ctx* ctx, void* ptr_d, int value, size_t bytes) {
    if (!ctx || (!ptr_d && bytes)) { spy::set_error("memset: null argument"); return -1; }
    if (bytes == 0) return 0;
    SPY_HIP_CHECK(hipSetDevice(ctx->device));
    SPY_HIP_CHECK(hipMemsetAsync(ptr_d, value, bytes, ctx->stream));
    return 0;
}

extern "C" int spyhip_upload(spyhip_ctx* ctx, void* dst_d, const void* src, size_t bytes) {
    if (!ctx || ((!dst_d || !src) && bytes)) { spy::set_error("upload: null argument"); return -1; }
    if (bytes == 0) return 0;
    SPY_HIP_CHECK(hipSetDevice(ctx->device));
    SPY_HIP_CHECK(hipMemcpyAsync(dst_d, src, bytes, hipMemcpyHostToDevice, ctx->stream));
    SPY_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return 0;
}

extern "C" int spyhip_download(spyhip_ctx* ctx, void* dst, const void* src_d, size_t bytes) {
    if (!ctx || ((!dst || !src_d) && bytes)) { spy::set_error("download: null argument"); return -1; }
    if (bytes == 0) return 0;
    SPY_HIP_CHECK(hipSetDevice(ctx->device));
    SPY_HIP_CHECK(hipMemcpyAsync(dst, src_d, bytes, hipMemcpyDeviceToHost, ctx->stream));
    SPY_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return 0;
}

// ---- the in-HBM trial queue ----------------------------------------------------------------------------------------
struct spyhip_queue {
    spyhip_ctx* ctx = nullptr;
    float* data = nullptr;          // (nrows x nchan)
    long long* seg = nullptr;       // [0, T): first rows, [T, 2T): stop rows
    int64_t nrows = 0;
    int nchan = 0, ntrials = 0;
};

extern "C" int spyhip_queue_upload(spyhip_ctx* ctx, const float* data, int64_t nrows, int nchan,
                                   const int64_t* sampleinfo, int ntrials, spyhip_queue** out) {
    if (!ctx || !data || !sampleinfo || !out) { spy::set_error("queue_upload: null argument"); return -1; }
    if (nrows < 1 || nchan < 1 || ntrials < 1) { spy::set_error("queue_upload: empty data (%lld rows, %d channels, %d trials)", (long long)nrows, nchan, ntrials); return -1; }
    std::vector<long long> seg((size_t)2 * ntrials);
    for (int t = 0; t < ntrials; ++t) {
        const int64_t a = sampleinfo[2 * t], b = sampleinfo[2 * t + 1];
        if (a < 0 || b < a || b > nrows) {       // exact indexing or nothing: a trial must lie inside the matrix
            spy::set_error("queue_upload: trial %d = rows [%lld, %lld) outside the %lld rows of the data", t,
                           (long long)a, (long long)b, (long long)nrows);
            return -1;
        }
        seg[t] = a;
        seg[(size_t)ntrials + t] = b;
    }
    SPY_HIP_CHECK(hipSetDevice(ctx->device));
    auto* q = new spyhip_queue();
    q->ctx = ctx; q->nrows = nrows; q->nchan = nchan; q->ntrials = ntrials;
    const size_t bytes = (size_t)nrows * nchan * sizeof(float);
    if (hipMalloc(reinterpret_cast<void**>(&q->data), bytes) != hipSuccess ||
        hipMalloc(reinterpret_cast<void**>(&q->seg), seg.size() * sizeof(long long)) != hipSuccess) {
        spy::set_error("queue_upload: cannot allocate %zu bytes of device memory", bytes);
        (void)spyhip_queue_destroy(q);
        return -2;
    }
    if (hipMemcpyAsync(q->data, data, bytes, hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
        hipMemcpyAsync(q->seg, seg.data(), seg.size() * sizeof(long long), hipMemcpyHostToDevice, ctx->stream) != hipSuccess ||
        hipStreamSynchronize(ctx->stream) != hipSuccess) {
        spy::set_error("queue_upload: host -> device copy failed");
        (void)spyhip_queue_destroy(q);
        return -2;
    }
    *out = q;
    return 0;
}

extern "C" int spyhip_queue_destroy(spyhip_queue* q) {
    if (!q) return 0;
    if (q->ctx) { (void)hipSetDevice(q->ctx->device); (void)hipStreamSynchronize(q->ctx->stream); }
    if (q->data) (void)hipFree(q->data);
    if (q->seg) (void)hipFree(q->seg);
    delete q;
    return 0;
}

extern "C" const float* spyhip_queue_data(const spyhip_queue* q) { return q ? q->data : nullptr; }

extern "C" int spyhip_queue_segments(const spyhip_queue* q, const int64_t** start_d, const int64_t** stop_d,
                                     int* ntrials) {
    if (!q) { spy::set_error("queue_segments: null queue"); return -1; }
    if (start_d) *start_d = reinterpret_cast<const int64_t*>(q->seg);
    if (stop_d) *stop_d = reinterpret_cast<const int64_t*>(q->seg + q->ntrials);
    if (ntrials) *ntrials = q->ntrials;
    return 0;
}

// ---- elementwise helpers for keeptrials=False accumulation ---------------------------------
// RECIP: the components of complex64 data - NumPy divides a complex array by a real count with Smith's algorithm, which
// for a real divisor is a MULTIPLICATION by the float32 reciprocal 1/T (loops.c.src, CFLOAT_divide: rat = 0, scl = 1/T,
// out = in * scl), not a division: the last bit differs
template <bool RECIP>
__global__ void trial_mean_kernel(const float* __restrict__ in, float* __restrict__ out, long long ntrials,
                                  long long n) {
    // sequential sum over trials in float32, one division at the end: the order of
    // ComputationalRoutine.compute_sequential (computational_routine.py:1022-1032)
    const long long stride = (long long)gridDim.x * blockDim.x;
    const float scl = __fdiv_rn(1.0f, (float)ntrials);
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float s = 0.f;
        for (long long t = 0; t < ntrials; ++t) s = __fadd_rn(s, in[t * n + i]);
        out[i] = RECIP ? __fmul_rn(s, scl) : __fdiv_rn(s, (float)ntrials);
    }
}

extern "C" int spyhip_trial_mean_f32(spyhip_ctx* ctx, const float* in_d, float* out_d, int64_t ntrials, int64_t n) {
    if (!ctx || !in_d || !out_d || ntrials < 1) { spy::set_error("trial_mean: bad argument"); return -1; }
    if (n <= 0) return 0;
    SPY_HIP_CHECK(hipSetDevice(ctx->device));
    long long blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(trial_mean_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, in_d, out_d,
                       (long long)ntrials, (long long)n);
    SPY_HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" int spyhip_trial_mean_c64(spyhip_ctx* ctx, const void* in_d, void* out_d, int64_t ntrials, int64_t n) {
    if (!ctx || !in_d || !out_d || ntrials < 1) { spy::set_error("trial_mean: bad argument"); return -1; }
    if (n <= 0) return 0;
    SPY_HIP_CHECK(hipSetDevice(ctx->device));
    long long blocks = (2 * n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(trial_mean_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream,
                       reinterpret_cast<const float*>(in_d), reinterpret_cast<float*>(out_d), (long long)ntrials, (long long)(2 * n));
    SPY_HIP_CHECK(hipGetLastError());
    return 0;
}

// ---- np.nanmean along one axis of a trial array (statistics/compRoutines.py:22-57: NumpyStatDim / npstats_cF) -----
// element k of a run with NaNs replaced by zero as np.nanmean does before it sums (pofs != 0: complex data - the element
// counts as NaN if either component is)
__device__ __forceinline__ float nan0(const float* a, long long idx, long long pofs) {
    const float v = a[idx];
    if (pofs == 0) return (v != v) ? 0.f : v;
    const float w = a[idx + pofs];
    return ((v != v) || (w != w)) ? 0.f : v;
}

// NumPy's float32 sum of n values that are contiguous in memory (pairwise_sum_FLOAT: eight running sums over blocks of
// at most 128 values, halves of longer runs added recursively) - followed literally so that means over the LAST axis
// agree with the reference to the last bit whenever its divide does
__device__ float np_pairwise_sum(const float* a, long long n) {
    if (n < 8) {
        float res = 0.f;
        for (long long i = 0; i < n; ++i) res = __fadd_rn(res, nan0(a, i, 0));
        return res;
    }
    if (n <= 128) {
        float r[8];
        for (int j = 0; j < 8; ++j) r[j] = nan0(a, j, 0);
        long long i = 8;
        for (; i < n - (n % 8); i += 8)
            for (int j = 0; j < 8; ++j) r[j] = __fadd_rn(r[j], nan0(a, i + j, 0));
        float res = __fadd_rn(__fadd_rn(__fadd_rn(r[0], r[1]), __fadd_rn(r[2], r[3])),
                              __fadd_rn(__fadd_rn(r[4], r[5]), __fadd_rn(r[6], r[7])));
        for (; i < n; ++i) res = __fadd_rn(res, nan0(a, i, 0));
        return res;
    }
    long long n2 = n / 2;
    n2 -= n2 % 8;
    return __fadd_rn(np_pairwise_sum(a, n2), np_pairwise_sum(a + n2, n - n2));
}

// one component of m complex values (interleaved floats): pairwise_sum_CFLOAT on 2m floats - fewer than 4 complex: plain
// loop; up to 64: this component's four running sums r[c], r[c+2], r[c+4], r[c+6]; longer: halves (multiples of 4)
__device__ float np_pairwise_sum_c(const float* a, long long m, long long pofs) {
    if (m < 4) {
        float r = 0.f;
        for (long long k = 0; k < m; ++k) r = __fadd_rn(r, nan0(a, 2 * k, pofs));
        return r;
    }
    if (m <= 64) {
        float r[4];
        for (int j = 0; j < 4; ++j) r[j] = nan0(a, 2 * j, pofs);
        long long k = 4;
        for (; k < m - (m % 4); k += 4)
            for (int j = 0; j < 4; ++j) r[j] = __fadd_rn(r[j], nan0(a, 2 * (k + j), pofs));
        float res = __fadd_rn(__fadd_rn(r[0], r[1]), __fadd_rn(r[2], r[3]));
        for (; k < m; ++k) res = __fadd_rn(res, nan0(a, 2 * k, pofs));
        return res;
    }
    long long m2 = m / 2;
    m2 -= m2 % 4;
    return __fadd_rn(np_pairwise_sum_c(a, m2, pofs), np_pairwise_sum_c(a + 2 * m2, m - m2, pofs));
}

// x (outer, n, inner) float32 (complex64 = inner doubled by the caller: components are independent except for the NaN
// test, which takes the complex element) -> out (outer, inner): NaNs skipped, sum in float32 in NumPy's order
// (axis not last: one accumulator per output walking the n rows in order; last axis: pairwise), division in float64
__global__ void axis_nanmean_kernel(const float* __restrict__ x, long long outer, long long n, long long inner, int cplx,
                                    float* __restrict__ out) {
    const long long tot = outer * inner, stride = (long long)gridDim.x * blockDim.x;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += stride) {
        const long long o = e / inner, i = e - o * inner;
        const float* p = x + o * n * inner + i;
        const long long partner = cplx ? ((i & 1) ? -1 : 1) : 0;       // the other component of a complex element
        long long cnt = 0;
        bool anynan = false;
        for (long long k = 0; k < n; ++k) {
            const float v = p[k * inner], w = cplx ? p[k * inner + partner] : 0.f;
            const bool bad = (v != v) || (w != w);
            anynan |= bad;
            cnt += bad ? 0 : 1;
        }
        float s;
        (void)anynan;
        if (inner == (cplx ? 2 : 1)) {
            // contiguous run (complex: NumPy sums the interleaved floats with the same eight accumulators, i.e. this
            // component with stride 2 over blocks of 64 complex values)
            s = cplx ? np_pairwise_sum_c(p, n, partner) : np_pairwise_sum(p, n);
        } else {
            s = 0.f;
            for (long long k = 0; k < n; ++k) {
                const float v = p[k * inner], w = cplx ? p[k * inner + partner] : 0.f;
                s = __fadd_rn(s, ((v != v) || (w != w)) ? 0.f : v);
            }
        }
        out[e] = (float)((double)s / (double)cnt);
    }
}

extern "C" int spyhip_axis_nanmean(spyhip_ctx* ctx, const void* x_d, int64_t outer, int64_t n, int64_t inner, int is_complex,
                                   void* out_d) {
    if (!ctx || !x_d || !out_d || outer < 1 || n < 1 || inner < 1) { spy::set_error("axis_nanmean: bad argument"); return -1; }
    SPY_HIP_CHECK(hipSetDevice(ctx->device));
    const long long in2 = is_complex ? 2 * inner : inner;
    long long blocks = (outer * in2 + 255) / 256;
    if (blocks > 65535) blocks = 65535;
    hipLaunchKernelGGL(axis_nanmean_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, reinterpret_cast<const float*>(x_d),
                       (long long)outer, (long long)n, in2, is_complex ? 1 : 0, reinterpret_cast<float*>(out_d));
    SPY_HIP_CHECK(hipGetLastError());
    return 0;
}
