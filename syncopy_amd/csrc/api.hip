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
    }
    delete ctx;
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
__global__ void trial_mean_kernel(const float* __restrict__ in, float* __restrict__ out, long long ntrials,
                                  long long n) {
    // sequential sum over trials in float32, one division at the end: the order of
    // ComputationalRoutine.compute_sequential (computational_routine.py:1022-1032)
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        float s = 0.f;
        for (long long t = 0; t < ntrials; ++t) s += in[t * n + i];
        out[i] = s / (float)ntrials;
    }
}

extern "C" int spyhip_trial_mean_f32(spyhip_ctx* ctx, const float* in_d, float* out_d, int64_t ntrials, int64_t n) {
    if (!ctx || !in_d || !out_d || ntrials < 1) { spy::set_error("trial_mean: bad argument"); return -1; }
    if (n <= 0) return 0;
    SPY_HIP_CHECK(hipSetDevice(ctx->device));
    long long blocks = (n + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(trial_mean_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, in_d, out_d,
                       (long long)ntrials, (long long)n);
    SPY_HIP_CHECK(hipGetLastError());
    return 0;
}
