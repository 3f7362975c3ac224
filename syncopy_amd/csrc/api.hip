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

extern "C" int spyhip_version(void) { return 100; }

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
    if (ctx && ctx->scratch) (void)hipFree(ctx->scratch);
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

// ---- elementwise helpers for keeptrials=False accumulation ---------------------------------
__global__ void axpy_kernel(const float* __restrict__ x, float* __restrict__ y, long long n, float alpha) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) y[i] += alpha * x[i];
}

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

extern "C" int spyhip_axpy_f32(spyhip_ctx* ctx, const float* x_d, float* y_d, int64_t n, float alpha) {
    if (!ctx || !x_d || !y_d) { spy::set_error("axpy: null argument"); return -1; }
    if (n <= 0) return 0;
    SPY_HIP_CHECK(hipSetDevice(ctx->device));
    long long blocks = (n + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(axpy_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, x_d, y_d, (long long)n, alpha);
    SPY_HIP_CHECK(hipGetLastError());
    return 0;
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
