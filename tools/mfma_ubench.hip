// fp32 MFMA issue-rate microbenchmark in the shape of csd_accum_kernel (development aid).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, int LDSREADS>
__global__ void __launch_bounds__(512) k(float* out, int iters) {
    __shared__ float2 X[4096];
    const int tid = threadIdx.x;
    for (int i = tid; i < 4096; i += 512) X[i] = make_float2(i * 1e-3f, 1.f - i * 1e-3f);
    __syncthreads();
    f32x16 acc[NACC];
    for (int t = 0; t < NACC; ++t)
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float2 a = X[tid], b = X[tid + 512];
    for (int it = 0; it < iters; ++it) {
        if (LDSREADS) {
            a = X[(tid + it * 64) & 4095];
            b = X[(tid + it * 64 + 1024) & 4095];
        }
#pragma unroll
        for (int t = 0; t < NACC; t += 2) {
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc[t], 0, 0, 0);
            acc[t + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.x, acc[t + 1], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc[t], 0, 0, 0);
            acc[t + 1] = __builtin_amdgcn_mfma_f32_32x32x2f32(-a.x, b.y, acc[t + 1], 0, 0, 0);
        }
    }
    float s = 0;
    for (int t = 0; t < NACC; ++t)
        for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * 512 + tid] = s;
}

template <int NACC, int L>
void run(const char* name, int threads, int blocks) {
    float* out;
    hipMalloc(&out, blocks * 512 * 4);
    const int iters = 4000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    k<NACC, L><<<blocks, threads>>>(out, 10);
    hipEventRecord(e0);
    k<NACC, L><<<blocks, threads>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    double flops = (double)blocks * (threads / 64) * iters * NACC * 2 * 4096.0;
    printf("%-40s threads=%d blocks=%d: %.3f ms  %.1f TF\n", name, threads, blocks, ms, flops / ms / 1e9);
    hipFree(out);
}

int main() {
    run<10, 0>("10 acc, no LDS, 8 waves/CU", 512, 256);
    run<10, 1>("10 acc, LDS reads, 8 waves/CU", 512, 256);
    run<8, 0>("8 acc, no LDS, 8 waves/CU", 512, 256);
    run<4, 0>("4 acc, no LDS, 8 waves/CU", 512, 256);
    run<10, 0>("10 acc, no LDS, 4 waves/CU", 256, 256);
    run<4, 0>("4 acc, no LDS, 4 waves/CU", 256, 256);
    run<10, 0>("10 acc, no LDS, 8 waves/CU, 2048 blocks", 512, 2048);
    return 0;
}
