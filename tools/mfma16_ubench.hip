// fp32 MFMA issue-rate microbenchmark, one or two waves per SIMD (development aid):
//   v_mfma_f32_16x16x4_f32 with NACC independent accumulators vs v_mfma_f32_32x32x2_f32
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NACC, int DISTINCT>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) k16(float* out, int iters) {
    f32x4 acc[NACC];
    for (int t = 0; t < NACC; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    float a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = threadIdx.x * 1e-3f + i; b[i] = 1.f - threadIdx.x * 1e-3f + i; }
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < NACC; ++t)
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[DISTINCT ? t % 8 : 0], b[DISTINCT ? (t / 3) % 8 : 0], acc[t], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(a[i]), "+v"(b[i]));
    }
    float s = 0;
    for (int t = 0; t < NACC; ++t) s += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int NACC>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(1, 2))) k32(float* out, int iters) {
    f32x16 acc[NACC];
    for (int t = 0; t < NACC; ++t)
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float a = threadIdx.x * 1e-3f, b = 1.f - a;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < NACC; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
        asm volatile("" : "+v"(a), "+v"(b));
    }
    float s = 0;
    for (int t = 0; t < NACC; ++t)
        for (int r = 0; r < 16; ++r) s += acc[t][r];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <typename K>
void run(const char* name, K kern, int nacc, double flop_per_mfma, int blocks_per_cu) {
    float* out;
    const int blocks = 256 * blocks_per_cu;
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    const int iters = 20000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    kern<<<blocks, 256>>>(out, 10);
    hipEventRecord(e0);
    kern<<<blocks, 256>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double nm = (double)blocks * 4 * iters * nacc;
    printf("%-44s %d wave(s)/SIMD: %7.3f ms  %6.1f TF  %.1f cycles per MFMA per SIMD at 2.35 GHz\n", name, blocks_per_cu, ms,
           nm * flop_per_mfma / ms / 1e9, ms * 1e-3 * 2.35e9 / ((double)iters * nacc * blocks_per_cu));
    hipFree(out);
}

int main() {
    for (int w : {1, 2}) {
        run("16x16x4, 51 accumulators, same operands", k16<51, 0>, 51, 2048, w);
        run("16x16x4, 51 accumulators, 8+8 operand regs", k16<51, 1>, 51, 2048, w);
        run("16x16x4, 12 accumulators", k16<12, 1>, 12, 2048, w);
        run("32x32x2, 13 accumulators", k32<13>, 13, 4096, w);
        run("32x32x2, 4 accumulators", k32<4>, 4, 4096, w);
    }
    return 0;
}
