// Microbenchmark behind DESIGN section 8 item 2 (K4 on the bf16 matrix cores with exactly split fp32 operands): issue rates of
//   (a) v_mfma_f32_16x16x4_f32          - what csd3m_kernel executes today (one fp32 value per lane and operand),
//   (b) v_mfma_f32_16x16x16_bf16 x 2    - the same lane <-> (row, k) mapping with 8 bf16 product slots per original k:
//                                         a1 b1, a1 b2, a2 b1, a1 b3, a2 b2, a3 b1 (+ a2 b3, a3 b2),
//   (c) = (b) with the fp32 -> 3 x bf16 split of both operands done in registers for every instruction pair,
//   (d) = (c) with each split fragment reused for FOUR instruction pairs (a 4 x 4 register tile per wave).
// Registers only (no memory): upper bounds of what a kernel could issue.  hipcc --offload-arch=gfx950 -O3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef short s16x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ unsigned short bf16_rn(float x) {
    unsigned u = __float_as_uint(x);
    u += 0x7fffu + ((u >> 16) & 1u);
    return (unsigned short)(u >> 16);
}
__device__ __forceinline__ float bf16_f(unsigned short h) { return __uint_as_float((unsigned)h << 16); }

// a -> (a1, a2, a3), a = a1 + a2 + a3 up to 2^-24 |a|
__device__ __forceinline__ void split3(float a, unsigned short& h1, unsigned short& h2, unsigned short& h3) {
    h1 = bf16_rn(a);
    const float r1 = a - bf16_f(h1);
    h2 = bf16_rn(r1);
    h3 = bf16_rn(r1 - bf16_f(h2));
}

__global__ void __launch_bounds__(256) k_f32(float* out, int iters) {
    f32x4 acc[4] = {};
    float a = threadIdx.x * 1e-3f, b = 1.f + threadIdx.x * 1e-4f;
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, acc[t], 0, 0, 0);
        a += 1e-6f;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
}

template <int MODE>   // 1: (b) no split, 2: (c) split per pair, 3: (d) split once per 4 pairs
__global__ void __launch_bounds__(256) k_bf16(float* out, int iters) {
    f32x4 acc[4] = {};
    float a = threadIdx.x * 1e-3f, b = 1.f + threadIdx.x * 1e-4f;
    s16x4 A0 = {1, 2, 3, 4}, A1 = {5, 6, 7, 8}, B0 = {1, 2, 3, 4}, B1 = {4, 3, 2, 1};
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            if (MODE == 2 || (MODE == 3 && t == 0)) {
                unsigned short a1, a2, a3, b1, b2, b3;
                split3(a + t, a1, a2, a3);
                split3(b + t, b1, b2, b3);
                A0 = s16x4{(short)a1, (short)a1, (short)a2, (short)a1};
                A1 = s16x4{(short)a2, (short)a3, (short)a2, (short)a3};
                B0 = s16x4{(short)b1, (short)b2, (short)b1, (short)b3};
                B1 = s16x4{(short)b2, (short)b1, (short)b3, (short)b2};
            }
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(A0, B0, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x16bf16_1k(A1, B1, acc[t], 0, 0, 0);
        }
        a += 1e-6f;
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3];
}

template <class F>
static double time_ms(F launch) {
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    launch();
    hipDeviceSynchronize();
    hipEventRecord(e0);
    launch();
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    return ms;
}

int main() {
    const int blocks = 256 * 8, iters = 20000;
    float* out;
    hipMalloc(&out, blocks * 256 * sizeof(float));
    // fp32-equivalent multiply-adds per loop iteration and wave: 4 tiles x 16 x 16 x 4
    const double macs = (double)blocks * 4 /*waves*/ * iters * 4.0 * 16 * 16 * 4;
    const double t0 = time_ms([&] { hipLaunchKernelGGL(k_f32, dim3(blocks), dim3(256), 0, 0, out, iters); });
    const double t1 = time_ms([&] { hipLaunchKernelGGL(k_bf16<1>, dim3(blocks), dim3(256), 0, 0, out, iters); });
    const double t2 = time_ms([&] { hipLaunchKernelGGL(k_bf16<2>, dim3(blocks), dim3(256), 0, 0, out, iters); });
    const double t3 = time_ms([&] { hipLaunchKernelGGL(k_bf16<3>, dim3(blocks), dim3(256), 0, 0, out, iters); });
    printf("fp32-equivalent multiply-adds of a 16x16x4 step, %d workgroups x 4 waves x %d iterations x 4 accumulator tiles\n", blocks, iters);
    printf("(a) v_mfma_f32_16x16x4_f32                         %8.2f ms  %7.1f TFLOP/s (fp32-equivalent)\n", t0, 2 * macs / t0 * 1e-9);
    printf("(b) 2 x v_mfma_f32_16x16x16_bf16, no split         %8.2f ms  %7.1f TFLOP/s   %.2f x (a)\n", t1, 2 * macs / t1 * 1e-9, t0 / t1);
    printf("(c) (b) + split of both operands per pair          %8.2f ms  %7.1f TFLOP/s   %.2f x (a)\n", t2, 2 * macs / t2 * 1e-9, t0 / t2);
    printf("(d) (b) + one split per four pairs (4 x 4 tile)    %8.2f ms  %7.1f TFLOP/s   %.2f x (a)\n", t3, 2 * macs / t3 * 1e-9, t0 / t3);
    return 0;
}
