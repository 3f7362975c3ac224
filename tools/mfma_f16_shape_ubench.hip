// v_mfma_f32_16x16x32_f16 against v_mfma_f32_32x32x16_f16 on REAL operand bits (random fp16 values, fresh registers per
// instruction group), registers only, two waves per SIMD on every CU: what the matrix pipe sustains under the chip's power
// management for the two shapes - the question behind "K4h on 32 x 32 x 16 instructions" (VERDICT r5, next 5).  Same flops
// per matrix-pipe cycle at peak (16384 flop in 4 passes against 32768 in 8); the 32 x 32 shape reads half the operand
// registers per flop.  Development aid:
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_f16_shape_ubench.hip -o /tmp/mfma_shape && /tmp/mfma_shape
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

__device__ inline f16x8 rnd8(uint32_t& s, bool zero) {
    f16x8 v;
    for (int i = 0; i < 8; ++i) {
        s = s * 1664525u + 1013904223u;
        v[i] = zero ? (_Float16)0.f : (_Float16)(((int)(s >> 9) % 2001 - 1000) * 1e-3f);
    }
    return v;
}

// NOP operand registers alternate per instruction: 16 A + 16 B fragments (16x16: 34 accumulators as in K4h's wave)
template <bool ZERO>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) k16(float* out, int iters) {
    uint32_t s = threadIdx.x * 2654435761u + blockIdx.x;
    f16x8 a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = rnd8(s, ZERO); b[i] = rnd8(s, ZERO); }
    f32x4 acc[32];
    for (int t = 0; t < 32; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int t = 0; t < 32; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[t % 8], b[(t / 4) % 8], acc[t], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(a[i]), "+v"(b[i]));
    }
    float r = 0;
    for (int t = 0; t < 32; ++t) r += acc[t][0] + acc[t][1] + acc[t][2] + acc[t][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <bool ZERO>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2))) k32(float* out, int iters) {
    uint32_t s = threadIdx.x * 2654435761u + blockIdx.x;
    f16x8 a[8], b[8];
    for (int i = 0; i < 8; ++i) { a[i] = rnd8(s, ZERO); b[i] = rnd8(s, ZERO); }
    f32x16 acc[8];                                   // the same 128 accumulator registers
    for (int t = 0; t < 8; ++t)
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int rep = 0; rep < 4; ++rep)
#pragma unroll
            for (int t = 0; t < 8; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[(t + rep) % 8], b[(t / 2 + rep) % 8], acc[t], 0, 0, 0);
#pragma unroll
        for (int i = 0; i < 8; ++i) asm volatile("" : "+v"(a[i]), "+v"(b[i]));
    }
    float r = 0;
    for (int t = 0; t < 8; ++t)
        for (int q = 0; q < 16; ++q) r += acc[t][q];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <typename K>
void run(const char* name, K kern, double flop_per_mfma) {
    float* out;
    const int blocks = 256 * 2;                      // two 4-wave workgroups per CU: two waves per SIMD
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    const int iters = 40000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    kern<<<blocks, 256>>>(out, 100);
    hipDeviceSynchronize();
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        kern<<<blocks, 256>>>(out, iters);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        const double flop = (double)blocks * 4 * iters * 32 * flop_per_mfma;
        printf("%-52s %8.2f ms  %7.1f TFLOP/s  (%.2f GHz if the pipe never idles)\n", name, ms, flop / ms / 1e9,
               flop / ms / 1e9 / 2500.0 * 2.4);
    }
    hipFree(out);
}

int main() {
    run("16x16x32 f16, random operands", k16<false>, 16384.0);
    run("32x32x16 f16, random operands", k32<false>, 32768.0);
    run("16x16x32 f16, zero operands", k16<true>, 16384.0);
    run("32x32x16 f16, zero operands", k32<true>, 32768.0);
    return 0;
}
