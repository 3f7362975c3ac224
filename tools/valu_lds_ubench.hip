// VALU / LDS issue-rate microbenchmark for the FFT engine design (development aid, not part of the library):
//   * is v_pk_fma_f32 / v_pk_add_f32 twice the work of v_fma_f32 / v_add_f32 per issue slot on gfx950, or the same?
//   * ds_write_b64 / b128 and ds_read_b64 / b128 rates per CU at 1, 2, 4 waves per SIMD
//   hipcc --offload-arch=gfx950 -O3 -o valu_lds_ubench tools/valu_lds_ubench.hip
#include <hip/hip_runtime.h>
#include <cstdio>

typedef float v2f __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ void __launch_bounds__(256) valu_kernel(float* out, int iters, float seed) {
    // 16 independent chains per lane so that the dependent-issue latency never binds
    v2f a[16];
    for (int i = 0; i < 16; ++i) a[i] = v2f{seed + i, seed - i};
    const v2f m = v2f{1.0000001f, 0.9999999f}, c = v2f{1e-7f, -1e-7f};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 16; ++i) {
            if (MODE == 0) {                     // v_pk_fma_f32
                a[i] = __builtin_elementwise_fma(a[i], m, c);
            } else if (MODE == 1) {              // 2 x v_fma_f32
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].x) : "v"(m.x), "v"(c.x));
                asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(a[i].y) : "v"(m.y), "v"(c.y));
            } else if (MODE == 2) {              // v_pk_add_f32
                asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(c));
            } else if (MODE == 3) {              // 2 x v_add_f32
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i].x) : "v"(c.x));
                asm volatile("v_add_f32 %0, %0, %1" : "+v"(a[i].y) : "v"(c.y));
            } else if (MODE == 4) {              // v_pk_fma_f32 via asm (no compiler scheduling)
                asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(a[i]) : "v"(m), "v"(c));
            } else {                             // v_pk_mul_f32
                asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(a[i]) : "v"(m));
            }
        }
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += a[i].x + a[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

// MODE 0: write b64, 1: write b128, 2: read b64, 3: read b128, 4: write b32
typedef float v4f __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ void __launch_bounds__(256) lds_kernel(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x;
    v4f v = v4f{(float)tid, tid + 1.f, tid + 2.f, tid + 3.f};
    v2f v2 = v2f{(float)tid, tid + 1.f};
    float v1 = (float)tid;
    v4f r4 = v4f{0.f, 0.f, 0.f, 0.f};
    v2f r2 = v2f{0.f, 0.f};
    const unsigned base = (unsigned)(size_t)smem;
    const unsigned esz = MODE == 0 || MODE == 2 ? 8u : (MODE == 4 ? 4u : 16u);
    // conflict-free: lane-contiguous addresses; 16 operations per wait
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const unsigned addr = base + (unsigned)(u * 256 + tid) * esz;
            if (MODE == 0) asm volatile("ds_write_b64 %0, %1" ::"v"(addr), "v"(v2) : "memory");
            else if (MODE == 1) asm volatile("ds_write_b128 %0, %1" ::"v"(addr), "v"(v) : "memory");
            else if (MODE == 2) asm volatile("ds_read_b64 %0, %1" : "=v"(r2) : "v"(addr) : "memory");
            else if (MODE == 3) asm volatile("ds_read_b128 %0, %1" : "=v"(r4) : "v"(addr) : "memory");
            else asm volatile("ds_write_b32 %0, %1" ::"v"(addr), "v"(v1) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    out[blockIdx.x * blockDim.x + tid] = r4.x + r2.x + reinterpret_cast<float*>(smem)[tid];
}

template <int MODE>
void run_valu(const char* name, int blocks_per_cu) {
    float* out;
    const int blocks = 256 * blocks_per_cu;
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    const int iters = 100000;
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    valu_kernel<MODE><<<blocks, 256>>>(out, 10, 1.f);
    hipEventRecord(e0);
    valu_kernel<MODE><<<blocks, 256>>>(out, iters, 1.f);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    // lane-ops: every inner step updates 2 floats per lane per chain
    const double laneops = (double)blocks * 256 * iters * 16 * 2;
    const double cyc_per_wave_step = ms * 1e-3 * 2.4e9 / ((double)blocks_per_cu * iters * 16);   // per SIMD: one wave of the block
    printf("%-28s %d waves/SIMD: %8.3f ms  %7.1f G float-updates/s  %.2f cycles per (2 floats x 64 lanes) per wave slot\n", name,
           blocks_per_cu, ms, laneops / ms / 1e6, cyc_per_wave_step);
    hipFree(out);
}

template <int MODE>
void run_lds(const char* name, int blocks_per_cu, int bytes_per_lane) {
    float* out;
    const int blocks = 256 * blocks_per_cu;
    hipMalloc(&out, (size_t)blocks * 256 * 4);
    const int iters = 40000;
    const size_t lds = 16 * 256 * 16;     // 64 KiB
    hipFuncSetAttribute(reinterpret_cast<const void*>(lds_kernel<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    lds_kernel<MODE><<<blocks, 256, lds>>>(out, 10);
    hipEventRecord(e0);
    lds_kernel<MODE><<<blocks, 256, lds>>>(out, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    const double bytes = (double)blocks * 256 * iters * 16 * bytes_per_lane;
    printf("%-28s %d blocks/CU: %8.3f ms  %7.1f B/clk/CU  (%.1f TB/s aggregate)\n", name, blocks_per_cu, ms,
           bytes / 256 / (ms * 1e-3 * 2.4e9), bytes / ms / 1e9);
    hipFree(out);
}

int main() {
    for (int w : {1, 2, 4}) {
        if (w == 1) { run_valu<0>("v_pk_fma_f32 (compiler)", 1); run_valu<4>("v_pk_fma_f32 (asm)", 1); run_valu<1>("2 x v_fma_f32", 1); run_valu<2>("v_pk_add_f32", 1); run_valu<3>("2 x v_add_f32", 1); run_valu<5>("v_pk_mul_f32", 1); }
        if (w == 2) { run_valu<0>("v_pk_fma_f32 (compiler)", 2); run_valu<4>("v_pk_fma_f32 (asm)", 2); run_valu<1>("2 x v_fma_f32", 2); run_valu<2>("v_pk_add_f32", 2); run_valu<3>("2 x v_add_f32", 2); }
        if (w == 4) { run_valu<0>("v_pk_fma_f32 (compiler)", 4); run_valu<4>("v_pk_fma_f32 (asm)", 4); run_valu<1>("2 x v_fma_f32", 4); run_valu<2>("v_pk_add_f32", 4); run_valu<3>("2 x v_add_f32", 4); }
    }
    for (int b : {1, 2}) {
        run_lds<4>("ds_write_b32", b, 4);
        run_lds<0>("ds_write_b64", b, 8);
        run_lds<1>("ds_write_b128", b, 16);
        run_lds<2>("ds_read_b64", b, 8);
        run_lds<3>("ds_read_b128", b, 16);
    }
    return 0;
}
