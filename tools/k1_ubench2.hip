// gfx950 micro-probes behind the K1 exchange design (development aid, not part of the library):
//  (1) v_permlane32_swap_b32 v, v  (same register): does lane l get lane l^32's value?
//  (2) do LDS stores issued by one wave of a SIMD slow the VALU stream of the OTHER wave of that SIMD?
//  (3) ds_write_addtid_b32: semantics (address = M0 + offset + 4*lane) and rate vs ds_write2_b64
//  hipcc --offload-arch=gfx950 -O3 -o k1_ubench2 tools/k1_ubench2.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float v2f __attribute__((ext_vector_type(2)));

__global__ void swap_kernel(unsigned* out) {
    unsigned v = threadIdx.x * 3u + 7u;
    asm volatile("v_permlane32_swap_b32 %0, %0" : "+v"(v));
    out[threadIdx.x] = v;
    unsigned a = threadIdx.x, b = 1000u + threadIdx.x;
    asm volatile("v_permlane32_swap_b32 %0, %1" : "+v"(a), "+v"(b));
    out[64 + threadIdx.x] = a;
    out[128 + threadIdx.x] = b;
}

// MODE bit 0: waves 0-3 run a packed-FMA stream; bit 1: waves 4-7 run LDS stores (KIND 0: ds_write2_b64, 1: addtid_b32 x4,
// 2: ds_write_b64 x2, 3: ds_read_b64 x2)
template <int MODE, int KIND>
__global__ void __launch_bounds__(512) mix_kernel(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, wave = tid >> 6;
    v2f a[16];
    for (int i = 0; i < 16; ++i) a[i] = v2f{1.f + i, 1.f - i};
    const v2f m = v2f{1.0000001f, 0.9999999f}, c = v2f{1e-7f, -1e-7f};
    v2f d0 = v2f{(float)tid, 1.f}, d1 = v2f{2.f, (float)tid};
    const unsigned base = (unsigned)(size_t)smem + (unsigned)(tid & 255) * 16u;
    v2f r0 = v2f{0.f, 0.f};
    if (wave < 4) {
        if (MODE & 1) {
            for (int it = 0; it < 3 * iters; ++it) {
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int i = 0; i < 16; ++i) a[i] = __builtin_elementwise_fma(a[i], m, c);
            }
        }
    } else if (MODE & 2) {
        const unsigned m0 = (unsigned)(size_t)smem + (unsigned)(wave - 4) * 256u;
        asm volatile("s_mov_b32 m0, %0" ::"s"(__builtin_amdgcn_readfirstlane(m0)));
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int u = 0; u < 16; ++u) {
                if (KIND == 0) asm volatile("ds_write2_b64 %0, %1, %2 offset1:1" ::"v"(base), "v"(d0), "v"(d1) : "memory");
                else if (KIND == 1) {
                    asm volatile("ds_write_addtid_b32 %0 offset:%1" ::"v"(d0.x), "n"(u * 4096) : "memory");
                    asm volatile("ds_write_addtid_b32 %0 offset:%1" ::"v"(d0.y), "n"(u * 4096 + 1024) : "memory");
                    asm volatile("ds_write_addtid_b32 %0 offset:%1" ::"v"(d1.x), "n"(u * 4096 + 2048) : "memory");
                    asm volatile("ds_write_addtid_b32 %0 offset:%1" ::"v"(d1.y), "n"(u * 4096 + 3072) : "memory");
                } else if (KIND == 2) {
                    asm volatile("ds_write_b64 %0, %1" ::"v"(base), "v"(d0) : "memory");
                    asm volatile("ds_write_b64 %0, %1 offset:8" ::"v"(base), "v"(d1) : "memory");
                } else {
                    asm volatile("ds_read_b64 %0, %1" : "=v"(r0) : "v"(base) : "memory");
                    asm volatile("ds_read_b64 %0, %1 offset:8" : "=v"(r0) : "v"(base) : "memory");
                }
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        }
    }
    float s = r0.x;
    for (int i = 0; i < 16; ++i) s += a[i].x + a[i].y;
    out[blockIdx.x * 512 + tid] = s + reinterpret_cast<float*>(smem)[tid];
}

__global__ void addtid_check(unsigned* out) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    unsigned* l = reinterpret_cast<unsigned*>(smem);
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) l[i] = 0xdeadbeefu;
    __syncthreads();
    const unsigned m0 = (unsigned)(size_t)smem + (threadIdx.x >> 6) * 1024u;
    asm volatile("s_mov_b32 m0, %0" ::"s"(__builtin_amdgcn_readfirstlane(m0)));
    unsigned v = 5000u + threadIdx.x;
    asm volatile("ds_write_addtid_b32 %0 offset:16\n s_waitcnt lgkmcnt(0)" ::"v"(v) : "memory");
    __syncthreads();
    for (int i = threadIdx.x; i < 1024; i += blockDim.x) out[i] = l[i];
}

template <int MODE, int KIND>
float run(int iters) {
    float* out; hipMalloc(&out, 256 * 512 * 4);
    auto k = mix_kernel<MODE, KIND>;
    hipFuncSetAttribute((const void*)k, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<<<256, 512, 80 * 1024>>>(out, 10);
    hipEventRecord(e0);
    k<<<256, 512, 80 * 1024>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipFree(out);
    return ms;
}

int main() {
    unsigned* d; hipMalloc(&d, 4096 * 4);
    std::vector<unsigned> h(4096);
    swap_kernel<<<1, 64>>>(d);
    hipMemcpy(h.data(), d, 192 * 4, hipMemcpyDeviceToHost);
    bool ok = true;
    for (int l = 0; l < 64; ++l) ok &= h[l] == (unsigned)((l ^ 32) * 3 + 7);
    printf("(1) v_permlane32_swap_b32 v,v : lane l <- lane l^32 : %s   (lane0=%u lane32=%u)\n", ok ? "YES" : "NO", h[0], h[32]);
    printf("    swap a,b: a[0]=%u a[32]=%u b[0]=%u b[32]=%u  (a=lane, b=1000+lane before)\n", h[64], h[96], h[128], h[160]);
    addtid_check<<<1, 128, 8192>>>(d);
    hipMemcpy(h.data(), d, 1024 * 4, hipMemcpyDeviceToHost);
    printf("(3) addtid: word[4]=%u (lane0 of wave0 expected 5000) word[4+63]=%u word[256+4]=%u (wave1 lane0 expected 5064) word[3]=%x\n",
           h[4], h[67], h[260], h[3]);
    const int it = 20000;
    const double clk = 2.4e9;
    float t;
    t = run<1, 0>(it); printf("(2) VALU only (4 waves, 1/SIMD)        : %.3f ms  %.2f cycles per pk_fma per wave\n", t, t * 1e-3 * clk / (3.0 * it * 64.0));
    const char* kn[4] = {"ds_write2_b64", "4 x ds_write_addtid_b32", "2 x ds_write_b64", "2 x ds_read_b64"};
    float tl[4], tb[4];
    tl[0] = run<2, 0>(it); tl[1] = run<2, 1>(it); tl[2] = run<2, 2>(it); tl[3] = run<2, 3>(it);
    tb[0] = run<3, 0>(it); tb[1] = run<3, 1>(it); tb[2] = run<3, 2>(it); tb[3] = run<3, 3>(it);
    for (int k = 0; k < 4; ++k)
        printf("    %-24s alone %.3f ms = %.1f cycles per 16 B x 64 lanes per wave (4 waves, %.1f B/clk/CU);  with the VALU waves: %.3f ms (VALU alone %.3f)\n",
               kn[k], tl[k], tl[k] * 1e-3 * clk / (it * 16.0), 4.0 * 1024.0 / (tl[k] * 1e-3 * clk / (it * 16.0)), tb[k], t);
    return 0;
}
