// SIMT emulation of the small HIP subset the kernels in syncopy_amd/csrc use.
//
// TEST INFRASTRUCTURE ONLY.  It lets `tests/` compile the *same* kernel
// sources with g++ and run single workgroups on the CPU (one OS thread per
// GPU thread, pthread barriers for __syncthreads) so that index arithmetic,
// LDS layouts and MFMA lane maps can be checked against the oracle without a
// GPU.  The product (`syncopy_amd/libspyhip.so`) is never built from this
// header and never falls back to it.
#pragma once
#include <pthread.h>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include <functional>
#include <type_traits>

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __restrict__

struct dim3 {
    unsigned x, y, z;
    dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {}
};
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct double2 { double x, y; };
static inline float2 make_float2(float a, float b) { return float2{a, b}; }
static inline float4 make_float4(float a, float b, float c, float d) { return float4{a, b, c, d}; }
static inline double2 make_double2(double a, double b) { return double2{a, b}; }
// IEEE round-to-nearest float32 add / divide (the host compiler does not contract or re-associate them)
static inline float __fadd_rn(float a, float b) { volatile float r = a + b; return r; }
static inline float __fdiv_rn(float a, float b) { volatile float r = a / b; return r; }
static inline float __fmul_rn(float a, float b) { volatile float r = a * b; return r; }
static inline void __threadfence() {}
static inline int __float_as_int(float f) { int i; std::memcpy(&i, &f, 4); return i; }
static inline float __int_as_float(int i) { float f; std::memcpy(&f, &i, 4); return f; }
static inline unsigned __umulhi(unsigned a, unsigned b) { return (unsigned)(((unsigned long long)a * b) >> 32); }

namespace emu {
struct BlockCtx {
    pthread_barrier_t bar;
    unsigned nthreads;
    char* dyn_smem;
    // scratch for wave-level exchange (shuffles / mfma): 64 lanes x 64 floats per wave
    std::vector<double> xchg;
    std::vector<pthread_barrier_t> wave_bar;
};
extern thread_local dim3 t_threadIdx;
extern thread_local dim3 t_blockIdx;
extern thread_local dim3 t_blockDim;
extern thread_local dim3 t_gridDim;
extern thread_local BlockCtx* t_ctx;
}  // namespace emu

#define threadIdx (emu::t_threadIdx)
#define blockIdx (emu::t_blockIdx)
#define blockDim (emu::t_blockDim)
#define gridDim (emu::t_gridDim)

static inline void __syncthreads() { pthread_barrier_wait(&emu::t_ctx->bar); }

// static __shared__ arrays: blocks run one after another, so a plain static is
// shared by the threads of the running block.
#define __shared__ static
// ---- csrc/spy_intrinsics.h, host version: the kernel headers reach every hardware idiom through these names; defining
// the header's include guard here keeps the device versions out of the emulator build
#define SPY_INTRINSICS_H
#define SPY_DYN_SMEM(type, name) type* name = reinterpret_cast<type*>(emu::t_ctx->dyn_smem)
#define SPY_WAVES_PER_EU(lo, hi)
#define SPY_MIN_WAVES_PER_EU(n)
static inline int spy_opaque(int v) { return v; }
static inline void spy_opaque2(double&, double&) {}
static inline void spy_sched_fence() {}
static inline int spy_wave_index(int tid) { return tid >> 6; }
static inline float spy_rsqrt(float x) { return 1.0f / sqrtf(x); }
static inline float spy_sqrt(float x) { return sqrtf(x); }
static inline float spy_log2(float x) { return log2f(x); }
static inline void spy_wait_vmem() {}
static inline void spy_glds16(const void* gsrc, char* lds_wave_base) {           // the same copy, lane by lane
    std::memcpy(lds_wave_base + 16 * (emu::t_threadIdx.x & 63), gsrc, 16);
}

// ---- wave-level primitives (wave = 64 consecutive threads of the block) ----
namespace emu {
inline unsigned lane() { return t_threadIdx.x & 63u; }
inline unsigned wave() { return t_threadIdx.x >> 6; }
inline void wave_sync() { pthread_barrier_wait(&t_ctx->wave_bar[wave()]); }
inline double* wave_scratch() { return t_ctx->xchg.data() + (size_t)wave() * 64 * 80; }
}  // namespace emu

template <typename T>
static inline T __shfl_xor(T v, int mask) {
    double* s = emu::wave_scratch();
    static_assert(sizeof(T) <= sizeof(double), "shfl type");
    emu::wave_sync();
    std::memcpy(&s[emu::lane()], &v, sizeof(T));
    emu::wave_sync();
    T r;
    unsigned src = (emu::lane() ^ (unsigned)mask) & 63u;
    // lanes beyond the block's thread count do not exist: return own value
    unsigned base = emu::wave() * 64;
    if (base + src >= emu::t_ctx->nthreads) src = emu::lane();
    std::memcpy(&r, &s[src], sizeof(T));
    emu::wave_sync();
    return r;
}

static inline float spy_lane_swap1(float v) { return __shfl_xor(v, 1); }
static inline float spy_lane_swap2(float v) { return __shfl_xor(v, 2); }

// v_mfma_f32_32x32x2_f32: lane l supplies A[i=l&31][k=l>>5], B[k=l>>5][j=l&31];
// D[row=(r&3)+8*(r>>2)+4*(l>>5)][col=l&31] for r in [0,16)   (cdna_hip_programming.md section 3)
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
static inline float __builtin_amdgcn_rcpf(float x) { return 1.0f / x; }
static inline float __builtin_amdgcn_rsqf(float x) { return 1.0f / sqrtf(x); }
static inline f32x16 __builtin_amdgcn_mfma_f32_32x32x2f32(float a, float b, f32x16 c, int, int, int) {
    double* s = emu::wave_scratch();
    float* A = reinterpret_cast<float*>(s);        // [2][32]
    float* B = A + 64;                             // [2][32]
    unsigned l = emu::lane();
    emu::wave_sync();
    A[(l >> 5) * 32 + (l & 31)] = a;
    B[(l >> 5) * 32 + (l & 31)] = b;
    emu::wave_sync();
    f32x16 d = c;
    for (int r = 0; r < 16; ++r) {
        int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
        int col = l & 31;
        float acc = d[r];
        for (int k = 0; k < 2; ++k) acc = fmaf(A[k * 32 + row], B[k * 32 + col], acc);
        d[r] = acc;
    }
    emu::wave_sync();
    return d;
}

// v_mfma_f32_16x16x4_f32: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15]; D[row=4*(l>>4)+r][col=l&15] for r in [0,4)
static inline f32x4 __builtin_amdgcn_mfma_f32_16x16x4f32(float a, float b, f32x4 c, int, int, int) {
    double* s = emu::wave_scratch();
    float* A = reinterpret_cast<float*>(s);        // [4][16]
    float* B = A + 64;                             // [4][16]
    unsigned l = emu::lane();
    emu::wave_sync();
    A[(l >> 4) * 16 + (l & 15)] = a;
    B[(l >> 4) * 16 + (l & 15)] = b;
    emu::wave_sync();
    f32x4 d = c;
    for (int r = 0; r < 4; ++r) {
        int row = 4 * (l >> 4) + r, col = l & 15;
        float acc = d[r];
        for (int k = 0; k < 4; ++k) acc = fmaf(A[k * 16 + row], B[k * 16 + col], acc);
        d[r] = acc;
    }
    emu::wave_sync();
    return d;
}

// v_mfma_f64_16x16x4_f64: A[i=l&15][k=l>>4], B[k=l>>4][j=l&15]; D[row=(l>>4)+4*r][col=l&15]
typedef double f64x4 __attribute__((ext_vector_type(4)));
static inline f64x4 __builtin_amdgcn_mfma_f64_16x16x4f64(double a, double b, f64x4 c, int, int, int) {
    double* s = emu::wave_scratch();
    double* A = s;        // [4][16]
    double* B = s + 64;   // [4][16]
    unsigned l = emu::lane();
    emu::wave_sync();
    A[(l >> 4) * 16 + (l & 15)] = a;
    B[(l >> 4) * 16 + (l & 15)] = b;
    emu::wave_sync();
    f64x4 d = c;
    for (int r = 0; r < 4; ++r) {
        int row = (l >> 4) + 4 * r, col = l & 15;
        double acc = d[r];
        for (int k = 0; k < 4; ++k) acc = fma(A[k * 16 + row], B[k * 16 + col], acc);
        d[r] = acc;
    }
    emu::wave_sync();
    return d;
}

static inline float atomicAdd(float* p, float v) {
    unsigned* u = reinterpret_cast<unsigned*>(p);
    unsigned old = __atomic_load_n(u, __ATOMIC_RELAXED), nw;
    float f;
    do {
        std::memcpy(&f, &old, 4);
        float g = f + v;
        std::memcpy(&nw, &g, 4);
    } while (!__atomic_compare_exchange_n(u, &old, nw, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
    return f;
}

static inline unsigned atomicMax(unsigned* p, unsigned v) {
    unsigned old = __atomic_load_n(p, __ATOMIC_RELAXED);
    while (old < v && !__atomic_compare_exchange_n(p, &old, v, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
    return old;
}
static inline unsigned __float_as_uint(float f) { unsigned u; std::memcpy(&u, &f, 4); return u; }

namespace emu {
// Run `kernel(args...)` for every block of `grid` (sequentially) with `block.x`
// OS threads each.  `only_block` >= 0 restricts the run to one block.
template <typename FF>
void launch(dim3 grid, dim3 block, size_t dyn_smem_bytes, FF&& body, long only_block = -1) {
    using F = typename std::remove_reference<FF>::type;
    unsigned nthr = block.x * block.y * block.z;
    for (unsigned bz = 0; bz < grid.z; ++bz)
        for (unsigned by = 0; by < grid.y; ++by)
            for (unsigned bx = 0; bx < grid.x; ++bx) {
                long lin = ((long)bz * grid.y + by) * grid.x + bx;
                if (only_block >= 0 && lin != only_block) continue;
                BlockCtx ctx;
                ctx.nthreads = nthr;
                std::vector<char> smem(dyn_smem_bytes + 64);
                ctx.dyn_smem = smem.data();
                unsigned nw = (nthr + 63) / 64;
                ctx.xchg.assign((size_t)nw * 64 * 80, 0.0);
                ctx.wave_bar.resize(nw);
                pthread_barrier_init(&ctx.bar, nullptr, nthr);
                for (unsigned w = 0; w < nw; ++w) {
                    unsigned cnt = (w + 1) * 64 <= nthr ? 64 : nthr - w * 64;
                    pthread_barrier_init(&ctx.wave_bar[w], nullptr, cnt);
                }
                struct Arg { F* f; BlockCtx* c; dim3 tid, bid, bd, gd; };
                std::vector<Arg> args(nthr);
                std::vector<pthread_t> th(nthr);
                pthread_attr_t attr;
                pthread_attr_init(&attr);
                pthread_attr_setstacksize(&attr, 1 << 20);
                for (unsigned t = 0; t < nthr; ++t) {
                    args[t] = Arg{&body, &ctx, dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y)),
                                  dim3(bx, by, bz), block, grid};
                    pthread_create(&th[t], &attr, [](void* p) -> void* {
                        Arg* a = static_cast<Arg*>(p);
                        t_threadIdx = a->tid; t_blockIdx = a->bid; t_blockDim = a->bd; t_gridDim = a->gd;
                        t_ctx = a->c;
                        (*a->f)();
                        return nullptr;
                    }, &args[t]);
                }
                for (unsigned t = 0; t < nthr; ++t) pthread_join(th[t], nullptr);
                pthread_attr_destroy(&attr);
                pthread_barrier_destroy(&ctx.bar);
                for (auto& b : ctx.wave_bar) pthread_barrier_destroy(&b);
            }
}
}  // namespace emu
