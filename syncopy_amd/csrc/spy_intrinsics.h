// The hardware idioms the kernel headers share.  Everything that is an AMDGCN builtin, an inline-assembly statement, a
// kernel attribute or a compiler vector type goes through one of the names below, so that the kernel sources themselves
// are plain C++: the host-side kernel emulator of the test suite (tests/emu/) defines SPY_INTRINSICS_H and its own
// versions of the same names before it includes a kernel header, and nothing in csrc/ knows about it.
// ONE EXCEPTION, stated rather than hidden: csdh_kernel.h (K4h) uses v_mfma_f32_16x16x32_f16, v_fma_mix inline assembly,
// s_setprio, sched_barrier and readfirstlane directly - its instruction stream IS the kernel - and is
// therefore not built by the emulator: K4h is covered on the GPU only (tests/test_gpu_k4h.py, test_gpu_depth.py, the smoke
// test, bench.py's selfcheck); its float32 stand-in csd3m_kernel is emulated.
#ifndef SPY_INTRINSICS_H
#define SPY_INTRINSICS_H
#include <hip/hip_runtime.h>

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef double f64x4 __attribute__((ext_vector_type(4)));

// the workgroup's dynamic LDS, 16-byte aligned (cdna_hip_programming.md Guideline 17)
#define SPY_DYN_SMEM(type, name) extern __shared__ __attribute__((aligned(16))) char name##_raw[]; \
    type* name = reinterpret_cast<type*>(name##_raw)
// register budget of a kernel: exactly / at least this many waves per SIMD
#define SPY_WAVES_PER_EU(lo, hi) __attribute__((amdgpu_waves_per_eu(lo, hi)))
#define SPY_MIN_WAVES_PER_EU(n) __attribute__((amdgpu_waves_per_eu(n)))

// a value the compiler must treat as unknown (keeps loop-invariant address arithmetic or twiddle powers from being
// hoisted into dozens of live registers)
__device__ __forceinline__ int spy_opaque(int v) { asm volatile("" : "+v"(v)); return v; }
__device__ __forceinline__ void spy_opaque2(double& a, double& b) { asm volatile("" : "+v"(a), "+v"(b)); }
// nothing moves across this point in the instruction schedule
__device__ __forceinline__ void spy_sched_fence() { __builtin_amdgcn_sched_barrier(0); }
// index of the wave inside its workgroup as a SCALAR (tests on it become s_cbranch, not exec masks)
__device__ __forceinline__ int spy_wave_index(int tid) { return __builtin_amdgcn_readfirstlane(tid >> 6); }
// value of the neighbouring lane (lane ^ 1): one DPP move, no LDS traffic
__device__ __forceinline__ float spy_lane_swap1(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0xB1, 0xF, 0xF, true));   // quad_perm:[1,0,3,2]
}
// ... and of the lane two away (lane ^ 2)
__device__ __forceinline__ float spy_lane_swap2(float v) {
    return __int_as_float(__builtin_amdgcn_mov_dpp(__float_as_int(v), 0x4E, 0xF, 0xF, true));   // quad_perm:[2,3,0,1]
}
// quarter-rate hardware approximations where 1 ulp is enough
__device__ __forceinline__ float spy_rsqrt(float x) { return __builtin_amdgcn_rsqf(x); }
__device__ __forceinline__ float spy_sqrt(float x) { return __builtin_amdgcn_sqrtf(x); }
__device__ __forceinline__ float spy_log2(float x) { return __log2f(x); }     // v_log_f32
// 16 bytes per lane global -> LDS: destination = wave-uniform LDS byte address + 16 * lane.  Issued as inline assembly ON
// PURPOSE: hipcc would otherwise guard the next ds_read of the loop with s_waitcnt vmcnt(0) and park the matrix pipe for
// a whole DMA latency once per chunk; the callers order the copies by hand (spy_wait_vmem before the barrier that
// publishes a buffer).  M0 (the LDS base of the copy) is saved and restored around the instruction.
__device__ __forceinline__ void spy_glds16(const void* gsrc, char* lds_wave_base) {
    unsigned keep;
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(size_t)lds_wave_base);
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep)
                 : "v"(gsrc), "s"(dst)
                 : "memory");
}
// every vector-memory operation of this wave (the copies above included) has completed
__device__ __forceinline__ void spy_wait_vmem() { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); }

#endif  // SPY_INTRINSICS_H
