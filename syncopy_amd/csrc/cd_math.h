// complex128 helpers shared by the Wilson kernels (granger_kernels.h, wilson_plus_kernel.h) and the reference-precision
// FFT (mtmfft_f64_kernel.h)
#pragma once

namespace spywil {

typedef double2 cd;

__device__ __forceinline__ cd cmul(cd a, cd b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }
__device__ __forceinline__ cd cmulc(cd a, cd b) { return make_double2(a.x * b.x + a.y * b.y, a.y * b.x - a.x * b.y); }  // a*conj(b)
__device__ __forceinline__ cd cadd(cd a, cd b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ cd csub(cd a, cd b) { return make_double2(a.x - b.x, a.y - b.y); }
__device__ __forceinline__ double cabs2(cd a) { return a.x * a.x + a.y * a.y; }

}  // namespace spywil
