// Launchers of K4h, the cross-spectral update on the half-precision matrix cores (csdh_kernel.h); csd.hip only sees this.
#pragma once
#include <hip/hip_runtime.h>

namespace spycsd {

// acc[f] += sum_r X[r, f, :] X[r, f, :]^H for f in [f0, f0 + nf) of (nrows, F, 256) spectra whose |re|, |im| stay below
// absmax[c] per channel: csdh_kernel, then the float32 kernel (3-multiplication, or 4-multiplication if phase_exact) on
// exactly the frequencies csdh_kernel flagged instead of adding (flags: F ints of device scratch, indexed by frequency).
// 0 or a negative spyhip error code.
int csdh_run(hipStream_t stream, const float2* spec, long long nrows, int F, float2* acc, const float* absmax, int* flags, int f0,
             int nf, bool phase_exact);

// absmax[c] = max(absmax[c], |re|, |im| of every spectrum value of channel c); nchan even
int csdh_absmax(hipStream_t stream, const float2* spec, long long nvalues, int nchan, float* absmax);

}  // namespace spycsd
