// Argument block of the reference-precision transform kernels (mtmfft_f64_kernel.h, mtmfft_dec64_kernel.h), on its own so
// that the host side (mtmfft.hip) shares the definition without pulling in the kernels.
#pragma once
#include "mtmfft_kernel.h"    // MtmArgs
#include "f64_stockham.h"     // PlusPlan

namespace spyfft {

struct F64Args {
    MtmArgs m;
    const double* tapers64;      // (ntaper x nsig) float64: the reference's windows, not rounded to float32
    const double2* tw64;         // exp(-2 pi i m / nfft)
    const double2* tw64_full;    // (CfgD64::HALF: tw64 then belongs to nfft / 2) exp(-2 pi i f / nfft), the half-step table
    double scale64;              // unused by the arithmetic (the float32 m.scale multiplies, as in the reference)
    // any-length variant (mtmfft_f64_any_kernel): factor schedule, two length-nfft work arrays per workgroup, first
    // work item of this launch
    spywil::PlusPlan plan;
    double2* work;
    long long wg0;
    // Bluestein form of the any-length kernel (a prime factor above 61): plan.L = M = 2^m >= 2 nfft - 1 and tw64 belongs
    // to M; chirp64[n] = exp(-i pi n^2 / nfft) (nfft entries), bhat64 = FFT_M of the wrapped conjugate chirp, / M
    int blue_n;                  // nfft of the Bluestein form, 0 otherwise
    const double2* chirp64;
    const double2* bhat64;
};

}  // namespace spyfft
