// K3 at the reference's own precision (spyhip_cwt_plan_set_precision): the wavelet transform as scipy.signal.fftconvolve
// computes it for cwt_time (specest/wavelets/transform.py:88-108) and cwtSL (specest/superlet.py:311-375) - a float64
// FFT convolution of the detrended float32 trial with the complex128 taps, rounded to complex64 where the reference
// stores it - instead of the float32 overlap-save kernels, whose ABSOLUTE error (~5e-7 of a trial's largest
// coefficient) becomes a relative one wherever a coefficient is small: the roots of superlet products amplify it.
//
// One workgroup of 256 threads per (segment, channel): ONE linear convolution of length L = 2^m >= nsig + taps - 1
// (no blocks, no halo bookkeeping: every kernel length is served by the same code), generic complex128 Stockham passes
// (f64_stockham.h) over three length-L work arrays in global memory - the zero-padded signal's spectrum is kept and
// multiplied by each scale's kernel spectrum (host float64, 1/L folded in), one inverse transform per scale.  Results
// go into the float32 kernels' staging layout (segment, scale, channel, time), so detrending, post-selection, trial
// sums and the transposition (cwt_scatter_kernel) are shared with them.  Speed is not the point: ~50x the float32
// kernels at 128 ch x 16384 - the per-trial route moves 210 MB per trial over PCIe in about the same time.
#pragma once
#include "cd_math.h"
#include "f64_stockham.h"
#include "cwt_kernel.h"

namespace spyfft {

struct Cwt64Args {
    CwtArgs c;                    // data / segments / trend / staging / out_kind as the float32 kernels take them
    int L;                        // transform length
    spywil::PlusPlan plan;        // radix schedule of L
    const double2* tw64;          // exp(-2 pi i m / L)
    const double2* hspec64;       // (nscales x L): FFT_L(h_s) / L
    const int* centre;            // per scale: c_s, the "same" offset: y[n] = full[n + c_s]
    double2* work;                // 3 L complex128 per workgroup of a launch
    long long wg0;                // first (segment, channel) item of this launch
};

// OUTK: 0 = power, 1 = any other real conversion, 2 = complex
template <int OUTK>
__global__ void __launch_bounds__(256) cwt64_kernel(Cwt64Args fa) {
    using spywil::cd;
    constexpr bool CPLX = (OUTK == 2);
    const CwtArgs& a = fa.c;
    const int L = fa.L, tid = threadIdx.x;
    const long long item = fa.wg0 + blockIdx.x;
    const int b = (int)(item / a.nchan), c = (int)(item % a.nchan);
    const long long col = a.chan_idx ? a.chan_idx[c] : c;
    const long long start = a.seg_start[b], tlo = a.trial_lo[b];
    cd* A = reinterpret_cast<cd*>(fa.work) + (size_t)blockIdx.x * 3 * (size_t)L;
    cd* B = A + L;
    cd* X = B + L;
    const cd* tw = reinterpret_cast<const cd*>(fa.tw64);

    double mean = 0.0, slope = 0.0, mid = 0.0;
    if (a.detrend >= 0) {
        const double* t = a.trend + ((size_t)b * a.nchan + c) * 2;
        mean = t[0];
        slope = t[1];
        mid = 0.5 * (double)(a.trial_hi[b] - tlo - 1);
    }
    // ---- the detrended float32 signal (as wavelet_cF hands it to cwt, compRoutines.py:582-595), zero-padded to L
    for (int n = tid; n < L; n += 256) {
        float x = 0.f;
        if (n < a.nsig) {
            const long long row = start + n;
            x = a.data[row * a.ld + col];
            if (a.detrend >= 0) x -= (float)(mean + slope * ((double)(row - tlo) - mid));
        }
        A[n] = make_double2((double)x, 0.0);
    }
    __syncthreads();
    cd *src = A, *dst = B;
    int Ns = 1;
    for (int q = 0; q < fa.plan.nfac; ++q) {
        spywil::po_pass_any(src, dst, L, fa.plan.radix[q], Ns, tw, -1, tid);
        __syncthreads();
        Ns *= fa.plan.radix[q];
        cd* t = src; src = dst; dst = t;
    }
    for (int n = tid; n < L; n += 256) X[n] = src[n];
    __syncthreads();

    const int nst = a.nscales_total ? a.nscales_total : a.nscales;
    for (int s = 0; s < a.nscales; ++s) {
        const cd* H = reinterpret_cast<const cd*>(fa.hspec64) + (size_t)s * L;
        for (int n = tid; n < L; n += 256) A[n] = spywil::cmul(X[n], H[n]);
        __syncthreads();
        src = A; dst = B; Ns = 1;
        for (int q = 0; q < fa.plan.nfac; ++q) {
            spywil::po_pass_any(src, dst, L, fa.plan.radix[q], Ns, tw, +1, tid);
            __syncthreads();
            Ns *= fa.plan.radix[q];
            cd* t = src; src = dst; dst = t;
        }
        const int cs = fa.centre[s];
        const size_t rowo = (((size_t)b * nst + (a.sidx ? a.sidx[s] : s)) * a.nchan + c) * (size_t)a.nsig;
        for (int n = tid; n < a.nsig; n += 256) {
            const cd y = src[n + cs];
            const float2 y32 = make_float2((float)y.x, (float)y.y);          // the reference's complex64 output array
            if (CPLX) reinterpret_cast<float2*>(a.stage)[rowo + n] = y32;
            else reinterpret_cast<float*>(a.stage)[rowo + n] = convert_real<OUTK>(y32, a.out_kind);
        }
        __syncthreads();          // the work arrays are rewritten by the next scale
    }
}

}  // namespace spyfft
