// K1 at the reference's own precision for lengths WITHOUT a compile-time schedule (spyhip_fft_plan_set_precision; the
// scheduled lengths run mtmfft_dec64_kernel.h / mtmfft_declong64.h): the taper product and the FFT run in float64 exactly
// where specest/mtmfft.py:96-127 runs them in float64 (`win *= data_arr` on a float64 window, np.fft.rfft of float64
// data), the spectrum is rounded to complex64 where the reference stores it (`ftr[taperIdx] = ...` into a complex64
// array, :104,117) and the normalisation factor multiplies in float32 (:119-127, _norm_spec.py:22).
//
// One workgroup = one channel PAIR of one segment: (c0, c1) are the real and imaginary part of one complex128
// transform, separated afterwards: X(c0)[f] = (Z[f] + conj(Z[N-f]))/2, X(c1)[f] = (Z[f] - conj(Z[N-f]))/(2i).
#pragma once
#include "cd_math.h"
#include "f64_stockham.h"
#include "wilson_plus_kernel.h"
#include "mtmfft_kernel.h"
#include "mtmfft_f64_args.h"

namespace spyfft {

// sum of NS doubles over the workgroup (T threads), broadcast; scratch = LDS (free at that point); two barriers
template <int NS, int T>
__device__ __forceinline__ void f64_block_sum(double (&s)[NS], double* scratch, int tid) {
    constexpr int NWAVES = (T + 63) / 64;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const double o = __shfl_xor(s[i], off);
            s[i] += ((tid ^ off) < T) ? o : 0.0;
        }
    }
    const int lane = tid & 63, w = tid >> 6;
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NS; ++i) scratch[w * NS + i] = s[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        double tot = 0.0;
        for (int ww = 0; ww < NWAVES; ++ww) tot += scratch[ww * NS + i];
        s[i] = tot;
    }
    __syncthreads();
}

// OUTK: 0 = power, 1 = any other real conversion, 2 = complex; MEAN: average over tapers
// The reference-precision transform for ANY length without a compile-time schedule (mtmfft_dec64_kernel.h): one
// workgroup of 256 threads per (segment, channel pair), the complex128 sequence in two length-nfft work arrays in
// global memory (L2-resident while the workgroup owns them), generic Stockham passes (f64_stockham.h: radix 2 / 4
// butterflies, O(R^2) passes for the other prime factors).  Identical rounding points; the taper mean accumulates in
// the output slab in float32 in taper order and is divided once.  Speed is
// not the point of this kernel (10-30 x the float32 kernels): it exists so that precision="reference" is not limited to
// power-of-two lengths up to 4096.
template <int OUTK, bool MEAN>
__global__ void __launch_bounds__(256) mtmfft_f64_any_kernel(F64Args fa) {
    using spywil::cd;
    constexpr bool CPLX = (OUTK == 2);
    const MtmArgs& a = fa.m;
    __shared__ double red[32];
    const int L = fa.plan.L, tid = threadIdx.x;         // work-array length: nfft, or the Bluestein length M
    const int N = fa.blue_n ? fa.blue_n : L;            // transform length
    const long long wg = fa.wg0 + blockIdx.x;
    const int npair = (a.nchan + 1) / 2;
    const int b = (int)(wg / npair), p = (int)(wg % npair);
    const int c0 = 2 * p;
    const bool has1 = c0 + 1 < a.nchan;
    const unsigned col0 = (unsigned)(a.chan_idx ? a.chan_idx[c0] : c0);
    const unsigned col1 = has1 ? (unsigned)(a.chan_idx ? a.chan_idx[c0 + 1] : c0 + 1) : 0u;
    const long long start = a.seg_start[b];
    const long long rl = a.seg_lo[b] - start, rh = a.seg_hi[b] - start;
    const int rlo = (int)(rl < 0 ? 0 : (rl > a.nsig ? a.nsig : rl));
    const int rhi = (int)(rh < 0 ? 0 : (rh > a.nsig ? a.nsig : rh));
    const float* seg = a.data + start * a.ld;
    // the two work arrays: LDS while they fit (nfft <= 5120: launched with 32 nfft bytes of dynamic LDS and work =
    // nullptr), global memory beyond
    SPY_DYN_SMEM(cd, ldsbuf);
    cd* A = fa.work ? reinterpret_cast<cd*>(fa.work) + (size_t)blockIdx.x * 2 * (size_t)L : ldsbuf;
    cd* B = A + L;

    // ---- polynomial removal in float32, exactly as the register kernel above
    float m0 = 0.f, m1 = 0.f;
    double s[4] = {0.0, 0.0, 0.0, 0.0};
    const float mid = 0.5f * (float)(a.nsig - 1);
    const bool fit = !(a.detrend == 0 && a.means) && a.detrend >= 0;
    if (a.detrend == 0 && a.means) {
        m0 = a.means[(size_t)b * a.nchan + c0];
        m1 = has1 ? a.means[(size_t)b * a.nchan + c0 + 1] : 0.f;
    } else if (fit) {
        for (int n = tid; n < a.nsig; n += 256) {
            const bool ok = (n >= rlo) && (n < rhi);
            const float x0 = ok ? seg[(size_t)n * a.ld + col0] : 0.f;
            const float x1 = (ok && has1) ? seg[(size_t)n * a.ld + col1] : 0.f;
            s[0] += (double)x0;
            s[1] += (double)x1;
            if (a.detrend == 1) {
                const double dn = (double)((float)n - mid);
                s[2] += dn * x0;
                s[3] += dn * x1;
            }
        }
        f64_block_sum<4, 256>(s, red, tid);
    }
    const double inv = 1.0 / a.nsig;
    const double den = (a.detrend == 1 && a.nsig > 1) ? 12.0 / ((double)a.nsig * ((double)a.nsig * a.nsig - 1.0)) : 0.0;

    const int kout = MEAN ? 1 : a.ntaper;
    constexpr unsigned OSZ = CPLX ? 8u : 4u;
    const int nf = N / 2 + 1;
    for (int k = 0; k < a.ntaper; ++k) {
        const double* w = fa.tapers64 + (size_t)k * a.nsig;
        double ds[2] = {0.0, 0.0};
        for (int n = tid; n < L; n += 256) {
            cd v = make_double2(0.0, 0.0);
            if (n < a.nsig) {
                const bool ok = (n >= rlo) && (n < rhi);
                float x0 = ok ? seg[(size_t)n * a.ld + col0] : 0.f;
                float x1 = (ok && has1) ? seg[(size_t)n * a.ld + col1] : 0.f;
                double d0, d1;
                if (fit && a.seg_f64) {                                            // float64 segments: float64 trend
                    const double dn = (double)((float)n - mid);
                    d0 = (double)x0 - (s[0] * inv + s[2] * den * dn);
                    d1 = (double)x1 - (s[1] * inv + s[3] * den * dn);
                } else if (fit) {
                    const double dn = (double)((float)n - mid);
                    d0 = (double)(x0 - (float)(s[0] * inv + s[2] * den * dn));
                    d1 = (double)(x1 - (float)(s[1] * inv + s[3] * den * dn));
                } else {
                    d0 = (double)(x0 - m0);
                    d1 = (double)(x1 - m1);
                }
                v = make_double2(w[n] * d0, w[n] * d1);                           // win *= data_arr (float64)
                ds[0] += v.x;
                ds[1] += v.y;
            }
            A[n] = v;
        }
        if (a.demean_taper) {                                                      // win -= win.mean(axis=0) (float64)
            f64_block_sum<2, 256>(ds, red, tid);
            const double d0 = ds[0] / a.nsig, d1 = ds[1] / a.nsig;
            for (int n = tid; n < a.nsig; n += 256) A[n] = make_double2(A[n].x - d0, A[n].y - d1);
        }
        __syncthreads();
        cd *src = A, *dst = B;
        if (fa.blue_n) {
            // chirp-z: Z[k] = c[k] IFFT_M(FFT_M(z c) Bhat)[k], c[n] = exp(-i pi n^2 / nfft) (phases reduced exactly on the host)
            const cd* ch = reinterpret_cast<const cd*>(fa.chirp64);
            const cd* bh = reinterpret_cast<const cd*>(fa.bhat64);
            for (int n = tid; n < a.nsig; n += 256) src[n] = spywil::cmul(src[n], ch[n]);
            __syncthreads();
            for (int dir = 0; dir < 2; ++dir) {
                int Ns = 1;
                for (int q = 0; q < fa.plan.nfac; ++q) {
                    spywil::po_pass_any(src, dst, L, fa.plan.radix[q], Ns, reinterpret_cast<const cd*>(fa.tw64), dir ? +1 : -1, tid);
                    __syncthreads();
                    Ns *= fa.plan.radix[q];
                    cd* t = src; src = dst; dst = t;
                }
                if (dir == 0) {
                    for (int n = tid; n < L; n += 256) src[n] = spywil::cmul(src[n], bh[n]);
                } else {
                    for (int n = tid; n < N; n += 256) src[n] = spywil::cmul(src[n], ch[n]);
                }
                __syncthreads();
            }
        } else {
            int Ns = 1;
            for (int q = 0; q < fa.plan.nfac; ++q) {
                spywil::po_pass_any(src, dst, L, fa.plan.radix[q], Ns, reinterpret_cast<const cd*>(fa.tw64), -1, tid);
                __syncthreads();
                Ns *= fa.plan.radix[q];
                cd* t = src; src = dst; dst = t;
            }
        }
        char* const slab = reinterpret_cast<char*>(a.out) +
                           ((size_t)b * kout + (MEAN ? 0 : k)) * (size_t)a.nfsel * (size_t)a.nchan * OSZ;
        const float nt = (float)a.ntaper;
        for (int f = tid; f < nf; f += 256) {
            const cd z = src[f];
            const cd zp = f == 0 ? z : src[N - f];
            const cd X0 = make_double2(0.5 * (z.x + zp.x), 0.5 * (z.y - zp.y));
            const cd X1 = make_double2(0.5 * (z.y + zp.y), 0.5 * (zp.x - z.x));
            const float2 s0 = make_float2(__fmul_rn((float)X0.x, a.scale), __fmul_rn((float)X0.y, a.scale));
            const float2 s1 = make_float2(__fmul_rn((float)X1.x, a.scale), __fmul_rn((float)X1.y, a.scale));
            const int fi = a.fpos ? a.fpos[f] : f;
            if (fi < 0) continue;
            const size_t o = ((size_t)fi * a.nchan + c0) * OSZ;
            const bool first = !MEAN || k == 0, last = MEAN && k == a.ntaper - 1;
            if (CPLX) {
                float2* q0 = reinterpret_cast<float2*>(slab + o);
                float2 v0 = s0, v1 = s1;
                if (!first) { v0.x += q0[0].x; v0.y += q0[0].y; if (has1) { v1.x += q0[1].x; v1.y += q0[1].y; } }
                if (last) { v0.x /= nt; v0.y /= nt; v1.x /= nt; v1.y /= nt; }
                q0[0] = v0;
                if (has1) q0[1] = v1;
            } else {
                float* q0 = reinterpret_cast<float*>(slab + o);
                float v0 = convert_real<OUTK>(s0, a.out_kind), v1 = convert_real<OUTK>(s1, a.out_kind);
                if (!first) { v0 += q0[0]; if (has1) v1 += q0[1]; }
                if (last) { v0 /= nt; v1 /= nt; }
                q0[0] = v0;
                if (has1) q0[1] = v1;
            }
        }
        __syncthreads();          // the work arrays are rewritten by the next taper
    }
}

}  // namespace spyfft
