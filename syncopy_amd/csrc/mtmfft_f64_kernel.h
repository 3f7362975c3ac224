// K1 at the reference's own precision (optional: spyhip_fft_plan_set_precision(plan, 1); power-of-two nfft 256 ... 4096):
// the taper product and the FFT run in float64 exactly where specest/mtmfft.py:96-127 runs them in float64
// (`win *= data_arr` on a float64 window, np.fft.rfft of float64 data), the spectrum is rounded to complex64 where the
// reference stores it (`ftr[taperIdx] = ...` into a complex64 array, :104,117) and the normalisation factor multiplies
// in float32 (:119-127, _norm_spec.py:22).  The default kernels transform in float32 (error ~1e-7 of a channel's
// largest bin, profiles/r2_fft_precision.txt): inside the parity criterion, but bins more than 40 dB below the peak
// miss a PURE rtol of 1e-5.  This kernel is for users who need every bin to 1e-5 of itself; it costs ~5x the time.
//
// One workgroup of T = nfft/16 threads = one channel PAIR of one segment: (c0, c1) are the real and imaginary part of
// one complex128 transform (spywil::p_fft, the radix-16 register/LDS FFT of the Wilson plus operator), separated
// afterwards: X(c0)[f] = (Z[f] + conj(Z[N-f]))/2, X(c1)[f] = (Z[f] - conj(Z[N-f]))/(2i).
#pragma once
#include "cd_math.h"
#include "f64_stockham.h"
#include "wilson_plus_kernel.h"
#include "mtmfft_kernel.h"

namespace spyfft {

struct F64Args {
    MtmArgs m;
    const double* tapers64;      // (ntaper x nsig) float64: the reference's windows, not rounded to float32
    const double2* tw64;         // exp(-2 pi i m / nfft)
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
template <int LOG2N, int OUTK, bool MEAN>
__global__ void __launch_bounds__((spywil::PCfg<LOG2N>::T)) mtmfft_f64_kernel(F64Args fa) {
    using C = spywil::PCfg<LOG2N>;
    using spywil::cd;
    constexpr bool CPLX = (OUTK == 2);
    constexpr int N = C::L, T = C::T;
    const MtmArgs& a = fa.m;
    SPY_DYN_SMEM(cd, lds);
    const int j = threadIdx.x;
    const int npair = (a.nchan + 1) / 2;
    const int b = (int)(blockIdx.x / (unsigned)npair), p = (int)(blockIdx.x % (unsigned)npair);
    const int c0 = 2 * p;
    const bool has1 = c0 + 1 < a.nchan;
    const unsigned col0 = (unsigned)(a.chan_idx ? a.chan_idx[c0] : c0);
    const unsigned col1 = has1 ? (unsigned)(a.chan_idx ? a.chan_idx[c0 + 1] : c0 + 1) : 0u;
    const long long start = a.seg_start[b];
    const long long rl = a.seg_lo[b] - start, rh = a.seg_hi[b] - start;
    const int rlo = (int)(rl < 0 ? 0 : (rl > a.nsig ? a.nsig : rl));
    const int rhi = (int)(rh < 0 ? 0 : (rh > a.nsig ? a.nsig : rh));
    const float* seg = a.data + start * a.ld;

    // ---- the segment (float32, as the reference holds it): x0 = channel c0, x1 = channel c1
    float x0[16], x1[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int n = j + T * e;
        const bool ok = (n >= rlo) && (n < rhi);
        x0[e] = ok ? seg[(size_t)n * a.ld + col0] : 0.f;
        x1[e] = (ok && has1) ? seg[(size_t)n * a.ld + col1] : 0.f;
    }
    // ---- polynomial removal in float32 (scipy.signal.detrend on the float32 trial, compRoutines.py:169-172)
    bool f64t = false;
    double t0c = 0.0, t1c = 0.0, t0s = 0.0, t1s = 0.0;       // float64 trend (constant, slope about the centre)
    const float midc = 0.5f * (float)(a.nsig - 1);
    if (a.detrend == 0 && a.means) {
        const float m0 = a.means[(size_t)b * a.nchan + c0], m1 = has1 ? a.means[(size_t)b * a.nchan + c0 + 1] : 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const bool in = j + T * e < a.nsig;
            x0[e] -= in ? m0 : 0.f;
            x1[e] -= in ? m1 : 0.f;
        }
    } else if (a.detrend >= 0) {
        const float mid = 0.5f * (float)(a.nsig - 1);
        double s[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int n = j + T * e;
            if (n < a.nsig) {
                s[0] += (double)x0[e];
                s[1] += (double)x1[e];
                if (a.detrend == 1) {
                    const double dn = (double)((float)n - mid);
                    s[2] += dn * x0[e];
                    s[3] += dn * x1[e];
                }
            }
        }
        f64_block_sum<4, T>(s, reinterpret_cast<double*>(lds), j);
        const double inv = 1.0 / a.nsig;
        const double den = (a.detrend == 1 && a.nsig > 1) ? 12.0 / ((double)a.nsig * ((double)a.nsig * a.nsig - 1.0)) : 0.0;
        if (a.seg_f64) {
            // float64 segments in the reference: the trend is subtracted in float64 (kept apart, applied with the taper)
            f64t = true;
            t0c = s[0] * inv; t1c = s[1] * inv; t0s = s[2] * den; t1s = s[3] * den;
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = j + T * e;
                if (n < a.nsig) {
                    const double dn = (double)((float)n - mid);
                    x0[e] -= (float)(s[0] * inv + s[2] * den * dn);
                    x1[e] -= (float)(s[1] * inv + s[3] * den * dn);
                }
            }
        }
    }

    float macc0[MEAN ? 9 : 1], macc1[MEAN ? 9 : 1], mim0[(MEAN && CPLX) ? 9 : 1], mim1[(MEAN && CPLX) ? 9 : 1];
    if (MEAN) {
#pragma unroll
        for (int e = 0; e < 9; ++e) {
            macc0[e] = macc1[e] = 0.f;
            if (CPLX) mim0[e] = mim1[e] = 0.f;
        }
    }
    const int kout = MEAN ? 1 : a.ntaper;
    constexpr unsigned OSZ = CPLX ? 8u : 4u;

    for (int k = 0; k < a.ntaper; ++k) {
        const double* w = fa.tapers64 + (size_t)k * a.nsig;
        cd v[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int n = j + T * e;
            const double wn = n < a.nsig ? w[n] : 0.0;
            double d0 = (double)x0[e], d1 = (double)x1[e];
            if (f64t) {
                const double dn = (double)((float)n - midc);
                d0 -= t0c + t0s * dn;
                d1 -= t1c + t1s * dn;
            }
            v[e] = make_double2(wn * d0, wn * d1);                               // win *= data_arr (float64)
        }
        if (a.demean_taper) {                                                  // win -= win.mean(axis=0) (float64)
            double s[2] = {0.0, 0.0};
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                s[0] += v[e].x;
                s[1] += v[e].y;
            }
            __syncthreads();
            f64_block_sum<2, T>(s, reinterpret_cast<double*>(lds), j);
            const double m0 = s[0] / a.nsig, m1 = s[1] / a.nsig;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                if (j + T * e < a.nsig) {
                    v[e].x -= m0;
                    v[e].y -= m1;
                }
            }
        }
        spywil::p_fft<LOG2N>(v, lds, j, fa.tw64);
        // ---- separation: partner bin N - f lives in the upper slots
        __syncthreads();
#pragma unroll
        for (int e = 8; e < 16; ++e) lds[C::idx(j + T * e)] = v[e];
        __syncthreads();
        char* const slab = reinterpret_cast<char*>(a.out) +
                           ((size_t)b * kout + (MEAN ? 0 : k)) * (size_t)a.nfsel * (size_t)a.nchan * OSZ;
#pragma unroll
        for (int e = 0; e < 9; ++e) {
            int f;
            cd X0, X1;
            if (e < 8) {
                f = j + T * e;
                const cd z = v[e];
                const cd zp = f == 0 ? z : lds[C::idx(N - f)];
                X0 = make_double2(0.5 * (z.x + zp.x), 0.5 * (z.y - zp.y));
                X1 = make_double2(0.5 * (z.y + zp.y), 0.5 * (zp.x - z.x));
            } else {
                if (j != 0) break;
                f = N / 2;
                X0 = make_double2(v[8].x, 0.0);
                X1 = make_double2(v[8].y, 0.0);
            }
            // complex64 storage, then the float32 normalisation factor (mtmfft.py:104,117-127)
            const float2 s0 = make_float2(__fmul_rn((float)X0.x, a.scale), __fmul_rn((float)X0.y, a.scale));
            const float2 s1 = make_float2(__fmul_rn((float)X1.x, a.scale), __fmul_rn((float)X1.y, a.scale));
            if (MEAN) {
                if (CPLX) {
                    macc0[e] += s0.x; mim0[e] += s0.y;
                    macc1[e] += s1.x; mim1[e] += s1.y;
                } else {
                    macc0[e] += convert_real<OUTK>(s0, a.out_kind);
                    macc1[e] += convert_real<OUTK>(s1, a.out_kind);
                }
                continue;
            }
            const int fi = a.fpos ? a.fpos[f] : f;
            if (fi < 0) continue;
            const size_t o = ((size_t)fi * a.nchan + c0) * OSZ;
            if (CPLX) {
                *reinterpret_cast<float2*>(slab + o) = s0;
                if (has1) *reinterpret_cast<float2*>(slab + o + 8) = s1;
            } else {
                *reinterpret_cast<float*>(slab + o) = convert_real<OUTK>(s0, a.out_kind);
                if (has1) *reinterpret_cast<float*>(slab + o + 4) = convert_real<OUTK>(s1, a.out_kind);
            }
        }
        __syncthreads();          // the partner reads are done before the next taper's FFT writes the buffer
    }

    if (MEAN) {
        char* const slab = reinterpret_cast<char*>(a.out) + (size_t)b * (size_t)a.nfsel * (size_t)a.nchan * OSZ;
        const float nt = (float)a.ntaper;
#pragma unroll
        for (int e = 0; e < 9; ++e) {
            if (e == 8 && j != 0) break;
            const int f = e < 8 ? j + T * e : N / 2;
            const int fi = a.fpos ? a.fpos[f] : f;
            if (fi < 0) continue;
            const size_t o = ((size_t)fi * a.nchan + c0) * OSZ;
            if (CPLX) {
                *reinterpret_cast<float2*>(slab + o) = make_float2(macc0[e] / nt, mim0[e] / nt);
                if (has1) *reinterpret_cast<float2*>(slab + o + 8) = make_float2(macc1[e] / nt, mim1[e] / nt);
            } else {
                *reinterpret_cast<float*>(slab + o) = macc0[e] / nt;
                if (has1) *reinterpret_cast<float*>(slab + o + 4) = macc1[e] / nt;
            }
        }
    }
}

// The same transform for ANY length the radix-16 register kernel does not serve (2000, 5000, 16384, 3000 ...): one
// workgroup of 256 threads per (segment, channel pair), the complex128 sequence in two length-nfft work arrays in
// global memory (L2-resident while the workgroup owns them), generic Stockham passes (f64_stockham.h: radix 2 / 4
// butterflies, O(R^2) passes for the other prime factors).  Identical rounding points; the taper mean accumulates in
// the output slab in float32 in taper order and is divided once, as the register kernel's accumulators are.  Speed is
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
