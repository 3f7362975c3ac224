// K1/K2: fused detrend -> taper -> packed real FFT -> scale -> output conversion
// (-> taper mean) for segments of a (rows x ld) float32 trial matrix.
//
// Reference semantics: specest/mtmfft.py:16-129 + specest/compRoutines.py:169-189
// (and specest/stft.py:101-154 when one segment = one STFT frame).
//
// One workgroup = G channel pairs of one segment.  Two real channels are packed
// as (re, im) of one complex FFT and separated afterwards:
//   Xa[f] = (Z[f] + conj(Z[N-f]))/2,  Xb[f] = (Z[f] - conj(Z[N-f]))/(2i).
// The segment is read from HBM exactly once (kept in registers across tapers);
// the spectra never leave registers/LDS before the output conversion.
#pragma once
#include "fft_device.h"

// development probes (tools/fft_probe.hip): extra kernel attributes / ablation bits, off in the library
#ifndef SPYFFT_KATTR
#define SPYFFT_KATTR
#endif
#ifndef SPYFFT_ABL
#define SPYFFT_ABL 0
#endif

namespace spyfft {

struct MtmArgs {
    const float* data;          // (rows x ld) float32
    long long ld;
    const int* chan_idx;        // nchan column ids or nullptr
    const long long* seg_start; // per segment: first row
    const long long* seg_lo;    // rows outside [lo, hi) read as 0
    const long long* seg_hi;
    int nseg;
    int nsig;                   // samples per segment (window length)
    int nchan;
    int ntaper;
    const float* tapers;        // (ntaper x nsig) float32
    const float2* tw;           // exp(-2 pi i m / N)
    float scale;
    int detrend;                // -1 none, 0 constant, 1 linear
    int demean_taper;
    const int* fpos;            // rfft bin -> output position or -1; nullptr = identity
    int nfsel;
    int out_kind;               // enum spyhip_output
    void* out;                  // (nseg, Kout, nfsel, nchan)
    int npg;                    // pair groups per segment
    int S;                      // pair groups sharing 128-byte lines (XCD cluster)
    int ncl;                    // clusters per segment
    // Bluestein (mtmfft_blue_kernel.h): logical FFT length and chirp tables; tw then belongs to the length-M FFT
    int nfft;
    const float2* chirp;        // nfft entries exp(-i pi n^2 / nfft)
    const float2* bhat;         // M entries: FFT_M of the wrapped conjugate chirp, / M
    int blocked;                // complex keeptapers output in the channel-quad-blocked layout
                                // (nseg*ntaper, ceil(nchan/4), nfsel, 4) instead of (nseg, ntaper, nfsel, nchan)
    const float* means;         // (nseg x nchan) per-channel means in the reference's summation order
                                // (seq_mean_kernel) used for detrend == 0, or nullptr: float64 block sums
    int seg_f64;                // the segments are FLOAT64 arrays in the reference (zero-extended / padded sliding windows,
                                // stft.py:101-117): the float64 kernels then subtract the trend in float64 as well
    unsigned* absmax;           // nullptr, or nchan bit patterns of non-negative floats raised to a bound of |re|, |im| of every
                                // complex value written for the channel (spyhip_fft_plan_set_absmax: the range K4h scales by)
    float wnorm;                // max over the tapers of || w * scale ||_2 (the bound is wnorm * ||detrended segment||_2)
};

// Per-channel mean of a segment exactly as the reference takes it.  scipy.signal.detrend(type="constant") on the
// float32 (time x channel) trial (specest/compRoutines.py:169-170, connectivity/ST_compRoutines.py:405-409) is
// `data - np.mean(data, axis=0)`: NumPy reduces the slow axis of a C-ordered array row by row, i.e. ONE float32
// accumulator per channel that takes the samples in time order, then one float32 division by the sample count.
// That rounding sequence cannot be re-associated, and for channels with an offset its error (~1e-6 of the offset)
// is what the bins next to DC are made of - so it is reproduced literally: one thread per (segment, channel), a
// serial chain of v_add_f32 over the rows (loads batched 64 rows ahead; lanes = adjacent channels: coalesced).
// Rows outside [seg_lo, seg_hi) count as +0 and leave the sum as it is.
//
// ONE channel is the exception: the reference's trial is then an (nSamples, 1) array, contiguous along the axis that is
// reduced, and NumPy sums such a run PAIRWISE (pairwise_sum_FLOAT: eight running sums over blocks of at most 128
// values, halves of longer runs recursively) - np_pairwise_rows below follows that order.
__device__ inline float np_pairwise_rows(const float* p, long long ld, int i0, int n, int rlo, int rhi) {
    auto at = [&](int i) { return (i >= rlo && i < rhi) ? p[(long long)i * ld] : 0.f; };
    if (n < 8) {
        float res = 0.f;
        for (int i = 0; i < n; ++i) res = __fadd_rn(res, at(i0 + i));
        return res;
    }
    if (n <= 128) {
        float r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = at(i0 + j);
        int i = 8;
        for (; i < n - (n % 8); i += 8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) r[j] = __fadd_rn(r[j], at(i0 + i + j));
        }
        float res = __fadd_rn(__fadd_rn(__fadd_rn(r[0], r[1]), __fadd_rn(r[2], r[3])),
                              __fadd_rn(__fadd_rn(r[4], r[5]), __fadd_rn(r[6], r[7])));
        for (; i < n; ++i) res = __fadd_rn(res, at(i0 + i));
        return res;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    return __fadd_rn(np_pairwise_rows(p, ld, i0, n2, rlo, rhi), np_pairwise_rows(p, ld, i0 + n2, n - n2, rlo, rhi));
}

static __global__ void __launch_bounds__(256) seq_mean_kernel(MtmArgs a, float* means) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;      // (up to 256 adjacent channels per workgroup: whole 1-KiB rows)
    const int b = blockIdx.y;
    if (c >= a.nchan) return;
    const long long col = a.chan_idx ? a.chan_idx[c] : c;
    const long long start = a.seg_start[b];
    const long long rl = a.seg_lo[b] - start, rh = a.seg_hi[b] - start;
    const int rlo = (int)(rl < 0 ? 0 : (rl > a.nsig ? a.nsig : rl));
    const int rhi = (int)(rh < 0 ? 0 : (rh > a.nsig ? a.nsig : rh));
    if (a.nchan == 1) {
        means[b] = __fdiv_rn(np_pairwise_rows(a.data + start * a.ld + col, a.ld, 0, a.nsig, rlo, rhi), (float)a.nsig);
        return;
    }
    const float* p = a.data + (start + rlo) * a.ld + col;
    float s = 0.f;
    int n = rlo;
    // 64 rows in flight per thread: with few (segment, channel) threads - a handful of long trials - the chain waits on
    // memory latency, not on the 64 dependent additions
    for (; n + 64 <= rhi; n += 64) {
        float t[64];
#pragma unroll
        for (int e = 0; e < 64; ++e) t[e] = p[(long long)e * a.ld];
#pragma unroll
        for (int e = 0; e < 64; ++e) s = __fadd_rn(s, t[e]);
        p += 64 * a.ld;
    }
    for (; n + 16 <= rhi; n += 16) {
        float t[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) t[e] = p[(long long)e * a.ld];
#pragma unroll
        for (int e = 0; e < 16; ++e) s = __fadd_rn(s, t[e]);
        p += 16 * a.ld;
    }
    for (; n < rhi; ++n) {
        s = __fadd_rn(s, *p);
        p += a.ld;
    }
    means[(size_t)b * a.nchan + c] = __fdiv_rn(s, (float)a.nsig);
}

// Range of the spectra a plan is about to write (spyhip_fft_plan_set_absmax) for the kernel families that do not form it
// from the samples they hold anyway: every bin of channel c obeys |X_c(f)| <= ||w scale||_2 ||x_c||_2 (Cauchy-Schwarz), and
// removing a mean or a line (orthogonal projections) or a taper-wise mean only shrinks the 2-norm - so the norm of the
// samples minus their mean (detrend >= 0; float64 sums) or of the samples as they are bounds all tapers and all bins.
// One thread per (segment, channel), adjacent lanes = adjacent channels (coalesced), 16 rows in flight.
static __global__ void __launch_bounds__(256) seg_range_kernel(MtmArgs a) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (c >= a.nchan) return;
    const long long col = a.chan_idx ? a.chan_idx[c] : c;
    const long long start = a.seg_start[b];
    const long long rl = a.seg_lo[b] - start, rh = a.seg_hi[b] - start;
    const int rlo = (int)(rl < 0 ? 0 : (rl > a.nsig ? a.nsig : rl));
    const int rhi = (int)(rh < 0 ? 0 : (rh > a.nsig ? a.nsig : rh));
    const float* p = a.data + (start + rlo) * a.ld + col;
    double s1 = 0.0, s2 = 0.0;
    int n = rlo;
    for (; n + 16 <= rhi; n += 16) {
        float t[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) t[e] = p[(long long)e * a.ld];
#pragma unroll
        for (int e = 0; e < 16; ++e) { s1 += (double)t[e]; s2 += (double)t[e] * (double)t[e]; }
        p += 16 * a.ld;
    }
    for (; n < rhi; ++n) {
        const double v = (double)*p;
        s1 += v;
        s2 += v * v;
        p += a.ld;
    }
    // (the mean is taken over the nsig samples of the window, zeros outside the trial included: compRoutines.py:169-172)
    double q = a.detrend >= 0 ? s2 - s1 * s1 / (double)a.nsig : s2;
    q = q > 0.0 ? q : 0.0;
    // cancellation in s2 - s1^2 / n leaves ~1e-16 s2 of doubt: part of the bound
    const float bound = (float)((sqrt(q) + 1e-7 * sqrt(s2)) * (double)a.wnorm * 1.001);
    atomicMax(a.absmax + c, __float_as_uint(bound));
}

// output conversions of const_def.py:25-37; `kind` is wave-uniform.  Kept out of
// line so that the compiler branches on `kind` instead of evaluating sqrt and
// atan2 for every bin and selecting afterwards.
__device__ __attribute__((noinline)) float convert_real_slow(float2 x, int kind) {
    switch (kind) {
        case SPYHIP_OUT_ABS: return sqrtf(x.x * x.x + x.y * x.y);
        case SPYHIP_OUT_REAL: return x.x;
        case SPYHIP_OUT_IMAG: return x.y;
        case SPYHIP_OUT_ANGLE: return atan2f(x.y, x.x);
        case SPYHIP_OUT_ABSREAL: return fabsf(x.x);
        case SPYHIP_OUT_ABSIMAG: return fabsf(x.y);
        default: return x.x * x.x + x.y * x.y;
    }
}
// OUTK: 0 = power (the hot path, inlined), 1 = any other real conversion, 2 = complex
template <int OUTK>
__device__ __forceinline__ float convert_real(float2 x, int kind) {
    if (OUTK == 0) return x.x * x.x + x.y * x.y;
    return convert_real_slow(x, kind);
}

// OUTK: see convert_real; MEAN: average over tapers (keeptapers=False)
//
// Register discipline (the kernel must fit 128 VGPRs at 1024 threads):
//  - addressing = wave-uniform 64-bit base + 32-bit per-lane byte offset (segment
//    base, taper row, twiddle table, output slab) -> saddr-form global accesses;
//  - LDS addresses = one lane base + compile-time offsets (fft_device.h);
//  - inside the taper loop the lane index goes through opaque() so the loop-
//    invariant address arithmetic is recomputed instead of hoisted into VGPRs;
//  - loads are branch-free (clamped index + select): the 16 row loads of a
//    thread are all in flight together.
template <int LOG2N, int G, int OUTK, bool MEAN>
__global__ void __launch_bounds__((Cfg<LOG2N, G>::NTHREADS)) SPYFFT_KATTR mtmfft_pow2_kernel(MtmArgs a) {
    using C = Cfg<LOG2N, G>;
    constexpr bool CPLX = (OUTK == 2);
    constexpr int N = C::N, T = C::T;
    SPY_DYN_SMEM(float2, lds);

    const int tid = threadIdx.x;
    const int h = tid % G, j0 = tid / G;

    // XCD-aware block -> (segment, pair group): the S workgroups that share
    // 128-byte lines of the (time x channel) rows get ids congruent mod 8
    // (same XCD / L2) and adjacent in dispatch order.
    const long long id = blockIdx.x;
    const int xcd = (int)(id & 7);
    const long long y = id >> 3;
    // each XCD walks a contiguous run of clusters (see mtmfft2_kernel.h): the rows of a spectrum meet in one L2
    const long long nclt = (long long)a.nseg * a.ncl, chunk = (nclt + 7) >> 3;
    const long long cidx = (long long)xcd * chunk + y / a.S;
    const int q = (int)(y % a.S);
    if (cidx >= nclt) return;
    const int b = (int)(cidx / a.ncl);
    const int pg = (int)(cidx % a.ncl) * a.S + q;
    if (pg >= a.npg) return;

    const int c0 = 2 * (pg * G + h), c1 = c0 + 1;
    const bool has0 = c0 < a.nchan, has1 = c1 < a.nchan;
    const unsigned col0 = has0 ? (unsigned)(a.chan_idx ? a.chan_idx[c0] : c0) : 0u;
    const unsigned col1 = has1 ? (unsigned)(a.chan_idx ? a.chan_idx[c1] : c1) : col0;
    const long long start = a.seg_start[b];
    // valid sample range [rlo, rhi) relative to the segment start
    const long long rl = a.seg_lo[b] - start, rh = a.seg_hi[b] - start;
    const int rlo = (int)(rl < 0 ? 0 : (rl > a.nsig ? a.nsig : rl));
    const int rhi = (int)(rh < 0 ? 0 : (rh > a.nsig ? a.nsig : rh));
    const unsigned rowb = (unsigned)a.ld * 4u;        // bytes per row
    const float* seg = a.data + start * a.ld;         // wave-uniform; only rows in [rlo, rhi) are dereferenced

    // ---- load the segment once: x[e] = sample n = j + T*e of both channels
    float x0[16], x1[16];
    if (rhi > rlo) {
        const bool vec2 = (a.chan_idx == nullptr) && has1 && ((a.ld & 1) == 0) &&
                          ((reinterpret_cast<size_t>(a.data) & 7) == 0);
        if (vec2) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = j0 + T * e;
                const int nc = min(max(n, rlo), rhi - 1);
                const float2 t = ldg<float2>(seg, (unsigned)nc * rowb + col0 * 4u);
                x0[e] = (n == nc) ? t.x : 0.f;
                x1[e] = (n == nc) ? t.y : 0.f;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = j0 + T * e;
                const int nc = min(max(n, rlo), rhi - 1);
                const float u0 = ldg<float>(seg, (unsigned)nc * rowb + col0 * 4u);
                const float u1 = ldg<float>(seg, (unsigned)nc * rowb + col1 * 4u);
                x0[e] = (n == nc && has0) ? u0 : 0.f;
                x1[e] = (n == nc && has1) ? u1 : 0.f;
            }
        }
    } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) x0[e] = x1[e] = 0.f;
    }

    // ---- polynomial removal over the nsig samples (float64 sums, branch-free; constant: the reference-order means)
    if (a.detrend == 0 && a.means) {
        const float* mp = a.means + (size_t)b * a.nchan;
        const float f0 = has0 ? mp[c0] : 0.f, f1 = has1 ? mp[c1] : 0.f;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const bool in = j0 + T * e < a.nsig;
            x0[e] -= in ? f0 : 0.f;
            x1[e] -= in ? f1 : 0.f;
        }
    } else if (a.detrend >= 0) {
        const float mid = 0.5f * (float)(a.nsig - 1);
        double s[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int n = j0 + T * e;
            const float m = (n < a.nsig) ? 1.f : 0.f;
            s[0] += (double)(m * x0[e]);
            s[1] += (double)(m * x1[e]);
            if (a.detrend == 1) {
                const double dn = (double)(m * ((float)n - mid));   // exact: half-integers < 2^23
                s[2] += dn * x0[e];
                s[3] += dn * x1[e];
            }
        }
        block_sum4<LOG2N, G>(s, reinterpret_cast<double*>(lds), tid, h);
        const double inv = 1.0 / a.nsig;
        const double m0 = s[0] * inv, m1 = s[1] * inv;
        if (a.detrend == 1 && a.nsig > 1) {
            const double den = 12.0 / ((double)a.nsig * ((double)a.nsig * a.nsig - 1.0));
            const double b0 = s[2] * den, b1 = s[3] * den;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = j0 + T * e;
                const double dn = (double)((float)n - mid);
                const float t0 = (float)(m0 + b0 * dn), t1 = (float)(m1 + b1 * dn);
                x0[e] -= (n < a.nsig) ? t0 : 0.f;
                x1[e] -= (n < a.nsig) ? t1 : 0.f;
            }
        } else {
            const float f0 = (float)m0, f1 = (float)m1;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const bool in = j0 + T * e < a.nsig;
                x0[e] -= in ? f0 : 0.f;
                x1[e] -= in ? f1 : 0.f;
            }
        }
    }

    // accumulators for the taper mean (bins e<8 plus the Nyquist bin on j == 0)
    float2 acc0[MEAN ? 9 : 1], acc1[MEAN ? 9 : 1];
    if (MEAN) {
#pragma unroll
        for (int e = 0; e < 9; ++e) acc0[e] = acc1[e] = make_float2(0.f, 0.f);
    }
    const int kout = MEAN ? 1 : a.ntaper;
    const float hs = 0.5f * a.scale;
    const unsigned nsig_m1 = (unsigned)(a.nsig - 1);
    constexpr unsigned OSZ = CPLX ? 8u : 4u;   // bytes per output element
    // complex outputs of a channel pair are 16 contiguous bytes: aligned when nchan is even
    const bool pair16 = ((a.nchan & 1) == 0) && ((reinterpret_cast<size_t>(a.out) & 15) == 0);

    for (int k = 0; k < a.ntaper; ++k) {
        const int j = opaque(j0);
        float2 v[16];
        const float* w = a.tapers + (size_t)k * a.nsig;   // wave-uniform
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const unsigned n = (unsigned)(j + T * e);
            const float wl = (SPYFFT_ABL & 1) ? a.scale : ldg<float>(w, min(n, nsig_m1) * 4u);
            const float wn = (n <= nsig_m1) ? wl : 0.f;
            v[e] = make_float2(wn * x0[e], wn * x1[e]);
        }
        if (a.demean_taper) {
            double s[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                s[0] += v[e].x;
                s[1] += v[e].y;
            }
            block_sum4<LOG2N, G>(s, reinterpret_cast<double*>(lds), tid, h);
            const float m0 = (float)(s[0] / a.nsig), m1 = (float)(s[1] / a.nsig);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const bool in = j + T * e < a.nsig;
                v[e].x -= in ? m0 : 0.f;
                v[e].y -= in ? m1 : 0.f;
            }
        }

        fft_forward<LOG2N, G>(v, lds, j, h, a.tw);

        // ---- separate the two real channels: partner bin N-f lives in the upper half
        {
            float2* const wr = lds + C::rbase(j, h);
#pragma unroll
            for (int e = 8; e < 16; ++e) wr[e * C::ESTRIDE] = v[e];
        }
        __syncthreads();
        // output slab of (segment b, taper k): wave-uniform base, 32-bit lane offsets
        char* const slab = reinterpret_cast<char*>(a.out) +
                           ((size_t)b * kout + (MEAN ? 0 : k)) * (size_t)a.nfsel * (size_t)a.nchan * OSZ;
        // partner of f = j + T*e is N - f: idx(N - j, h) - e*ESTRIDE (index N = spare slot, unused value)
        const float2* const pr = lds + C::idx(N - j, h) - 7 * C::ESTRIDE;
#pragma unroll
        for (int e = 0; e < 9; ++e) {
            float2 xa, xb;
            int f;
            if (e < 8) {
                f = j + T * e;
                const float2 z = v[e];
                const float2 zl = pr[(7 - e) * C::ESTRIDE];
                const float2 zp = (f == 0) ? z : zl;
                xa = make_float2(hs * (z.x + zp.x), hs * (z.y - zp.y));
                xb = make_float2(hs * (z.y + zp.y), hs * (zp.x - z.x));
            } else {
                if (j != 0) break;
                f = N / 2;
                xa = make_float2(a.scale * v[8].x, 0.f);
                xb = make_float2(a.scale * v[8].y, 0.f);
            }
            if (MEAN) {
                if (CPLX) {
                    acc0[e] = cadd(acc0[e], xa);
                    acc1[e] = cadd(acc1[e], xb);
                } else {
                    acc0[e].x += convert_real<OUTK>(xa, a.out_kind);
                    acc1[e].x += convert_real<OUTK>(xb, a.out_kind);
                }
            } else {
                const int fi = a.fpos ? ldg<int>(a.fpos, (unsigned)f * 4u) : f;
                if (fi >= 0 && !((SPYFFT_ABL & 4) && xa.x != 12345.f)) {
                    const unsigned o = ((unsigned)fi * (unsigned)a.nchan + (unsigned)c0) * OSZ;
                    if (CPLX) {
                        if (has1 && pair16) {
                            stg<float4>(slab, o, make_float4(xa.x, xa.y, xb.x, xb.y));   // one 16-byte store per bin
                        } else {
                            if (has0) stg<float2>(slab, o, xa);
                            if (has1) stg<float2>(slab, o + OSZ, xb);
                        }
                    } else {
                        if (has0) stg<float>(slab, o, convert_real<OUTK>(xa, a.out_kind));
                        if (has1) stg<float>(slab, o + OSZ, convert_real<OUTK>(xb, a.out_kind));
                    }
                }
            }
        }
        __syncthreads();  // LDS is reused by the next taper
    }

    if (MEAN) {
        const float kk = (float)a.ntaper;
        char* const slab = reinterpret_cast<char*>(a.out) + (size_t)b * (size_t)a.nfsel * (size_t)a.nchan * OSZ;
#pragma unroll
        for (int e = 0; e < 9; ++e) {
            if (e == 8 && j0 != 0) break;
            const int f = (e < 8) ? j0 + T * e : N / 2;
            const int fi = a.fpos ? ldg<int>(a.fpos, (unsigned)f * 4u) : f;
            if (fi < 0) continue;
            const unsigned o = ((unsigned)fi * (unsigned)a.nchan + (unsigned)c0) * OSZ;
            if (CPLX) {
                if (has0) stg<float2>(slab, o, make_float2(acc0[e].x / kk, acc0[e].y / kk));
                if (has1) stg<float2>(slab, o + OSZ, make_float2(acc1[e].x / kk, acc1[e].y / kk));
            } else {
                if (has0) stg<float>(slab, o, acc0[e].x / kk);
                if (has1) stg<float>(slab, o + OSZ, acc1[e].x / kk);
            }
        }
    }
}

}  // namespace spyfft
