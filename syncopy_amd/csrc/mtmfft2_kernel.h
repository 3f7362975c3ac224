// K1/K2 (power-of-two lengths 256..8192): fused detrend -> taper -> packed real FFT -> scale ->
// output conversion (-> taper mean) for segments of a (rows x ld) float32 trial matrix.
//
// Reference semantics: specest/mtmfft.py:16-129 + specest/compRoutines.py:169-189 (and
// specest/stft.py:101-154 when one segment = one STFT frame).
//
// One thread serves FOUR adjacent real channels c0..c3 of one segment: channels (c0, c2) are packed
// as (re, im) of complex FFT "a", (c1, c3) of FFT "b", and a/b travel in the two halves of packed
// fp32 registers (fft2_device.h), i.e. r = samples of (c0, c1), i = samples of (c2, c3): exactly the
// halves of the 16-byte row read, so neither the load nor the butterflies need a shuffle.
// Separation afterwards:  X(c0,c1)[f] = (Z[f] + conj(Z[N-f]))/2,  X(c2,c3)[f] = (Z[f] - conj(Z[N-f]))/(2i).
// The segment is read from HBM exactly once (kept in registers across tapers); the spectra never
// leave registers/LDS before the output conversion.
// `a.tapers` of THIS kernel carry the factor scale/2 already (the plan uploads w * scale/2 rounded from float64):
// the separation X = (Z[f] +- conj Z[N-f]) * scale/2 then needs no multiplication at all (32 packed
// instructions per thread and taper).  When the window fills the transform (nsig == N, every c2/c3-type
// call) the taper weights are fetched without the zero-padding clamps (64 of ~950 vector instructions).
#pragma once
#include "fft2_device.h"
#include "mtmfft_kernel.h"

namespace spyfft {

// OUTK: 0 = power (inlined), 1 = any other real conversion, 2 = complex; MEAN: average over tapers
// HALF (2^14 samples, the only power of two beyond a quad's LDS): a thread set carries a channel PAIR, z[m] = x[2 m] +
// i x[2 m + 1] through the transform of N = nfft / 2, and the epilogue forms the bins f and N - f of the real transform
// from Z[f], Z[N - f] - exactly CfgD::HALF of mtmfft_dec_kernel.h, on this kernel's radix-16 engine
template <int LOG2N, int G, int OUTK, bool MEAN, bool HALF = false>
__global__ void __launch_bounds__((Cfg2<LOG2N, G>::NTHREADS)) SPYFFT_KATTR mtmfft_quad_kernel(MtmArgs a) {
    using C = Cfg2<LOG2N, G>;
    constexpr bool CPLX = (OUTK == 2);
    constexpr int N = C::N, T = C::T;
    constexpr int CW = HALF ? 2 : 4;              // channels of a thread set
    constexpr int SM = HALF ? 2 : 1;              // sample index of value e: SM (j + T e) (+ 1 for the imaginary part, HALF)
    SPY_DYN_SMEM(v2f, lds);
    v2f* const lre = lds;
    v2f* const lim = lds + C::PLANE;

    const int tid = threadIdx.x;
    const int h = tid % G, j0 = tid / G;
#ifdef SPYFFT_STAMPS
    int sn = 0;
#endif
    SPY_STAMP(sn);

    // XCD-aware block -> (segment, quad group): the S workgroups that share 128-byte lines of the
    // (time x channel) rows get ids congruent mod 8 (same XCD / L2) and adjacent in dispatch order.
    // Each XCD walks a contiguous run of clusters: consecutive segments (overlapping sliding-window frames share half
    // of their rows) then find those rows in the L2 they were just fetched into.
    // (32-bit arithmetic: the host launches at most 2^31 - 1 workgroups, so nseg * ncl * S fits)
    const unsigned id = blockIdx.x;
    const unsigned xcd = id & 7u, y = id >> 3;
    const unsigned nclt = (unsigned)a.nseg * (unsigned)a.ncl, chunk = (nclt + 7u) >> 3;
    const unsigned yS = y / (unsigned)a.S;
    const unsigned cidx = xcd * chunk + yS;
    const int q = (int)(y - yS * (unsigned)a.S);
    if (cidx >= nclt) return;
    const int b = (int)(cidx / (unsigned)a.ncl);
    const int pg = (int)(cidx - (unsigned)b * (unsigned)a.ncl) * a.S + q;
    if (pg >= a.npg) return;

    const int c0 = CW * (pg * G + h);
    bool has[4];
    unsigned col[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        has[i] = i < CW && c0 + i < a.nchan;
        col[i] = has[i] ? (unsigned)(a.chan_idx ? a.chan_idx[c0 + i] : c0 + i) : 0u;
    }
    const bool full = has[CW - 1];
    const long long start = a.seg_start[b];
    const long long rl = a.seg_lo[b] - start, rh = a.seg_hi[b] - start;
    const int rlo = (int)(rl < 0 ? 0 : (rl > a.nsig ? a.nsig : rl));
    const int rhi = (int)(rh < 0 ? 0 : (rh > a.nsig ? a.nsig : rh));
    const unsigned rowb = (unsigned)a.ld * 4u;        // bytes per row
    const float* seg = a.data + start * a.ld;         // wave-uniform; only rows in [rlo, rhi) are dereferenced

    // ---- load the segment once: x[e] = sample n = j + T*e; r = (c0, c1), i = (c2, c3)
    C2 x[16];
    if constexpr (HALF) {
        // sample pairs (2 m, 2 m + 1), m = j + T e: the even sample in .r, the odd one in .i, channels (c0, c1) in the halves
        if (a.xpair != nullptr) {
            // ... from the pair-major copy of the segment (pair_stage_kernel): 16 contiguous bytes = both samples
            const float2* xs = a.xpair + ((size_t)b * ((a.nchan + 1) >> 1) + (size_t)(c0 >> 1)) * (size_t)a.xstride;
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const long long n0 = 2LL * (j0 + T * e);
                float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
                if (n0 < a.xstride) t = *reinterpret_cast<const float4*>(xs + n0);
                x[e].r = v2f{t.x, t.y};
                x[e].i = v2f{t.z, t.w};
            }
        } else if (rhi > rlo) {
            const bool vec2 = (a.chan_idx == nullptr) && full && ((a.ld & 1) == 0) &&
                              ((reinterpret_cast<size_t>(a.data) & 7) == 0);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n0 = 2 * (j0 + T * e), n1 = n0 + 1;
                const int nc0 = min(max(n0, rlo), rhi - 1), nc1 = min(max(n1, rlo), rhi - 1);
                float u0[2], u1[2];
                if (vec2) {
                    const float2 t0 = ldg<float2>(seg, (unsigned)nc0 * rowb + col[0] * 4u);
                    const float2 t1 = ldg<float2>(seg, (unsigned)nc1 * rowb + col[0] * 4u);
                    u0[0] = t0.x; u0[1] = t0.y; u1[0] = t1.x; u1[1] = t1.y;
                } else {
#pragma unroll
                    for (int i = 0; i < 2; ++i) {
                        const float t0 = ldg<float>(seg, (unsigned)nc0 * rowb + col[i] * 4u);
                        const float t1 = ldg<float>(seg, (unsigned)nc1 * rowb + col[i] * 4u);
                        u0[i] = has[i] ? t0 : 0.f;
                        u1[i] = has[i] ? t1 : 0.f;
                    }
                }
                const bool ok0 = (n0 == nc0), ok1 = (n1 == nc1);
                x[e].r = v2f{ok0 ? u0[0] : 0.f, ok0 ? u0[1] : 0.f};
                x[e].i = v2f{ok1 ? u1[0] : 0.f, ok1 ? u1[1] : 0.f};
            }
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) x[e].r = x[e].i = splat(0.f);
        }
    } else if (rhi > rlo) {
        const bool vec4 = (a.chan_idx == nullptr) && full && ((a.ld & 3) == 0) &&
                          ((reinterpret_cast<size_t>(a.data) & 15) == 0);
        if (vec4 && rlo == 0 && rhi == N) {
            // the segment lies inside its trial and fills the transform (every frame but the first and last of a sliding
            // window, every unpadded trial): no clamping, no zero extension (~150 of ~1500 instructions per segment)
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const float4 t = ldg<float4>(seg, (unsigned)(j0 + T * e) * rowb + col[0] * 4u);
                x[e].r = v2f{t.x, t.y};
                x[e].i = v2f{t.z, t.w};
            }
        } else if (vec4) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = j0 + T * e;
                const int nc = min(max(n, rlo), rhi - 1);
                const float4 t = ldg<float4>(seg, (unsigned)nc * rowb + col[0] * 4u);
                const bool ok = (n == nc);
                x[e].r = v2f{ok ? t.x : 0.f, ok ? t.y : 0.f};
                x[e].i = v2f{ok ? t.z : 0.f, ok ? t.w : 0.f};
            }
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = j0 + T * e;
                const int nc = min(max(n, rlo), rhi - 1);
                float u[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float t = ldg<float>(seg, (unsigned)nc * rowb + col[i] * 4u);
                    u[i] = (n == nc && has[i]) ? t : 0.f;
                }
                x[e].r = v2f{u[0], u[1]};
                x[e].i = v2f{u[2], u[3]};
            }
        }
    } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) x[e].r = x[e].i = splat(0.f);
    }

    // ---- polynomial removal over the nsig samples (float64 sums, branch-free; constant: the reference-order means)
    if (a.detrend == 0 && a.means) {
        const float* mp = a.means + (size_t)b * a.nchan + c0;
        float f[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) f[i] = has[i] ? mp[i] : 0.f;
        const v2f mr = v2f{f[0], f[1]}, mi = HALF ? mr : v2f{f[2], f[3]};
        if (a.nsig == SM * N) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                x[e].r -= mr;
                x[e].i -= mi;
            }
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n0 = SM * (j0 + T * e);
                x[e].r -= n0 < a.nsig ? mr : splat(0.f);
                x[e].i -= n0 + (HALF ? 1 : 0) < a.nsig ? mi : splat(0.f);
            }
        }
    } else if (HALF && a.detrend >= 0) {
        // (HALF) the float64 sums of the two channels run over the even AND the odd samples
        const float mid = 0.5f * (float)(a.nsig - 1);
        double s[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int n0 = 2 * (j0 + T * e), n1 = n0 + 1;
            const float m0 = (n0 < a.nsig) ? 1.f : 0.f, m1 = (n1 < a.nsig) ? 1.f : 0.f;
            s[0] += (double)(m0 * x[e].r[0]);
            s[1] += (double)(m0 * x[e].r[1]);
            s[0] += (double)(m1 * x[e].i[0]);
            s[1] += (double)(m1 * x[e].i[1]);
            if (a.detrend == 1) {
                const double d0 = (double)(m0 * ((float)n0 - mid)), d1 = (double)(m1 * ((float)n1 - mid));
                s[4] += d0 * x[e].r[0];
                s[5] += d0 * x[e].r[1];
                s[4] += d1 * x[e].i[0];
                s[5] += d1 * x[e].i[1];
            }
        }
        block_sum<C::NTHREADS, G, 8>(s, reinterpret_cast<double*>(lds), tid, h);
        const double inv = 1.0 / a.nsig;
        const double den = (a.detrend == 1 && a.nsig > 1) ? 12.0 / ((double)a.nsig * ((double)a.nsig * a.nsig - 1.0)) : 0.0;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int n0 = 2 * (j0 + T * e), n1 = n0 + 1;
            const double d0 = (double)((float)n0 - mid), d1 = (double)((float)n1 - mid);
            float t0[2], t1[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                t0[i] = n0 < a.nsig ? (float)(s[i] * inv + s[4 + i] * den * d0) : 0.f;
                t1[i] = n1 < a.nsig ? (float)(s[i] * inv + s[4 + i] * den * d1) : 0.f;
            }
            x[e].r -= v2f{t0[0], t0[1]};
            x[e].i -= v2f{t1[0], t1[1]};
        }
    } else if (a.detrend >= 0) {
        const float mid = 0.5f * (float)(a.nsig - 1);
        double s[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int n = j0 + T * e;
            const float m = (n < a.nsig) ? 1.f : 0.f;
            const float u[4] = {m * x[e].r[0], m * x[e].r[1], m * x[e].i[0], m * x[e].i[1]};
#pragma unroll
            for (int i = 0; i < 4; ++i) s[i] += (double)u[i];
            if (a.detrend == 1) {
                const double dn = (double)(m * ((float)n - mid));   // exact: half-integers < 2^23
                s[4] += dn * x[e].r[0];
                s[5] += dn * x[e].r[1];
                s[6] += dn * x[e].i[0];
                s[7] += dn * x[e].i[1];
            }
        }
        block_sum<C::NTHREADS, G, 8>(s, reinterpret_cast<double*>(lds), tid, h);
        const double inv = 1.0 / a.nsig;
        if (a.detrend == 1 && a.nsig > 1) {
            const double den = 12.0 / ((double)a.nsig * ((double)a.nsig * a.nsig - 1.0));
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = j0 + T * e;
                const double dn = (double)((float)n - mid);
                const bool in = n < a.nsig;
                float t[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) t[i] = in ? (float)(s[i] * inv + s[4 + i] * den * dn) : 0.f;
                x[e].r -= v2f{t[0], t[1]};
                x[e].i -= v2f{t[2], t[3]};
            }
        } else {
            const v2f mr = v2f{(float)(s[0] * inv), (float)(s[1] * inv)};
            const v2f mi = v2f{(float)(s[2] * inv), (float)(s[3] * inv)};
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const bool in = j0 + T * e < a.nsig;
                x[e].r -= in ? mr : splat(0.f);
                x[e].i -= in ? mi : splat(0.f);
            }
        }
    }

#ifdef SPYFFT_STAMPS
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
    SPY_STAMP(sn);
    // accumulators for the taper mean (bins e<8 plus the Nyquist bin on j == 0):
    // real outputs: ma.r = sum conv(X(c0,c1)), ma.i = sum conv(X(c2,c3)); complex: ma = X(c0,c1), mb = X(c2,c3)
    C2 ma[MEAN ? 9 : 1], mb[(MEAN && CPLX) ? 9 : 1];
    if (MEAN) {
#pragma unroll
        for (int e = 0; e < 9; ++e) {
            ma[e].r = ma[e].i = splat(0.f);
            if (CPLX) mb[e].r = mb[e].i = splat(0.f);
        }
    }
    const int kout = MEAN ? 1 : a.ntaper;
    const unsigned nsig_m1 = (unsigned)(a.nsig - 1);
    const bool wfull = (a.nsig == SM * N);     // uniform: no zero padding behind the window
    constexpr unsigned OSZ = CPLX ? 8u : 4u;   // bytes per output element
    // straight-line epilogue: all four channels present, every bin kept, 16-byte aligned rows
    const bool fast = full && (a.fpos == nullptr) && ((reinterpret_cast<size_t>(a.out) & 15) == 0) &&
                      ((a.nchan & ((CPLX || HALF) ? 1 : 3)) == 0);

    // ---- range of the spectra for K4h (spyhip_fft_plan_set_absmax; complex all-taper output only): every bin of channel c
    // obeys |X_c(f)| = |sum_n w[n] x_c[n] e^(...)| <= ||w||_2 ||x_c||_2, so ONE sum of squares of the detrended samples
    // per segment (they are in registers) bounds all tapers and all bins - 32 packed multiply-adds per thread and segment
    // instead of a maximum over every value written.  a.wnorm = max over the tapers of ||w scale||_2 (a per-taper mean
    // removed after the product, demean_taper, only shrinks the norm).
    if (CPLX && !MEAN && !HALF && a.absmax != nullptr) {          // uniform (HALF: the host runs seg_range_kernel)
        v2f qr = splat(0.f), qi = splat(0.f);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            qr += x[e].r * x[e].r;
            qi += x[e].i * x[e].i;
        }
        double q[4] = {(double)qr[0], (double)qr[1], (double)qi[0], (double)qi[1]};
        block_sum<C::NTHREADS, G, 4>(q, reinterpret_cast<double*>(lds), tid, h);
        if (j0 == 0) {
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (has[i]) {
                    // (float32 partial sums: 1e-6 relative; the margin covers them and the rounding of the product)
                    const float bound = sqrtf((float)q[i]) * a.wnorm * 1.001f;
                    atomicMax(a.absmax + c0 + i, __float_as_uint(bound));    // non-negative floats order like their bits
                }
        }
    }

    // taper weights of the first taper; later tapers are prefetched while the previous FFT runs
    float wn[16], wo[HALF ? 16 : 1];       // (HALF) wo: the weights of the odd samples
    if (wfull) {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            wn[e] = (SPYFFT_ABL & 1) ? a.scale : ldg<float>(a.tapers, (unsigned)(SM * (j0 + T * e)) * 4u);
            if constexpr (HALF) wo[e] = ldg<float>(a.tapers, (unsigned)(2 * (j0 + T * e) + 1) * 4u);
        }
    } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const unsigned n = (unsigned)(SM * (j0 + T * e));
            const float wl = (SPYFFT_ABL & 1) ? a.scale : ldg<float>(a.tapers, min(n, nsig_m1) * 4u);
            wn[e] = (n <= nsig_m1) ? wl : 0.f;
            if constexpr (HALF) {
                const float wl1 = ldg<float>(a.tapers, min(n + 1u, nsig_m1) * 4u);
                wo[e] = (n + 1u <= nsig_m1) ? wl1 : 0.f;
            }
        }
    }

    for (int k = 0; k < a.ntaper; ++k) {
        const int j = opaque(j0);
        C2 v[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            v[e].r = x[e].r * wn[e];
            v[e].i = x[e].i * (HALF ? wo[HALF ? e : 0] : wn[e]);
        }
        if (k + 1 < a.ntaper) {
            const float* w = a.tapers + (size_t)(k + 1) * a.nsig;   // wave-uniform
            if (wfull) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    wn[e] = (SPYFFT_ABL & 1) ? a.scale : ldg<float>(w, (unsigned)(SM * (j + T * e)) * 4u);
                    if constexpr (HALF) wo[e] = ldg<float>(w, (unsigned)(2 * (j + T * e) + 1) * 4u);
                }
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const unsigned n = (unsigned)(SM * (j + T * e));
                    const float wl = (SPYFFT_ABL & 1) ? a.scale : ldg<float>(w, min(n, nsig_m1) * 4u);
                    wn[e] = (n <= nsig_m1) ? wl : 0.f;
                    if constexpr (HALF) {
                        const float wl1 = ldg<float>(w, min(n + 1u, nsig_m1) * 4u);
                        wo[e] = (n + 1u <= nsig_m1) ? wl1 : 0.f;
                    }
                }
            }
        }
        if (a.demean_taper) {
            __syncthreads();          // block_sum writes its scratch into the planes other waves may still be reading
            double s[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                s[0] += v[e].r[0];
                s[1] += v[e].r[1];
                s[HALF ? 0 : 2] += v[e].i[0];
                s[HALF ? 1 : 3] += v[e].i[1];
            }
            block_sum<C::NTHREADS, G, 4>(s, reinterpret_cast<double*>(lds), tid, h);
            const v2f mr = v2f{(float)(s[0] / a.nsig), (float)(s[1] / a.nsig)};
            const v2f mi = HALF ? mr : v2f{(float)(s[2] / a.nsig), (float)(s[3] / a.nsig)};
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n0 = SM * (j + T * e);
                v[e].r -= n0 < a.nsig ? mr : splat(0.f);
                v[e].i -= n0 + (HALF ? 1 : 0) < a.nsig ? mi : splat(0.f);
            }
        }

        SPY_STAMP(sn);
        fft2_forward<LOG2N, G>(v, lds, j, h, a.tw SPY_STAMP_ARG);
        SPY_STAMP(sn);

        // ---- separate the real channels: partner bin N-f lives in the upper half
        __syncthreads();              // the FFT's last reads of the planes are done everywhere
        SPY_STAMP(sn);
        {
            const int wb = C::rbase(j, h);
#pragma unroll
            for (int e = 8; e < 16; ++e) {
                lre[wb + e * C::ESTRIDE] = v[e].r;
                lim[wb + e * C::ESTRIDE] = v[e].i;
            }
        }
        SPY_STAMP(sn);
        __syncthreads();
        SPY_STAMP(sn);
        // output slab of (segment b, taper k): wave-uniform base, 32-bit lane offsets
        char* const slab = reinterpret_cast<char*>(a.out) +
                           ((size_t)b * kout + (MEAN ? 0 : k)) * (size_t)a.nfsel * (size_t)a.nchan * OSZ;
        // partner of f = j + T*e is N - f: idx(N - j, h) - e*ESTRIDE (index N = spare slot, unused value)
        const int pb = C::idx(N - j, h) - 7 * C::ESTRIDE;
#pragma unroll
        for (int e = 0; e < 9; ++e) {
            C2 xa, xb;   // xa = X(c0, c1), xb = X(c2, c3)
            int f;
            if (e < 8) {
                f = j + T * e;
                const C2 z = v[e];
                C2 zp;
                zp.r = lre[pb + (7 - e) * C::ESTRIDE];
                zp.i = lim[pb + (7 - e) * C::ESTRIDE];
                if (f == 0) zp = z;
                xa.r = z.r + zp.r;          // (the tapers carry scale / 2)
                xa.i = z.i - zp.i;
                xb.r = z.i + zp.i;
                xb.i = zp.r - z.r;
                if constexpr (HALF) {
                    // xa = E, xb = O of the pair's real transform: bins f and N - f (f = 0: DC and the Nyquist bin)
                    const C2 t = cmul_s(xb, ldg<float2>(a.twh, (unsigned)f * 8u));
                    const C2 d = csub(xa, t);
                    xa = cadd(xa, t);
                    xb.r = d.r;
                    xb.i = -d.i;
                }
            } else {
                if (j != 0) break;
                f = N / 2;
                xa.r = v[8].r * 2.f;
                xb.r = v[8].i * 2.f;
                xa.i = xb.i = splat(0.f);
                if constexpr (HALF) {            // the middle bin is its own partner: X[N / 2] = conj Z[N / 2]
                    xa.i = -xb.r;
                    xb.r = splat(0.f);
                }
            }
            if (MEAN) {
                if (CPLX) {
                    ma[e] = cadd(ma[e], xa);
                    mb[e] = cadd(mb[e], xb);
                } else if (OUTK == 0) {
                    ma[e].r += xa.r * xa.r + xa.i * xa.i;
                    ma[e].i += xb.r * xb.r + xb.i * xb.i;
                } else {
                    ma[e].r += v2f{convert_real_slow(make_float2(xa.r[0], xa.i[0]), a.out_kind),
                                   convert_real_slow(make_float2(xa.r[1], xa.i[1]), a.out_kind)};
                    ma[e].i += v2f{convert_real_slow(make_float2(xb.r[0], xb.i[0]), a.out_kind),
                                   convert_real_slow(make_float2(xb.r[1], xb.i[1]), a.out_kind)};
                }
                continue;
            }
            if constexpr (HALF) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    if (q == 1 && e == 8) break;
                    const int fb = q ? N - f : f;
                    const C2 X = q ? xb : xa;
                    if (fast) {
                        const unsigned o = ((unsigned)fb * (unsigned)a.nchan + (unsigned)c0) * OSZ;
                        if (CPLX) {
                            stg<float4>(slab, o, make_float4(X.r[0], X.i[0], X.r[1], X.i[1]));
                        } else if (OUTK == 0) {
                            const v2f pw = X.r * X.r + X.i * X.i;
                            stg<float2>(slab, o, make_float2(pw[0], pw[1]));
                        } else {
                            stg<float2>(slab, o, make_float2(convert_real_slow(make_float2(X.r[0], X.i[0]), a.out_kind),
                                                             convert_real_slow(make_float2(X.r[1], X.i[1]), a.out_kind)));
                        }
                    } else {
                        const int fi = a.fpos ? ldg<int>(a.fpos, (unsigned)fb * 4u) : fb;
                        if (fi >= 0) {
                            const float2 Xc[2] = {make_float2(X.r[0], X.i[0]), make_float2(X.r[1], X.i[1])};
                            const unsigned o = ((unsigned)fi * (unsigned)a.nchan + (unsigned)c0) * OSZ;
#pragma unroll
                            for (int i = 0; i < 2; ++i) {
                                if (!has[i]) continue;
                                if (CPLX) stg<float2>(slab, o + i * OSZ, Xc[i]);
                                else stg<float>(slab, o + i * OSZ, convert_real<OUTK>(Xc[i], a.out_kind));
                            }
                        }
                    }
                }
                continue;
            }
            if (CPLX && a.blocked) {
                // channel-quad-blocked layout for the CSD kernel: the 32 bytes of this thread's four channels
                // are contiguous in f, so a wave writes whole 2-KiB runs instead of 64 scattered pieces
                const int fi = a.fpos ? ldg<int>(a.fpos, (unsigned)f * 4u) : f;
                if (fi >= 0 && has[0]) {               // has[0]: the quad exists (padding lanes of a group write nothing)
                    char* const bs = reinterpret_cast<char*>(a.out) +
                                     (((size_t)b * kout + k) * (size_t)((a.nchan + 3) >> 2) + (size_t)(c0 >> 2)) *
                                         (size_t)a.nfsel * 32u;
                    stg<float4>(bs, (unsigned)fi * 32u, make_float4(xa.r[0], xa.i[0], xa.r[1], xa.i[1]));
                    stg<float4>(bs, (unsigned)fi * 32u + 16u, make_float4(xb.r[0], xb.i[0], xb.r[1], xb.i[1]));
                }
                continue;
            }
            // (G == 2, complex spectra) the neighbouring row's halves for the 64-byte stores below: fetched by EVERY lane, in
            // front of the lane-dependent branch (a ragged last quad takes the slow path; its partner two lanes away has
            // the same quad and goes with it)
            float4 got2 = make_float4(0.f, 0.f, 0.f, 0.f);
            if constexpr (CPLX && !MEAN && G == 2 && !HALF) {
                if (e < 8) {
                    const bool odd = (j & 1) != 0;
                    const float4 give = odd ? make_float4(xa.r[0], xa.i[0], xa.r[1], xa.i[1]) : make_float4(xb.r[0], xb.i[0], xb.r[1], xb.i[1]);
                    got2 = make_float4(spy_lane_swap2(give.x), spy_lane_swap2(give.y), spy_lane_swap2(give.z), spy_lane_swap2(give.w));
                }
            }
            if (fast) {
                const unsigned o = ((unsigned)f * (unsigned)a.nchan + (unsigned)c0) * OSZ;
                if ((SPYFFT_ABL & 4) && xa.r[0] + xa.r[1] + xa.i[0] + xa.i[1] + xb.r[0] + xb.r[1] + xb.i[0] + xb.i[1] != 12345.f)
                    continue;
                if (CPLX) {
                    const float4 lo = make_float4(xa.r[0], xa.i[0], xa.r[1], xa.i[1]);
                    const float4 hi = make_float4(xb.r[0], xb.i[0], xb.r[1], xb.i[1]);
                    if (G == 1 && e < 8) {
                        // The L2 accepts one write request per line and clock whatever its size, and every lane
                        // owns a different row (bin): let lanes (2i, 2i+1) write the two 16-byte halves of the
                        // SAME row with one instruction (one 32-byte request instead of two 16-byte ones):
                        // the even lane hands over its upper half, the odd lane its lower half.
                        const bool odd = (j & 1) != 0;
                        const float4 give = odd ? lo : hi;
                        const float4 got = make_float4(lane_swap1(give.x), lane_swap1(give.y), lane_swap1(give.z),
                                                       lane_swap1(give.w));
                        const unsigned rb = (unsigned)a.nchan * OSZ;     // bytes per bin row
                        const unsigned oe = odd ? o - rb + 16u : o;      // row of the even lane, this lane's half
                        stg<float4>(slab, oe, odd ? got : lo);
                        stg<float4>(slab, oe + rb, odd ? hi : got);
                    } else if (G == 2 && e < 8) {
                        // Two quads per workgroup: lanes (j, 0), (j, 1), (j + 1, 0), (j + 1, 1) are neighbours and hold the
                        // 64 bytes of row j and the 64 bytes of row j + 1.  The same trade two lanes apart (even j hands over
                        // its upper halves, odd j its lower ones) lets ONE instruction write the whole 64-byte piece of a row:
                        // 16 line requests per instruction instead of 32
                        const bool odd = (j & 1) != 0;
                        const float4 got = got2;
                        const unsigned rb = (unsigned)a.nchan * OSZ;     // bytes per bin row
                        const unsigned oe = odd ? o - rb + 16u : o;      // row of the even lane, this lane's half
                        stg<float4>(slab, oe, odd ? got : lo);
                        stg<float4>(slab, oe + rb, odd ? hi : got);
                    } else {
                        stg<float4>(slab, o, lo);
                        stg<float4>(slab, o + 16u, hi);
                    }
                } else if (OUTK == 0) {
                    const v2f pa = xa.r * xa.r + xa.i * xa.i, pb2 = xb.r * xb.r + xb.i * xb.i;
                    stg<float4>(slab, o, make_float4(pa[0], pa[1], pb2[0], pb2[1]));
                } else {
                    stg<float4>(slab, o, make_float4(convert_real_slow(make_float2(xa.r[0], xa.i[0]), a.out_kind),
                                                     convert_real_slow(make_float2(xa.r[1], xa.i[1]), a.out_kind),
                                                     convert_real_slow(make_float2(xb.r[0], xb.i[0]), a.out_kind),
                                                     convert_real_slow(make_float2(xb.r[1], xb.i[1]), a.out_kind)));
                }
            } else {
                const int fi = a.fpos ? ldg<int>(a.fpos, (unsigned)f * 4u) : f;
                if (fi >= 0) {
                    const float2 X[4] = {make_float2(xa.r[0], xa.i[0]), make_float2(xa.r[1], xa.i[1]),
                                         make_float2(xb.r[0], xb.i[0]), make_float2(xb.r[1], xb.i[1])};
                    const unsigned o = ((unsigned)fi * (unsigned)a.nchan + (unsigned)c0) * OSZ;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (!has[i]) continue;
                        if (CPLX) stg<float2>(slab, o + i * OSZ, X[i]);
                        else stg<float>(slab, o + i * OSZ, convert_real<OUTK>(X[i], a.out_kind));
                    }
                }
            }
        }
        // no barrier here: the next taper's first LDS write sits behind one (fft2_forward / block_sum)
        SPY_STAMP(sn);
    }

    if (MEAN) {
        const float kk = 1.0f / (float)a.ntaper;
        char* const slab = reinterpret_cast<char*>(a.out) + (size_t)b * (size_t)a.nfsel * (size_t)a.nchan * OSZ;
#pragma unroll
        for (int e = 0; e < 9; ++e) {
            if (e == 8 && j0 != 0) break;
            const int f = (e < 8) ? j0 + T * e : N / 2;
            const float nt = (float)a.ntaper;
            if constexpr (HALF) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    if (q == 1 && e == 8) break;
                    const int fb = q ? N - f : f;
                    const int fi = a.fpos ? ldg<int>(a.fpos, (unsigned)fb * 4u) : fb;
                    if (fi < 0) continue;
                    const unsigned o = ((unsigned)fi * (unsigned)a.nchan + (unsigned)c0) * OSZ;
                    if (CPLX) {
                        const C2 A = q ? mb[e] : ma[e];
                        const float2 X[2] = {make_float2(A.r[0] / nt, A.i[0] / nt), make_float2(A.r[1] / nt, A.i[1] / nt)};
#pragma unroll
                        for (int i = 0; i < 2; ++i)
                            if (has[i]) stg<float2>(slab, o + i * OSZ, X[i]);
                    } else {
                        const v2f A = q ? ma[e].i : ma[e].r;
                        if (fast) {
                            stg<float2>(slab, o, make_float2(A[0] / nt, A[1] / nt));
                        } else {
#pragma unroll
                            for (int i = 0; i < 2; ++i)
                                if (has[i]) stg<float>(slab, o + i * OSZ, A[i] / nt);
                        }
                    }
                }
                continue;
            }
            const int fi = a.fpos ? ldg<int>(a.fpos, (unsigned)f * 4u) : f;
            if (fi < 0) continue;
            const unsigned o = ((unsigned)fi * (unsigned)a.nchan + (unsigned)c0) * OSZ;
            if (CPLX) {
                const float2 X[4] = {make_float2(ma[e].r[0] / nt, ma[e].i[0] / nt), make_float2(ma[e].r[1] / nt, ma[e].i[1] / nt),
                                     make_float2(mb[e].r[0] / nt, mb[e].i[0] / nt), make_float2(mb[e].r[1] / nt, mb[e].i[1] / nt)};
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (has[i]) stg<float2>(slab, o + i * OSZ, X[i]);
            } else if (fast) {
                stg<float4>(slab, o, make_float4(ma[e].r[0] / nt, ma[e].r[1] / nt, ma[e].i[0] / nt, ma[e].i[1] / nt));
            } else {
                const float X[4] = {ma[e].r[0] / nt, ma[e].r[1] / nt, ma[e].i[0] / nt, ma[e].i[1] / nt};
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (has[i]) stg<float>(slab, o + i * OSZ, X[i]);
            }
            (void)kk;
        }
    }
}

}  // namespace spyfft
