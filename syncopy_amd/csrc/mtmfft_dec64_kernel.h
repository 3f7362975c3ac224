// K1r, second generation: the reference-precision tapered FFT (float64 taper product and transform, complex64 rounding
// exactly where specest/mtmfft.py:96-127 rounds; spyhip_fft_plan_set_precision) with COMPILE-TIME radix schedules - the
// structure of mtmfft_dec_kernel.h carried over to complex128:
//
//   * a work item is a channel PAIR of one segment: (c0, c1) are the real and the imaginary part of ONE complex128
//     transform of length N = V R1 R2 R3, separated afterwards (X(c0)[f] = (Z[f] + conj(Z[N-f]))/2,
//     X(c1)[f] = (Z[f] - conj(Z[N-f]))/(2i));
//   * T = N / V threads per pair, G pairs per workgroup (thread id = j G + h); thread j keeps the float32 samples
//     n = j + T e (e < V) of its two channels in registers across all tapers (the segment is read from HBM once) -
//     except where the register file cannot hold them (XRES = false: N = 16384 with 1024 threads re-reads them per taper);
//   * Stockham passes with immediates: the first (radix V) takes the tapered samples straight from the registers, the
//     last leaves bin j + T e in register e: P - 1 exchanges through LDS for P passes.  An exchange moves complex128
//     values (16 bytes, the float4 layout of the float32 kernel) or, with SPLIT, the real parts and then the imaginary
//     parts through ONE 8-byte plane (half the LDS: N = 16384 needs it to fit at all);
//   * XCD-aware block -> (segment, pair group) map: the 16 / G workgroups whose 8-byte pieces share a 128-byte line
//     of a trial row run on one XCD back to back, as in the float32 kernels.
//
// The generic kernels this replaces on the common lengths (mtmfft_f64_kernel.h): the radix-16 register kernel without
// the XCD map and with 4-byte loads (powers of two 256 ... 4096) and the run-time-radix Stockham passes over work arrays
// in LDS / global memory (everything else: 7.6x the float32 time at N = 2000, 8.8x at 5000, 9x at 16384).
#pragma once
#include "cd_math.h"
#include "f64_dft.h"
#include "fft2_device.h"          // block_sum
#include "mtmfft_kernel.h"        // MtmArgs, convert_real, ldg / stg
#include "mtmfft_f64_kernel.h"    // F64Args

namespace spyfft {

template <int V_, int R1_, int R2_, int R3_, int G_, bool SPLIT_ = false, bool XRES_ = true, bool HOIST_ = true, int P_ = 1,
          bool HALF_ = false>
struct CfgD64 {
    static constexpr int V = V_, R1 = R1_, R2 = R2_, R3 = R3_, G = G_;
    // P > 1: decimation in time in front of the schedule - N = P M: P groups of T threads transform the sub-sequences
    // x[P n + r] (length M = V R1 R2 R3, one LDS region each) side by side, then ONE radix-P combine through LDS:
    // X[k + M q] = sum_r w_P^(r q) W_N^(r k) F_r[k].  Trial lengths 3 x (a length with a schedule): 3000, 6000, 1500, 3072 ...
    static constexpr int P = P_;
    static constexpr bool SPLIT = SPLIT_, XRES = XRES_;
    // HOIST: the base twiddles of the later passes stay in registers across the tapers and their powers are formed by
    // multiplication (d64_pass); false: every power comes from the table inside the taper loop (N = 8192: the four
    // passes' base twiddles do not fit next to 16 resident values per thread - measured 61.9 vs 53.8 us/trial)
    static constexpr bool HOIST = HOIST_;
    // HALF (as CfgD::HALF of the float32 kernel): a work item is ONE channel, z[m] = x[2 m] + i x[2 m + 1], and the epilogue
    // turns Z[f], Z[N - f] into the bins f and N - f of the real transform of 2 N samples - trials of 10240 < nfft <= 20480
    // samples stay in LDS (before: N = P M through HBM scratch, mtmfft_declong64.h: 40 x the algorithmic traffic at 12000)
    static constexpr bool HALF = HALF_;
    static constexpr int M = V * R1 * R2 * R3;               // length of one sub-transform (= N without decimation)
    static constexpr int N = P * M;
    static constexpr int T = M / V;                          // threads per sub-transform
    static constexpr int TT = P * T;                         // threads per channel pair
    static constexpr int NPASS = 2 + (R2 > 1 ? 1 : 0) + (R3 > 1 ? 1 : 0);
    static constexpr int NTHREADS = ((TT * G + 63) / 64) * 64;
    static constexpr int PL1 = M + M / V + 1;                // LDS elements of one sub-transform: one pad per V values
    static constexpr int PLANE = P * PL1;                    // elements per pair
    static constexpr int ESTRIDE = (T + T / V) * G;          // LDS distance of e -> e + 1
    static constexpr size_t LDS_BYTES = (size_t)PLANE * G * (SPLIT ? 8 : 16);
    // waves per SIMD the register allocation is held to: what LDS lets co-reside, capped by what the schedule needs
    // (2 V sample + 4 V value + 4 R butterfly registers and ~50 others: V = 10 -> 168 registers, V = 16 / 20 -> 256);
    // a 1024-thread workgroup has no choice (128)
    static constexpr int WG_PER_CU = (int)((160 * 1024) / LDS_BYTES) < 1 ? 1 : (int)((160 * 1024) / LDS_BYTES);
    static constexpr int WPE_RAW = (NTHREADS / 64) * WG_PER_CU / 4;
    static constexpr int WPE_CAP = NTHREADS == 1024 ? 4 : (V <= 10 ? 3 : 2);
    static constexpr int WPE = WPE_RAW < 1 ? 1 : (WPE_RAW > WPE_CAP ? WPE_CAP : WPE_RAW);
    static_assert(R1 > 1 && V % R1 == 0 && V % R2 == 0 && V % R3 == 0, "every radix divides the values per thread");
    static_assert(T % V == 0, "T multiple of V: idx(j + T e) stays affine in e");
    static_assert(NTHREADS <= 1024 && (64 % G) == 0 && (V % 2) == 0, "workgroup shape");
    static_assert(P == 1 || (P == 3 && !SPLIT && XRES), "decimation: radix 3, plain exchanges, resident samples");
    static_assert(!HALF || (XRES && NTHREADS < 1024 && V <= 20), "HALF form: resident samples, taper mean in registers");
    __device__ static __forceinline__ int idx(int i, int h) { return (i + i / V) * G + h; }
};

// exchange through LDS: thread-private slots wbase[m] + r WS out, rb + e ESTRIDE back
template <class C, int R, int MB, int WS>
__device__ __forceinline__ void d64_exchange(spywil::cd (&v)[C::V], void* ldsv, const int (&wbase)[MB], int rb, bool active) {
    using spywil::cd;
    constexpr int V = C::V;
    if constexpr (!C::SPLIT) {
        cd* lds = reinterpret_cast<cd*>(ldsv);
        __syncthreads();              // (write-after-read: earlier reads of the buffer by any thread are done)
        if (active) {
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int r = 0; r < R; ++r) lds[wbase[m] + r * WS] = v[m + MB * r];
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < V; ++e) v[e] = lds[rb + e * C::ESTRIDE];
    } else {
        double* lds = reinterpret_cast<double*>(ldsv);
        __syncthreads();
        if (active) {
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int r = 0; r < R; ++r) lds[wbase[m] + r * WS] = v[m + MB * r].x;
        }
        __syncthreads();
        double re[V];
#pragma unroll
        for (int e = 0; e < V; ++e) re[e] = lds[rb + e * C::ESTRIDE];
        __syncthreads();
        if (active) {
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int r = 0; r < R; ++r) lds[wbase[m] + r * WS] = v[m + MB * r].y;
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < V; ++e) v[e] = make_double2(re[e], lds[rb + e * C::ESTRIDE]);
    }
}

// One pass.  In: v[e] = in[j + T e] (pass 0: the tapered samples).  Out: LAST - v[e] = X[j + T e] in registers;
// otherwise the outputs go through LDS and v[e] = out[j + T e] comes back.  w1[m] = exp(-2 pi i k_m / (Ns R)), the base
// twiddle of butterfly m = j + T m of this pass: it does not depend on the taper, so the kernel fetches it ONCE per
// workgroup (d64_twiddles) and every power w^r, r < R, is formed from it in registers (w^2 = w w, w^3, w^4 = (w^2)^2,
// w^8 ..., then w^(4a + l) = w^(4a) w^l: ~R complex products instead of (R + 3) / 4 + 2 table reads per butterfly and
// taper - with two waves per SIMD the kernel waits on L2 latency, not on the vector pipe).
template <class C, int R, int Ns, bool FIRST, bool LAST>
__device__ __forceinline__ void d64_pass(spywil::cd (&v)[C::V], void* lds, int j, int h, bool active,
                                         const spywil::cd* w1, int region = 0) {
    using spywil::cd;
    using spywil::cmul;
    constexpr int V = C::V, T = C::T, G = C::G, MB = V / R;
    int wbase[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) {
        const int b = j + T * m;
        const int q = b / Ns, k = b - q * Ns;
        cd u[R];
#pragma unroll
        for (int r = 0; r < R; ++r) u[r] = v[m + MB * r];
        if (!FIRST) {
            constexpr int NA = (R + 3) / 4;
            cd wb[4], wa[NA > 1 ? NA : 2];
            if constexpr (C::HOIST) {
                wb[1] = w1[m];
                // (the powers are taper-invariant too: without this the compiler hoists all of them out of the taper loop
                // and spills ~100 registers; only the base twiddle is meant to stay live)
                spy_opaque2(wb[1].x, wb[1].y);
            }
            if constexpr (C::HOIST) {
                wb[2] = cmul(wb[1], wb[1]);
                wb[3] = cmul(wb[2], wb[1]);
                wa[1] = cmul(wb[2], wb[2]);
#pragma unroll
                for (int a = 2; a < NA; ++a) wa[a] = (a % 2 == 0) ? cmul(wa[a / 2], wa[a / 2]) : cmul(wa[a - 1], wa[1]);
            } else {
                // w1 = the table itself (d64_twiddles_none): powers by table reads, as the first generation did
                const int st = k * (C::M / (Ns * R)) * C::P;
#pragma unroll
                for (int l = 1; l < 4; ++l) wb[l] = (l < R) ? w1[l * st] : make_double2(1.0, 0.0);
#pragma unroll
                for (int a = 1; a < NA; ++a) wa[a] = w1[4 * a * st];
            }
#pragma unroll
            for (int r = 1; r < R; ++r) {
                const int hi = r >> 2, lo = r & 3;
                const cd w = (hi == 0) ? wb[lo] : (lo == 0 ? wa[hi] : cmul(wa[hi], wb[lo]));
                u[r] = cmul(u[r], w);
            }
        }
        spywil::d_dft<R>(u);
#pragma unroll
        for (int r = 0; r < R; ++r) v[m + MB * r] = u[r];
        // LDS slot of output r: idx(q Ns R + k + r Ns); Ns is 1 (first pass, R = V) or a multiple of V
        wbase[m] = region + (FIRST ? (b * (V + 1)) * G + h : (q * (Ns * R + Ns * R / V) + k + k / V) * G + h);
    }
    if (LAST) return;
    constexpr int WS = FIRST ? G : (Ns + Ns / V) * G;
    d64_exchange<C, R, MB, WS>(v, lds, wbase, region + C::idx(j, h), active);
}

// (P = 3) bin k + M q of the length-3M transform from the twiddled sub-transforms in the three LDS regions:
// X[k + M q] = g0 + w3^q g1 + w3^(2q) g2 = g0 + a (g1 + g2) + c (-i)(g1 - g2); a = 1, c = 0 (q = 0); a = -1/2, c = +-sqrt(3)/2
template <class C>
__device__ __forceinline__ spywil::cd d64_dit3_bin(const void* lds, int k, int q, int h) {
    using spywil::cd;
    const cd* const L = reinterpret_cast<const cd*>(lds);
    const int ki = C::idx(k, h);
    const cd g0 = L[ki], g1 = L[C::PL1 * C::G + ki], g2 = L[2 * C::PL1 * C::G + ki];
    const double ca = q == 0 ? 1.0 : -0.5;
    const double cc = q == 0 ? 0.0 : (q == 1 ? 0.86602540378443864676 : -0.86602540378443864676);
    const double sx = g1.x + g2.x, sy = g1.y + g2.y, dx = g1.x - g2.x, dy = g1.y - g2.y;
    return make_double2(fma(cc, dy, fma(ca, sx, g0.x)), fma(-cc, dx, fma(ca, sy, g0.y)));
}

// the base twiddles of one pass for thread j: w1[m] = tw[k_m N / (Ns R)], k_m = (j + T m) mod Ns
template <class C, int R, int Ns>
__device__ __forceinline__ void d64_twiddles(spywil::cd (&w1)[C::V / R], int j, const spywil::cd* __restrict__ tw) {
    constexpr int MB = C::V / R;
#pragma unroll
    for (int m = 0; m < MB; ++m) {
        const int b = j + C::T * m;
        w1[m] = tw[(b % Ns) * (C::M / (Ns * R)) * C::P];
    }
}

// taper weight of sample n (0 beyond the window): branch-free - a clamped 32-bit offset from the wave-uniform row pointer
// and a select, instead of one exec-masked branch per value
__device__ __forceinline__ double tapw(const double* w, unsigned n, unsigned nsig_m1) {
    const double wl = ldg<double>(w, min(n, nsig_m1) * 8u);
    return n <= nsig_m1 ? wl : 0.0;
}

// sample n of the two channels of a pair (float32, as the reference holds the trial); zero outside [rlo, rhi)
struct D64Src {
    const float* seg;
    long long ld;
    unsigned col0, col1;
    int rlo, rhi;
    bool some, vec2, has0, has1;
    // the V samples n = j + T e of a thread, branches outside the unrolled loops (straight-line loads: the register
    // allocator copes badly with sixteen diamonds)
    template <int V, int T, bool HALF = false>
    __device__ __forceinline__ void get_all(int j, float (&u0)[V], float (&u1)[V]) const {      // samples j + T e
        if (HALF && some) {
            // (HALF) u0 = the even, u1 = the odd sample of the pair 2 (j + T e), 2 (j + T e) + 1 of channel col0
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const int n0 = 2 * (j + T * e), n1 = n0 + 1;
                const int nc0 = min(max(n0, rlo), rhi - 1), nc1 = min(max(n1, rlo), rhi - 1);
                const float t0 = seg[(size_t)nc0 * ld + col0], t1 = seg[(size_t)nc1 * ld + col0];
                u0[e] = (n0 == nc0 && has0) ? t0 : 0.f;
                u1[e] = (n1 == nc1 && has0) ? t1 : 0.f;
            }
        } else if (some) {
            if (vec2) {
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    const int n = j + T * e;
                    const int nc = min(max(n, rlo), rhi - 1);
                    const float2 t = *reinterpret_cast<const float2*>(seg + (size_t)nc * ld + col0);
                    u0[e] = (n == nc) ? t.x : 0.f;
                    u1[e] = (n == nc) ? t.y : 0.f;
                }
            } else {
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    const int n = j + T * e;
                    const int nc = min(max(n, rlo), rhi - 1);
                    const float t0 = seg[(size_t)nc * ld + col0], t1 = seg[(size_t)nc * ld + col1];
                    u0[e] = (n == nc && has0) ? t0 : 0.f;
                    u1[e] = (n == nc && has1) ? t1 : 0.f;
                }
            }
        } else {
#pragma unroll
            for (int e = 0; e < V; ++e) u0[e] = u1[e] = 0.f;
        }
    }
};

// OUTK: 0 = power, 1 = any other real conversion, 2 = complex; MEAN: average over tapers
template <class C, int OUTK, bool MEAN>
__global__ void __launch_bounds__((C::NTHREADS), (C::WPE)) mtmfft_dec64_kernel(F64Args fa) {
    using spywil::cd;
    constexpr bool CPLX = (OUTK == 2);
    constexpr int V = C::V, N = C::N, T = C::T, TT = C::TT, P = C::P, G = C::G, HV = V / 2;
    constexpr bool HALF = C::HALF;
    constexpr int SM = HALF ? 2 : 1;               // sample n of value e: SM (jn + TT e) for the real part, + (HALF ? 1 : 0) for the imaginary
    const MtmArgs& a = fa.m;
    SPY_DYN_SMEM(char, ldsraw);
    void* const lds = ldsraw;

    const int tid = threadIdx.x;
    const int h = tid % G, jt = tid / G;
    const bool active = jt < TT;                  // the workgroup is padded to whole waves
    const int j0 = active ? jt : 0;               // thread of the pair: bins j0 + TT e in the epilogue
    // decimation (P > 1): group r0 = j0 / T transforms the samples P n + r0; js0 = its thread index inside the group.
    // Thread j0 then holds the SAMPLES jn0 + TT e with jn0 = P js0 + r0 (P = 1: jn0 = js0 = j0, TT = T)
    const int r0 = j0 / T, js0 = j0 - r0 * T, jn0 = P * js0 + r0;

    // XCD-aware block -> (segment, pair group), as mtmfft_dec_kernel
    const long long id = blockIdx.x;
    const int xcd = (int)(id & 7);
    const long long y = id >> 3;
    const long long nclt = (long long)a.nseg * a.ncl, chunk = (nclt + 7) >> 3;
    const long long cidx = (long long)xcd * chunk + y / a.S;
    const int qs = (int)(y % a.S);
    if (cidx >= nclt) return;
    const int b = (int)(cidx / a.ncl);
    const int pg = (int)(cidx % a.ncl) * a.S + qs;
    if (pg >= a.npg) return;

    const int c0 = (HALF ? 1 : 2) * (pg * G + h);
    const int c1 = HALF ? c0 : c0 + 1;            // (HALF: both halves of the complex value belong to channel c0)
    const bool has0 = active && c0 < a.nchan, has1 = active && c1 < a.nchan;
    D64Src src;
    src.col0 = has0 ? (unsigned)(a.chan_idx ? a.chan_idx[c0] : c0) : 0u;
    src.col1 = has1 ? (unsigned)(a.chan_idx ? a.chan_idx[c1] : c1) : 0u;
    src.has0 = has0; src.has1 = has1;
    const long long start = a.seg_start[b];
    const long long rl = a.seg_lo[b] - start, rh = a.seg_hi[b] - start;
    src.rlo = (int)(rl < 0 ? 0 : (rl > a.nsig ? a.nsig : rl));
    src.rhi = (int)(rh < 0 ? 0 : (rh > a.nsig ? a.nsig : rh));
    src.seg = a.data + start * a.ld;              // only rows in [rlo, rhi) are dereferenced
    src.ld = a.ld;
    src.vec2 = !HALF && (a.chan_idx == nullptr) && has1 && ((a.ld & 1) == 0) && ((reinterpret_cast<size_t>(a.data) & 7) == 0);
    src.some = src.rhi > src.rlo;

    float x0[V], x1[V];           // XRES: resident across the tapers; otherwise refilled per taper
    const bool fit = !(a.detrend == 0 && a.means) && a.detrend >= 0;
    if (C::XRES || fit) src.template get_all<V, TT, HALF>(jn0, x0, x1);

    // ---- polynomial removal in float32 (scipy.signal.detrend on the float32 trial, compRoutines.py:169-172): the
    // reference-order means (detrend 0) or a float64 fit whose trend is rounded to float32 before it is subtracted;
    // float64 segments (seg_f64: padded sliding windows, stft.py:101-117) subtract the float64 trend
    const float mid = 0.5f * (float)(a.nsig - 1);
    float m0 = 0.f, m1 = 0.f;                          // constant detrending with the reference-order means
    double t0c = 0.0, t1c = 0.0, t0s = 0.0, t1s = 0.0; // fitted trend: constant and slope about the centre
    if (a.detrend == 0 && a.means) {
        m0 = has0 ? a.means[(size_t)b * a.nchan + c0] : 0.f;
        m1 = has1 ? a.means[(size_t)b * a.nchan + c1] : 0.f;
    } else if (fit) {
        double s[4] = {0.0, 0.0, 0.0, 0.0};
        const float lin = a.detrend == 1 ? 1.f : 0.f;
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const int n = SM * (jn0 + TT * e), n1 = n + (HALF ? 1 : 0);
            const float m = (active && n < a.nsig) ? 1.f : 0.f;           // branch-free masks (exact: 0 or 1)
            const float mb = (active && n1 < a.nsig) ? 1.f : 0.f;
            const float u0 = m * x0[e], u1 = mb * x1[e];
            const double dn = (double)(lin * ((float)n - mid));           // exact: half-integers < 2^23
            const double dn1 = (double)(lin * ((float)n1 - mid));
            s[0] += (double)u0;
            s[HALF ? 0 : 1] += (double)u1;                                // (HALF: one channel, even and odd samples)
            s[2] += dn * u0;
            s[HALF ? 2 : 3] += dn1 * u1;
        }
        block_sum<C::NTHREADS, G, 4>(s, reinterpret_cast<double*>(lds), tid, h);
        const double inv = 1.0 / a.nsig;
        const double den = (a.detrend == 1 && a.nsig > 1) ? 12.0 / ((double)a.nsig * ((double)a.nsig * a.nsig - 1.0)) : 0.0;
        t0c = s[0] * inv; t1c = s[HALF ? 0 : 1] * inv; t0s = s[2] * den; t1s = s[HALF ? 2 : 3] * den;
    }
    const bool f64t = fit && a.seg_f64;
    // float32 trend of sample n (what the reference subtracts from the float32 trial); float64 segments: see the taper loop
    if constexpr (C::XRES) {
        if (fit && !f64t) {
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const int n = SM * (jn0 + TT * e), n1 = n + (HALF ? 1 : 0);
                const double dn = (double)((float)n - mid), dn1 = (double)((float)n1 - mid);
                const float q0 = (float)(t0c + t0s * dn), q1 = (float)(t1c + t1s * dn1);
                x0[e] -= n < a.nsig ? q0 : 0.f;
                x1[e] -= n1 < a.nsig ? q1 : 0.f;
            }
        } else if (!fit) {
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const int n = SM * (jn0 + TT * e), n1 = n + (HALF ? 1 : 0);
                x0[e] -= n < a.nsig ? m0 : 0.f;
                x1[e] -= n1 < a.nsig ? m1 : 0.f;
            }
        }
    }

    // taper mean: float32 accumulators in registers - or, where a 1024-thread workgroup leaves no room for them, in
    // the output slab itself (every thread owns its bins; same float32 additions in taper order, one division at the end)
    constexpr bool MREG = MEAN && C::NTHREADS < 1024 && V <= 20;          // (V = 32: 128 value registers leave no room either)
    float macc0[MREG ? HV + 1 : 1], macc1[MREG ? HV + 1 : 1], mim0[(MREG && CPLX) ? HV + 1 : 1], mim1[(MREG && CPLX) ? HV + 1 : 1];
    if constexpr (MREG) {
#pragma unroll
        for (int e = 0; e <= HV; ++e) {
            macc0[e] = macc1[e] = 0.f;
            if constexpr (CPLX) mim0[e] = mim1[e] = 0.f;
        }
    }
    const int kout = MEAN ? 1 : a.ntaper;
    constexpr unsigned OSZ = CPLX ? 8u : 4u;
    const bool fast = has1 && (a.fpos == nullptr) && ((reinterpret_cast<size_t>(a.out) & 15) == 0) && ((a.nchan & 1) == 0);
    const cd* const tw = reinterpret_cast<const cd*>(fa.tw64);
    const cd wdit = P > 1 ? tw[r0 * js0] : make_double2(1.0, 0.0), wdit_step = P > 1 ? tw[r0 * T] : make_double2(1.0, 0.0);
    // base twiddles of the later passes: taper-invariant, fetched once (4 registers per butterfly and pass)
    cd w1a[V / C::R1], w1b[C::R2 > 1 ? V / C::R2 : 1], w1c[C::R3 > 1 ? V / C::R3 : 1];
    if constexpr (C::HOIST) {
        d64_twiddles<C, C::R1, V>(w1a, js0, tw);
        if constexpr (C::NPASS >= 3) d64_twiddles<C, C::R2, V * C::R1>(w1b, js0, tw);
        if constexpr (C::NPASS >= 4) d64_twiddles<C, C::R3, V * C::R1 * C::R2>(w1c, js0, tw);
    }

    const unsigned nsig_m1 = (unsigned)(a.nsig - 1);
    for (int k = 0; k < a.ntaper; ++k) {
        const int je = opaque(j0);    // (keeps the index arithmetic of the passes inside the loop)
        const int r = je / T, j = je - r * T, jn = P * j + r;        // group, thread inside it, first sample (P = 1: r = 0, jn = j = je)
        const int region = r * C::PL1 * G;
        const double* w = fa.tapers64 + (size_t)k * a.nsig;
        cd v[V];
        bool filled = false;
        if constexpr (!C::XRES) {
            // not resident: the samples come back from L2 for every taper, detrended in float32 as above
            if (src.some && src.vec2 && !f64t) {
                // the common case in one straight pass: nothing but v[] stays live (a 1024-thread workgroup has 128 registers)
                filled = true;
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    const int n = jn + TT * e;
                    const int nc = min(max(n, src.rlo), src.rhi - 1);
                    const float2 t = *reinterpret_cast<const float2*>(src.seg + (size_t)nc * src.ld + src.col0);
                    const double dn = (double)((float)n - mid);
                    const float r0 = fit ? (float)(t0c + t0s * dn) : m0, r1 = fit ? (float)(t1c + t1s * dn) : m1;
                    const float u0 = ((n == nc) ? t.x : 0.f) - (n < a.nsig ? r0 : 0.f);
                    const float u1 = ((n == nc) ? t.y : 0.f) - (n < a.nsig ? r1 : 0.f);
                    const double wn = tapw(w, (unsigned)n, nsig_m1);
                    v[e] = make_double2(wn * (double)u0, wn * (double)u1);
                }
            } else {
                src.template get_all<V, TT>(jn, x0, x1);
                if (!f64t) {
#pragma unroll
                    for (int e = 0; e < V; ++e) {
                        const int n = jn + TT * e;
                        const double dn = (double)((float)n - mid);
                        const float r0 = fit ? (float)(t0c + t0s * dn) : m0, r1 = fit ? (float)(t1c + t1s * dn) : m1;
                        x0[e] -= n < a.nsig ? r0 : 0.f;
                        x1[e] -= n < a.nsig ? r1 : 0.f;
                    }
                }
            }
        }
        if (filled) {
        } else if (f64t) {
            // float64 segments in the reference: the trend is subtracted in float64
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const int n = SM * (jn + TT * e), n1 = n + (HALF ? 1 : 0);
                const double wn = tapw(w, (unsigned)n, nsig_m1), wn1 = HALF ? tapw(w, (unsigned)n1, nsig_m1) : wn;
                const double dn = (double)((float)n - mid), dn1 = (double)((float)n1 - mid);
                v[e] = make_double2(wn * ((double)x0[e] - (t0c + t0s * dn)), wn1 * ((double)x1[e] - (t1c + t1s * dn1)));
            }
        } else {
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const int n = SM * (jn + TT * e);
                const double wn = tapw(w, (unsigned)n, nsig_m1), wn1 = HALF ? tapw(w, (unsigned)n + 1u, nsig_m1) : wn;
                v[e] = make_double2(wn * (double)x0[e], wn1 * (double)x1[e]);     // win *= data_arr (float64)
            }
        }
        if (a.demean_taper) {                                                  // win -= win.mean(axis=0) (float64)
            double ds[2] = {0.0, 0.0};
#pragma unroll
            for (int e = 0; e < V; ++e) {
                ds[0] += v[e].x;
                ds[HALF ? 0 : 1] += v[e].y;
            }
            if (!active) ds[0] = ds[1] = 0.0;
            __syncthreads();          // block_sum writes its scratch into the buffer other waves may still be reading
            block_sum<C::NTHREADS, G, 2>(ds, reinterpret_cast<double*>(lds), tid, h);
            const double dm0 = ds[0] / a.nsig, dm1 = ds[HALF ? 0 : 1] / a.nsig;
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const int n = SM * (jn + TT * e);
                if (n < a.nsig) v[e].x -= dm0;
                if (n + (HALF ? 1 : 0) < a.nsig) v[e].y -= dm1;
            }
        }

        // ---- the passes: radix V from the registers, then R1 (R2, R3); the last one leaves v[e] = Z[j + T e]
        d64_pass<C, V, 1, true, false>(v, lds, j, h, active, nullptr, region);
        d64_pass<C, C::R1, V, false, C::NPASS == 2>(v, lds, j, h, active, C::HOIST ? w1a : tw, region);
        if constexpr (C::NPASS >= 3) d64_pass<C, C::R2, V * C::R1, false, C::NPASS == 3>(v, lds, j, h, active, C::HOIST ? w1b : tw, region);
        if constexpr (C::NPASS >= 4) d64_pass<C, C::R3, V * C::R1 * C::R2, false, true>(v, lds, j, h, active, C::HOIST ? w1c : tw, region);

        if constexpr (P == 3) {
            // ---- radix-3 combine of the three sub-transforms: v[e] = F_r[k], k = j + T e  ->  X[k + M r]
            cd* const L = reinterpret_cast<cd*>(lds);
            if (r > 0) {
                // W_N^(r k) = W_N^(r j) (W_N^(r T))^e: two table values per thread, fetched ONCE per workgroup (an L2
                // round trip between two barriers of every taper otherwise), the rest by multiplication
                cd w = wdit, step = wdit_step;
                spy_opaque2(w.x, w.y);                          // (keeps the powers inside the taper loop, see d64_pass)
#pragma unroll
                for (int e = 0; e < V; ++e) {
                    v[e] = spywil::cmul(v[e], w);
                    if (e + 1 < V) w = spywil::cmul(w, step);
                }
            }
            __syncthreads();              // (the last exchange's reads are done everywhere)
            if (active) {
#pragma unroll
                for (int e = 0; e < V; ++e) L[region + C::idx(j + T * e, h)] = v[e];
            }
            __syncthreads();
            // No second exchange: the epilogue forms the bins of its mapping, f = je + TT e, AND their partners N - f straight
            // from the three regions (d64_dit3_bin).  Against a per-thread combine (q = r) + natural-order write + read-back:
            // 3000 35.2 -> 24.5, 6000 79 -> 49, 3072 58 -> 16.6, 768 11.5 -> 3.3 us/trial (the 16-value schedules spilled)
        }

        // ---- separation: partner bin N - f lives in the upper slots
        double zpx[C::SPLIT ? HV : 1];
        {
            const int wb = C::idx(j, h);
            if constexpr (P > 1) {
                // (all N bins sit in LDS in natural order already)
            } else if constexpr (!C::SPLIT) {
                cd* L = reinterpret_cast<cd*>(lds);
                __syncthreads();              // the FFT's last reads of the buffer are done everywhere
                if (active) {
#pragma unroll
                    for (int e = HV; e < V; ++e) L[wb + e * C::ESTRIDE] = v[e];
                }
                __syncthreads();
            } else {
                double* L = reinterpret_cast<double*>(lds);
                __syncthreads();
                if (active) {
#pragma unroll
                    for (int e = HV; e < V; ++e) L[wb + e * C::ESTRIDE] = v[e].x;
                }
                __syncthreads();
#pragma unroll
                for (int e = 0; e < HV; ++e) {
                    const int f = j + T * e;
                    zpx[e] = f == 0 ? v[0].x : L[C::idx(N - f, h)];
                }
                __syncthreads();
                if (active) {
#pragma unroll
                    for (int e = HV; e < V; ++e) L[wb + e * C::ESTRIDE] = v[e].y;
                }
                __syncthreads();
            }
        }
        char* const slab = reinterpret_cast<char*>(a.out) +
                           ((size_t)b * kout + (MEAN ? 0 : k)) * (size_t)a.nfsel * (size_t)a.nchan * OSZ;
#pragma unroll
        for (int e = 0; e <= HV; ++e) {
            int f;
            cd X0, X1;
            if (e < HV) {
                if (!active) break;
                f = je + TT * e;
                cd z = v[e];
                cd p;
                if constexpr (P == 3) {
                    // N - f = (M - k) + M (2 - q), or M (3 - q) for k = 0
                    const int q = (f >= C::M ? 1 : 0) + (f >= 2 * C::M ? 1 : 0), kk = f - q * C::M;
                    z = d64_dit3_bin<C>(lds, kk, q, h);
                    p = f == 0 ? z : d64_dit3_bin<C>(lds, kk == 0 ? 0 : C::M - kk, kk == 0 ? 3 - q : 2 - q, h);
                } else if constexpr (!C::SPLIT) {
                    p = f == 0 ? z : reinterpret_cast<const cd*>(lds)[C::idx(N - f, h)];
                } else {
                    p = make_double2(zpx[C::SPLIT ? e : 0], f == 0 ? z.y : reinterpret_cast<const double*>(lds)[C::idx(N - f, h)]);
                }
                X0 = make_double2(0.5 * (z.x + p.x), 0.5 * (z.y - p.y));
                X1 = make_double2(0.5 * (z.y + p.y), 0.5 * (p.x - z.x));
                if constexpr (HALF) {
                    // X0 = E, X1 = O of the channel's real transform: bins f and N - f (f = 0: DC and the Nyquist bin)
                    const cd t = spywil::cmul(X1, reinterpret_cast<const cd*>(fa.tw64_full)[f]);
                    X1 = make_double2(X0.x - t.x, t.y - X0.y);
                    X0 = make_double2(X0.x + t.x, X0.y + t.y);
                }
            } else {
                if (je != 0 || !active) break;
                f = N / 2;
                const cd zn = (P == 3) ? d64_dit3_bin<C>(lds, C::M / 2, 1, h) : v[HV];
                X0 = make_double2(zn.x, HALF ? -zn.y : 0.0);        // (HALF: the middle bin is its own partner, X = conj Z)
                X1 = make_double2(HALF ? 0.0 : zn.y, 0.0);
            }
            // complex64 storage, then the float32 normalisation factor (mtmfft.py:104,117-127)
            const float2 s0 = make_float2(__fmul_rn((float)X0.x, a.scale), __fmul_rn((float)X0.y, a.scale));
            const float2 s1 = make_float2(__fmul_rn((float)X1.x, a.scale), __fmul_rn((float)X1.y, a.scale));
            if constexpr (MREG) {
                if constexpr (CPLX) {
                    macc0[e] += s0.x; mim0[e] += s0.y;
                    macc1[e] += s1.x; mim1[e] += s1.y;
                } else {
                    macc0[e] += convert_real<OUTK>(s0, a.out_kind);
                    macc1[e] += convert_real<OUTK>(s1, a.out_kind);
                }
                continue;
            }
            if (MEAN) {
                const int fi = a.fpos ? a.fpos[f] : f;
                if (fi < 0) continue;
                const size_t o = ((size_t)fi * a.nchan + c0) * OSZ;
                const bool first = k == 0, last = k == a.ntaper - 1;
                const float nt = (float)a.ntaper;
                if (CPLX) {
                    float2* q = reinterpret_cast<float2*>(slab + o);
                    float2 v0 = s0, v1 = s1;
                    if (!first) {
                        if (has0) { v0.x += q[0].x; v0.y += q[0].y; }
                        if (has1) { v1.x += q[1].x; v1.y += q[1].y; }
                    }
                    if (last) { v0.x /= nt; v0.y /= nt; v1.x /= nt; v1.y /= nt; }
                    if (has0) q[0] = v0;
                    if (has1) q[1] = v1;
                } else {
                    float* q = reinterpret_cast<float*>(slab + o);
                    float v0 = convert_real<OUTK>(s0, a.out_kind), v1 = convert_real<OUTK>(s1, a.out_kind);
                    if (!first) {
                        if (has0) v0 += q[0];
                        if (has1) v1 += q[1];
                    }
                    if (last) { v0 /= nt; v1 /= nt; }
                    if (has0) q[0] = v0;
                    if (has1) q[1] = v1;
                }
                continue;
            }
            if constexpr (HALF) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    if ((q == 1 && e == HV) || !has0) break;
                    const int fb = q ? N - f : f;
                    const int fi = a.fpos ? a.fpos[fb] : fb;
                    if (fi < 0) continue;
                    const size_t o = ((size_t)fi * a.nchan + c0) * OSZ;
                    const float2 sq = q ? s1 : s0;
                    if (CPLX) *reinterpret_cast<float2*>(slab + o) = sq;
                    else *reinterpret_cast<float*>(slab + o) = convert_real<OUTK>(sq, a.out_kind);
                }
                continue;
            }
            if (fast) {
                const size_t o = ((size_t)f * a.nchan + c0) * OSZ;
                if (CPLX) *reinterpret_cast<float4*>(slab + o) = make_float4(s0.x, s0.y, s1.x, s1.y);
                else *reinterpret_cast<float2*>(slab + o) = make_float2(convert_real<OUTK>(s0, a.out_kind),
                                                                        convert_real<OUTK>(s1, a.out_kind));
                continue;
            }
            const int fi = a.fpos ? a.fpos[f] : f;
            if (fi < 0) continue;
            const size_t o = ((size_t)fi * a.nchan + c0) * OSZ;
            if (CPLX) {
                if (has0) *reinterpret_cast<float2*>(slab + o) = s0;
                if (has1) *reinterpret_cast<float2*>(slab + o + 8) = s1;
            } else {
                if (has0) *reinterpret_cast<float*>(slab + o) = convert_real<OUTK>(s0, a.out_kind);
                if (has1) *reinterpret_cast<float*>(slab + o + 4) = convert_real<OUTK>(s1, a.out_kind);
            }
        }
        // no barrier here: the next taper's first LDS write sits behind one (d64_exchange / block_sum)
    }

    if constexpr (MREG) {
        char* const slab = reinterpret_cast<char*>(a.out) + (size_t)b * (size_t)a.nfsel * (size_t)a.nchan * OSZ;
        const float nt = (float)a.ntaper;
#pragma unroll
        for (int e = 0; e <= HV; ++e) {
            if (!active || (e == HV && j0 != 0)) break;
            const int f = e < HV ? j0 + TT * e : N / 2;
            if constexpr (HALF) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    if ((q == 1 && e == HV) || !has0) break;
                    const int fb = q ? N - f : f;
                    const int fi = a.fpos ? a.fpos[fb] : fb;
                    if (fi < 0) continue;
                    const size_t o = ((size_t)fi * a.nchan + c0) * OSZ;
                    if constexpr (CPLX) *reinterpret_cast<float2*>(slab + o) = make_float2((q ? macc1[e] : macc0[e]) / nt, (q ? mim1[e] : mim0[e]) / nt);
                    else *reinterpret_cast<float*>(slab + o) = (q ? macc1[e] : macc0[e]) / nt;
                }
                continue;
            }
            const int fi = a.fpos ? a.fpos[f] : f;
            if (fi < 0) continue;
            const size_t o = ((size_t)fi * a.nchan + c0) * OSZ;
            if constexpr (CPLX) {
                if (has0) *reinterpret_cast<float2*>(slab + o) = make_float2(macc0[e] / nt, mim0[e] / nt);
                if (has1) *reinterpret_cast<float2*>(slab + o + 8) = make_float2(macc1[e] / nt, mim1[e] / nt);
            } else {
                if (has0) *reinterpret_cast<float*>(slab + o) = macc0[e] / nt;
                if (has1) *reinterpret_cast<float*>(slab + o + 4) = macc1[e] / nt;
            }
        }
    }
}

}  // namespace spyfft
