// K1d: the packed tapered FFT with COMPILE-TIME radix schedules and V values per thread - decimal trial lengths
// (1000 = 10 x 10 x 10, 2000 = 10 x 10 x 10 x 2, 5000 = 10 x 10 x 10 x 5: 1 kHz x 1 / 2 / 5 s; BASELINE config 1 is
// N = 2000) and, with V = 8, a high-occupancy variant of the power-of-two lengths.
//
// Reference semantics: specest/mtmfft.py:16-129 + specest/compRoutines.py:169-189 (and specest/stft.py:101-154 when one
// segment = one STFT frame) - the same argument block, options and output layouts as mtmfft_quad_kernel.
//
// Structure of the packed power-of-two kernel (mtmfft2_kernel.h) with its radix-16 network generalised: N = V R1 R2 R3,
// T = N / V threads per channel quad; thread j keeps the samples n = j + T e (e < V) of its FOUR channels in registers
// across all tapers; every Stockham pass of radix R (R divides V, V / R butterflies per thread) reads in[j + T e] and
// writes out[(b / Ns) Ns R + b % Ns + r Ns] - with immediates instead of the run-time index arithmetic of the
// mixed-radix engine (mtmfft_mixed.h: 2.1x the vector instructions and 2.9x the LDS cycles of the power-of-two kernel
// per sample, profiles/r2_pmc_mixed2000_*); the first pass takes the tapered samples straight from the registers and
// the last pass leaves bin j + T e in register e, so a transform of P passes makes P - 1 exchanges through LDS
// (the mixed-radix engine: 2 P + 3 array passes).  With V = 10 a thread holds 40 + 40 data registers instead of the
// 64 + 64 of the radix-16 kernel: four waves per SIMD instead of two.
#pragma once
#include "fft2_device.h"
#include "mtmfft_kernel.h"
#include "mtmfft_mixed.h"      // dft3p / dft5p / dft_pq: the composite butterflies

namespace spyfft {

template <int R>
__device__ __forceinline__ void dec_dft(C2 (&t)[R]) {
    if constexpr (R == 2) dft2p(t);
    else if constexpr (R == 4) dft4p(t);
    else if constexpr (R == 5) dft5p(t);
    else if constexpr (R == 8) dft8p(t);
    else if constexpr (R == 10) dft_pq<2, 5>(t);
    else if constexpr (R == 16) dft16p(t);
    else if constexpr (R == 20) dft_pq<4, 5>(t);
    else static_assert(R == 2, "radix of the compile-time schedules: 2, 4, 5, 8, 10, 16, 20");
}

template <int V_, int R1_, int R2_, int R3_, int G_, int P_ = 1, bool SPLIT_ = false, bool HALF_ = false>
struct CfgD {
    static constexpr int V = V_, R1 = R1_, R2 = R2_, R3 = R3_, G = G_;
    // P = 3: radix-3 decimation in time in FRONT of the schedule - N = 3 M: three groups of T threads transform the
    // sub-sequences x[3 n + r] (length M = V R1 R2 R3, one LDS region each) side by side, then one combine through LDS:
    // X[k + M q] = sum_r w_3^(r q) W_N^(r k) F_r[k]  (3000 = 3 x 1000, 6000 = 3 x 2000, 1500, 7500, 600: lengths no
    // radix dividing 10 or 20 ends; mtmfft_dec64_kernel.h carries the same wrapper)
    static constexpr int P = P_;
    // SPLIT: an exchange moves the real parts and then the imaginary parts of the packed values through ONE 8-byte plane
    // (half the LDS, twice the barriers): N = 10000 needs it to fit at all, N = 5000 to hold two workgroups per CU
    static constexpr bool SPLIT = SPLIT_;
    // HALF: a real transform of 2 N samples through the complex length-N schedule.  A thread set carries channel PAIRS
    // instead of quads: z[m] = x[2 m] + i x[2 m + 1] (the two channels in the packed halves), and the epilogue turns
    // Z[f], Z[N - f] into the bins f AND N - f of the length-2N real transform: E = (Z[f] + conj Z[N - f]) / 2,
    // O = (Z[f] - conj Z[N - f]) / 2i, X[f] = E + W O, X[N - f] = conj(E - W O), W = exp(-2 pi i f / 2N).  Same flops per
    // channel as the two-channels-per-complex-transform packing, HALF the LDS per workgroup: trials of 10240 < nfft <=
    // 20480 samples stay in LDS (before: decimation in time through HBM, mtmfft_declong.h: a scratch round trip of
    // nfft packed values per taper - 16 x the algorithmic traffic at nfft = 12000)
    static constexpr bool HALF = HALF_;
    static constexpr int M = V * R1 * R2 * R3;               // length of one sub-transform (= N without decimation)
    static constexpr int N = P * M;
    static constexpr int T = M / V;                          // threads per sub-transform
    static constexpr int TT = P * T;                         // threads per channel quad
    static constexpr int NPASS = 2 + (R2 > 1 ? 1 : 0) + (R3 > 1 ? 1 : 0);
    static constexpr int NTHREADS = ((TT * G + 63) / 64) * 64;
    static constexpr int PL1 = M + M / V + 1;                // float4 units of one sub-transform: one pad per V values
    static constexpr int PLANE = P * PL1;                    // ... per quad
    static constexpr int ESTRIDE = (T + T / V) * G;          // LDS distance of e -> e + 1
    static constexpr size_t LDS_BYTES = (size_t)PLANE * G * (SPLIT ? 8 : 16);
    // one workgroup per CU (nothing else hides the twiddle reads): base twiddles requested a pass ahead (DecTw)
    static constexpr bool TWAHEAD = LDS_BYTES > 80 * 1024;
    static_assert(R1 > 1 && V % R1 == 0 && V % R2 == 0 && V % R3 == 0, "every radix divides the values per thread");
    static_assert(T % V == 0, "T multiple of V: idx(j + T e) stays affine in e");
    static_assert(NTHREADS <= 1024 && (64 % G) == 0, "workgroup shape");
    static_assert(P == 1 || (P == 3 && !SPLIT), "decimation: radix 3, plain exchanges");
    static_assert(LDS_BYTES <= 160 * 1024, "one workgroup's LDS");
    __device__ static __forceinline__ int idx(int i, int h) { return (i + i / V) * G + h; }
};

// Twiddles of one pass for thread j: butterfly m = j + T m needs w^r, w = tw[k N / (Ns R)], r < R.  As in the power-of-two
// engine (fft2_device.h: Tw6) only the BASE powers come from the table - w, w^2, w^3 and w^4, w^8, ... - and are requested
// one pass ahead (DecTw::load sits in front of the previous pass's exchange, whose barriers hide the L2 round trip); the
// other powers are products w^(4a) w^l.  Measured against R - 1 table reads per butterfly right before its use
// (tools/dec_probe.hip, reads switched off as the bound): 8192 25.1 (bound 18.4), 4096 8.6 (6.8), 2000 4.9 (4.3) us/trial.
__device__ __forceinline__ float2 dec_tw_read(const float2* __restrict__ tw, unsigned byte_offset) {
#if defined(SPYFFT_ABL) && (SPYFFT_ABL & 2)
    return make_float2(__uint_as_float(byte_offset), 0.5f);      // (tools/dec_probe.hip: the kernel without its table reads)
#else
    return ldg<float2>(tw, byte_offset);
#endif
}

template <class C, int R, int Ns, bool AHEAD = C::TWAHEAD>
struct DecTw {
    static constexpr int MB = C::V / R;
    static constexpr int NLO = (R - 1 < 3) ? R - 1 : 3;
    static constexpr int NHI = (R + 3) / 4 - 1;
    float2 lo[MB][NLO > 0 ? NLO : 1];
    float2 hi[MB][NHI > 0 ? NHI : 1];
    __device__ __forceinline__ void load(int j, const float2* __restrict__ tw) {
#pragma unroll
        for (int m = 0; m < MB; ++m) {
            const int b = j + C::T * m;
            const int k = b - (b / Ns) * Ns;
            const unsigned kb = (unsigned)(k * (C::N / (Ns * R))) * 8u;          // byte offset of tw[k N / (Ns R)]
#pragma unroll
            for (int l = 1; l <= NLO; ++l)
                lo[m][l - 1] = dec_tw_read(tw, kb * (unsigned)l);
#pragma unroll
            for (int a = 1; a <= NHI; ++a)
                hi[m][a - 1] = dec_tw_read(tw, kb * (unsigned)(4 * a));
        }
    }
    __device__ __forceinline__ float2 w(int m, int r) const {        // (r is a compile-time constant once unrolled)
        const int h = r >> 2, l = r & 3;
        if (h == 0) return lo[m][l - 1];
        if (l == 0) return hi[m][h - 1];
        return cmul(hi[m][h - 1], lo[m][l - 1]);
    }
};
// ... where several workgroups share a CU the other workgroups hide that latency already and the registers of the early
// request cost occupancy (2000: three -> two workgroups per CU, 4.9 -> 5.3 us/trial): R - 1 table reads at the point of use
template <class C, int R, int Ns>
struct DecTw<C, R, Ns, false> {
    const float2* tw;
    int j;
    __device__ __forceinline__ void load(int j_, const float2* __restrict__ tw_) { j = j_; tw = tw_; }
    __device__ __forceinline__ float2 w(int m, int r) const {
        const int b = j + C::T * m;
        const int k = b - (b / Ns) * Ns;
        const unsigned kb = (unsigned)(k * (C::N / (Ns * R))) * 8u;
        return dec_tw_read(tw, kb * (unsigned)r);
    }
};
struct DecTwNone {           // the first pass has no twiddles; the last pass has no successor to fetch for
    __device__ __forceinline__ void load(int, const float2*) {}
};

// One pass.  In: v[e] = in[j + T e] (pass 0: the tapered samples).  Out: LAST - v[e] = X[j + T e] in registers;
// otherwise the outputs go through LDS and v[e] = out[j + T e] comes back.
template <class C, int R, int Ns, bool FIRST, bool LAST, class TwNow, class TwNext>
__device__ __forceinline__ void dec_pass(C2 (&v)[C::V], float4* lds, int j, int h, bool active,
                                         const float2* __restrict__ tw, const TwNow& now, TwNext& next, int region = 0) {
    constexpr int V = C::V, N = C::N, T = C::T, G = C::G, MB = V / R;      // (tw has N entries: W_M^k = tw[P k])
    int wbase[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) {
        const int b = j + T * m;
        const int q = b / Ns, k = b - q * Ns;
        C2 u[R];
#pragma unroll
        for (int r = 0; r < R; ++r) u[r] = v[m + MB * r];
        if constexpr (!FIRST) {
#pragma unroll
            for (int r = 1; r < R; ++r) u[r] = cmul_s(u[r], now.w(m, r));
        }
        dec_dft<R>(u);
#pragma unroll
        for (int r = 0; r < R; ++r) v[m + MB * r] = u[r];
        // LDS slot of output r: idx(q Ns R + k + r Ns); Ns is 1 (first pass, R = V) or a multiple of V
        wbase[m] = region + (FIRST ? (b * (V + 1)) * G + h : (q * (Ns * R + Ns * R / V) + k + k / V) * G + h);
    }
    if (LAST) return;
    next.load(j, tw);                 // the next pass's base twiddles: in flight across this pass's exchange
    constexpr int WS = FIRST ? G : (Ns + Ns / V) * G;
    const int rb = region + C::idx(j, h);
    if constexpr (!C::SPLIT) {
        __syncthreads();              // (write-after-read: earlier reads of the buffer by any thread are done)
        if (active) {
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int r = 0; r < R; ++r) {
                    const C2 t = v[m + MB * r];
                    lds[wbase[m] + r * WS] = make_float4(t.r[0], t.r[1], t.i[0], t.i[1]);
                }
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const float4 t = lds[rb + e * C::ESTRIDE];
            v[e].r = v2f{t.x, t.y};
            v[e].i = v2f{t.z, t.w};
        }
    } else {
        float2* const L = reinterpret_cast<float2*>(lds);
        __syncthreads();
        if (active) {
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int r = 0; r < R; ++r) L[wbase[m] + r * WS] = make_float2(v[m + MB * r].r[0], v[m + MB * r].r[1]);
        }
        __syncthreads();
        v2f re[V];
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const float2 t = L[rb + e * C::ESTRIDE];
            re[e] = v2f{t.x, t.y};
        }
        __syncthreads();
        if (active) {
#pragma unroll
            for (int m = 0; m < MB; ++m)
#pragma unroll
                for (int r = 0; r < R; ++r) L[wbase[m] + r * WS] = make_float2(v[m + MB * r].i[0], v[m + MB * r].i[1]);
        }
        __syncthreads();
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const float2 t = L[rb + e * C::ESTRIDE];
            v[e].r = re[e];
            v[e].i = v2f{t.x, t.y};
        }
    }
}

// The passes of one scheduled transform: radix V from the registers, then R1 (R2, R3); the last one leaves v[e] = Z[j + T e]
template <class C>
__device__ __forceinline__ void dec_transform(C2 (&v)[C::V], float4* lds, int j, int h, bool active,
                                              const float2* __restrict__ tw, int region = 0) {
    constexpr int V = C::V;
    DecTwNone none;
    DecTw<C, C::R1, V> t1;
    dec_pass<C, V, 1, true, false>(v, lds, j, h, active, tw, none, t1, region);
    if constexpr (C::NPASS == 2) {
        dec_pass<C, C::R1, V, false, true>(v, lds, j, h, active, tw, t1, none, region);
    } else {
        DecTw<C, C::R2, V * C::R1> t2;
        dec_pass<C, C::R1, V, false, false>(v, lds, j, h, active, tw, t1, t2, region);
        if constexpr (C::NPASS == 3) {
            dec_pass<C, C::R2, V * C::R1, false, true>(v, lds, j, h, active, tw, t2, none, region);
        } else {
            DecTw<C, C::R3, V * C::R1 * C::R2> t3;
            dec_pass<C, C::R2, V * C::R1, false, false>(v, lds, j, h, active, tw, t2, t3, region);
            dec_pass<C, C::R3, V * C::R1 * C::R2, false, true>(v, lds, j, h, active, tw, t3, none, region);
        }
    }
}

// (P = 3) bin k + M q of the length-3M transform from the twiddled sub-transforms in the three LDS regions
template <class C>
__device__ __forceinline__ C2 dit3_bin(const float4* lds, int k, int q, int h) {
    const int ki = C::idx(k, h);
    C2 g[3];
#pragma unroll
    for (int rr = 0; rr < 3; ++rr) {
        const float4 t = lds[rr * C::PL1 * C::G + ki];
        g[rr].r = v2f{t.x, t.y};
        g[rr].i = v2f{t.z, t.w};
    }
    const float ca = q == 0 ? 1.f : -0.5f;
    const float cc = q == 0 ? 0.f : (q == 1 ? 0.8660254037844386f : -0.8660254037844386f);
    const v2f sr = g[1].r + g[2].r, si = g[1].i + g[2].i, dr = g[1].r - g[2].r, di = g[1].i - g[2].i;
    C2 x;
    x.r = g[0].r + sr * ca + di * cc;
    x.i = g[0].i + si * ca - dr * cc;
    return x;
}

// OUTK: 0 = power (inlined), 1 = any other real conversion, 2 = complex; MEAN: average over tapers
template <class C, int OUTK, bool MEAN>
__global__ void __launch_bounds__((C::NTHREADS)) SPYFFT_KATTR mtmfft_dec_kernel(MtmArgs a) {
    constexpr bool CPLX = (OUTK == 2);
    constexpr int V = C::V, N = C::N, T = C::T, TT = C::TT, P = C::P, G = C::G, HV = V / 2;
    constexpr bool HALF = C::HALF;
    constexpr int CW = HALF ? 2 : 4;              // channels of a thread set
    SPY_DYN_SMEM(float4, lds);

    const int tid = threadIdx.x;
    const int h = tid % G, jt = tid / G;
    const bool active = jt < TT;                  // the workgroup is padded to whole waves
    const int j0 = active ? jt : 0;               // thread of the quad: bins j0 + TT e in the epilogue
    // decimation (P = 3): group r0 = j0 / T transforms the samples 3 n + r0, js0 = its thread index inside the group;
    // thread j0 holds the SAMPLES jn0 + TT e with jn0 = P js0 + r0 (P = 1: jn0 = js0 = j0, TT = T)
    const int r0 = j0 / T, js0 = j0 - r0 * T, jn0 = P * js0 + r0;

    // XCD-aware block -> (segment, quad group), as mtmfft_quad_kernel
    const long long id = blockIdx.x;
    const int xcd = (int)(id & 7);
    const long long y = id >> 3;
    const long long nclt = (long long)a.nseg * a.ncl, chunk = (nclt + 7) >> 3;
    const long long cidx = (long long)xcd * chunk + y / a.S;
    const int q = (int)(y % a.S);
    if (cidx >= nclt) return;
    const int b = (int)(cidx / a.ncl);
    const int pg = (int)(cidx % a.ncl) * a.S + q;
    if (pg >= a.npg) return;

    const int c0 = CW * (pg * G + h);
    bool has[4];
    unsigned col[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        has[i] = active && i < CW && c0 + i < a.nchan;
        col[i] = has[i] ? (unsigned)(a.chan_idx ? a.chan_idx[c0 + i] : c0 + i) : 0u;
    }
    const bool full = has[CW - 1];
    const long long start = a.seg_start[b];
    const long long rl = a.seg_lo[b] - start, rh = a.seg_hi[b] - start;
    const int rlo = (int)(rl < 0 ? 0 : (rl > a.nsig ? a.nsig : rl));
    const int rhi = (int)(rh < 0 ? 0 : (rh > a.nsig ? a.nsig : rh));
    const unsigned rowb = (unsigned)a.ld * 4u;        // bytes per row
    const float* seg = a.data + start * a.ld;         // wave-uniform; only rows in [rlo, rhi) are dereferenced

    // ---- load the segment once: x[e] = sample n = jn0 + TT*e; r = (c0, c1), i = (c2, c3)
    C2 x[V];
    if constexpr (HALF) {
        // sample pairs (2 m, 2 m + 1), m = jn0 + TT e: the even sample in .r, the odd one in .i, channels (c0, c1) in the halves
        if (a.xpair != nullptr) {
            // ... from the pair-major copy of the segment (pair_stage_kernel): 16 contiguous bytes = both samples
            const float2* xs = a.xpair + ((size_t)b * ((a.nchan + 1) >> 1) + (size_t)(c0 >> 1)) * (size_t)a.xstride;
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const long long n0 = 2LL * (jn0 + TT * e);
                float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
                if (active && n0 < a.xstride) t = *reinterpret_cast<const float4*>(xs + n0);
                x[e].r = v2f{t.x, t.y};
                x[e].i = v2f{t.z, t.w};
            }
        } else if (rhi > rlo) {
            const bool vec2 = (a.chan_idx == nullptr) && full && ((a.ld & 1) == 0) &&
                              ((reinterpret_cast<size_t>(a.data) & 7) == 0);
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const int n0 = 2 * (jn0 + TT * e), n1 = n0 + 1;
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
            for (int e = 0; e < V; ++e) x[e].r = x[e].i = splat(0.f);
        }
    } else if (rhi > rlo) {
        const bool vec4 = (a.chan_idx == nullptr) && full && ((a.ld & 3) == 0) &&
                          ((reinterpret_cast<size_t>(a.data) & 15) == 0);
        if (vec4) {
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const int n = jn0 + TT * e;
                const int nc = min(max(n, rlo), rhi - 1);
                const float4 t = ldg<float4>(seg, (unsigned)nc * rowb + col[0] * 4u);
                const bool ok = (n == nc);
                x[e].r = v2f{ok ? t.x : 0.f, ok ? t.y : 0.f};
                x[e].i = v2f{ok ? t.z : 0.f, ok ? t.w : 0.f};
            }
        } else {
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const int n = jn0 + TT * e;
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
        for (int e = 0; e < V; ++e) x[e].r = x[e].i = splat(0.f);
    }

    // ---- polynomial removal over the nsig samples (float64 sums, branch-free; constant: the reference-order means)
    if (a.detrend == 0 && a.means) {
        const float* mp = a.means + (size_t)b * a.nchan + c0;
        float f[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) f[i] = has[i] ? mp[i] : 0.f;
        const v2f mr = v2f{f[0], f[1]}, mi = HALF ? mr : v2f{f[2], f[3]};
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const int n0 = HALF ? 2 * (jn0 + TT * e) : jn0 + TT * e;
            x[e].r -= n0 < a.nsig ? mr : splat(0.f);
            x[e].i -= n0 + (HALF ? 1 : 0) < a.nsig ? mi : splat(0.f);
        }
    } else if (HALF && a.detrend >= 0) {
        // (HALF) the float64 sums of the two channels run over the even AND the odd samples
        const float mid = 0.5f * (float)(a.nsig - 1);
        double s[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const int n0 = 2 * (jn0 + TT * e), n1 = n0 + 1;
            const float m0 = (active && n0 < a.nsig) ? 1.f : 0.f, m1 = (active && n1 < a.nsig) ? 1.f : 0.f;
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
        for (int e = 0; e < V; ++e) {
            const int n0 = 2 * (jn0 + TT * e), n1 = n0 + 1;
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
        for (int e = 0; e < V; ++e) {
            const int n = jn0 + TT * e;
            const float m = (active && n < a.nsig) ? 1.f : 0.f;
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
            for (int e = 0; e < V; ++e) {
                const int n = jn0 + TT * e;
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
            for (int e = 0; e < V; ++e) {
                const bool in = jn0 + TT * e < a.nsig;
                x[e].r -= in ? mr : splat(0.f);
                x[e].i -= in ? mi : splat(0.f);
            }
        }
    }

    // accumulators for the taper mean (bins e < V/2 plus the Nyquist bin on j == 0)
    C2 ma[MEAN ? HV + 1 : 1], mb[(MEAN && CPLX) ? HV + 1 : 1];
    if (MEAN) {
#pragma unroll
        for (int e = 0; e <= HV; ++e) {
            ma[e].r = ma[e].i = splat(0.f);
            if (CPLX) mb[e].r = mb[e].i = splat(0.f);
        }
    }
    const int kout = MEAN ? 1 : a.ntaper;
    const float hs = 0.5f * a.scale;
    const unsigned nsig_m1 = (unsigned)(a.nsig - 1);
    constexpr unsigned OSZ = CPLX ? 8u : 4u;   // bytes per output element
    const bool fast = full && (a.fpos == nullptr) && ((reinterpret_cast<size_t>(a.out) & 15) == 0) &&
                      ((a.nchan & ((CPLX || HALF) ? 1 : 3)) == 0);

    for (int k = 0; k < a.ntaper; ++k) {
        const int je = opaque(j0);
        const int r = je / T, j = je - r * T, jn = P * j + r;        // group, thread inside it, first sample (P = 1: jn = j = je)
        const int region = r * C::PL1 * G;
        const float* w = a.tapers + (size_t)k * a.nsig;   // wave-uniform
        C2 v[V];
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const unsigned n = (unsigned)(jn + TT * e) * (HALF ? 2u : 1u);
            const float wl = ldg<float>(w, min(n, nsig_m1) * 4u);
            const float wn = (n <= nsig_m1) ? wl : 0.f;
            float wo = wn;
            if constexpr (HALF) {
                const float wl1 = ldg<float>(w, min(n + 1u, nsig_m1) * 4u);
                wo = (n + 1u <= nsig_m1) ? wl1 : 0.f;
            }
            v[e].r = x[e].r * wn;
            v[e].i = x[e].i * wo;
        }
        if (a.demean_taper) {
            __syncthreads();          // block_sum writes its scratch into the buffer other waves may still be reading
            double s[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int e = 0; e < V; ++e) {
                s[0] += v[e].r[0];
                s[1] += v[e].r[1];
                s[HALF ? 0 : 2] += v[e].i[0];
                s[HALF ? 1 : 3] += v[e].i[1];
            }
            if (!active) s[0] = s[1] = s[2] = s[3] = 0.0;
            block_sum<C::NTHREADS, G, 4>(s, reinterpret_cast<double*>(lds), tid, h);
            const v2f mr = v2f{(float)(s[0] / a.nsig), (float)(s[1] / a.nsig)};
            const v2f mi = HALF ? mr : v2f{(float)(s[2] / a.nsig), (float)(s[3] / a.nsig)};
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const int n0 = (jn + TT * e) * (HALF ? 2 : 1);
                v[e].r -= n0 < a.nsig ? mr : splat(0.f);
                v[e].i -= n0 + (HALF ? 1 : 0) < a.nsig ? mi : splat(0.f);
            }
        }

        C2 zpart[C::SPLIT ? HV : 1];      // (SPLIT, P = 3) the partner bins Z[N - f] of this thread's bins
        // ---- the passes: radix V from the registers, then R1 (R2, R3); the last one leaves v[e] = Z[j + T e]
        dec_transform<C>(v, lds, j, h, active, a.tw, region);

        if constexpr (P == 3) {
            // ---- radix-3 combine of the three sub-transforms: v[e] = F_r[k], k = j + T e  ->  X[k + M r]
            if (r > 0) {
#pragma unroll
                for (int e = 0; e < V; ++e) v[e] = cmul_s(v[e], ldg<float2>(a.tw, (unsigned)(r * (j + T * e)) * 8u));   // W_N^(r k)
            }
            __syncthreads();              // (the last exchange's reads are done everywhere)
            if (active) {
#pragma unroll
                for (int e = 0; e < V; ++e)
                    lds[region + C::idx(j + T * e, h)] = make_float4(v[e].r[0], v[e].r[1], v[e].i[0], v[e].i[1]);
            }
            __syncthreads();
            // No second exchange: the epilogue below forms the bins of its mapping, f = je + TT e (e < V / 2, and N / 2 on
            // je = 0), AND their partners N - f straight from the three regions (dit3_bin).  Measured against a version that
            // combined per thread (q = r), wrote X in natural order and read bins and partners back: 3000 14.6 -> 13.1,
            // 6000 33.9 -> 27.4, 7500 40.6 -> 35.6 us/trial (profiles/r4_precision_probe_dit3.txt)
        } else if constexpr (!C::SPLIT) {
            // ---- separate the real channels: partner bin N - f lives in the upper half
            __syncthreads();              // the FFT's last reads of the buffer are done everywhere
            if (active) {
                const int wb = C::idx(j, h);
#pragma unroll
                for (int e = HV; e < V; ++e) lds[wb + e * C::ESTRIDE] = make_float4(v[e].r[0], v[e].r[1], v[e].i[0], v[e].i[1]);
            }
            __syncthreads();
        } else {
            // ... through the 8-byte plane: the partners' real parts, then their imaginary parts
            float2* const L = reinterpret_cast<float2*>(lds);
            const int wb = C::idx(j, h);
            __syncthreads();
            if (active) {
#pragma unroll
                for (int e = HV; e < V; ++e) L[wb + e * C::ESTRIDE] = make_float2(v[e].r[0], v[e].r[1]);
            }
            __syncthreads();
#pragma unroll
            for (int e = 0; e < HV; ++e) {
                const int f = j + T * e;
                const float2 t = L[C::idx(f == 0 ? N / 2 : N - f, h)];
                zpart[e].r = v2f{t.x, t.y};
            }
            __syncthreads();
            if (active) {
#pragma unroll
                for (int e = HV; e < V; ++e) L[wb + e * C::ESTRIDE] = make_float2(v[e].i[0], v[e].i[1]);
            }
            __syncthreads();
#pragma unroll
            for (int e = 0; e < HV; ++e) {
                const int f = j + T * e;
                const float2 t = L[C::idx(f == 0 ? N / 2 : N - f, h)];
                zpart[e].i = v2f{t.x, t.y};
            }
        }
        char* const slab = reinterpret_cast<char*>(a.out) +
                           ((size_t)b * kout + (MEAN ? 0 : k)) * (size_t)a.nfsel * (size_t)a.nchan * OSZ;
#pragma unroll
        for (int e = 0; e <= HV; ++e) {
            C2 xa, xb;   // xa = X(c0, c1), xb = X(c2, c3)
            int f;
            if (e < HV) {
                if (!active) break;
                f = je + TT * e;
                C2 z = v[e];
                if constexpr (P == 3) z = dit3_bin<C>(lds, f >= C::M ? f - C::M : f, f >= C::M ? 1 : 0, h);   // (f < N / 2)
                C2 zp = z;
                if (f != 0) {
                    if constexpr (P == 3) {
                        // N - f = (M - k) + M (2 - q), or M (3 - q) for k = 0
                        const int q = f >= C::M ? 1 : 0, kk = f - q * C::M;
                        zp = dit3_bin<C>(lds, kk == 0 ? 0 : C::M - kk, kk == 0 ? 3 - q : 2 - q, h);
                    } else if constexpr (C::SPLIT) {
                        zp = zpart[e];
                    } else {
                        const float4 t = lds[C::idx(N - f, h)];
                        zp.r = v2f{t.x, t.y};
                        zp.i = v2f{t.z, t.w};
                    }
                }
                xa.r = (z.r + zp.r) * hs;
                xa.i = (z.i - zp.i) * hs;
                xb.r = (z.i + zp.i) * hs;
                xb.i = (zp.r - z.r) * hs;
                if constexpr (HALF) {
                    // xa = E, xb = O of the pair's real transform: bins f and N - f (f = 0: DC and the Nyquist bin)
                    const C2 t = cmul_s(xb, ldg<float2>(a.twh, (unsigned)f * 8u));
                    const C2 d = csub(xa, t);
                    xa = cadd(xa, t);
                    xb.r = d.r;
                    xb.i = -d.i;
                }
            } else {
                if (je != 0 || !active) break;
                f = N / 2;
                const C2 zn = (P == 3) ? dit3_bin<C>(lds, C::M / 2, 1, h) : v[HV];
                xa.r = zn.r * a.scale;
                xb.r = zn.i * a.scale;
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
                    if (q == 1 && e == HV) break;
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
            if (fast) {
                const unsigned o = ((unsigned)f * (unsigned)a.nchan + (unsigned)c0) * OSZ;
                if (CPLX) {
                    stg<float4>(slab, o, make_float4(xa.r[0], xa.i[0], xa.r[1], xa.i[1]));
                    stg<float4>(slab, o + 16u, make_float4(xb.r[0], xb.i[0], xb.r[1], xb.i[1]));
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
        // no barrier here: the next taper's first LDS write sits behind one (dec_pass / block_sum)
    }

    if (MEAN) {
        char* const slab = reinterpret_cast<char*>(a.out) + (size_t)b * (size_t)a.nfsel * (size_t)a.nchan * OSZ;
        const float nt = (float)a.ntaper;
#pragma unroll
        for (int e = 0; e <= HV; ++e) {
            if (!active || (e == HV && j0 != 0)) break;
            const int f = (e < HV) ? j0 + TT * e : N / 2;
            if constexpr (HALF) {
#pragma unroll
                for (int q = 0; q < 2; ++q) {
                    if (q == 1 && e == HV) break;
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
        }
    }
}

}  // namespace spyfft
