// K1p (power-of-two lengths 1024 / 2048 / 4096): the packed-quad tapered FFT of mtmfft2_kernel.h as a TWO-STAGE
// SOFTWARE PIPELINE inside one workgroup.
//
// Reference semantics: specest/mtmfft.py:16-129 + specest/compRoutines.py:169-189 (and specest/stft.py:101-154 when
// one segment = one STFT frame) - identical to mtmfft_quad_kernel, which stays the kernel for the other lengths.
//
// Why: the timeline of mtmfft_quad_kernel (tools/fft_stamp_probe.hip, profiles/r3_k1_*) shows butterflies (vector
// pipe) and LDS exchanges strictly alternating inside a workgroup, and two independent workgroups per CU do not fall
// into opposite phases by themselves: the vector pipe was busy ~1/3 of the time, the LDS store path (61 B/clk/CU for
// 8/16-byte stores, measured) ~1/4, hardly ever together.  Here a workgroup is TWO halves of T = N/16 threads, each
// half owns one channel quad (four real channels = two packed complex FFTs, fft2_device.h) of the same segment, and
// half 1 runs exactly one slot behind half 0:
//
//      slot          s        s+1        s+2        s+3        s+4 ...
//      half 0    V2(k-1)+V0(k)  L0(k)     V1(k)      L1(k)   V2(k)+V0(k+1)
//      half 1       L1(k-1)  V2(k-1)+V0(k) L0(k)     V1(k)      L1(k)
//
//   V0 = taper multiply + radix-16 butterflies of pass 0      L0 = exchange through LDS (write | read)
//   V1 = twiddles + pass 1                                     L1 = exchange (write | read in the folded column order)
//   V2 = twiddles + last pass, channel separation, output conversion / stores / taper-mean accumulation
//
// so in every slot one half computes while the other moves data through LDS.  Every slot is [part a | barrier |
// part b | barrier] for both halves (the LDS half needs the barrier between its stores and its loads; s_barrier is
// workgroup-wide, the computing half simply passes it in the middle of its butterflies).  The halves have their own
// LDS planes; barriers order write -> read inside a half and pace the two halves against each other.
//
// Channel separation without LDS: the partner bin N-f of column j lives in column T-j.  The last pass runs in a FOLDED
// column order - lane l < 32 of wave w takes column 32w+l, lane l+32 takes column T-(32w+l) - so partners sit in lanes
// l and l+32 of one wave and v_permlane32_swap_b32 exchanges the upper eight values of the two lanes (32 VALU
// instructions instead of 16 LDS stores + 16 LDS loads + two barriers per taper).  Columns 0 and T/2 are their own
// partners: they share the lane pair (0, 32) of wave 0, keep their own values and take the partner from their own
// registers (column 0: bin T(16-e), column T/2: bin T/2 + T(15-e)).
#pragma once
#include "fft2_device.h"
#include "mtmfft_kernel.h"

#ifndef SPY_HOST_EMU
#define SPY_WAVES_PER_EU2 __attribute__((amdgpu_waves_per_eu(2)))   // at most 256 registers: two waves per SIMD
#else
#define SPY_WAVES_PER_EU2
#endif

namespace spyfft {

// exchange a value between lanes l and l ^ 32 of the wave
__device__ __forceinline__ float swap32(float x) {
#ifndef SPY_HOST_EMU
    asm("v_permlane32_swap_b32 %0, %0" : "+v"(x));
    return x;
#else
    return __shfl_xor(x, 32);
#endif
}
__device__ __forceinline__ v2f swap32(v2f x) { return v2f{swap32(x[0]), swap32(x[1])}; }
__device__ __forceinline__ C2 swap32(C2 x) { return C2{swap32(x.r), swap32(x.i)}; }
// a value that is the same in every lane of the wave, moved to a scalar register (branches on it are s_cbranch)
__device__ __forceinline__ int wave_uniform(int v) {
#ifndef SPY_HOST_EMU
    return __builtin_amdgcn_readfirstlane(v);
#else
    return v;
#endif
}
__device__ __forceinline__ int lane_swap1(int v) { return __float_as_int(lane_swap1(__int_as_float(v))); }

template <int LOG2N, int NH = 2>
struct CfgP {
    using C = Cfg2<LOG2N, 1>;
    static constexpr int N = C::N, T = C::T;
    static constexpr int NTHREADS = NH * T;
    static constexpr int NW = T / 64;                        // waves per half
    static constexpr int KMAX = 64;                          // tapers whose post-taper means fit the LDS tail
    static constexpr size_t HALF_UNITS = C::LDS_BYTES / 8;   // v2f units per half (two planes)
    static constexpr size_t LDS_BYTES = NH * C::LDS_BYTES + NH * KMAX * 16;
    static_assert(LOG2N >= 10 && LOG2N <= 12, "pipelined kernel: N = 1024, 2048, 4096");
    static_assert((C::LDS_BYTES % 16) == 0, "half regions stay 16-byte aligned");
};

// Sum NS doubles over the T threads of one half; result in every thread of the half.  Both halves call it together
// (it contains two workgroup barriers); `scratch` = the start of the dynamic LDS buffer (unused at that point).
template <int NS, int NW>
__device__ __forceinline__ void half_sum(double (&s)[NS], double* scratch, int half, int wv, int lane) {
#pragma unroll
    for (int i = 0; i < NS; ++i) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) s[i] += __shfl_xor(s[i], off);
    }
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < NS; ++i) scratch[(half * NW + wv) * NS + i] = s[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        double tot = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) tot += scratch[(half * NW + w) * NS + i];
        s[i] = tot;
    }
    __syncthreads();
}

// OUTK: 0 = power (inlined), 1 = any other real conversion, 2 = complex; MEAN: average over tapers
template <int LOG2N, int OUTK, bool MEAN, int NH = 2>
__global__ void __launch_bounds__((CfgP<LOG2N, NH>::NTHREADS)) SPY_WAVES_PER_EU2 SPYFFT_KATTR mtmfft_pipe_kernel(MtmArgs a) {
    using P = CfgP<LOG2N, NH>;
    using C = typename P::C;
    constexpr bool CPLX = (OUTK == 2);
    constexpr int N = P::N, T = P::T;
    SPY_DYN_SMEM(v2f, lds_all);

    const int tid = threadIdx.x;
    const int half = NH == 1 ? 0 : wave_uniform(tid / T);
    const int t_ = tid - half * T;                // thread of the half = column of passes 0 and 1
    const int t = t_;
    const int lane = t & 63;
    const int wv = wave_uniform(t >> 6);
    v2f* const lre = lds_all + half * P::HALF_UNITS;
    v2f* const lim = lre + C::PLANE;
    // column of the last pass (folded order, see the header)
    const int j2_ = (lane < 32) ? 32 * wv + lane : ((wv == 0 && lane == 32) ? T / 2 : T - 32 * wv - (lane - 32));

    const int j2 = j2_;

    // XCD-aware block -> (segment, pair of quads), as mtmfft_quad_kernel with two quads per workgroup
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

    const int c0 = 4 * (pg * NH + half);
    bool has[4];
    unsigned col[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        has[i] = c0 + i < a.nchan;
        col[i] = has[i] ? (unsigned)(a.chan_idx ? a.chan_idx[c0 + i] : c0 + i) : 0u;
    }
    const bool full = has[3];
    const long long start = a.seg_start[b];
    const long long rl = a.seg_lo[b] - start, rh = a.seg_hi[b] - start;
    const int rlo = (int)(rl < 0 ? 0 : (rl > a.nsig ? a.nsig : rl));
    const int rhi = (int)(rh < 0 ? 0 : (rh > a.nsig ? a.nsig : rh));
    const unsigned rowb = (unsigned)a.ld * 4u;        // bytes per row
    const float* seg = a.data + start * a.ld;         // wave-uniform; only rows in [rlo, rhi) are dereferenced

    // ---- load the segment once: x[e] = sample n = t + T*e; r = (c0, c1), i = (c2, c3)
    C2 x[16];
    if (rhi > rlo) {
        const bool vec4 = (a.chan_idx == nullptr) && full && ((a.ld & 3) == 0) &&
                          ((reinterpret_cast<size_t>(a.data) & 15) == 0);
        if (vec4) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = t + T * e;
                const int nc = min(max(n, rlo), rhi - 1);
                const float4 u = ldg<float4>(seg, (unsigned)nc * rowb + col[0] * 4u);
                const bool ok = (n == nc);
                x[e].r = v2f{ok ? u.x : 0.f, ok ? u.y : 0.f};
                x[e].i = v2f{ok ? u.z : 0.f, ok ? u.w : 0.f};
            }
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = t + T * e;
                const int nc = min(max(n, rlo), rhi - 1);
                float u[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float w = ldg<float>(seg, (unsigned)nc * rowb + col[i] * 4u);
                    u[i] = (n == nc && has[i]) ? w : 0.f;
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
        const v2f mr = v2f{f[0], f[1]}, mi = v2f{f[2], f[3]};
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const bool in = t + T * e < a.nsig;
            x[e].r -= in ? mr : splat(0.f);
            x[e].i -= in ? mi : splat(0.f);
        }
    } else if (a.detrend >= 0) {
        const float mid = 0.5f * (float)(a.nsig - 1);
        double s[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int n = t + T * e;
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
        half_sum<8, P::NW>(s, reinterpret_cast<double*>(lds_all), half, wv, lane);
        const double inv = 1.0 / a.nsig;
        if (a.detrend == 1 && a.nsig > 1) {
            const double den = 12.0 / ((double)a.nsig * ((double)a.nsig * a.nsig - 1.0));
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = t + T * e;
                const double dn = (double)((float)n - mid);
                const bool in = n < a.nsig;
                float u[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) u[i] = in ? (float)(s[i] * inv + s[4 + i] * den * dn) : 0.f;
                x[e].r -= v2f{u[0], u[1]};
                x[e].i -= v2f{u[2], u[3]};
            }
        } else {
            const v2f mr = v2f{(float)(s[0] * inv), (float)(s[1] * inv)};
            const v2f mi = v2f{(float)(s[2] * inv), (float)(s[3] * inv)};
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const bool in = t + T * e < a.nsig;
                x[e].r -= in ? mr : splat(0.f);
                x[e].i -= in ? mi : splat(0.f);
            }
        }
    }

    const unsigned nsig_m1 = (unsigned)(a.nsig - 1);
    // ---- demean_taper (mtmfft.py:115-116): the mean of the tapered segment of every taper, ahead of the pipeline
    // (one reduction per taper while both halves are still in lockstep), kept in the LDS tail
    float4* const dmean = reinterpret_cast<float4*>(lds_all + NH * P::HALF_UNITS) + half * P::KMAX;
    if (a.demean_taper) {
        for (int k = 0; k < a.ntaper; ++k) {
            const float* w = a.tapers + (size_t)k * a.nsig;
            double s[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const unsigned n = (unsigned)(t + T * e);
                const float wl = ldg<float>(w, min(n, nsig_m1) * 4u);
                const float wn = (n <= nsig_m1) ? wl : 0.f;
                const v2f pr = x[e].r * wn, pi = x[e].i * wn;
                s[0] += pr[0];
                s[1] += pr[1];
                s[2] += pi[0];
                s[3] += pi[1];
            }
            half_sum<4, P::NW>(s, reinterpret_cast<double*>(lds_all), half, wv, lane);
            if (t == 0)
                dmean[k] = make_float4((float)(s[0] / a.nsig), (float)(s[1] / a.nsig), (float)(s[2] / a.nsig),
                                       (float)(s[3] / a.nsig));
        }
        __syncthreads();
    }

    // accumulators for the taper mean (bins e<8 plus the Nyquist bin on column 0):
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
    const float hs = 0.5f * a.scale;
    constexpr unsigned OSZ = CPLX ? 8u : 4u;   // bytes per output element
    // straight-line epilogue: all four channels present, every bin kept, 16-byte aligned rows
    const bool fast = full && (a.fpos == nullptr) && ((reinterpret_cast<size_t>(a.out) & 15) == 0) &&
                      ((a.nchan & (CPLX ? 1 : 3)) == 0);
    // lanes (2i, 2i+1) hold neighbouring columns: complex rows leave as 32 contiguous bytes per store instruction
    const int drow = (lane_swap1(j2) - j2) * (int)((unsigned)a.nchan * OSZ);   // partner's row minus mine, bytes

    // taper weights of the first taper; the next taper's are requested while the previous FFT is in its second exchange
    float wn[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const unsigned n = (unsigned)(t + T * e);
        const float wl = ldg<float>(a.tapers, min(n, nsig_m1) * 4u);
        wn[e] = (n <= nsig_m1) ? wl : 0.f;
    }

    C2 v[16];
    Tw6 tn;
    tn.b1 = tn.b2 = tn.b3 = tn.a1 = tn.a2 = tn.a3 = make_float2(1.f, 0.f);

    // v = x * w_k (- its mean), first half of the pass-0 butterflies | barrier | second half
    // (inside the slots the lane indices go through opaque(): index arithmetic that depends on them is redone per slot
    // instead of being hoisted out of the slot loop into dozens of live VGPRs)
    auto stage_v0 = [&](int k) {
        const int t = opaque(t_);
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            v[e].r = x[e].r * wn[e];
            v[e].i = x[e].i * wn[e];
        }
        if (a.demean_taper) {
            const float4 m = dmean[k];
            const v2f mr = v2f{m.x, m.y}, mi = v2f{m.z, m.w};
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const bool in = t + T * e < a.nsig;
                v[e].r -= in ? mr : splat(0.f);
                v[e].i -= in ? mi : splat(0.f);
            }
        }
        dft16p_a(v);
    };

    // Both halves run the same straight-line sequence of slots (two barriers each); half 1 enters it one slot later
    // and half 0 leaves it one slot earlier, so the barriers pair slot n of half 0 with slot n - 1 of half 1.
#ifndef SPYFFT_PIPE_LOCKSTEP
    if (NH == 2 && half == 1) {
        __syncthreads();
        __syncthreads();
    }
#endif
    stage_v0(0);
    __syncthreads();
    dft16p_b(v);
    __syncthreads();
    for (int k = 0; k < a.ntaper; ++k) {
        const int t = opaque(t_);
        const int j2 = opaque(j2_);
        {

            // ---- L0: exchange after pass 0 (Ns = 1): out[16 t + r] -> in[t + T e]
            tn = load_tw6(a.tw, (unsigned)((t & 15) * (N / 256)) * 8u);          // twiddles of pass 1, used next slot
            const int wb = C::idx(16 * t, 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                lre[wb + r] = v[r].r;
                lim[wb + r] = v[r].i;
            }
            __syncthreads();
            const int rb = C::rbase(t, 0);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                v[e].r = lre[rb + e * C::ESTRIDE];
                v[e].i = lim[rb + e * C::ESTRIDE];
            }
            __syncthreads();
        }
        {
            // ---- V1: twiddles + butterflies of pass 1
            apply_tw(v, tn);
            dft16p_a(v);
            __syncthreads();
            dft16p_b(v);
            __syncthreads();
        }
        {
            // ---- L1: exchange after pass 1 (Ns = 16); the loads follow the folded column order of the last pass
            if (LOG2N == 12) tn = load_tw6(a.tw, (unsigned)j2 * 8u);             // twiddles of pass 2 (Ns = 256, k = j2)
            const int wb = C::idx(((t >> 4) << 8) + (t & 15), 0);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                lre[wb + r * 17] = v[r].r;
                lim[wb + r * 17] = v[r].i;
            }
            __syncthreads();
            const int rb = C::rbase(j2, 0);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                v[e].r = lre[rb + e * C::ESTRIDE];
                v[e].i = lim[rb + e * C::ESTRIDE];
            }
            __syncthreads();
        }
        {
            // ---- V2: last pass in the folded order -> v[e] = Z[j2 + T e]
            if constexpr (LOG2N == 12) {
                apply_tw(v, tn);
                dft16p_a(v);
                __syncthreads();
                dft16p_b(v);
            } else {
                constexpr int R = C::RLAST;
                constexpr int M = 16 / R;
                __syncthreads();
#pragma unroll
                for (int m = 0; m < M; ++m) {
                    C2 u[R];
                    const unsigned jb = (unsigned)(j2 + T * m) * 8u;
#pragma unroll
                    for (int r = 0; r < R; ++r) {
                        u[r] = v[m + r * M];
                        if (r > 0) u[r] = cmul_s(u[r], ldg<float2>(a.tw, jb * (unsigned)r));
                    }
                    if constexpr (R == 4) dft4p(u);
                    else dft8p(u);
#pragma unroll
                    for (int r = 0; r < R; ++r) v[m + r * M] = u[r];
                }
            }
            if (!MEAN) {
            // taper weights of the next taper: requested here, used behind the separation / conversion of this one
            if (k + 1 < a.ntaper) {
                const float* w = a.tapers + (size_t)(k + 1) * a.nsig;            // wave-uniform
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const unsigned n = (unsigned)(t + T * e);
                    const float wl = ldg<float>(w, min(n, nsig_m1) * 4u);
                    wn[e] = (n <= nsig_m1) ? wl : 0.f;
                }
            }
            }
            // ---- partner bins: afterwards v[15 - e] = Z[N - f] for f = j2 + T e, e < 8 (column 0: v[15] = its Nyquist bin)
            // Columns 0 and T/2 (lanes 0 and 32 of wave 0) are their own partners: their upper values take a detour
            // through the half's idle LDS planes (two lanes, same wave: no barrier) instead of the lane exchange -
            // column 0 reads them back shifted by one: v[15 - e] = Z[T (16 - e)], v[15] = Z[8 T] (Nyquist).
            const bool own = (wv == 0) && ((lane == 0) || (lane == 32));
            float4* const sp = reinterpret_cast<float4*>(lre) + (lane == 0 ? 0 : 8);
            if (wv == 0) {
                if (own) {
#pragma unroll
                    for (int i = 0; i < 8; ++i) sp[i] = make_float4(v[8 + i].r[0], v[8 + i].r[1], v[8 + i].i[0], v[8 + i].i[1]);
                }
            }
#pragma unroll
            for (int i = 8; i < 16; ++i) v[i] = swap32(v[i]);
            if (wv == 0) {
                if (own) {
                    const int sh = (lane == 0) ? 1 : 0;
#pragma unroll
                    for (int i = 0; i < 8; ++i) {
                        const float4 u = sp[(i + sh) & 7];
                        v[8 + i].r = v2f{u.x, u.y};
                        v[8 + i].i = v2f{u.z, u.w};
                    }
                }
            }

            // ---- separate the real channels, convert, store / accumulate
            char* const slab = reinterpret_cast<char*>(a.out) +
                               ((size_t)b * kout + (MEAN ? 0 : k)) * (size_t)a.nfsel * (size_t)a.nchan * OSZ;
#pragma unroll
            for (int e = 0; e < 9; ++e) {
                C2 xa, xb;   // xa = X(c0, c1), xb = X(c2, c3)
                int f;
                if (e < 8) {
                    f = j2 + T * e;
                    const C2 z = v[e];
                    C2 zp = v[15 - e];
                    if (f == 0) zp = z;
                    xa.r = (z.r + zp.r) * hs;
                    xa.i = (z.i - zp.i) * hs;
                    xb.r = (z.i + zp.i) * hs;
                    xb.i = (zp.r - z.r) * hs;
                } else {
                    if (j2 != 0) break;
                    f = N / 2;
                    xa.r = v[15].r * a.scale;
                    xb.r = v[15].i * a.scale;
                    xa.i = xb.i = splat(0.f);
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
                if (CPLX && a.blocked) {
                    // channel-quad-blocked layout for the CSD kernel (mtmfft2_kernel.h)
                    const int fi = a.fpos ? ldg<int>(a.fpos, (unsigned)f * 4u) : f;
                    if (fi >= 0 && has[0]) {
                        char* const bs = reinterpret_cast<char*>(a.out) +
                                         (((size_t)b * kout + k) * (size_t)((a.nchan + 3) >> 2) + (size_t)(c0 >> 2)) *
                                             (size_t)a.nfsel * 32u;
                        stg<float4>(bs, (unsigned)fi * 32u, make_float4(xa.r[0], xa.i[0], xa.r[1], xa.i[1]));
                        stg<float4>(bs, (unsigned)fi * 32u + 16u, make_float4(xb.r[0], xb.i[0], xb.r[1], xb.i[1]));
                    }
                    continue;
                }
                if (fast) {
                    const unsigned o = ((unsigned)f * (unsigned)a.nchan + (unsigned)c0) * OSZ;
                    if (CPLX) {
                        const float4 lo = make_float4(xa.r[0], xa.i[0], xa.r[1], xa.i[1]);
                        const float4 hi = make_float4(xb.r[0], xb.i[0], xb.r[1], xb.i[1]);
                        if (e < 8) {
                            // The L2 accepts one write request per line and clock whatever its size, and every lane owns a
                            // different row (bin): lanes (2i, 2i+1) write the two 16-byte halves of the SAME row with one
                            // instruction - first the even lane's row, then the odd lane's (rows `drow` bytes apart).
                            const bool second = (lane & 1) != 0;
                            const float4 give = second ? lo : hi;
                            const float4 got = make_float4(lane_swap1(give.x), lane_swap1(give.y), lane_swap1(give.z),
                                                           lane_swap1(give.w));
                            const unsigned o1 = second ? (unsigned)((int)o + drow) + 16u : o;
                            const unsigned o2 = second ? o + 16u : (unsigned)((int)o + drow);
                            stg<float4>(slab, o1, second ? got : lo);
                            stg<float4>(slab, o2, second ? hi : got);
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
            if (MEAN) {
            // taper weights of the next taper: requested here, used behind the separation / conversion of this one
            if (k + 1 < a.ntaper) {
                const float* w = a.tapers + (size_t)(k + 1) * a.nsig;            // wave-uniform
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    const unsigned n = (unsigned)(t + T * e);
                    const float wl = ldg<float>(w, min(n, nsig_m1) * 4u);
                    wn[e] = (n <= nsig_m1) ? wl : 0.f;
                }
            }
            }
            // ---- V0 of the next taper rides in the same slot
            if (k + 1 < a.ntaper) {
                stage_v0(k + 1);
                dft16p_b(v);
            }
            __syncthreads();
        }
    }
#ifndef SPYFFT_PIPE_LOCKSTEP
    if (NH == 2 && half == 0) {
        __syncthreads();
        __syncthreads();
    }
#endif

    if (MEAN) {
        char* const slab = reinterpret_cast<char*>(a.out) + (size_t)b * (size_t)a.nfsel * (size_t)a.nchan * OSZ;
#pragma unroll
        for (int e = 0; e < 9; ++e) {
            if (e == 8 && j2 != 0) break;
            const int f = (e < 8) ? j2 + T * e : N / 2;
            const int fi = a.fpos ? ldg<int>(a.fpos, (unsigned)f * 4u) : f;
            if (fi < 0) continue;
            const unsigned o = ((unsigned)fi * (unsigned)a.nchan + (unsigned)c0) * OSZ;
            const float nt = (float)a.ntaper;
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
