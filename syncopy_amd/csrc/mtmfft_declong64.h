// K1L2 at the reference's precision: trial lengths beyond one workgroup's LDS, N = P M with M a length the float64
// compile-time schedules serve (mtmfft_dec64_kernel.h) and P <= 8 - the structure of mtmfft_declong.h on channel PAIRS
// (one complex128 transform per pair; one scratch element = double2 = 16 bytes):
//
//   declong64_sub_kernel<C>  : (segment, pair) x r < P, tapers in a loop: float32 trend subtraction, float64 taper product
//                              and scheduled transform of x[P m + r] (mtmfft.py:96-117 in the reference's own types),
//                              times W_N^(r k) -> scratch[item][r M + k], item = (segment of the chunk, pair, taper)
//   declong64_post_kernel<P> : per (segment, pair, k < M): bins k + M q <= N / 2 and their partners from the P regions
//                              (the radix-P step in float64), channel separation, complex64 rounding, float32 scale
//                              (mtmfft.py:104,117-127), conversion, taper mean (float32 additions in taper order), store
//
// Replaces mtmfft_f64_any_kernel (generic Stockham passes over work arrays in global memory) for such lengths.
#pragma once
#include "mtmfft_dec64_kernel.h"

namespace spyfft {

struct Long64Args {
    MtmArgs m;                  // trial matrix, segments, float32 tapers (the statistics), output description
    const double* tapers64;     // (ntaper x nsig) float64
    const double2* twM;         // exp(-2 pi i m / M), M entries
    const double2* twN;         // exp(-2 pi i m / N), N entries
    const double2* twP;         // exp(-2 pi i m / P), P entries
    double2* scratch;           // [item][N]
    const double* stats;        // long_stats_kernel: [seg][chan][2 + ntaper]: sum x, sum (n - mid) x, sum w_k x
    const double* wsum;         // [ntaper][2]: sum w_k, sum w_k (n - mid)
    int seg0, nsegc;            // segments [seg0, seg0 + nsegc) are in flight
    int npair;
};

template <class C>
__global__ void __launch_bounds__((C::NTHREADS), (C::WPE)) declong64_sub_kernel(Long64Args a, int P) {
    static_assert(C::P == 1, "the sub-transforms are plain schedules");
    using spywil::cd;
    constexpr int V = C::V, M = C::N, T = C::T, G = C::G;
    SPY_DYN_SMEM(char, ldsraw);
    void* const lds = ldsraw;
    const MtmArgs& m = a.m;
    const int tid = threadIdx.x, h = tid % G, jt = tid / G;
    const int ngrp = (a.npair + G - 1) / G;
    long long id = blockIdx.x;
    const int pg = (int)(id % ngrp); id /= ngrp;
    const int r = (int)(id % P);
    const int bl = (int)(id / P);
    const int q = pg * G + h;
    const bool active = jt < T;                       // the workgroup is padded to whole waves
    const bool valid = active && q < a.npair;
    const int j = active ? jt : 0;
    const int b = a.seg0 + bl, c0 = 2 * q;
    const long long start = m.seg_start[b], lo = m.seg_lo[b], hi = m.seg_hi[b];

    // trend of the two channels: the reference-order float32 mean, or the float64 fit of long_stats_kernel - rounded to
    // float32 before it is subtracted from the float32 samples (scipy.signal.detrend on the float32 trial)
    bool has[2];
    long long col[2];
    double mean[2] = {0.0, 0.0}, slope[2] = {0.0, 0.0};
    const float mid = 0.5f * (float)(m.nsig - 1);
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        has[i] = valid && c0 + i < m.nchan;
        col[i] = has[i] ? (m.chan_idx ? m.chan_idx[c0 + i] : c0 + i) : 0;
        if (!has[i]) continue;
        const double* st = a.stats + ((size_t)b * m.nchan + c0 + i) * (2 + m.ntaper);
        if (m.detrend == 0 && m.means) {
            mean[i] = (double)m.means[(size_t)b * m.nchan + c0 + i];
        } else if (m.detrend >= 0) {
            mean[i] = st[0] / m.nsig;
            if (m.detrend == 1 && m.nsig > 1)
                slope[i] = st[1] * 12.0 / ((double)m.nsig * ((double)m.nsig * m.nsig - 1.0));
        }
    }
    float x0[V], x1[V];
#pragma unroll
    for (int e = 0; e < V; ++e) {
        const long long n = (long long)P * (j + T * e) + r;
        float u[2] = {0.f, 0.f};
        if (valid && n < m.nsig) {
            const long long row = start + n;
            const bool in = row >= lo && row < hi;
            const double dn = (double)((float)n - mid);
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (!has[i]) continue;
                u[i] = in ? m.data[row * m.ld + col[i]] : 0.f;
                if (m.detrend >= 0) u[i] -= (float)(mean[i] + slope[i] * dn);
            }
        }
        x0[e] = u[0];
        x1[e] = u[1];
    }

    // base twiddles of the later passes: taper-invariant, fetched once (d64_pass)
    cd w1a[V / C::R1], w1b[C::R2 > 1 ? V / C::R2 : 1], w1c[C::R3 > 1 ? V / C::R3 : 1];
    if constexpr (C::HOIST) {
        d64_twiddles<C, C::R1, V>(w1a, j, a.twM);
        if constexpr (C::NPASS >= 3) d64_twiddles<C, C::R2, V * C::R1>(w1b, j, a.twM);
        if constexpr (C::NPASS >= 4) d64_twiddles<C, C::R3, V * C::R1 * C::R2>(w1c, j, a.twM);
    }

    for (int k = 0; k < m.ntaper; ++k) {
        const int jo = opaque(j);     // (keeps the index arithmetic of the passes inside the loop)
        const double* w = a.tapers64 + (size_t)k * m.nsig;
        double dm[2] = {0.0, 0.0};
        if (m.demean_taper) {         // win -= win.mean(axis=0): mean of w_k (x - trend) over the nsig samples
#pragma unroll
            for (int i = 0; i < 2; ++i)
                if (has[i]) {
                    const double* st = a.stats + ((size_t)b * m.nchan + c0 + i) * (2 + m.ntaper);
                    dm[i] = (st[2 + k] - mean[i] * a.wsum[2 * k] - slope[i] * a.wsum[2 * k + 1]) / m.nsig;
                }
        }
        cd v[V];
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const long long n = (long long)P * (jo + T * e) + r;
            const bool in = n < m.nsig;
            const double wn = in ? w[n] : 0.0;
            v[e] = make_double2(wn * (double)x0[e] - (in ? dm[0] : 0.0), wn * (double)x1[e] - (in ? dm[1] : 0.0));
        }

        d64_pass<C, V, 1, true, false>(v, lds, jo, h, active, nullptr);
        d64_pass<C, C::R1, V, false, C::NPASS == 2>(v, lds, jo, h, active, C::HOIST ? w1a : a.twM);
        if constexpr (C::NPASS >= 3) d64_pass<C, C::R2, V * C::R1, false, C::NPASS == 3>(v, lds, jo, h, active, C::HOIST ? w1b : a.twM);
        if constexpr (C::NPASS >= 4) d64_pass<C, C::R3, V * C::R1 * C::R2, false, true>(v, lds, jo, h, active, C::HOIST ? w1c : a.twM);

        if (valid) {
            const size_t item = ((size_t)bl * a.npair + q) * m.ntaper + k;
            double2* const F = a.scratch + item * ((size_t)P * M) + (size_t)r * M;
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const int kk = jo + T * e;
                F[kk] = r == 0 ? v[e] : spywil::cmul(v[e], a.twN[(size_t)r * kk]);          // W_N^(r k)
            }
        }
        // (the next taper's first LDS write sits behind a barrier of d64_pass)
    }
}

// bin k + M q from the P twiddled sub-transforms: sum_r w_P^(r q) F_r[k].  q is a constant of an unrolled loop at every call:
// (r q) mod P folds, and the trivial factors 1, -1, -i, +i cost no multiplication
template <int P>
__device__ __forceinline__ spywil::cd declong64_bin(const spywil::cd (&g)[P], const spywil::cd (&wp)[P], int q) {
    using spywil::cd;
    cd s = g[0];
#pragma unroll
    for (int r = 1; r < P; ++r) {
        const int t = (r * q) % P;
        if (t == 0) s = spywil::cadd(s, g[r]);
        else if (2 * t == P) s = make_double2(s.x - g[r].x, s.y - g[r].y);
        else if (4 * t == P) s = make_double2(s.x + g[r].y, s.y - g[r].x);           // w = -i
        else if (4 * t == 3 * P) s = make_double2(s.x - g[r].y, s.y + g[r].x);       // w = +i
        else s = spywil::cadd(s, spywil::cmul(g[r], wp[t]));
    }
    return s;
}

template <int P, int OUTK, bool MEAN>
__global__ void __launch_bounds__(256) declong64_post_kernel(Long64Args a, int M) {
    using spywil::cd;
    constexpr bool CPLX = (OUTK == 2);
    constexpr int NQ = P / 2 + 1;                     // bins per thread: q <= (N / 2 - k) / M
    const MtmArgs& m = a.m;
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long tot = (long long)a.nsegc * a.npair * M;
    if (gid >= tot) return;
    const int kk = (int)(gid % M);
    const int q2 = (int)((gid / M) % a.npair);
    const int bl = (int)(gid / ((long long)M * a.npair));
    const int b = a.seg0 + bl, c0 = 2 * q2;
    const int N = P * M, kb = kk == 0 ? 0 : M - kk;
    const bool has1 = c0 + 1 < m.nchan;
    const int kout = MEAN ? 1 : m.ntaper;
    cd wp[P];
#pragma unroll
    for (int r = 0; r < P; ++r) wp[r] = a.twP[r];
    float2 acc[NQ][2];
#pragma unroll
    for (int s = 0; s < NQ; ++s) acc[s][0] = acc[s][1] = make_float2(0.f, 0.f);
    for (int k = 0; k < m.ntaper; ++k) {
        const double2* const base = a.scratch + (((size_t)bl * a.npair + q2) * m.ntaper + k) * (size_t)N;
        cd ga[P], gb[P];
#pragma unroll
        for (int r = 0; r < P; ++r) {
            ga[r] = base[(size_t)r * M + kk];
            gb[r] = base[(size_t)r * M + kb];
        }
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int f = kk + M * q;
            if (2 * f > N) break;
            const int fi = m.fpos ? m.fpos[f] : f;
            if (fi < 0) continue;
            const cd z = declong64_bin<P>(ga, wp, q);
            // partner N - f = (M - k) + M (P - 1 - q); in the k = 0 thread M (P - q)  (a branch: both indices stay constants)
            cd p;
            if (kk == 0) p = declong64_bin<P>(gb, wp, (P - q) % P);
            else p = declong64_bin<P>(gb, wp, P - 1 - q);
            const cd X0 = make_double2(0.5 * (z.x + p.x), 0.5 * (z.y - p.y));
            const cd X1 = make_double2(0.5 * (z.y + p.y), 0.5 * (p.x - z.x));
            // complex64 storage, then the float32 normalisation factor (mtmfft.py:104,117-127)
            const float2 s0 = make_float2(__fmul_rn((float)X0.x, m.scale), __fmul_rn((float)X0.y, m.scale));
            const float2 s1 = make_float2(__fmul_rn((float)X1.x, m.scale), __fmul_rn((float)X1.y, m.scale));
            if (MEAN) {
                if (CPLX) {
                    acc[q][0].x += s0.x; acc[q][0].y += s0.y;
                    acc[q][1].x += s1.x; acc[q][1].y += s1.y;
                } else {
                    acc[q][0].x += convert_real<OUTK>(s0, m.out_kind);
                    acc[q][1].x += convert_real<OUTK>(s1, m.out_kind);
                }
            } else {
                const size_t o = (((size_t)b * kout + k) * m.nfsel + fi) * m.nchan + c0;
                if (CPLX) {
                    reinterpret_cast<float2*>(m.out)[o] = s0;
                    if (has1) reinterpret_cast<float2*>(m.out)[o + 1] = s1;
                } else {
                    reinterpret_cast<float*>(m.out)[o] = convert_real<OUTK>(s0, m.out_kind);
                    if (has1) reinterpret_cast<float*>(m.out)[o + 1] = convert_real<OUTK>(s1, m.out_kind);
                }
            }
        }
    }
    if (MEAN) {
        const float nt = (float)m.ntaper;
#pragma unroll
        for (int q = 0; q < NQ; ++q) {
            const int f = kk + M * q;
            if (2 * f > N) break;
            const int fi = m.fpos ? m.fpos[f] : f;
            if (fi < 0) continue;
            const size_t o = ((size_t)b * m.nfsel + fi) * m.nchan + c0;
            if (CPLX) {
                reinterpret_cast<float2*>(m.out)[o] = make_float2(acc[q][0].x / nt, acc[q][0].y / nt);
                if (has1) reinterpret_cast<float2*>(m.out)[o + 1] = make_float2(acc[q][1].x / nt, acc[q][1].y / nt);
            } else {
                reinterpret_cast<float*>(m.out)[o] = acc[q][0].x / nt;
                if (has1) reinterpret_cast<float*>(m.out)[o + 1] = acc[q][1].x / nt;
            }
        }
    }
}

}  // namespace spyfft
