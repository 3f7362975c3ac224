// K1L2: trial lengths beyond one workgroup's LDS (a channel quad of more than 10240 samples) that factor as N = P M with M a
// length the compile-time schedules serve (mtmfft_dec_kernel.h) and P <= 8: decimation in time through HBM.
//
//   declong_sub_kernel<C>   : (segment, quad) x r < P, tapers in a loop: the scheduled transform of x[P m + r] w_k[P m + r]
//                             (polynomial removal / demean_taper from the statistics of long_stats_kernel, as long_cols_kernel),
//                             times W_N^(r k), stored at scratch[item][r M + k]
//   declong_post_kernel<P>  : per (segment, quad, k <= M / 2): the bins k + M q, (M - k) + M q <= N / 2 and their partners from the P regions
//                             (Z[k + M q] = sum_r w_P^(r q) F_r[k]), channel separation, scale, conversion, taper mean, store
//
// Against the path this replaces for such lengths (Bluestein with three power-of-two four-step transforms of length
// >= 2 N through HBM): one scheduled transform of N points, ONE write and one read of N packed values per taper instead
// of ~7 passes over 2 - 4 N.
//
// Reference semantics: specest/mtmfft.py:16-129 (np.fft.rfft takes any nSamples).
#pragma once
#include "mtmfft_dec_kernel.h"
#include "mtmfft_long_args.h"

namespace spyfft {

// LongArgs fields as used here: M1 = N, M2 = 1 (long_post_kernel's natural-order addressing), tw1 = exp(-2 pi i m / M)
// (M entries), twM = exp(-2 pi i m / N) (N entries), tw2 = exp(-2 pi i m / P) (P entries); scratch [item][N]
template <class C>
__global__ void __launch_bounds__((C::NTHREADS)) declong_sub_kernel(LongArgs a, int P) {
    static_assert(C::P == 1, "the sub-transforms are plain schedules");
    constexpr int V = C::V, M = C::N, T = C::T, G = C::G;
    SPY_DYN_SMEM(float4, lds);
    const MtmArgs& m = a.m;
    const int tid = threadIdx.x, h = tid % G, jt = tid / G;
    const int ngrp = (a.nquad + G - 1) / G;
    long long id = blockIdx.x;
    const int qg = (int)(id % ngrp); id /= ngrp;
    const int r = (int)(id % P);
    const int bl = (int)(id / P);
    const int q = qg * G + h;
    const bool active = jt < T;                       // the workgroup is padded to whole waves
    const bool valid = active && q < a.nquad;         // (groups of G quads: the last one may be ragged)
    const int j = active ? jt : 0;
    const int b = a.seg0 + bl, c0 = 4 * q;
    const long long start = m.seg_start[b], lo = m.seg_lo[b], hi = m.seg_hi[b];

    // per-channel polynomial removal and post-taper mean (float64 on the statistics of long_stats_kernel)
    double mean[4] = {0, 0, 0, 0}, slope[4] = {0, 0, 0, 0};
    bool has[4];
    long long col[4];
    const double mid = 0.5 * (double)(m.nsig - 1);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        has[i] = valid && c0 + i < m.nchan;
        col[i] = has[i] ? (m.chan_idx ? m.chan_idx[c0 + i] : c0 + i) : 0;
        if (!has[i]) continue;
        const double* st = a.stats + ((size_t)b * m.nchan + c0 + i) * (2 + m.ntaper);
        if (m.detrend == 0 && m.means) {
            mean[i] = (double)m.means[(size_t)b * m.nchan + c0 + i];   // the reference-order float32 mean
        } else if (m.detrend >= 0) {
            mean[i] = st[0] / m.nsig;
            if (m.detrend == 1 && m.nsig > 1)
                slope[i] = st[1] * 12.0 / ((double)m.nsig * ((double)m.nsig * m.nsig - 1.0));
        }
    }
    const bool vec4 = (m.chan_idx == nullptr) && has[3] && ((m.ld & 3) == 0) &&
                      ((reinterpret_cast<size_t>(m.data) & 15) == 0);
    // the thread's samples n = P (j + T e) + r, trend removed, resident across the tapers (the segment is read once)
    C2 x[V];
#pragma unroll
    for (int e = 0; e < V; ++e) {
        const long long n = (long long)P * (j + T * e) + r;
        C2 z{splat(0.f), splat(0.f)};
        if (valid && n < m.nsig) {
            const long long row = start + n;
            const bool in = row >= lo && row < hi;
            float raw[4] = {0.f, 0.f, 0.f, 0.f}, u[4] = {0.f, 0.f, 0.f, 0.f};
            if (in && vec4) {
                const float4 t = *reinterpret_cast<const float4*>(m.data + row * m.ld + c0);
                raw[0] = t.x; raw[1] = t.y; raw[2] = t.z; raw[3] = t.w;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (!has[i]) continue;
                u[i] = vec4 ? raw[i] : (in ? m.data[row * m.ld + col[i]] : 0.f);
                if (m.detrend >= 0) u[i] -= (float)(mean[i] + slope[i] * ((double)n - mid));
            }
            z = C2{v2f{u[0], u[1]}, v2f{u[2], u[3]}};
        }
        x[e] = z;
    }

    for (int k = 0; k < m.ntaper; ++k) {
        const float* w = m.tapers + (size_t)k * m.nsig;
        v2f dr = splat(0.f), di = splat(0.f);
        if (m.demean_taper) {     // mean of w_k (x - trend) over the nsig samples
            float dm[4] = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int i = 0; i < 4; ++i)
                if (has[i]) {
                    const double* st = a.stats + ((size_t)b * m.nchan + c0 + i) * (2 + m.ntaper);
                    dm[i] = (float)((st[2 + k] - mean[i] * a.wsum[2 * k] - slope[i] * a.wsum[2 * k + 1]) / m.nsig);
                }
            dr = v2f{dm[0], dm[1]};
            di = v2f{dm[2], dm[3]};
        }
        C2 v[V];
#pragma unroll
        for (int e = 0; e < V; ++e) {
            const long long n = (long long)P * (j + T * e) + r;
            const bool in = n < m.nsig;
            const float wn = in ? w[n] : 0.f;
            v[e].r = x[e].r * wn - (in ? dr : splat(0.f));
            v[e].i = x[e].i * wn - (in ? di : splat(0.f));
        }

        dec_transform<C>(v, lds, j, h, active, a.tw1);

        if (valid) {
            const size_t item = ((size_t)bl * a.nquad + q) * m.ntaper + k;
            float4* const F = a.scratch + item * ((size_t)P * M) + (size_t)r * M;
#pragma unroll
            for (int e = 0; e < V; ++e) {
                const int kk = j + T * e;
                st_c2(F + kk, r == 0 ? v[e] : cmul_s(v[e], a.twM[(size_t)r * kk]));          // W_N^(r k)
            }
        }
        // (the next taper's first LDS write sits behind a barrier of dec_pass)
    }
}

// bin k + M q of the length-N transform from the P twiddled sub-transforms of one (item): sum_r w_P^(r q) F_r[k].  q is a
// constant of an unrolled loop at every call: (r q) mod P folds, and the factors 1, -1, -i, +i cost no multiplication
template <int P>
__device__ __forceinline__ C2 declong_bin(const C2 (&g)[P], const float2 (&wp)[P], int q) {
    C2 s = g[0];
#pragma unroll
    for (int r = 1; r < P; ++r) {
        const int t = (r * q) % P;
        if (t == 0) s = cadd(s, g[r]);
        else if (2 * t == P) s = C2{s.r - g[r].r, s.i - g[r].i};
        else if (4 * t == P) s = C2{s.r + g[r].i, s.i - g[r].r};           // w = -i
        else if (4 * t == 3 * P) s = C2{s.r - g[r].i, s.i + g[r].r};       // w = +i
        else s = cadd(s, cmul_s(g[r], wp[t]));
    }
    return s;
}

// the output bins ko + M q <= N / 2 of one taper from the regions' values at ko (`own`) and at M - ko (`other`): radix-P step for
// the bin and for its partner N - f = (M - ko) + M (P - 1 - q) (ko = 0: M (P - q), from `own`), channel separation, scale,
// conversion, then the taper-mean accumulators or the store
template <int P, int OUTK, bool MEAN>
__device__ __forceinline__ void declong_emit(const MtmArgs& m, const C2 (&own)[P], const C2 (&other)[P], const float2 (&wp)[P],
                                             int ko, int M, int b, int k, int c0, float2 (&acc)[P / 2 + 1][4]) {
    constexpr bool CPLX = (OUTK == 2);
    constexpr int NQ = P / 2 + 1;
    const int N = P * M;
    const float hs = 0.5f * m.scale;
    const int kout = MEAN ? 1 : m.ntaper;
#pragma unroll
    for (int q = 0; q < NQ; ++q) {
        const int f = ko + M * q;
        if (2 * f > N) break;
        const int fi = m.fpos ? m.fpos[f] : f;
        if (fi < 0) continue;
        const C2 z = declong_bin<P>(own, wp, q);
        C2 zp;                        // (a branch: both indices stay constants of the unrolled loop)
        if (ko == 0) zp = declong_bin<P>(own, wp, (P - q) % P);
        else zp = declong_bin<P>(other, wp, P - 1 - q);
        C2 xa, xb;
        xa.r = (z.r + zp.r) * hs;
        xa.i = (z.i - zp.i) * hs;
        xb.r = (z.i + zp.i) * hs;
        xb.i = (zp.r - z.r) * hs;
        const float2 X[4] = {make_float2(xa.r[0], xa.i[0]), make_float2(xa.r[1], xa.i[1]),
                             make_float2(xb.r[0], xb.i[0]), make_float2(xb.r[1], xb.i[1])};
        if (MEAN) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (CPLX) acc[q][i] = cadd(acc[q][i], X[i]);
                else acc[q][i].x += convert_real<OUTK>(X[i], m.out_kind);
            }
        } else {
            const size_t o = (((size_t)b * kout + k) * m.nfsel + fi) * m.nchan + c0;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (c0 + i >= m.nchan) continue;
                if (CPLX) reinterpret_cast<float2*>(m.out)[o + i] = X[i];
                else reinterpret_cast<float*>(m.out)[o + i] = convert_real<OUTK>(X[i], m.out_kind);
            }
        }
    }
}

template <int P, int OUTK>
__device__ __forceinline__ void declong_store_mean(const MtmArgs& m, int ko, int M, int b, int c0, const float2 (&acc)[P / 2 + 1][4]) {
    constexpr bool CPLX = (OUTK == 2);
    const float nt = (float)m.ntaper;
#pragma unroll
    for (int q = 0; q < P / 2 + 1; ++q) {
        const int f = ko + M * q;
        if (2 * f > P * M) break;
        const int fi = m.fpos ? m.fpos[f] : f;
        if (fi < 0) continue;
        const size_t o = ((size_t)b * m.nfsel + fi) * m.nchan + c0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (c0 + i >= m.nchan) continue;
            if (CPLX) reinterpret_cast<float2*>(m.out)[o + i] = make_float2(acc[q][i].x / nt, acc[q][i].y / nt);
            else reinterpret_cast<float*>(m.out)[o + i] = acc[q][i].x / nt;
        }
    }
}

// ---- radix-P step + channel separation + scale + conversion + taper mean + store, one thread per (segment, quad, k <= M / 2):
// the thread reads the P regions at k and at M - k ONCE per taper and forms the bins k + M q and (M - k) + M q that lie in
// [0, N / 2] - each with its partner, which is a bin of the other index - the spectrum is never written in natural order.
template <int P, int OUTK, bool MEAN>
__global__ void __launch_bounds__(256) declong_post_kernel(LongArgs a, int M) {
    constexpr int NQ = P / 2 + 1;                     // bins per index: q <= (N / 2 - k) / M
    const MtmArgs& m = a.m;
    const int Mh = M / 2 + 1;
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long tot = (long long)a.nsegc * a.nquad * Mh;
    if (gid >= tot) return;
    const int kk = (int)(gid % Mh);
    const int q4 = (int)((gid / Mh) % a.nquad);
    const int bl = (int)(gid / ((long long)Mh * a.nquad));
    const int b = a.seg0 + bl, c0 = 4 * q4;
    const int kb = kk == 0 ? 0 : M - kk;
    const bool both = kk != 0 && 2 * kk != M;         // (k = 0 and k = M / 2 are their own mirror index)
    float2 wp[P];
#pragma unroll
    for (int r = 0; r < P; ++r) wp[r] = a.tw2[r];
    float2 acca[NQ][4], accb[NQ][4];
#pragma unroll
    for (int s = 0; s < NQ; ++s)
#pragma unroll
        for (int i = 0; i < 4; ++i) acca[s][i] = accb[s][i] = make_float2(0.f, 0.f);
    for (int k = 0; k < m.ntaper; ++k) {
        const float4* const base = a.scratch + (((size_t)bl * a.nquad + q4) * m.ntaper + k) * ((size_t)P * M);
        C2 ga[P], gb[P];
#pragma unroll
        for (int r = 0; r < P; ++r) {
            ga[r] = ld_c2(base + (size_t)r * M + kk);
            gb[r] = ld_c2(base + (size_t)r * M + kb);
        }
        declong_emit<P, OUTK, MEAN>(m, ga, gb, wp, kk, M, b, k, c0, acca);
        if (both) declong_emit<P, OUTK, MEAN>(m, gb, ga, wp, kb, M, b, k, c0, accb);
    }
    if (MEAN) {
        declong_store_mean<P, OUTK>(m, kk, M, b, c0, acca);
        if (both) declong_store_mean<P, OUTK>(m, kb, M, b, c0, accb);
    }
}

}  // namespace spyfft
