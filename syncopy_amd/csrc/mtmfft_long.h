// K1 for FFT lengths no single workgroup can hold (nfft > 4096 and not a power of two, or nfft > 16384):
// Bluestein's chirp-z with the length-M transforms (M = M1*M2 = 2^m >= 2 nfft - 1) done as four-step FFTs through
// HBM on the packed engine.  Same semantics as mtmfft_quad_kernel (specest/mtmfft.py:16-129,
// specest/compRoutines.py:169-189): four real channels per item, FFT "a" = c0 + i c2, FFT "b" = c1 + i c3 packed
// in the halves of fp32 register pairs; one scratch element = float4 (re_a, re_b, im_a, im_b).
//
//   n = n1*M2 + n2 (time-like index),  k = k1 + M1*k2 (frequency-like index of the length-M transform)
//   long_stats_kernel : per (segment, channel) sums for the polynomial removal and the post-taper mean
//   long_cols_kernel  : a[n] = x[n] w[n] c[n]; for every n2: FFT over n1 (length M1), * W_M^(n2 k1) -> Y[k1][n2]
//   long_rows_kernel  : for every k1: FFT over n2 (length M2) -> A[k1 + M1 k2]; * Bhat (/M folded in);
//                       inverse FFT over k2; * W_M^-(n2 k1) -> Z[k1][n2]           (the spectrum never leaves LDS)
//   long_cols_inv_kernel: for every n2: inverse FFT over k1 -> conv[n1*M2 + n2]; * c[n] for n < nfft -> Zb[n]
// Power-of-two nfft (> 16384) skip the chirp: cols -> rows (forward only) -> post on the [k1][k2]-ordered spectrum.
//   long_post_kernel  : channel separation X(c0,c1)[f] = (Zb[f] + conj Zb[nfft-f])/2, ..., scale, conversion,
//                       taper mean, store
// c[n] = exp(-i pi n^2 / nfft) (phases reduced exactly on the host), Bhat = FFT_M(conj(c) wrapped)/M stored in
// [k1][k2] order.
#pragma once
#include "fft2_device.h"
#include "mtmfft_kernel.h"
#include "mtmfft_long_args.h"

namespace spyfft {

// ---- sums over the valid samples of every (segment, channel): z = 0 -> (sum x, sum (n-mid) x); z = k+1 -> sum w_k x.
// Two deterministic stages: LONG_SPLITS slices of the segment in parallel (a single workgroup walking 30000 rows
// is latency-bound), then a fixed-order reduction.
constexpr int LONG_SPLITS = 32;
__global__ void __launch_bounds__(256) long_stats_kernel(MtmArgs a, double* part, int nz) {
    __shared__ double red[4][64][2];
    const int tid = threadIdx.x, cl = tid & 63, ph = tid >> 6;
    const int b = blockIdx.y, c = blockIdx.x * 64 + cl;
    const int z = blockIdx.z / LONG_SPLITS, sp = blockIdx.z % LONG_SPLITS;
    const long long start = a.seg_start[b], lo = a.seg_lo[b], hi = a.seg_hi[b];
    const double mid = 0.5 * (double)(a.nsig - 1);
    const int per = (a.nsig + LONG_SPLITS - 1) / LONG_SPLITS;
    const int n0 = sp * per, n1 = min(a.nsig, n0 + per);
    double s0 = 0.0, s1 = 0.0;
    if (c < a.nchan) {
        const long long col = a.chan_idx ? a.chan_idx[c] : c;
        const float* w = z > 0 ? a.tapers + (size_t)(z - 1) * a.nsig : nullptr;
#pragma unroll 4
        for (int n = n0 + ph; n < n1; n += 4) {
            const long long row = start + n;
            if (row < lo || row >= hi) continue;
            const double x = a.data[row * a.ld + col];
            if (z == 0) {
                s0 += x;
                s1 += ((double)n - mid) * x;
            } else {
                s0 += (double)w[n] * x;
            }
        }
    }
    red[ph][cl][0] = s0;
    red[ph][cl][1] = s1;
    __syncthreads();
    if (ph == 0 && c < a.nchan) {
        double t0 = 0.0, t1 = 0.0;
        for (int p = 0; p < 4; ++p) {
            t0 += red[p][cl][0];
            t1 += red[p][cl][1];
        }
        double* o = part + ((((size_t)b * nz + z) * LONG_SPLITS + sp) * a.nchan + c) * 2;
        o[0] = t0;
        o[1] = t1;
    }
}

// stats[seg][chan][0..1] = sum x, sum (n-mid) x; [2 + k] = sum w_k x
__global__ void __launch_bounds__(256) long_stats_final_kernel(MtmArgs a, const double* part, int nz, double* stats) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)a.nseg * a.nchan) return;
    const int b = (int)(i / a.nchan), c = (int)(i % a.nchan);
    double* o = stats + (size_t)i * (2 + a.ntaper);
    for (int z = 0; z < nz; ++z) {
        double t0 = 0.0, t1 = 0.0;
        for (int sp = 0; sp < LONG_SPLITS; ++sp) {
            const double* q = part + ((((size_t)b * nz + z) * LONG_SPLITS + sp) * a.nchan + c) * 2;
            t0 += q[0];
            t1 += q[1];
        }
        if (z == 0) {
            o[0] = t0;
            o[1] = t1;
        } else {
            o[1 + z] = t0;
        }
    }
}

// item -> (segment in chunk, quad, taper)
__device__ __forceinline__ void long_item(const LongArgs& a, long long item, int& bl, int& q, int& k) {
    k = (int)(item % a.m.ntaper);
    item /= a.m.ntaper;
    q = (int)(item % a.nquad);
    bl = (int)(item / a.nquad);
}

// ---- columns: taper * chirp on the fly, FFT over n1 (length M1 = 2^LOG2L), twiddle, transposed store
template <int LOG2L, int G>
__global__ void __launch_bounds__((Cfg2<LOG2L, G>::NTHREADS)) long_cols_kernel(LongArgs a) {
    using C = Cfg2<LOG2L, G>;
    constexpr int T = C::T;
    SPY_DYN_SMEM(v2f, lds);
    const int tid = threadIdx.x, h = tid % G, j = tid / G;
    const int ngrp = a.M2 / G;
    const long long item = blockIdx.x / ngrp;
    const int n2 = (int)(blockIdx.x % ngrp) * G + h;
    int bl, q, k;
    long_item(a, item, bl, q, k);
    const MtmArgs& m = a.m;
    const int b = a.seg0 + bl;
    const int c0 = 4 * q;
    const long long start = m.seg_start[b], lo = m.seg_lo[b], hi = m.seg_hi[b];
    // per-channel polynomial removal and post-taper mean (float64 on the statistics of long_stats_kernel)
    double mean[4] = {0, 0, 0, 0}, slope[4] = {0, 0, 0, 0};
    float dm[4] = {0.f, 0.f, 0.f, 0.f};
    bool has[4];
    long long col[4];
    const double mid = 0.5 * (double)(m.nsig - 1);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        has[i] = c0 + i < m.nchan;
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
        if (m.demean_taper)     // mean of w_k (x - trend) over the nsig samples
            dm[i] = (float)((st[2 + k] - mean[i] * a.wsum[2 * k] - slope[i] * a.wsum[2 * k + 1]) / m.nsig);
    }
    const float* w = m.tapers + (size_t)k * m.nsig;
    // all four channels adjacent and 16-byte aligned: one load per sample instead of four
    const bool vec4 = (m.chan_idx == nullptr) && has[3] && ((m.ld & 3) == 0) &&
                      ((reinterpret_cast<size_t>(m.data) & 15) == 0);
    C2 v[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int n1 = j + T * e;
        const long long n = (long long)n1 * a.M2 + n2;
        C2 z{splat(0.f), splat(0.f)};
        if (n < m.nsig) {
            const long long row = start + n;
            float x[4] = {0.f, 0.f, 0.f, 0.f};
            const bool in = row >= lo && row < hi;
            float raw[4] = {0.f, 0.f, 0.f, 0.f};
            if (in && vec4) {
                const float4 t = *reinterpret_cast<const float4*>(m.data + row * m.ld + c0);
                raw[0] = t.x; raw[1] = t.y; raw[2] = t.z; raw[3] = t.w;
            }
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (!has[i]) continue;
                float u = vec4 ? raw[i] : (in ? m.data[row * m.ld + col[i]] : 0.f);
                if (m.detrend >= 0) u -= (float)(mean[i] + slope[i] * ((double)n - mid));
                x[i] = w[n] * u - dm[i];
            }
            z = C2{v2f{x[0], x[1]}, v2f{x[2], x[3]}};
            if (!a.direct) z = cmul_s(z, a.chirp[n]);
        }
        v[e] = z;
    }
    fft2_forward<LOG2L, G>(v, lds, j, h, a.tw1);
    float4* const Y = a.scratch + (size_t)item * a.M1 * a.M2;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int k1 = j + T * e;
        st_c2(Y + (size_t)k1 * a.M2 + n2, cmul_s(v[e], a.twM[(size_t)n2 * k1]));
    }
}

// ---- rows: FFT over n2, * Bhat, inverse FFT over k2, conjugate twiddle; in place.  Thread id = row*T + j so that
// a wave reads consecutive elements of a row.
template <int LOG2L, int G>
__global__ void __launch_bounds__((Cfg2<LOG2L, G>::NTHREADS)) long_rows_kernel(LongArgs a) {
    using C = Cfg2<LOG2L, G>;
    constexpr int T = C::T;
    SPY_DYN_SMEM(v2f, lds);
    const int tid = threadIdx.x, j = tid % T, h = tid / T;
    const int ngrp = a.M1 / G;
    const long long item = blockIdx.x / ngrp;
    const int k1 = (int)(blockIdx.x % ngrp) * G + h;
    float4* const row = a.scratch + (size_t)item * a.M1 * a.M2 + (size_t)k1 * a.M2;
    C2 v[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] = ld_c2(row + j + T * e);
    fft2_forward<LOG2L, G>(v, lds, j, h, a.tw2);
    if (a.direct) {                                   // v[e] = X[k1 + M1*(j + T*e)]: done
#pragma unroll
        for (int e = 0; e < 16; ++e) st_c2(row + j + T * e, v[e]);
        return;
    }
    const float2* bh = a.bhat + (size_t)k1 * a.M2;
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] = cmul_s(v[e], bh[j + T * e]);
    fft2_inverse<LOG2L, G>(v, lds, j, h, a.tw2);
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int n2 = j + T * e;
        float2 t = a.twM[(size_t)n2 * k1];
        t.y = -t.y;
        st_c2(row + n2, cmul_s(v[e], t));
    }
}

// ---- columns, inverse: FFT over k1 -> conv[n1*M2 + n2], * chirp -> Zb[n] (natural order, n < nfft)
template <int LOG2L, int G>
__global__ void __launch_bounds__((Cfg2<LOG2L, G>::NTHREADS)) long_cols_inv_kernel(LongArgs a) {
    using C = Cfg2<LOG2L, G>;
    constexpr int T = C::T;
    SPY_DYN_SMEM(v2f, lds);
    const int tid = threadIdx.x, h = tid % G, j = tid / G;
    const int ngrp = a.M2 / G;
    const long long item = blockIdx.x / ngrp;
    const int n2 = (int)(blockIdx.x % ngrp) * G + h;
    float4* const Y = a.scratch + (size_t)item * a.M1 * a.M2;
    C2 v[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e] = ld_c2(Y + (size_t)(j + T * e) * a.M2 + n2);
    fft2_inverse<LOG2L, G>(v, lds, j, h, a.tw1);
    // every element of the column has been read by its own thread before the transform: writing the column back
    // (same addresses, natural-order index n = n1*M2 + n2) races with nobody
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const long long n = (long long)(j + T * e) * a.M2 + n2;
        if (n < a.m.nfft) st_c2(Y + n, cmul_s(v[e], a.chirp[n]));
    }
}

// ---- separation, scale, conversion, taper mean, store.  One thread per (segment, quad, bin).
template <int OUTK, bool MEAN>
__global__ void __launch_bounds__(256) long_post_kernel(LongArgs a) {
    constexpr bool CPLX = (OUTK == 2);
    const MtmArgs& m = a.m;
    const int nf = m.nfft / 2 + 1;
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long tot = (long long)a.nsegc * a.nquad * nf;
    if (gid >= tot) return;
    const int f = (int)(gid % nf);
    const int q = (int)((gid / nf) % a.nquad);
    const int bl = (int)(gid / ((long long)nf * a.nquad));
    const int b = a.seg0 + bl, c0 = 4 * q;
    const int fi = m.fpos ? m.fpos[f] : f;
    if (fi < 0) return;
    const int p = (f == 0) ? 0 : m.nfft - f;
    const float hs = 0.5f * m.scale;
    const int kout = MEAN ? 1 : m.ntaper;
    const size_t M = (size_t)a.M1 * a.M2;
    float2 acc[4] = {make_float2(0.f, 0.f), make_float2(0.f, 0.f), make_float2(0.f, 0.f), make_float2(0.f, 0.f)};
    for (int k = 0; k < m.ntaper; ++k) {
        const float4* Z = a.scratch + (((size_t)bl * a.nquad + q) * m.ntaper + k) * M;
        const size_t pf = a.direct ? (size_t)(f % a.M1) * a.M2 + f / a.M1 : (size_t)f;
        const size_t pp = a.direct ? (size_t)(p % a.M1) * a.M2 + p / a.M1 : (size_t)p;
        const C2 z = ld_c2(Z + pf), zp = ld_c2(Z + pp);
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
                if (CPLX) acc[i] = cadd(acc[i], X[i]);
                else acc[i].x += convert_real<OUTK>(X[i], m.out_kind);
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
    if (MEAN) {
        const float nt = (float)m.ntaper;
        const size_t o = ((size_t)b * m.nfsel + fi) * m.nchan + c0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            if (c0 + i >= m.nchan) continue;
            if (CPLX) reinterpret_cast<float2*>(m.out)[o + i] = make_float2(acc[i].x / nt, acc[i].y / nt);
            else reinterpret_cast<float*>(m.out)[o + i] = acc[i].x / nt;
        }
    }
}

}  // namespace spyfft
