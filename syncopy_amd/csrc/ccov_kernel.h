// K8: cross-covariance / cross-correlation lags from the accumulated cross spectra of zero-padded trials.
//
// The reference convolves every channel pair of every trial in the time domain order
// (connectivity/ST_compRoutines.py:466-584: fftconvolve(x_i, x_j[::-1], "same"), cropped to the lags
// 0 .. N/2 and divided by the overlap N - lag), then averages over trials and normalises
// (AV_compRoutines.py:166-228).  The trial average commutes with the inverse transform: the cross spectra
// X_a conj(X_b) of the zero-padded trials are summed over trials by the MFMA kernel (K4), and ONE inverse
// transform per channel pair yields the trial-averaged R_ab(tau) = sum_n x_a[n] x_b[n - tau] for all lags.
//
// Index conventions of the reference, kept bit for bit in meaning:
//   out[l, a, b] = R_ab(l) / (N - l)                      a >= b
//   out[l, a, b] = R_ab(l + q) / (N - l),  a < b          q = 1 for an even number of samples (the reversed
//                                                         "same" crop starts one sample late), 0 for odd.
#pragma once
#include "spy_intrinsics.h"
#include "../../include/spyhip.h"
#include "fft2_device.h"

namespace spyfft {

struct CcovArgs {
    const float2* acc;    // (L/2 + 1, C, C) complex64: sum over trials of X_a conj(X_b), lower triangle valid
    const float2* tw;     // exp(-2 pi i m / L), m < L
    int C, nsamples, nlag, q;
    long long npairs;     // C (C + 1) / 2
    float scale;          // 1 / (L * ntrials * fft_scale^2)
    float* out;           // (nlag, C, C)
};

__device__ __forceinline__ void pair_of(long long p, int& a, int& b) {
    long long r = (long long)((sqrt(8.0 * (double)p + 1.0) - 1.0) * 0.5);
    while (r * (r + 1) / 2 > p) --r;
    while ((r + 1) * (r + 2) / 2 <= p) ++r;
    a = (int)r;
    b = (int)(p - r * (r + 1) / 2);
}

// Workgroup = G thread sets; a set transforms TWO channel pairs (the halves of the packed registers) of the lower
// triangle: Hermitian extension on load (index k > L/2 reads conj(acc[L - k])), inverse transform in LDS, real part
// = R_ab(tau) at tau = k (lags >= 0) and tau = k - L (lags < 0, which are the mirrored element's lags).
template <int LOG2N, int G>
__global__ void __launch_bounds__((Cfg2<LOG2N, G>::NTHREADS)) ccov_lags_kernel(CcovArgs a) {
    using C = Cfg2<LOG2N, G>;
    constexpr int L = C::N, T = C::T;
    SPY_DYN_SMEM(v2f, lds);
    const int tid = threadIdx.x;
    const int h = tid % G, j = tid / G;
    // neighbouring channel pairs share 128-byte lines of every accumulator row: each XCD takes a contiguous run of
    // pair blocks (ids congruent mod 8 run on one XCD, one after the other), so a line is fetched into one L2 only
    const long long nblk = (a.npairs + 2 * G - 1) / (2 * G), chunk = (nblk + 7) >> 3;
    const long long blk = (long long)(blockIdx.x & 7u) * chunk + (blockIdx.x >> 3);
    if ((long long)(blockIdx.x >> 3) >= chunk || blk >= nblk) return;
    const long long p0 = (blk * G + h) * 2;
    int ca[2] = {0, 0}, cb[2] = {0, 0};
    bool has[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        has[i] = p0 + i < a.npairs;
        if (has[i]) pair_of(p0 + i, ca[i], cb[i]);
    }
    const size_t cc = (size_t)a.C * a.C;
    C2 v[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int k = j + T * e;
        const bool mir = k > L / 2;
        const size_t row = (size_t)(mir ? L - k : k) * cc;
        float2 s[2];
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            s[i] = make_float2(0.f, 0.f);
            if (has[i]) s[i] = a.acc[row + (size_t)ca[i] * a.C + cb[i]];
            if (mir) s[i].y = -s[i].y;
        }
        v[e].r = v2f{s[0].x, s[1].x};
        v[e].i = v2f{s[0].y, s[1].y};
    }
    fft2_inverse<LOG2N, G>(v, lds, j, h, a.tw);
    const size_t Cn = (size_t)a.C;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int tau = j + T * e;
        // lags >= 0: the element itself
        if (tau < a.nlag) {
            const float w = a.scale / (float)(a.nsamples - tau);
#pragma unroll
            for (int i = 0; i < 2; ++i)
                if (has[i]) a.out[((size_t)tau * Cn + ca[i]) * Cn + cb[i]] = v[e].r[i] * w;
        }
        // lags < 0 (tau - L): R_ba(l + q) of the mirrored element with l + q = L - tau
        const int l = (a.q == 0 && tau == 0) ? 0 : L - tau - a.q;
        if (l >= 0 && l < a.nlag) {
            const float w = a.scale / (float)(a.nsamples - l);
#pragma unroll
            for (int i = 0; i < 2; ++i)
                if (has[i] && ca[i] != cb[i]) a.out[((size_t)l * Cn + cb[i]) * Cn + ca[i]] = v[e].r[i] * w;
        }
    }
}

// ---- trials longer than 5461 samples (L > 8192: the inverse transform no longer fits the in-LDS engine).
// R_ab = irfft(S_ab) is taken with the FORWARD real transform the mtmfft kernels already provide for any length:
// S_ext[k] (Hermitian extension) = p[k] + i q[k] with p even and q odd, so
//     R(+tau) = (Re FFT(p)[tau] + Im FFT(q)[tau]) / L,      R(-tau) = (Re FFT(p)[tau] - Im FFT(q)[tau]) / L.
// ccov_extend_kernel lays p and q of a chunk of channel pairs out as the "channels" of one (L x 2 npairs) real
// segment; spyhip_fft_exec transforms it (2^14: the in-LDS kernel, longer: the four-step path); ccov_combine_kernel
// applies the reference's lag conventions exactly as ccov_lags_kernel does.
__global__ void __launch_bounds__(256) ccov_extend_kernel(const float2* acc, int Cn, int L, long long pair0, int npc, int cap,
                                                          float* pq) {
    const long long tot = (long long)L * cap;
    const size_t cc = (size_t)Cn * Cn;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < tot; idx += (long long)gridDim.x * 256) {
        const int k = (int)(idx / cap), p = (int)(idx % cap);
        float2 s = make_float2(0.f, 0.f);
        if (p < npc) {
            int ca, cb;
            pair_of(pair0 + p, ca, cb);
            const bool mir = k > L / 2;
            s = acc[(size_t)(mir ? L - k : k) * cc + (size_t)ca * Cn + cb];
            if (mir) s.y = -s.y;
        }
        reinterpret_cast<float2*>(pq)[idx] = s;            // (k, 2p) = p[k], (k, 2p + 1) = q[k]
    }
}

__global__ void __launch_bounds__(256) ccov_combine_kernel(const float2* spec, int L, long long pair0, int npc, int cap, int Cn,
                                                           int nsamples, int nlag, int q, float scale, float* out) {
    const long long tot = (long long)(L / 2 + 1) * cap;
    const size_t C = (size_t)Cn;
    for (long long idx = (long long)blockIdx.x * 256 + threadIdx.x; idx < tot; idx += (long long)gridDim.x * 256) {
        const int tau = (int)(idx / cap), p = (int)(idx % cap);
        if (p >= npc) continue;
        int ca, cb;
        pair_of(pair0 + p, ca, cb);
        const float2 P = spec[(size_t)tau * (2 * cap) + 2 * p], Q = spec[(size_t)tau * (2 * cap) + 2 * p + 1];
        if (tau < nlag) out[((size_t)tau * C + ca) * C + cb] = (P.x + Q.y) * (scale / (float)(nsamples - tau));
        const int l = tau - q;                               // the mirrored element holds R_ab(-(l + q))
        if (l >= 0 && l < nlag && ca != cb) out[((size_t)l * C + cb) * C + ca] = (P.x - Q.y) * (scale / (float)(nsamples - l));
    }
}

// d[a]: what the normalisation divides by, per channel (before the square root of the product of two):
//   mode 1  CC[0,a,a]                                  zero-lag auto-covariance of the trial average
//                                                       (normalize_ccov_cF, AV_compRoutines.py:215-225)
//   mode 2  CC[0,a,a] - mean_a^2 = np.std(x_a)^2        single trials, cross_covariance_cF(norm=True)
//                                                       (ST_compRoutines.py:575-579); mean_a^2 = |X_a(0)|^2 / N^2
__global__ void __launch_bounds__(256) ccov_diag_kernel(const float* out, const float2* acc, int Cn, int mode,
                                                        float dc_scale, float* d) {
    const int c = blockIdx.x * 256 + threadIdx.x;
    if (c >= Cn) return;
    float v = out[(size_t)c * Cn + c];
    if (mode == 2) v -= acc[(size_t)c * Cn + c].x * dc_scale;
    d[c] = v;
}

__global__ void __launch_bounds__(256) ccov_normalize_kernel(float* out, long long n, int Cn, const float* d) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    const int b = (int)(idx % Cn), a = (int)((idx / Cn) % Cn);
    out[idx] = out[idx] / sqrtf(d[a] * d[b]);
}

}  // namespace spyfft
