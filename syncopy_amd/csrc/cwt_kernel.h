// K3: Morlet continuous wavelet transform by overlap-save FFT convolution.
//
// Reference semantics: cwt_time (specest/wavelets/transform.py:88-108) = for every scale s the
// full linear convolution of each channel with the sampled, amplitude-normalised complete Morlet
// kernel h_s (specest/wavelets/wavelets.py:27-86), cropped like scipy.signal.fftconvolve(mode=
// "same"):  y_s[n] = sum_m h_s[m] x[n + c_s - m],  c_s = (L_s-1)//2,  x = 0 outside the signal;
// wavelet_cF (specest/compRoutines.py:582-595) detrends the whole trial first, convolves the
// pre-selected samples and keeps the post-selected ones.
//
// One workgroup = G channels of one block of one segment: the block (NB samples incl. the halo
// of the longest kernel) is transformed ONCE, its spectrum stays in registers, and every scale
// costs one spectral multiply (H_s = FFT of the zero-padded kernel, built in fp64 on the host)
// and one inverse FFT; wrapped samples are discarded (overlap-save), so the result is the exact
// linear convolution the reference computes.
#pragma once
#include "mtmfft_kernel.h"

namespace spyfft {

struct CwtArgs {
    const float* data;            // (rows x ld) float32
    long long ld;
    const int* chan_idx;          // nchan column ids or nullptr
    const long long* seg_start;   // per segment: row of sample 0 of the (pre-selected) signal
    const long long* trial_lo;    // per segment: rows [lo, hi) of the whole trial (detrending range)
    const long long* trial_hi;
    int nseg, nsig, nchan, nscales;
    const float2* tw;             // exp(-2 pi i m / NB)
    const float2* hspec;          // (nscales x NB): FFT_NB(h_s) / NB
    const int* cshift;            // per scale: halo + c_s : output n of block o0 sits at q = n - o0 + cshift[s]
    int V;                        // outputs per block
    int halo;                     // samples read before o0
    int nblocks;
    int detrend;
    const double* trend;          // (nseg x nchan x 2): mean, slope about the trial centre
    int out_kind;
    const int* tpos;              // nsig: output slot of sample n, or -1; nullptr = identity
    int ntime_out;
    void* out;                    // (nseg, ntime_out, nscales, nchan)
    int accumulate;
};

// per (segment, channel): mean and least-squares slope over the trial rows [lo, hi)
__global__ void __launch_bounds__(256) cwt_trend_kernel(CwtArgs a, double* trend) {
    __shared__ double red[4][64][2];
    const int tid = threadIdx.x, cl = tid & 63, ph = tid >> 6;
    const int b = blockIdx.y, c = blockIdx.x * 64 + cl;
    const long long lo = a.trial_lo[b], hi = a.trial_hi[b];
    const double mid = 0.5 * (double)(hi - lo - 1);
    double s0 = 0.0, s1 = 0.0;
    if (c < a.nchan) {
        const long long col = a.chan_idx ? a.chan_idx[c] : c;
        for (long long r = lo + ph; r < hi; r += 4) {
            const double x = a.data[r * a.ld + col];
            s0 += x;
            s1 += ((double)(r - lo) - mid) * x;
        }
    }
    red[ph][cl][0] = s0;
    red[ph][cl][1] = s1;
    __syncthreads();
    if (ph == 0 && c < a.nchan) {
        const double n = (double)(hi - lo);
        double t0 = 0.0, t1 = 0.0;
        for (int p = 0; p < 4; ++p) {
            t0 += red[p][cl][0];
            t1 += red[p][cl][1];
        }
        double* o = trend + ((size_t)b * a.nchan + c) * 2;
        o[0] = t0 / n;
        o[1] = (a.detrend == 1 && n > 1.0) ? t1 * 12.0 / (n * (n * n - 1.0)) : 0.0;
    }
}

template <int LOG2N, int G, int OUTK>
__global__ void __launch_bounds__((Cfg<LOG2N, G>::NTHREADS)) cwt_kernel(CwtArgs a) {
    using C = Cfg<LOG2N, G>;
    constexpr int N = C::N, T = C::T;
    constexpr bool CPLX = (OUTK == 2);
    SPY_DYN_SMEM(float2, lds);
    const int tid = threadIdx.x;
    const int h = tid % G, j = tid / G;
    const int ngrp = (a.nchan + G - 1) / G;
    long long id = blockIdx.x;
    const int blk = (int)(id % a.nblocks);
    id /= a.nblocks;
    const int cg = (int)(id % ngrp);
    const int b = (int)(id / ngrp);
    const int c = cg * G + h;
    const bool has = c < a.nchan;
    const long long col = has ? (a.chan_idx ? a.chan_idx[c] : c) : 0;
    const long long start = a.seg_start[b], tlo = a.trial_lo[b];
    const int o0 = blk * a.V;

    double mean = 0.0, slope = 0.0, mid = 0.0;
    if (a.detrend >= 0 && has) {
        const double* t = a.trend + ((size_t)b * a.nchan + c) * 2;
        mean = t[0];
        slope = t[1];
        mid = 0.5 * (double)(a.trial_hi[b] - tlo - 1);
    }

    // ---- block samples u = o0 - halo + i, zero outside the signal (fftconvolve's zero padding)
    float2 v[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int u = o0 - a.halo + j + T * e;
        float x = 0.f;
        if (has && u >= 0 && u < a.nsig) {
            const long long row = start + u;
            x = a.data[row * a.ld + col];
            if (a.detrend >= 0) x -= (float)(mean + slope * ((double)(row - tlo) - mid));
        }
        v[e] = make_float2(x, 0.f);
    }
    fft_forward<LOG2N, G>(v, lds, j, h, a.tw);
    float2 Z[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) Z[e] = v[e];

    const int nend = min(o0 + a.V, a.nsig);
    constexpr unsigned OSZ = CPLX ? 8u : 4u;
    for (int s = 0; s < a.nscales; ++s) {
        const float2* H = a.hspec + (size_t)s * N;
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = cmul(Z[e], ldg<float2>(H, (unsigned)(j + T * e) * 8u));
        fft_inverse<LOG2N, G>(v, lds, j, h, a.tw);
        const int sh = a.cshift[s];
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int n = j + T * e - sh + o0;
            if (!has || n < o0 || n >= nend) continue;
            const int slot = a.tpos ? a.tpos[n] : n;
            if (slot < 0) continue;
            const size_t o = (((size_t)b * a.ntime_out + slot) * a.nscales + s) * a.nchan + c;
            if (CPLX) {
                float2* out = reinterpret_cast<float2*>(a.out) + o;
                *out = a.accumulate ? cadd(*out, v[e]) : v[e];
            } else {
                float* out = reinterpret_cast<float*>(a.out) + o;
                const float val = convert_real<OUTK>(v[e], a.out_kind);
                *out = a.accumulate ? *out + val : val;
            }
        }
        (void)OSZ;
    }
}

}  // namespace spyfft
