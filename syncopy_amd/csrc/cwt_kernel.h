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
#include "fft2_device.h"

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
    const int* tfloor;            // nsig: the largest slot among the samples 0 ... n (0 if none) - direct kernels' tile reference
    int ntime_out;
    void* out;                    // (nseg, ntime_out, nscales, nchan)
    int accumulate;
    void* stage;                  // (chunk segments, nscales, nchan, nsig): time-contiguous staging of the
                                  // converted values; cwt_scatter_kernel transposes it into `out`
    int seg0;                     // first segment of the chunk being processed
    const int* sidx;              // scale s of this launch -> scale index of the plan (nullptr = identity): scales
    int nscales_total;            // are grouped by the block length their kernel support needs (0 = nscales)
    int stage_add;                // 1: add to the staging values (later pieces of a kernel longer than one block)
    const float* xt;              // nullptr, or the pre-selected signals of this launch's segments channel-major and
                                  // time-contiguous: xt[(b * nchan + c) * nsig + n] (cwt_stage_input_kernel) - a transform
                                  // workgroup needs 4 ... 32 bytes of every 4 nchan-byte row of the trial; from the copy its
                                  // lanes read consecutive samples (c4 wavelet: 430 MB of fetches per trial for 46 MB of input)
    const int* smap;              // scatter kernels: staging row s of a segment -> scale index of the output (nullptr = identity)
    int nscales_out;              // scatter kernels: scales of the output (0 = nscales: every scale is staged)
};

// per (segment, channel): mean and least-squares slope over the trial rows [lo, hi), in two
// deterministic stages: partial sums over CWT_TREND_SPLITS slices of the trial (many workgroups in
// flight: the trial is read at HBM rate instead of one dependent row at a time), then a fixed-order
// reduction.  Trial lengths live on the device, so the slice length is derived in the kernel.
constexpr int CWT_TREND_SPLITS = 64;
__global__ void __launch_bounds__(256) cwt_trend_partial_kernel(CwtArgs a, double* part) {
    constexpr int nsplit = CWT_TREND_SPLITS;
    __shared__ double red[4][64][2];
    const int tid = threadIdx.x, cl = tid & 63, ph = tid >> 6;
    const int b = blockIdx.z, sp = blockIdx.y, c = blockIdx.x * 64 + cl;
    const long long lo = a.trial_lo[b], hi = a.trial_hi[b];
    const double mid = 0.5 * (double)(hi - lo - 1);
    const long long rows = (hi - lo + nsplit - 1) / nsplit;
    const long long r0 = lo + (long long)sp * rows;
    const long long r1 = (r0 + rows < hi) ? r0 + rows : hi;
    double s0 = 0.0, s1 = 0.0;
    if (c < a.nchan) {
        const long long col = a.chan_idx ? a.chan_idx[c] : c;
#pragma unroll 8
        for (long long r = r0 + ph; r < r1; r += 4) {
            const double x = a.data[r * a.ld + col];
            s0 += x;
            s1 += ((double)(r - lo) - mid) * x;
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
        double* o = part + (((size_t)b * nsplit + sp) * a.nchan + c) * 2;
        o[0] = t0;
        o[1] = t1;
    }
}

__global__ void __launch_bounds__(256) cwt_trend_final_kernel(CwtArgs a, const double* part, double* trend) {
    constexpr int nsplit = CWT_TREND_SPLITS;
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)a.nseg * a.nchan) return;
    const int b = (int)(i / a.nchan), c = (int)(i % a.nchan);
    const long long len = a.trial_hi[b] - a.trial_lo[b];
    double t0 = 0.0, t1 = 0.0;
    for (int sp = 0; sp < nsplit; ++sp) {
        const double* q = part + (((size_t)b * nsplit + sp) * a.nchan + c) * 2;
        t0 += q[0];
        t1 += q[1];
    }
    const double n = (double)len;
    trend[i * 2] = t0 / n;
    trend[i * 2 + 1] = (a.detrend == 1 && n > 1.0) ? t1 * 12.0 / (n * (n * n - 1.0)) : 0.0;
}

// Constant detrending (polyremoval = 0): the mean scipy.signal.detrend(type="constant") subtracts is NumPy's float32
// mean over the rows of the trial - one float32 accumulator per channel in time order, PAIRWISE for a one-channel
// trial (mtmfft_kernel.h: seq_mean_kernel) - and its rounding (~1e-6 of a channel's offset) is what the first and last
// samples of a wavelet transform are made of (the kernel has zero mean, the truncated convolution at the edges has
// not).  Reproduced literally; one thread per (segment, channel).
__global__ void __launch_bounds__(64) cwt_mean_np_kernel(CwtArgs a, double* trend) {
    const int c = blockIdx.x * 64 + threadIdx.x;
    const int b = blockIdx.y;
    if (c >= a.nchan) return;
    const long long col = a.chan_idx ? a.chan_idx[c] : c;
    const long long lo = a.trial_lo[b], n = a.trial_hi[b] - lo;
    const float* p = a.data + lo * a.ld + col;
    float s = 0.f;
    if (a.nchan == 1) {
        s = np_pairwise_rows(p, a.ld, 0, (int)n, 0, (int)n);
    } else {
        // few threads (segments x channels), long trials: the next 32 rows are in flight while the current 32 are added
        // (two register sets in turn, no copies), rows addressed as a wave-uniform row pointer + this lane's column so
        // that the loads take the scalar-base form: the time is the dependent chain of 16384 additions
        constexpr int D = 32;
        const unsigned colb = (unsigned)col * 4u;
        const char* row = reinterpret_cast<const char*>(a.data + lo * a.ld);       // wave-uniform
        const long long rstride = a.ld * 4;                                          // bytes per row (uniform)
        float t[D], u[D];
        long long k = 0;
        auto fetch = [&](float (&d)[D], const char* r) {
#pragma unroll
            for (int e = 0; e < D; ++e) d[e] = ldg<float>(r + (long long)e * rstride, colb);
        };
        auto add = [&](const float (&d)[D]) {
#pragma unroll
            for (int e = 0; e < D; ++e) s = __fadd_rn(s, d[e]);
        };
        if (n >= D) fetch(t, row);
        while (k + D <= n) {
            row += D * rstride;
            if (k + 2 * D <= n) fetch(u, row);
            add(t);
            k += D;
            if (k + D > n) break;
            row += D * rstride;
            if (k + 2 * D <= n) fetch(t, row);
            add(u);
            k += D;
        }
        for (; k < n; ++k) {
            s = __fadd_rn(s, ldg<float>(row, colb));
            row += rstride;
        }
    }
    double* o = trend + ((size_t)b * a.nchan + c) * 2;
    o[0] = (double)__fdiv_rn(s, (float)n);
    o[1] = 0.0;
}

// The pre-selected signals of a chunk of segments, turned channel-major (CwtArgs::xt): tiles of 64 samples x 64 channels
// read along the channels (256 contiguous bytes per row of the trial), turned in LDS, written along time.  One pass over
// the input (8.4 MB read + written per trial at 128 channels x 16384 samples) instead of a gather by every block group.
__global__ void __launch_bounds__(256) cwt_stage_input_kernel(CwtArgs a, float* xt) {
    __shared__ float tile[64][65];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int n0 = blockIdx.x * 64, c0 = blockIdx.y * 64, b = blockIdx.z;
    const long long start = a.seg_start[b];
    const int c = c0 + tx;
    const long long col = c < a.nchan ? (a.chan_idx ? a.chan_idx[c] : c) : 0;
#pragma unroll 4
    for (int r = ty; r < 64; r += 4) {
        const int n = n0 + r;
        if (n < a.nsig && c < a.nchan) tile[r][tx] = a.data[(start + n) * a.ld + col];
    }
    __syncthreads();
    const int n = n0 + tx;
#pragma unroll 4
    for (int r = ty; r < 64; r += 4) {
        const int cc = c0 + r;
        if (n < a.nsig && cc < a.nchan) xt[((size_t)b * a.nchan + cc) * (size_t)a.nsig + n] = tile[tx][r];
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
            x = a.xt ? a.xt[((size_t)b * a.nchan + c) * (size_t)a.nsig + u] : a.data[row * a.ld + col];
            if (a.detrend >= 0) x -= (float)(mean + slope * ((double)(row - tlo) - mid));
        }
        v[e] = make_float2(x, 0.f);
    }
    fft_forward<LOG2N, G>(v, lds, j, h, a.tw);
    float2 Z[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) Z[e] = v[e];

    // Results leave the workgroup time-contiguous (lanes = consecutive samples of one (scale, channel)
    // row of the staging buffer: full 256-byte wave stores); the channel-fastest layout of the
    // reference's output is produced by cwt_scatter_kernel.
    const int nend = min(o0 + a.V, a.nsig);
    for (int s = 0; s < a.nscales; ++s) {
        const float2* H = a.hspec + (size_t)s * N;
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = cmul(Z[e], ldg<float2>(H, (unsigned)(j + T * e) * 8u));
        fft_inverse<LOG2N, G>(v, lds, j, h, a.tw);
        const int sh = a.cshift[s];
        const size_t rowo = (((size_t)b * (a.nscales_total ? a.nscales_total : a.nscales) + (a.sidx ? a.sidx[s] : s)) * a.nchan + c) *
                            (size_t)a.nsig;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int n = j + T * e - sh + o0;
            if (!has || n < o0 || n >= nend) continue;
            if (CPLX) {
                float2* const d = reinterpret_cast<float2*>(a.stage) + rowo + n;
                *d = a.stage_add ? cadd(*d, v[e]) : v[e];
            } else {
                reinterpret_cast<float*>(a.stage)[rowo + n] = convert_real<OUTK>(v[e], a.out_kind);
            }
        }
    }
}

// Packed variant (block lengths up to 8192): one thread carries TWO channels in the halves of packed fp32
// registers (fft2_device.h) - both share the kernel spectrum H_s, so the spectral multiply and every butterfly
// of the forward and the nscales inverse transforms run on v_pk_* instructions for two channels at once.
// PAIRT (trial sums, accumulate == 2): the two halves carry the SAME channel of two consecutive segments instead, and
// what leaves the thread is the sum of their converted values - half the staging rows to write, half for the
// transposition pass to read and add (the staging round trip is what the trial average costs: 420 -> 210 MB per trial at
// 128 channels x 16384 samples x 25 scales).  Staging row set bp holds segments 2 bp and 2 bp + 1.
template <int LOG2N, int G, int OUTK, bool PAIRT = false>
__global__ void __launch_bounds__((Cfg2<LOG2N, G>::NTHREADS)) cwt2_kernel(CwtArgs a) {
    using C = Cfg2<LOG2N, G>;
    constexpr int N = C::N, T = C::T;
    constexpr bool CPLX = (OUTK == 2);
    SPY_DYN_SMEM(v2f, lds);
    const int tid = threadIdx.x;
    const int h = tid % G, j = tid / G;
    const int nunit = PAIRT ? a.nchan : (a.nchan + 1) / 2;
    const int ngrp = (nunit + G - 1) / G;
    long long id = blockIdx.x;
    const int blk = (int)(id % a.nblocks);
    id /= a.nblocks;
    const int cg = (int)(id % ngrp);
    const int b = (int)(id / ngrp);                 // segment, or (PAIRT) pair of segments
    const int c0 = PAIRT ? cg * G + h : 2 * (cg * G + h);
    const int bs[2] = {PAIRT ? 2 * b : b, PAIRT ? 2 * b + 1 : b};
    const bool has[2] = {c0 < a.nchan, PAIRT ? (c0 < a.nchan && bs[1] < a.nseg) : (c0 + 1 < a.nchan)};
    long long col[2], start[2], tlo[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int c = PAIRT ? c0 : c0 + i;
        col[i] = has[i] ? (a.chan_idx ? a.chan_idx[c] : c) : 0;
        start[i] = has[i] ? a.seg_start[bs[i]] : 0;
        tlo[i] = has[i] ? a.trial_lo[bs[i]] : 0;
    }
    const int o0 = blk * a.V;

    double mean[2] = {0.0, 0.0}, slope[2] = {0.0, 0.0}, mid[2] = {0.0, 0.0};
    if (a.detrend >= 0) {
#pragma unroll
        for (int i = 0; i < 2; ++i)
            if (has[i]) {
                const double* t = a.trend + ((size_t)bs[i] * a.nchan + (PAIRT ? c0 : c0 + i)) * 2;
                mean[i] = t[0];
                slope[i] = t[1];
                mid[i] = 0.5 * (double)(a.trial_hi[bs[i]] - tlo[i] - 1);
            }
    }

    // ---- block samples u = o0 - halo + i, zero outside the signal (fftconvolve's zero padding)
    C2 v[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int u = o0 - a.halo + j + T * e;
        float x[2] = {0.f, 0.f};
        if (u >= 0 && u < a.nsig) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
                if (has[i]) {
                    const long long row = start[i] + u;
                    x[i] = a.xt ? a.xt[((size_t)bs[i] * a.nchan + (PAIRT ? c0 : c0 + i)) * (size_t)a.nsig + u] : a.data[row * a.ld + col[i]];
                    if (a.detrend >= 0) x[i] -= (float)(mean[i] + slope[i] * ((double)(row - tlo[i]) - mid[i]));
                }
        }
        v[e].r = v2f{x[0], x[1]};
        v[e].i = splat(0.f);
    }
    fft2_forward<LOG2N, G>(v, lds, j, h, a.tw);
    C2 Z[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) Z[e] = v[e];

    const int nend = min(o0 + a.V, a.nsig);
    for (int s = 0; s < a.nscales; ++s) {
        const float2* H = a.hspec + (size_t)s * N;
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = cmul_s(Z[e], ldg<float2>(H, (unsigned)(j + T * e) * 8u));
        fft2_inverse<LOG2N, G>(v, lds, j, h, a.tw);
        const int sh = a.cshift[s];
        const size_t rowo = (((size_t)b * (a.nscales_total ? a.nscales_total : a.nscales) + (a.sidx ? a.sidx[s] : s)) * a.nchan + c0) *
                            (size_t)a.nsig;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int n = j + T * e - sh + o0;
            if (n < o0 || n >= nend) continue;
            if (PAIRT) {
                if (!has[0]) continue;
                if (CPLX) {
                    reinterpret_cast<float2*>(a.stage)[rowo + n] = make_float2(v[e].r[0] + v[e].r[1], v[e].i[0] + v[e].i[1]);   // (the absent half is 0)
                } else {
                    float y = convert_real<OUTK>(make_float2(v[e].r[0], v[e].i[0]), a.out_kind);
                    if (has[1]) y += convert_real<OUTK>(make_float2(v[e].r[1], v[e].i[1]), a.out_kind);
                    reinterpret_cast<float*>(a.stage)[rowo + n] = y;
                }
                continue;
            }
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                if (!has[i]) continue;
                const float2 y = make_float2(v[e].r[i], v[e].i[i]);
                if (CPLX) reinterpret_cast<float2*>(a.stage)[rowo + (size_t)i * a.nsig + n] = y;
                else reinterpret_cast<float*>(a.stage)[rowo + (size_t)i * a.nsig + n] = convert_real<OUTK>(y, a.out_kind);
            }
        }
    }
}

// Direct variant (round 6): the results leave the workgroup in the output's own layout (segment, slot(time), scale,
// channel) - no staging buffer, no transposition pass - for per-segment outputs (accumulate 0 / 1: keeptrials).  A
// workgroup carries G = 8 (1024-point blocks) or 4 (2048) channel PAIRS, the pair index fastest across lanes: the 2 G
// channels of one time sample are 8 G contiguous bytes (64 / 32), one store instruction of a wave writes 64 / G such runs.
// c4 wavelet, keeptrials: 293 -> 245 us/trial.  Trial sums stay on the staged kernels (PAIRT above): summing in the
// output layout is a read-modify-write per trial whose latency the two waves of a SIMD do not hide (measured with the
// workgroup owning its tile and walking the segments: 309 us/trial against 215 staged).
template <int LOG2N, int G, int OUTK>
__global__ void __launch_bounds__((Cfg2<LOG2N, G>::NTHREADS)) cwt2d_kernel(CwtArgs a) {
    using C = Cfg2<LOG2N, G>;
    constexpr int N = C::N, T = C::T;
    constexpr bool CPLX = (OUTK == 2);
    constexpr unsigned ESZ = CPLX ? 8u : 4u;
    SPY_DYN_SMEM(v2f, lds);
    const int tid = threadIdx.x;
    const int h = tid % G, j = tid / G;
    const int npair = (a.nchan + 1) / 2;
    const int ngrp = (npair + G - 1) / G;
    long long id = blockIdx.x;
    const int blk = (int)(id % a.nblocks);
    id /= a.nblocks;
    const int cg = (int)(id % ngrp);
    const int b = (int)(id / ngrp);
    const int c0 = 2 * (cg * G + h);
    const bool has[2] = {c0 < a.nchan, c0 + 1 < a.nchan};
    const bool vec = has[1] && !(a.nchan & 1);      // both channels, 8-byte aligned pairs
    long long col[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) col[i] = has[i] ? (a.chan_idx ? a.chan_idx[c0 + i] : c0 + i) : 0;
    const long long start = a.seg_start[b], tlo = a.trial_lo[b];
    const int o0 = blk * a.V;
    const int nend = min(o0 + a.V, a.nsig);
    const int nsc = a.nscales_total ? a.nscales_total : a.nscales;

    double mean[2] = {0.0, 0.0}, slope[2] = {0.0, 0.0}, mid = 0.0;
    if (a.detrend >= 0) {
        mid = 0.5 * (double)(a.trial_hi[b] - tlo - 1);
#pragma unroll
        for (int i = 0; i < 2; ++i)
            if (has[i]) {
                const double* t = a.trend + ((size_t)b * a.nchan + c0 + i) * 2;
                mean[i] = t[0];
                slope[i] = t[1];
            }
    }
    C2 v[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int u = o0 - a.halo + j + T * e;
        float x[2] = {0.f, 0.f};
        if (u >= 0 && u < a.nsig) {
            const long long row = start + u;
#pragma unroll
            for (int i = 0; i < 2; ++i)
                if (has[i]) {
                    x[i] = a.xt ? a.xt[((size_t)b * a.nchan + c0 + i) * (size_t)a.nsig + u] : a.data[row * a.ld + col[i]];
                    if (a.detrend >= 0) x[i] -= (float)(mean[i] + slope[i] * ((double)(row - tlo) - mid));
                }
        }
        v[e].r = v2f{x[0], x[1]};
        v[e].i = splat(0.f);
    }
    fft2_forward<LOG2N, G>(v, lds, j, h, a.tw);
    C2 Z[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) Z[e] = v[e];

    // Stores: a wave-uniform base (segment, reference slot of the tile, scale) + a 32-bit byte offset per value (the
    // host admits a plan to this kernel only if a block's rows span less than 4 GiB): no 64-bit address per value
    const unsigned rowb = (unsigned)nsc * (unsigned)a.nchan * ESZ;        // bytes from one time slot to the next
    const unsigned cb = (unsigned)c0 * ESZ;
    const int sref = a.tpos ? a.tfloor[o0] : o0;    // a slot at or below every slot of the tile (tpos is increasing)
    char* const tile = reinterpret_cast<char*>(a.out) + ((size_t)(a.seg0 + b) * a.ntime_out + sref) * (size_t)rowb;
    for (int s = 0; s < a.nscales; ++s) {
        const float2* H = a.hspec + (size_t)s * N;
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = cmul_s(Z[e], ldg<float2>(H, (unsigned)(j + T * e) * 8u));
        fft2_inverse<LOG2N, G>(v, lds, j, h, a.tw);
        const int sh = a.cshift[s];
        char* const sb = tile + (size_t)(a.sidx ? a.sidx[s] : s) * a.nchan * ESZ;
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int n = j + T * e - sh + o0;
            if (!has[0] || n < o0 || n >= nend) continue;
            const int slot = a.tpos ? ldg<int>(a.tpos, (unsigned)n * 4u) : n;
            if (slot < 0) continue;
            const unsigned off = (unsigned)(slot - sref) * rowb + cb;
            if (CPLX) {
                float4 y = make_float4(v[e].r[0], v[e].i[0], v[e].r[1], v[e].i[1]);
                if (vec) {
                    if (a.accumulate) { const float4 q = ldg<float4>(sb, off); y.x += q.x; y.y += q.y; y.z += q.z; y.w += q.w; }
                    stg<float4>(sb, off, y);
                } else {
                    if (a.accumulate) { const float2 q = ldg<float2>(sb, off); y.x += q.x; y.y += q.y; }
                    stg<float2>(sb, off, make_float2(y.x, y.y));
                    if (has[1]) {
                        if (a.accumulate) { const float2 q = ldg<float2>(sb, off + 8u); y.z += q.x; y.w += q.y; }
                        stg<float2>(sb, off + 8u, make_float2(y.z, y.w));
                    }
                }
            } else {
                float2 y = make_float2(convert_real<OUTK>(make_float2(v[e].r[0], v[e].i[0]), a.out_kind),
                                       has[1] ? convert_real<OUTK>(make_float2(v[e].r[1], v[e].i[1]), a.out_kind) : 0.f);
                if (vec) {
                    if (a.accumulate) { const float2 q = ldg<float2>(sb, off); y.x += q.x; y.y += q.y; }
                    stg<float2>(sb, off, y);
                } else {
                    if (a.accumulate) y.x += ldg<float>(sb, off);
                    stg<float>(sb, off, y.x);
                    if (has[1]) {
                        if (a.accumulate) y.y += ldg<float>(sb, off + 4u);
                        stg<float>(sb, off + 4u, y.y);
                    }
                }
            }
        }
    }
}

// Kernels longer than one block (more than 8191 taps after trimming) are cut into pieces of <= 8192 taps: each piece
// is an overlap-save convolution of its own (own launch, own halo), the complex results of the pieces add up in a
// complex side buffer (segment, long scale, channel, time); this kernel converts the sums into the staging rows of
// those scales.
__global__ void __launch_bounds__(256) cwt_long_convert_kernel(const float2* lng, const int* lidx, int nlong, int nseg,
                                                               int nscales, int nchan, int nsig, int out_kind,
                                                               float* stage) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long per = (long long)nchan * nsig;
    if (i >= (long long)nseg * nlong * per) return;
    const long long r = i % per, q = i / per;
    const int li = (int)(q % nlong), b = (int)(q / nlong);
    stage[((long long)b * nscales + lidx[li]) * per + r] = convert_real_slow(lng[i], out_kind);
}

// staging (segment, scale, channel, time) -> out (segment, slot(time), scale, channel), 64 x 64 tiles
// through LDS so that both sides move >= 256 contiguous bytes per wave; applies the post-selection
// (tpos) and the trial accumulation.
template <typename V>
__global__ void __launch_bounds__(256) cwt_scatter_kernel(CwtArgs a) {
    __shared__ V tile[64][65];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int n0 = blockIdx.x * 64, s = blockIdx.y;
    // accumulate == 2: the segments of the chunk are summed into output slot 0 (trial averaging with ONE
    // read-modify-write of the output per chunk); otherwise blockIdx.z = segment of the chunk
    const bool sum_segs = a.accumulate == 2;
    const int bl0 = sum_segs ? 0 : blockIdx.z, bl1 = sum_segs ? a.nseg : bl0 + 1;
    const size_t seg_stride = (size_t)a.nscales * a.nchan * (size_t)a.nsig;     // (launched with nscales = plan total)
    const V* const st0 = reinterpret_cast<const V*>(a.stage) + (size_t)s * a.nchan * (size_t)a.nsig;
    V* const out = reinterpret_cast<V*>(a.out);
    const int oseg = sum_segs ? 0 : a.seg0 + bl0;
    const int nso = a.nscales_out ? a.nscales_out : a.nscales, so = a.smap ? a.smap[s] : s;   // staging row -> output scale
    const int n = n0 + tx;
    for (int c0 = 0; c0 < a.nchan; c0 += 64) {
        if (c0) __syncthreads();
        if (sizeof(V) == 4 && (a.nsig & 3) == 0) {
            // 16-byte loads: 16 lanes cover the 64 samples of one channel row, a wave covers 4 rows
            const int q = threadIdx.x & 15, rr = threadIdx.x >> 4;          // rr in [0, 16)
            const int nq = n0 + 4 * q;
#pragma unroll
            for (int r = rr; r < 64; r += 16) {
                if (c0 + r < a.nchan && nq < a.nsig) {
                    const float* src = reinterpret_cast<const float*>(st0) + (size_t)(c0 + r) * a.nsig + nq;
                    float4 acc = *reinterpret_cast<const float4*>(src + (size_t)bl0 * seg_stride);
                    for (int bl = bl0 + 1; bl < bl1; ++bl) {
                        const float4 x = *reinterpret_cast<const float4*>(src + (size_t)bl * seg_stride);
                        acc.x += x.x; acc.y += x.y; acc.z += x.z; acc.w += x.w;
                    }
                    float* t = reinterpret_cast<float*>(&tile[r][4 * q]);
                    t[0] = acc.x; t[1] = acc.y; t[2] = acc.z; t[3] = acc.w;
                }
            }
        } else {
#pragma unroll 4
            for (int r = ty; r < 64; r += 4) {
                if (c0 + r < a.nchan && n < a.nsig) {
                    V acc = st0[(size_t)bl0 * seg_stride + (size_t)(c0 + r) * a.nsig + n];
                    for (int bl = bl0 + 1; bl < bl1; ++bl) {
                        const V x = st0[(size_t)bl * seg_stride + (size_t)(c0 + r) * a.nsig + n];
                        if constexpr (sizeof(V) == 8) acc = cadd(acc, x);
                        else acc = acc + x;
                    }
                    tile[r][tx] = acc;
                }
            }
        }
        __syncthreads();
        const int c = c0 + tx;
#pragma unroll 4
        for (int r = ty; r < 64; r += 4) {
            const int m = n0 + r;
            if (m >= a.nsig || c >= a.nchan) continue;
            const int slot = a.tpos ? a.tpos[m] : m;
            if (slot < 0) continue;
            V* const o = out + (((size_t)oseg * a.ntime_out + slot) * nso + so) * a.nchan + c;
            V val = tile[tx][r];
            if (a.accumulate) {
                const V old = *o;
                if constexpr (sizeof(V) == 8) val = cadd(old, val);
                else val = old + val;
            }
            *o = val;
        }
    }
}

// The same transposition with tiles of 256 samples x 16 channels for real outputs (nsig a multiple of 4): a wave reads ONE
// KiB of a staging row at a time instead of 256 bytes from four rows - the staging side is where the bytes are (with
// accumulate == 2 every segment of the chunk is read, the output touched once) - and writes 64-byte runs of the output.
// c4 wavelet: 51 -> 46 us/trial of this kernel.  (Requesting four segments at a time before adding them, in the same
// order, measured worse: 51 - more streams in flight cost more in DRAM locality than the latency they hide.)
__global__ void __launch_bounds__(256) cwt_scatter_wide_kernel(CwtArgs a) {
    constexpr int TN = 256, TC = 16, LD = TN + 4;      // (row stride 260 floats: the 16 x 4 lanes of a store hit every bank twice)
    __shared__ float tile[TC * LD];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int n0 = blockIdx.x * TN, s = blockIdx.y;
    const bool sum_segs = a.accumulate == 2;
    const int bl0 = sum_segs ? 0 : blockIdx.z, bl1 = sum_segs ? a.nseg : bl0 + 1;
    const size_t seg_stride = (size_t)a.nscales * a.nchan * (size_t)a.nsig;     // (launched with nscales = plan total)
    const float* const st0 = reinterpret_cast<const float*>(a.stage) + (size_t)s * a.nchan * (size_t)a.nsig;
    float* const out = reinterpret_cast<float*>(a.out);
    const int oseg = sum_segs ? 0 : a.seg0 + bl0;
    const int nso = a.nscales_out ? a.nscales_out : a.nscales, so = a.smap ? a.smap[s] : s;   // staging row -> output scale
    const int nq = n0 + 4 * lane;
    const int cc = threadIdx.x & 15, mm = threadIdx.x >> 4;        // output side: 16 channels x 16 samples per pass
    for (int c0 = 0; c0 < a.nchan; c0 += TC) {
        if (c0) __syncthreads();
#pragma unroll
        for (int r = w; r < TC; r += 4) {                         // one wave = one channel row of the tile
            if (c0 + r < a.nchan && nq < a.nsig) {
                const float* src = st0 + (size_t)(c0 + r) * a.nsig + nq;
                float4 acc = *reinterpret_cast<const float4*>(src + (size_t)bl0 * seg_stride);
                for (int bl = bl0 + 1; bl < bl1; ++bl) {
                    const float4 x = *reinterpret_cast<const float4*>(src + (size_t)bl * seg_stride);
                    acc.x += x.x; acc.y += x.y; acc.z += x.z; acc.w += x.w;
                }
                *reinterpret_cast<float4*>(&tile[r * LD + 4 * lane]) = acc;
            }
        }
        __syncthreads();
        const int c = c0 + cc;
#pragma unroll 4
        for (int m0 = mm; m0 < TN; m0 += 16) {
            const int m = n0 + m0;
            if (m >= a.nsig || c >= a.nchan) continue;
            const int slot = a.tpos ? a.tpos[m] : m;
            if (slot < 0) continue;
            float* const o = out + (((size_t)oseg * a.ntime_out + slot) * nso + so) * a.nchan + c;
            float val = tile[cc * LD + m0];
            if (a.accumulate) val = *o + val;
            *o = val;
        }
    }
}

// ---- superlets (specest/superlet.py:97-211): geometric mean over the wavelet set, one order at a time ----------
// acc[r, s0+q, c] = (init ? 1 : acc[...]) * spec[r, q, c] ^ expo[q]   (principal branch, 0^e = 0 for e > 0).
// The modulus goes through a split log2 / exp2 (see below) so that its error does not scale with log|z|; the phase
// (atan2f, sincosf) is skipped altogether for outputs that only need the modulus.
#define spy_log2f spy_log2         // v_log_f32 (spy_intrinsics.h): the argument is a mantissa in [0.5, 1)
constexpr int SLT_MAX_SCALES = 128;
struct SltArgs {
    float2* acc;
    const float2* spec;
    long long nrows;
    int nscales, nsub, s0, nchan, init;
    int nsub_total, q0;          // this launch covers the scales [q0, q0 + nsub) of spec's nsub_total
    int modulus_only;            // 1: |spec|^expo instead of spec^expo (no phase work); 2: acc and spec are REAL
                                 // float32 arrays holding moduli (ABS output of the plan) - half the traffic
    int square;                  // modulus_only == 2: store the square of the product (last factor, POW output)
    double expo[SLT_MAX_SCALES];
};

__global__ void __launch_bounds__(256) slt_combine_kernel(SltArgs a) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long n = a.nrows * a.nsub * a.nchan;
    if (idx >= n) return;
    const int c = (int)(idx % a.nchan), q = (int)((idx / a.nchan) % a.nsub);
    const long long r = idx / ((long long)a.nchan * a.nsub);
    const double e = a.expo[q];
    const size_t di = ((size_t)r * a.nscales + a.s0 + q) * a.nchan + c;
    const size_t si = ((size_t)r * a.nsub_total + a.q0 + q) * a.nchan + c;
    if (a.modulus_only == 2) {
        float* const dst = reinterpret_cast<float*>(a.acc) + di;
        if (e == 0.0 && !a.init && !a.square) return;
        const float m = reinterpret_cast<const float*>(a.spec)[si];
        float p = 1.f;
        if (e != 0.0) {
            p = 0.f;
            if (m > 0.f) {
                int ex;
                const float mant = frexpf(m, &ex);
                const double t = e * ((double)ex + (double)spy_log2f(mant));
                const double ti = floor(t);
                p = ldexpf(exp2f((float)(t - ti)), (int)ti);
            }
        }
        float g = a.init ? p : *dst * p;
        if (a.square) g *= g;
        *dst = g;
        return;
    }
    float2* dst = a.acc + di;
    if (e == 0.0 && !a.init) return;                    // z^0 = 1
    const float2 z = a.spec[si];
    float px = 1.f, py = 0.f;
    if (e != 0.0) {
        const double m2 = (double)z.x * z.x + (double)z.y * z.y;
        if (m2 > 0.0) {
            // |z|^e = 2^(e/2 * log2 |z|^2): exponent and mantissa of |z|^2 apart (log2f of [1,2) is good to 1e-7
            // absolute whatever the magnitude), the product with e/2 and the split into integer + fraction in fp64,
            // 2^fraction in fp32, the integer part by ldexp
            int ex;
            const float mant = frexpf((float)m2, &ex);                      // (float)m2 never flushes: |z| is fp32
            const double t = 0.5 * e * ((double)ex + (double)spy_log2f(mant));
            const double ti = floor(t);
            const float mag = ldexpf(exp2f((float)(t - ti)), (int)ti);
            if (a.modulus_only) {
                px = mag;
            } else {
                float sn, cs;
                sincosf((float)(e * (double)atan2f(z.y, z.x)), &sn, &cs);
                px = mag * cs;
                py = mag * sn;
            }
        } else {
            px = 0.f;
        }
    }
    if (a.init) {
        *dst = make_float2(px, py);
    } else {
        const float2 g = *dst;
        *dst = make_float2(g.x * px - g.y * py, g.x * py + g.y * px);
    }
}

// spectralConversions (shared/const_def.py:25-33) of a complex64 array: any real output kind -> float32
__global__ void __launch_bounds__(256) spec_convert_kernel(const float2* in, long long n, int kind, float* out) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    out[idx] = convert_real_slow(in[idx], kind);
}

}  // namespace spyfft
