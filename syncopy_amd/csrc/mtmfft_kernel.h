// K1/K2: fused detrend -> taper -> packed real FFT -> scale -> output conversion
// (-> taper mean) for segments of a (rows x ld) float32 trial matrix.
//
// Reference semantics: specest/mtmfft.py:16-129 + specest/compRoutines.py:169-189
// (and specest/stft.py:101-154 when one segment = one STFT frame).
//
// One workgroup = G channel pairs of one segment.  Two real channels are packed
// as (re, im) of one complex FFT and separated afterwards:
//   Xa[f] = (Z[f] + conj(Z[N-f]))/2,  Xb[f] = (Z[f] - conj(Z[N-f]))/(2i).
// The segment is read from HBM exactly once (kept in registers across tapers);
// the spectra never leave registers/LDS before the output conversion.
#pragma once
#include "fft_device.h"

// development probes (tools/fft_probe.hip): extra kernel attributes / ablation bits, off in the library
#ifndef SPYFFT_KATTR
#define SPYFFT_KATTR
#endif
#ifndef SPYFFT_ABL
#define SPYFFT_ABL 0
#endif

namespace spyfft {

struct MtmArgs {
    const float* data;          // (rows x ld) float32
    long long ld;
    const int* chan_idx;        // nchan column ids or nullptr
    const long long* seg_start; // per segment: first row
    const long long* seg_lo;    // rows outside [lo, hi) read as 0
    const long long* seg_hi;
    int nseg;
    int nsig;                   // samples per segment (window length)
    int nchan;
    int ntaper;
    const float* tapers;        // (ntaper x nsig) float32
    const float2* tw;           // exp(-2 pi i m / N)
    float scale;
    int detrend;                // -1 none, 0 constant, 1 linear
    int demean_taper;
    const int* fpos;            // rfft bin -> output position or -1; nullptr = identity
    int nfsel;
    int out_kind;               // enum spyhip_output
    void* out;                  // (nseg, Kout, nfsel, nchan)
    int npg;                    // pair groups per segment
    int S;                      // pair groups sharing 128-byte lines (XCD cluster)
    int ncl;                    // clusters per segment
    // Bluestein (mtmfft_blue_kernel.h): logical FFT length and chirp tables; tw then belongs to the length-M FFT
    int nfft;
    const float2* chirp;        // nfft entries exp(-i pi n^2 / nfft)
    const float2* bhat;         // M entries: FFT_M of the wrapped conjugate chirp, / M
    int blocked;                // complex keeptapers output in the channel-quad-blocked layout
                                // (nseg*ntaper, ceil(nchan/4), nfsel, 4) instead of (nseg, ntaper, nfsel, nchan)
    const float* means;         // (nseg x nchan) per-channel means in the reference's summation order
                                // (seq_mean_kernel) used for detrend == 0, or nullptr: float64 block sums
    int seg_f64;                // the segments are FLOAT64 arrays in the reference (zero-extended / padded sliding windows,
                                // stft.py:101-117): the float64 kernels then subtract the trend in float64 as well
    unsigned* absmax;           // nullptr, or nchan bit patterns of non-negative floats raised to a bound of |re|, |im| of every
                                // complex value written for the channel (spyhip_fft_plan_set_absmax: the range K4h scales by)
    float wnorm;                // max over the tapers of || w * scale ||_2 (the bound is wnorm * ||detrended segment||_2)
    const float2* twh;          // (CfgD::HALF) exp(-2 pi i f / nfft), f <= nfft / 4: the step from the half-length complex
                                // transform of (even, odd) samples to the bins of the real transform
    const float2* xpair;        // (HALF forms of long trials) nullptr, or the segments of this launch channel-PAIR-major:
                                // xpair[(b * npair + pair) * xstride + n] = (x_c0[n], x_c1[n]), zeros outside the trial and
                                // behind nsig (pair_stage_kernel): the 8-byte-per-row gather becomes a stream
    long long xstride;          // samples per (segment, pair) of xpair (nsig rounded up to even)
};

// Per-channel mean of a segment exactly as the reference takes it.  scipy.signal.detrend(type="constant") on the
// float32 (time x channel) trial (specest/compRoutines.py:169-170, connectivity/ST_compRoutines.py:405-409) is
// `data - np.mean(data, axis=0)`: NumPy reduces the slow axis of a C-ordered array row by row, i.e. ONE float32
// accumulator per channel that takes the samples in time order, then one float32 division by the sample count.
// That rounding sequence cannot be re-associated, and for channels with an offset its error (~1e-6 of the offset)
// is what the bins next to DC are made of - so it is reproduced literally: one thread per (segment, channel), a
// serial chain of v_add_f32 over the rows (loads batched 64 rows ahead; lanes = adjacent channels: coalesced).
// Rows outside [seg_lo, seg_hi) count as +0 and leave the sum as it is.
//
// ONE channel is the exception: the reference's trial is then an (nSamples, 1) array, contiguous along the axis that is
// reduced, and NumPy sums such a run PAIRWISE (pairwise_sum_FLOAT: eight running sums over blocks of at most 128
// values, halves of longer runs recursively) - np_pairwise_rows below follows that order.
__device__ inline float np_pairwise_rows(const float* p, long long ld, int i0, int n, int rlo, int rhi) {
    auto at = [&](int i) { return (i >= rlo && i < rhi) ? p[(long long)i * ld] : 0.f; };
    if (n < 8) {
        float res = 0.f;
        for (int i = 0; i < n; ++i) res = __fadd_rn(res, at(i0 + i));
        return res;
    }
    if (n <= 128) {
        float r[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) r[j] = at(i0 + j);
        int i = 8;
        for (; i < n - (n % 8); i += 8) {
#pragma unroll
            for (int j = 0; j < 8; ++j) r[j] = __fadd_rn(r[j], at(i0 + i + j));
        }
        float res = __fadd_rn(__fadd_rn(__fadd_rn(r[0], r[1]), __fadd_rn(r[2], r[3])),
                              __fadd_rn(__fadd_rn(r[4], r[5]), __fadd_rn(r[6], r[7])));
        for (; i < n; ++i) res = __fadd_rn(res, at(i0 + i));
        return res;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    return __fadd_rn(np_pairwise_rows(p, ld, i0, n2, rlo, rhi), np_pairwise_rows(p, ld, i0 + n2, n - n2, rlo, rhi));
}

static __global__ void __launch_bounds__(256) seq_mean_kernel(MtmArgs a, float* means) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;      // (up to 256 adjacent channels per workgroup: whole 1-KiB rows)
    const int b = blockIdx.y;
    if (c >= a.nchan) return;
    const long long col = a.chan_idx ? a.chan_idx[c] : c;
    const long long start = a.seg_start[b];
    const long long rl = a.seg_lo[b] - start, rh = a.seg_hi[b] - start;
    const int rlo = (int)(rl < 0 ? 0 : (rl > a.nsig ? a.nsig : rl));
    const int rhi = (int)(rh < 0 ? 0 : (rh > a.nsig ? a.nsig : rh));
    if (a.nchan == 1) {
        means[b] = __fdiv_rn(np_pairwise_rows(a.data + start * a.ld + col, a.ld, 0, a.nsig, rlo, rhi), (float)a.nsig);
        return;
    }
    const float* p = a.data + (start + rlo) * a.ld + col;
    float s = 0.f;
    int n = rlo;
    // 64 rows in flight per thread: with few (segment, channel) threads - a handful of long trials - the chain waits on
    // memory latency, not on the 64 dependent additions
    for (; n + 64 <= rhi; n += 64) {
        float t[64];
#pragma unroll
        for (int e = 0; e < 64; ++e) t[e] = p[(long long)e * a.ld];
#pragma unroll
        for (int e = 0; e < 64; ++e) s = __fadd_rn(s, t[e]);
        p += 64 * a.ld;
    }
    for (; n + 16 <= rhi; n += 16) {
        float t[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) t[e] = p[(long long)e * a.ld];
#pragma unroll
        for (int e = 0; e < 16; ++e) s = __fadd_rn(s, t[e]);
        p += 16 * a.ld;
    }
    for (; n < rhi; ++n) {
        s = __fadd_rn(s, *p);
        p += a.ld;
    }
    means[(size_t)b * a.nchan + c] = __fdiv_rn(s, (float)a.nsig);
}

// Range of the spectra a plan is about to write (spyhip_fft_plan_set_absmax) for the kernel families that do not form it
// from the samples they hold anyway: every bin of channel c obeys |X_c(f)| <= ||w scale||_2 ||x_c||_2 (Cauchy-Schwarz), and
// removing a mean or a line (orthogonal projections) or a taper-wise mean only shrinks the 2-norm - so the norm of the
// samples minus their mean (detrend >= 0; float64 sums) or of the samples as they are bounds all tapers and all bins.
// One thread per (segment, channel), adjacent lanes = adjacent channels (coalesced), 16 rows in flight.
static __global__ void __launch_bounds__(256) seg_range_kernel(MtmArgs a) {
    const int c = blockIdx.x * blockDim.x + threadIdx.x;
    const int b = blockIdx.y;
    if (c >= a.nchan) return;
    const long long col = a.chan_idx ? a.chan_idx[c] : c;
    const long long start = a.seg_start[b];
    const long long rl = a.seg_lo[b] - start, rh = a.seg_hi[b] - start;
    const int rlo = (int)(rl < 0 ? 0 : (rl > a.nsig ? a.nsig : rl));
    const int rhi = (int)(rh < 0 ? 0 : (rh > a.nsig ? a.nsig : rh));
    const float* p = a.data + (start + rlo) * a.ld + col;
    double s1 = 0.0, s2 = 0.0;
    int n = rlo;
    for (; n + 16 <= rhi; n += 16) {
        float t[16];
#pragma unroll
        for (int e = 0; e < 16; ++e) t[e] = p[(long long)e * a.ld];
#pragma unroll
        for (int e = 0; e < 16; ++e) { s1 += (double)t[e]; s2 += (double)t[e] * (double)t[e]; }
        p += 16 * a.ld;
    }
    for (; n < rhi; ++n) {
        const double v = (double)*p;
        s1 += v;
        s2 += v * v;
        p += a.ld;
    }
    // (the mean is taken over the nsig samples of the window, zeros outside the trial included: compRoutines.py:169-172)
    double q = a.detrend >= 0 ? s2 - s1 * s1 / (double)a.nsig : s2;
    q = q > 0.0 ? q : 0.0;
    // cancellation in s2 - s1^2 / n leaves ~1e-16 s2 of doubt: part of the bound
    const float bound = (float)((sqrt(q) + 1e-7 * sqrt(s2)) * (double)a.wnorm * 1.001);
    atomicMax(a.absmax + c, __float_as_uint(bound));
}

// output conversions of const_def.py:25-37; `kind` is wave-uniform.  Kept out of
// line so that the compiler branches on `kind` instead of evaluating sqrt and
// atan2 for every bin and selecting afterwards.
__device__ __attribute__((noinline)) float convert_real_slow(float2 x, int kind) {
    switch (kind) {
        case SPYHIP_OUT_ABS: return sqrtf(x.x * x.x + x.y * x.y);
        case SPYHIP_OUT_REAL: return x.x;
        case SPYHIP_OUT_IMAG: return x.y;
        case SPYHIP_OUT_ANGLE: return atan2f(x.y, x.x);
        case SPYHIP_OUT_ABSREAL: return fabsf(x.x);
        case SPYHIP_OUT_ABSIMAG: return fabsf(x.y);
        default: return x.x * x.x + x.y * x.y;
    }
}
// Segments -> channel-pair-major copy for the pair forms of long trials (MtmArgs::xpair).  A pair workgroup needs 8 bytes of
// every row; gathered directly that is one 64-byte L2 request per row and lane (19 of the 54 us per trial at 16384 samples
// x 256 channels).  Here a tile of 64 rows x 64 channels is read along the channels (256 contiguous bytes per row), turned
// in LDS, and written along the rows: 512 contiguous bytes per pair.  Channel selection and the zero extension outside
// [seg_lo, seg_hi) happen here, so the transform kernel reads its samples without clamps.
static __global__ void __launch_bounds__(256) pair_stage_kernel(MtmArgs a, float2* __restrict__ xp, long long xstride, int npair) {
    __shared__ float tile[64][65];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const int r0 = blockIdx.x * 64, c0 = blockIdx.y * 64, b = blockIdx.z;
    const long long start = a.seg_start[b], lo = a.seg_lo[b], hi = a.seg_hi[b];
    const int c = c0 + lane;
    const long long col = c < a.nchan ? (a.chan_idx ? a.chan_idx[c] : c) : 0;
#pragma unroll 4
    for (int r = w; r < 64; r += 4) {
        const long long n = r0 + r, row = start + n;
        float v = 0.f;
        if (c < a.nchan && n < a.nsig && row >= lo && row < hi) v = a.data[row * a.ld + col];
        tile[r][lane] = v;
    }
    __syncthreads();
    const long long n = r0 + lane;                       // this lane's row of the tile
    if (n >= xstride) return;
#pragma unroll 4
    for (int q = w; q < 32; q += 4) {                    // 32 pairs of the tile, one per wave and pass
        const int pair = (c0 >> 1) + q;
        if (pair >= npair) break;
        xp[((size_t)b * npair + pair) * (size_t)xstride + n] = make_float2(tile[lane][2 * q], tile[lane][2 * q + 1]);
    }
}

// OUTK: 0 = power (the hot path, inlined), 1 = any other real conversion, 2 = complex
template <int OUTK>
__device__ __forceinline__ float convert_real(float2 x, int kind) {
    if (OUTK == 0) return x.x * x.x + x.y * x.y;
    return convert_real_slow(x, kind);
}

}  // namespace spyfft
