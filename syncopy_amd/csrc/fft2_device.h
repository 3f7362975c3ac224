// Packed-pair radix-16 Stockham FFT for gfx950: every thread carries TWO independent complex
// FFTs in the two halves of 64-bit register pairs, so that the whole butterfly network runs on
// v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 (2 flops per lane per operand) without a single
// shuffle: a value is C2 = { r = (re of FFT a, re of FFT b), i = (im of a, im of b) } and both
// FFTs share every twiddle factor (a scalar pair that the packed ops broadcast).
//
// Decomposition (length N = 2^LOG2N, T = N/16 threads per FFT pair), as in fft_device.h:
//   every pass, thread j holds v[e] = in[j + T*e], e = 0..15; radix-16 passes with Ns = 16^p
//   write out[(j/Ns)*16*Ns + j%Ns + r*Ns]; a final radix-R pass (R = N / 16^p in {2,4,8}) leaves
//   the spectrum in place: v[e] = Z[j + T*e].
// G FFT pairs are interleaved in one workgroup (thread id = j*G + h).  LDS holds a plane of real
// parts and a plane of imaginary parts, element (i,h) at (i + i/16)*G + h in 8-byte units (the pad
// per 16 keeps the radix-16 scatter conflict free for ds_write_b64 lane groups).
#pragma once
#include "fft_device.h"
#ifndef SPYFFT_ABL
#define SPYFFT_ABL 0
#endif

// development timeline probe (tools/fft_stamp_probe.hip): s_memtime stamps of one lane per wave, off in the library
#ifdef SPYFFT_STAMPS
#define SPY_STAMP(sn) do { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
        if ((threadIdx.x & 63) == 0 && blockIdx.x >= SPYFFT_STAMP_B0 && blockIdx.x < SPYFFT_STAMP_B0 + 8 * SPYFFT_STAMP_NB && (blockIdx.x & 7) == 0) \
            spy_stamp_buf[((((blockIdx.x - SPYFFT_STAMP_B0) >> 3) * 16 + (threadIdx.x >> 6)) << 10) + (sn)] = __builtin_readcyclecounter(); \
        ++(sn); } while (0)
#define SPY_STAMP_PARAM , int& sn
#define SPY_STAMP_ARG , sn
__device__ unsigned long long* spy_stamp_buf;
#else
#define SPY_STAMP(sn) do { } while (0)
#define SPY_STAMP_PARAM
#define SPY_STAMP_ARG
#endif

namespace spyfft {

typedef float v2f __attribute__((ext_vector_type(2)));
struct C2 {
    v2f r, i;
};

__device__ __forceinline__ v2f splat(float s) { return v2f{s, s}; }
__device__ __forceinline__ C2 cadd(C2 a, C2 b) { return C2{a.r + b.r, a.i + b.i}; }
__device__ __forceinline__ C2 csub(C2 a, C2 b) { return C2{a.r - b.r, a.i - b.i}; }
// multiply by -i
__device__ __forceinline__ C2 mul_mi(C2 a) { return C2{a.i, -a.r}; }
// multiply both FFTs by the same twiddle w
__device__ __forceinline__ C2 cmul_s(C2 a, float2 w) {
    C2 o;
    o.r = a.r * w.x - a.i * w.y;
    o.i = a.r * w.y + a.i * w.x;
    return o;
}

__device__ __forceinline__ void dft4(C2& t0, C2& t1, C2& t2, C2& t3) {
    const C2 a0 = cadd(t0, t2), a1 = csub(t0, t2), a2 = cadd(t1, t3), a3 = mul_mi(csub(t1, t3));
    t0 = cadd(a0, a2);
    t1 = cadd(a1, a3);
    t2 = csub(a0, a2);
    t3 = csub(a1, a3);
}

// in-place DFTs of R values in natural order (t[k] = sum_n t[n] W_R^(nk))
__device__ __forceinline__ void dft2p(C2 (&t)[2]) {
    const C2 a = t[0], b = t[1];
    t[0] = cadd(a, b);
    t[1] = csub(a, b);
}
__device__ __forceinline__ void dft4p(C2 (&t)[4]) { dft4(t[0], t[1], t[2], t[3]); }
__device__ __forceinline__ void dft8p(C2 (&t)[8]) {
    const float h = 0.70710678118654752440f;
    C2 a[4] = {t[0], t[2], t[4], t[6]};
    C2 b[4] = {t[1], t[3], t[5], t[7]};
    dft4p(a);
    dft4p(b);
    b[1] = C2{(b[1].r + b[1].i) * h, (b[1].i - b[1].r) * h};    // * W8^1
    b[2] = mul_mi(b[2]);                                         // * W8^2
    b[3] = C2{(b[3].i - b[3].r) * h, (b[3].r + b[3].i) * -h};   // * W8^3
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        t[k] = cadd(a[k], b[k]);
        t[k + 4] = csub(a[k], b[k]);
    }
}
// 16-point DFT in two halves (the pipelined kernel puts a workgroup barrier between them): a = four 4-point DFTs
// over n1 + the internal twiddles, b = four 4-point DFTs over n2 + the index transposition
__device__ __forceinline__ void dft16p_a(C2 (&t)[16]) {
    // n = 4*n1 + n2, k = k1 + 4*k2
    const float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f;
    const float h = 0.70710678118654752440f;
#pragma unroll
    for (int n2 = 0; n2 < 4; ++n2) dft4(t[n2], t[4 + n2], t[8 + n2], t[12 + n2]);
    // now t[4*k1 + n2] = y[n2][k1]; twiddles W16^(n2*k1)
    t[4 * 1 + 1] = cmul_s(t[4 * 1 + 1], make_float2(c1, -s1));
    t[4 * 2 + 1] = C2{(t[4 * 2 + 1].r + t[4 * 2 + 1].i) * h, (t[4 * 2 + 1].i - t[4 * 2 + 1].r) * h};
    t[4 * 3 + 1] = cmul_s(t[4 * 3 + 1], make_float2(s1, -c1));
    t[4 * 1 + 2] = C2{(t[4 * 1 + 2].r + t[4 * 1 + 2].i) * h, (t[4 * 1 + 2].i - t[4 * 1 + 2].r) * h};
    t[4 * 2 + 2] = mul_mi(t[4 * 2 + 2]);
    t[4 * 3 + 2] = C2{(t[4 * 3 + 2].i - t[4 * 3 + 2].r) * h, (t[4 * 3 + 2].r + t[4 * 3 + 2].i) * -h};
    t[4 * 1 + 3] = cmul_s(t[4 * 1 + 3], make_float2(s1, -c1));
    t[4 * 2 + 3] = C2{(t[4 * 2 + 3].i - t[4 * 2 + 3].r) * h, (t[4 * 2 + 3].r + t[4 * 2 + 3].i) * -h};
    t[4 * 3 + 3] = cmul_s(t[4 * 3 + 3], make_float2(-c1, s1));
}
__device__ __forceinline__ void dft16p_b(C2 (&t)[16]) {
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) dft4(t[4 * k1], t[4 * k1 + 1], t[4 * k1 + 2], t[4 * k1 + 3]);
    // t[4*k1 + k2] = X[k1 + 4*k2]: transpose the 4x4 register tile (pure renaming once unrolled)
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) {
#pragma unroll
        for (int k2 = k1 + 1; k2 < 4; ++k2) {
            const C2 tmp = t[4 * k1 + k2];
            t[4 * k1 + k2] = t[4 * k2 + k1];
            t[4 * k2 + k1] = tmp;
        }
    }
}
__device__ __forceinline__ void dft16p(C2 (&t)[16]) {
    dft16p_a(t);
    dft16p_b(t);
}

template <int LOG2N, int G>
struct Cfg2 {
    static constexpr int N = 1 << LOG2N;
    static constexpr int T = N / 16;             // threads per FFT pair
    static constexpr int NTHREADS = T * G;
    static constexpr int NWAVES = (NTHREADS + 63) / 64;
    static constexpr int NP16 = LOG2N / 4;       // radix-16 passes
    static constexpr int RLAST = 1 << (LOG2N % 4);
    // one pad element per 16 (conflict-free radix-16 scatter) needs T to be a multiple of 16 for the reads
    // idx(j + T*e) = rbase(j) + e*ESTRIDE to stay affine: the short factor lengths 64 and 128 go unpadded
    static constexpr bool PAD = (T % 16) == 0;
    static constexpr int ESTRIDE = (T + (PAD ? T / 16 : 0)) * G;       // LDS distance of e -> e+1 (8-byte units)
    static constexpr int PLANE = (N + (PAD ? N / 16 : 0)) * G + G;      // 8-byte units per plane (+G: slot of index N)
    static constexpr size_t LDS_BYTES = (size_t)PLANE * 16; // real plane + imaginary plane
    static_assert(LOG2N >= 6 && LOG2N <= 13, "supported FFT lengths: 64..8192 (64 and 128 only as factors of a long transform)");
    static_assert(NTHREADS >= 64 && NTHREADS <= 1024, "workgroup size");
    static_assert((64 % G) == 0, "G must divide the wave size");
    __device__ static __forceinline__ int idx(int i, int h) { return (i + (PAD ? (i >> 4) : 0)) * G + h; }
    __device__ static __forceinline__ int rbase(int j, int h) { return (j + (PAD ? (j >> 4) : 0)) * G + h; }
};

// six table loads per pass (w^1, w^2, w^3, w^4, w^8, w^12); the other nine twiddles are products
struct Tw6 {
    float2 b1, b2, b3, a1, a2, a3;
};
__device__ __forceinline__ Tw6 load_tw6(const float2* __restrict__ tw, unsigned kb) {
    Tw6 t;
#if defined(SPYFFT_ABL) && (SPYFFT_ABL & 2)
    t.b1 = t.b2 = t.b3 = t.a1 = t.a2 = t.a3 = make_float2(__uint_as_float(kb), 0.5f);
    return t;
#endif
    t.b1 = ldg<float2>(tw, kb);
    t.b2 = ldg<float2>(tw, kb * 2u);
    t.b3 = ldg<float2>(tw, kb * 3u);
    t.a1 = ldg<float2>(tw, kb * 4u);
    t.a2 = ldg<float2>(tw, kb * 8u);
    t.a3 = ldg<float2>(tw, kb * 12u);
    return t;
}
__device__ __forceinline__ void apply_tw(C2 (&v)[16], const Tw6& t) {
    const float2 wb[4] = {make_float2(1.f, 0.f), t.b1, t.b2, t.b3};
    const float2 wa[4] = {make_float2(1.f, 0.f), t.a1, t.a2, t.a3};
#pragma unroll
    for (int r = 1; r < 16; ++r) {
        const int hi = r >> 2, lo = r & 3;
        const float2 w = (hi == 0) ? wb[lo] : (lo == 0 ? wa[hi] : cmul(wa[hi], wb[lo]));
        v[r] = cmul_s(v[r], w);
    }
}

// Forward FFT of the 16 packed values per thread; on return v[e] = Z[j + T*e] (both FFTs).
// `tw[m] = exp(-2 pi i m / N)`.  Contains __syncthreads(): every thread of the workgroup must call
// it.  `re`/`im` = the two LDS planes.  Barrier discipline: every LDS write phase is PRECEDED by a barrier
// (so earlier reads of the planes by any thread - also the caller's - are finished) and followed by one
// before the reads; there is NO barrier after the last reads: a caller that writes the planes itself
// must put a __syncthreads() in front of its writes.
template <int LOG2N, int G>
__device__ __forceinline__ void fft2_forward(C2 (&v)[16], v2f* re, int j, int h, const float2* __restrict__ tw SPY_STAMP_PARAM) {
    using C = Cfg2<LOG2N, G>;
    v2f* const im = re + C::PLANE;
    const int rb = C::rbase(j, h);
    Tw6 tnext;                       // twiddles of the NEXT radix-16 pass: requested before the exchange of the
                                     // current one, so their L2 round trip hides behind the LDS traffic and barriers
#pragma unroll
    for (int p = 0; p < C::NP16; ++p) {
        const int Ns = 1 << (4 * p);
        const int k = j & (Ns - 1);
        if (p > 0) apply_tw(v, tnext);
        dft16p(v);
        SPY_STAMP(sn);
        if (p + 1 < C::NP16) {
            const int Ns1 = Ns * 16;
            const int k1 = j & (Ns1 - 1);
            tnext = load_tw6(tw, (unsigned)(k1 * (C::N / (Ns1 * 16))) * 8u);   // byte offset of tw[k*stride]
        }
        const bool last = (p == C::NP16 - 1) && (C::RLAST == 1);
        if (!last) {
            const int B = ((j >> (4 * p)) << (4 * p + 4)) + k;
            const int wb = C::idx(B, h);
            const int ws = (p == 0) ? G : (Ns + (C::PAD ? Ns / 16 : 0)) * G;
            // write-after-read barrier placed HERE, behind this pass's butterflies, instead of right after the
            // previous reads: a wave that is done reading starts computing at once and meets the others later
            if (!(SPYFFT_ABL & 16)) __syncthreads();
            SPY_STAMP(sn);
            if (!(SPYFFT_ABL & 8)) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    re[wb + r * ws] = v[r].r;
                    im[wb + r * ws] = v[r].i;
                }
            }
            SPY_STAMP(sn);
            if (!(SPYFFT_ABL & 16)) __syncthreads();
            SPY_STAMP(sn);
            if (!(SPYFFT_ABL & 32)) {
#pragma unroll
                for (int e = 0; e < 16; ++e) {
                    v[e].r = re[rb + e * C::ESTRIDE];
                    v[e].i = im[rb + e * C::ESTRIDE];
                }
            } else {
#pragma unroll
                for (int e = 0; e < 16; ++e) { const C2 t = v[e]; v[e].r = t.i * 0.5f; v[e].i = t.r; }
            }
            SPY_STAMP(sn);
        }
    }
    if constexpr (C::RLAST > 1) {
        constexpr int R = C::RLAST;
        constexpr int M = 16 / R;
#pragma unroll
        for (int m = 0; m < M; ++m) {
            C2 t[R];
            const unsigned jb = (unsigned)(j + C::T * m) * 8u;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                t[r] = v[m + r * M];
                if (r > 0) t[r] = cmul_s(t[r], ldg<float2>(tw, jb * (unsigned)r));
            }
            if constexpr (R == 2) dft2p(t);
            else if constexpr (R == 4) dft4p(t);
            else dft8p(t);
#pragma unroll
            for (int r = 0; r < R; ++r) v[m + r * M] = t[r];
        }
    }
}

// Inverse FFT (unnormalised): conj -> forward -> conj.
template <int LOG2N, int G>
__device__ __forceinline__ void fft2_inverse(C2 (&v)[16], v2f* re, int j, int h, const float2* __restrict__ tw) {
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e].i = -v[e].i;
#ifdef SPYFFT_STAMPS
    int sn = 1000;
#endif
    fft2_forward<LOG2N, G>(v, re, j, h, tw SPY_STAMP_ARG);
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e].i = -v[e].i;
}

// Sum NS doubles over the T threads that share `h`; result broadcast to all of them.
// `scratch` = the (currently unused) dynamic LDS buffer.
template <int NTHREADS, int G, int NS>
__device__ __forceinline__ void block_sum(double (&s)[NS], double* scratch, int tid, int h) {
    constexpr int NWAVES = (NTHREADS + 63) / 64;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
#pragma unroll
        for (int off = 32; off >= G; off >>= 1) s[i] += __shfl_xor(s[i], off);
    }
    const int lane = tid & 63, w = tid >> 6;
    if (lane < G) {
#pragma unroll
        for (int i = 0; i < NS; ++i) scratch[(w * G + lane) * NS + i] = s[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < NS; ++i) {
        double tot = 0.0;
        for (int ww = 0; ww < NWAVES; ++ww) tot += scratch[(ww * G + h) * NS + i];
        s[i] = tot;
    }
    __syncthreads();
}

}  // namespace spyfft
