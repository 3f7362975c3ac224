// Register/LDS radix-16 Stockham FFT for gfx950, shared by the mtmfft / STFT /
// CWT kernels.
//
// Decomposition (one complex length-N FFT, N = 2^LOG2N, by T = N/16 threads):
//   every pass, thread j holds v[e] = in[j + T*e], e = 0..15   (same read
//   pattern in every pass, lanes -> consecutive LDS words, conflict free);
//   radix-16 passes with Ns = 16^p write out[(j/Ns)*16*Ns + j%Ns + r*Ns];
//   a final radix-R pass (R = N / 16^p in {2,4,8}) does 16/R butterflies per
//   thread and leaves the spectrum in place: v[e] = Z[j + T*e].
// G independent FFTs ("pairs") are interleaved in one workgroup: thread id
// = j*G + h, LDS element (i,h) at (i + i/16)*G + h (float2 units).  The +1
// element pad per 16 keeps the radix-16 scatter of pass 0 conflict free for
// ds_write_b64 lane groups (MI355X_MICROARCH.md, LDS table).
#pragma once

namespace spyfft {

__device__ __forceinline__ float2 cmul(float2 a, float2 b) {
    return make_float2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x);
}
__device__ __forceinline__ float2 cadd(float2 a, float2 b) { return make_float2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ float2 csub(float2 a, float2 b) { return make_float2(a.x - b.x, a.y - b.y); }
// multiply by -i (forward-transform quarter turn)
__device__ __forceinline__ float2 mul_mi(float2 a) { return make_float2(a.y, -a.x); }

// load through a wave-uniform base + 32-bit byte offset (saddr-form global load)
template <typename T>
__device__ __forceinline__ T ldg(const void* base, unsigned byte_off) {
    return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off);
}
template <typename T>
__device__ __forceinline__ void stg(void* base, unsigned byte_off, T v) {
    *reinterpret_cast<T*>(reinterpret_cast<char*>(base) + byte_off) = v;
}
// hide a value from loop-invariant code motion: index arithmetic that depends on
// it is redone per iteration instead of being kept live in dozens of VGPRs
__device__ __forceinline__ int opaque(int v) { return spy_opaque(v); }

// value of the neighbouring lane (lane ^ 1): one DPP move, no LDS traffic
__device__ __forceinline__ float lane_swap1(float v) { return spy_lane_swap1(v); }

__device__ __forceinline__ void sched_fence() { spy_sched_fence(); }

template <int R>
__device__ __forceinline__ void dft(float2 (&t)[R]);

template <>
__device__ __forceinline__ void dft<2>(float2 (&t)[2]) {
    float2 a = t[0], b = t[1];
    t[0] = cadd(a, b);
    t[1] = csub(a, b);
}

template <>
__device__ __forceinline__ void dft<4>(float2 (&t)[4]) {
    float2 a0 = cadd(t[0], t[2]), a1 = csub(t[0], t[2]);
    float2 a2 = cadd(t[1], t[3]), a3 = mul_mi(csub(t[1], t[3]));
    t[0] = cadd(a0, a2);
    t[1] = cadd(a1, a3);
    t[2] = csub(a0, a2);
    t[3] = csub(a1, a3);
}

template <>
__device__ __forceinline__ void dft<8>(float2 (&t)[8]) {
    // n = 2*n1 + n2, k = k1 + 4*k2
    const float h = 0.70710678118654752440f;
    float2 a[4] = {t[0], t[2], t[4], t[6]};
    float2 b[4] = {t[1], t[3], t[5], t[7]};
    dft<4>(a);
    dft<4>(b);
    b[1] = make_float2(h * (b[1].x + b[1].y), h * (b[1].y - b[1].x));   // * W8^1
    b[2] = mul_mi(b[2]);                                                  // * W8^2
    b[3] = make_float2(h * (b[3].y - b[3].x), -h * (b[3].x + b[3].y));  // * W8^3
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) {
        t[k1] = cadd(a[k1], b[k1]);
        t[k1 + 4] = csub(a[k1], b[k1]);
    }
}

template <>
__device__ __forceinline__ void dft<16>(float2 (&t)[16]) {
    // n = 4*n1 + n2, k = k1 + 4*k2 ; y[n2][k1] = W16^(n2*k1) * DFT4_{n1}(t[4*n1+n2])[k1]
    const float c1 = 0.92387953251128675613f, s1 = 0.38268343236508977173f;
    const float h = 0.70710678118654752440f;
    float2 y[4][4];
#pragma unroll
    for (int n2 = 0; n2 < 4; ++n2) {
        float2 a[4] = {t[n2], t[4 + n2], t[8 + n2], t[12 + n2]};
        dft<4>(a);
#pragma unroll
        for (int k1 = 0; k1 < 4; ++k1) y[n2][k1] = a[k1];
    }
    // twiddles W16^m = exp(-2 pi i m / 16)
    y[1][1] = cmul(y[1][1], make_float2(c1, -s1));   // m = 1
    y[1][2] = cmul(y[1][2], make_float2(h, -h));     // m = 2
    y[1][3] = cmul(y[1][3], make_float2(s1, -c1));   // m = 3
    y[2][1] = cmul(y[2][1], make_float2(h, -h));     // m = 2
    y[2][2] = mul_mi(y[2][2]);                       // m = 4
    y[2][3] = cmul(y[2][3], make_float2(-h, -h));    // m = 6
    y[3][1] = cmul(y[3][1], make_float2(s1, -c1));   // m = 3
    y[3][2] = cmul(y[3][2], make_float2(-h, -h));    // m = 6
    y[3][3] = cmul(y[3][3], make_float2(-c1, s1));   // m = 9
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) {
        float2 b[4] = {y[0][k1], y[1][k1], y[2][k1], y[3][k1]};
        dft<4>(b);
#pragma unroll
        for (int k2 = 0; k2 < 4; ++k2) t[k1 + 4 * k2] = b[k2];
    }
}

template <int LOG2N, int G>
struct Cfg {
    static constexpr int N = 1 << LOG2N;
    static constexpr int T = N / 16;             // threads per FFT
    static constexpr int NTHREADS = T * G;
    static constexpr int NWAVES = (NTHREADS + 63) / 64;
    static constexpr int NP16 = LOG2N / 4;       // radix-16 passes
    static constexpr int RLAST = 1 << (LOG2N % 4);
    static constexpr int ESTRIDE = (T + T / 16) * G;       // LDS distance of e -> e+1 (float2 units)
    static constexpr int LDS_ELEMS = (N + N / 16) * G + G;  // float2 units (+G: slot of index N)
    static constexpr size_t LDS_BYTES = (size_t)LDS_ELEMS * 8;
    static_assert(LOG2N >= 8 && LOG2N <= 14, "supported FFT lengths: 256..16384");
    static_assert(NTHREADS >= 64 && NTHREADS <= 1024, "workgroup size");
    static_assert((64 % G) == 0, "G must divide the wave size");
    __device__ static __forceinline__ int idx(int i, int h) { return (i + (i >> 4)) * G + h; }
    // idx(j + T*e, h) = rbase(j,h) + e*ESTRIDE  (T is a multiple of 16)
    __device__ static __forceinline__ int rbase(int j, int h) { return (j + (j >> 4)) * G + h; }
};

// Forward FFT of the 16 values per thread; on return v[e] = Z[j + T*e].
// `tw[m] = exp(-2 pi i m / N)`, m < N.  Contains __syncthreads(): must be
// called by every thread of the workgroup.  The LDS buffer may be reused by
// the caller after return (the last pass does not touch it).
template <int LOG2N, int G>
__device__ __forceinline__ void fft_forward(float2 (&v)[16], float2* lds, int j, int h,
                                            const float2* __restrict__ tw) {
    using C = Cfg<LOG2N, G>;
    // All LDS addresses are (one lane-dependent base) + (compile-time constant):
    // reads  idx(j + T*e)      = rbase + e*ESTRIDE
    // writes idx(B + r*Ns)     = idx(B) + r*(Ns + Ns/16)*G   (Ns >= 16; for Ns = 1: + r*G)
    float2* const rd = lds + C::rbase(j, h);
#pragma unroll
    for (int p = 0; p < C::NP16; ++p) {
        const int Ns = 1 << (4 * p);
        const int k = j & (Ns - 1);
        if (p > 0) {
            // twiddles w^r, r = 1..15, w = tw[k*stride]: six table loads (w^1,w^2,w^3,w^4,w^8,w^12) issued
            // together, the other nine as w^(4a) * w^b - one extra fp32 rounding (~6e-8) instead of nine
            // more dependent L2 round trips per pass
            const unsigned kb = (unsigned)(k * (C::N / (Ns * 16))) * 8u;   // byte offset of tw[k*stride]
            float2 wb[4], wa[4];
#if defined(SPYFFT_ABL) && (SPYFFT_ABL & 2)
            wb[1] = wb[2] = wb[3] = wa[1] = wa[2] = wa[3] = make_float2(__uint_as_float(kb), 0.5f);
#else
            wb[1] = ldg<float2>(tw, kb);
            wb[2] = ldg<float2>(tw, kb * 2u);
            wb[3] = ldg<float2>(tw, kb * 3u);
            wa[1] = ldg<float2>(tw, kb * 4u);
            wa[2] = ldg<float2>(tw, kb * 8u);
            wa[3] = ldg<float2>(tw, kb * 12u);
#endif
#pragma unroll
            for (int r = 1; r < 16; ++r) {
                const int hi = r >> 2, lo = r & 3;
                const float2 w = (hi == 0) ? wb[lo] : (lo == 0 ? wa[hi] : cmul(wa[hi], wb[lo]));
                v[r] = cmul(v[r], w);
            }
        }
        dft<16>(v);
        const bool last = (p == C::NP16 - 1) && (C::RLAST == 1);
        if (!last) {
            const int B = ((j >> (4 * p)) << (4 * p + 4)) + k;
            float2* const wr = lds + C::idx(B, h);
            const int ws = (p == 0) ? G : (Ns + Ns / 16) * G;
#pragma unroll
            for (int r = 0; r < 16; ++r) wr[r * ws] = v[r];
            __syncthreads();
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] = rd[e * C::ESTRIDE];
            __syncthreads();
        }
    }
    if constexpr (C::RLAST > 1) {
        constexpr int R = C::RLAST;
        constexpr int M = 16 / R;
#pragma unroll
        for (int m = 0; m < M; ++m) {
            float2 t[R];
            const unsigned jb = (unsigned)(j + C::T * m) * 8u;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                t[r] = v[m + r * M];
                if (r > 0) t[r] = cmul(t[r], ldg<float2>(tw, jb * (unsigned)r));
            }
            dft<R>(t);
#pragma unroll
            for (int r = 0; r < R; ++r) v[m + r * M] = t[r];
        }
    }
}

// Inverse FFT (unnormalised): conj -> forward -> conj.
template <int LOG2N, int G>
__device__ __forceinline__ void fft_inverse(float2 (&v)[16], float2* lds, int j, int h,
                                            const float2* __restrict__ tw) {
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e].y = -v[e].y;
    fft_forward<LOG2N, G>(v, lds, j, h, tw);
#pragma unroll
    for (int e = 0; e < 16; ++e) v[e].y = -v[e].y;
}

// Sum 4 doubles over the T threads that share `h`; result broadcast to all of
// them.  `scratch` = the (currently unused) dynamic LDS buffer.
template <int LOG2N, int G>
__device__ __forceinline__ void block_sum4(double (&s)[4], double* scratch, int tid, int h) {
    using C = Cfg<LOG2N, G>;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int off = 32; off >= G; off >>= 1) s[i] += __shfl_xor(s[i], off);
    }
    const int lane = tid & 63, w = tid >> 6;
    if (lane < G) {
#pragma unroll
        for (int i = 0; i < 4; ++i) scratch[(w * G + lane) * 4 + i] = s[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        double tot = 0.0;
        for (int ww = 0; ww < C::NWAVES; ++ww) tot += scratch[(ww * G + h) * 4 + i];
        s[i] = tot;
    }
    __syncthreads();
}

}  // namespace spyfft
