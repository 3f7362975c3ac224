// K6, the plus operator of Wilson's algorithm for lag-domain lengths L = 2^m (wilson_sf.py:154-184):
//
//     beta = real(ifft(g, axis=0));  beta[0] *= 0.5;  beta[L/2] *= 0.5;  beta[L/2+1:] = 0;  g+ = fft(beta, axis=0)
//
// for every entry (i, j) of the (F, n, n) half spectrum g, F = L/2 + 1 (the negative frequencies are the complex
// conjugates and never stored, granger_kernels.h).  plus_kernel (granger_kernels.h) gives one workgroup to one entry,
// gathers its F values 16 bytes at a time from lines 16 n^2 bytes apart and runs twelve radix-4 passes through
// 128 KiB of LDS: 12.4 ms per call at 2049 x 256 x 256, 14x the time of the 4.3 GB it has to move.  Here:
//  * a workgroup takes FOUR adjacent entries: 64 contiguous bytes per frequency row;
//  * two REAL lag sequences travel as ONE complex transform: z = beta_a + i beta_b has the spectrum
//    Z(f) = Ga(f) + i Gb(f) with G(L - f) = conj(G(f)), so one inverse FFT of length L serves two entries and one
//    forward FFT of the windowed z gives both results back: g+_a(f) = (Y(f) + conj(Y(L-f)))/2,
//    g+_b(f) = (Y(f) - conj(Y(L-f)))/(2i).  (real(ifft(.)) ignores the imaginary parts of the DC and Nyquist bins;
//    so does this: they are dropped when Z is built.)
//  * the transforms are register/LDS radix-16 Stockham FFTs in complex128 (thread j holds the points j + T e,
//    T = L/16 threads; the structure of fft_device.h) - three passes and two exchanges for L = 4096 instead of twelve
//    passes; the two pairs of a workgroup go through the same 64 KiB exchange buffer one after the other: 70 KiB of
//    LDS, two workgroups per CU.
// add_S (S = triu(g0) - triu(g0)^H added to every frequency, wilson_sf.py:97-98) is NOT fused: g0 of all entries is
// only known when this kernel has finished.
#pragma once
#include "cd_math.h"

#define SPY_PLUS_KATTR SPY_MIN_WAVES_PER_EU(2)      // two workgroups of 4 waves per CU

namespace spywil {

template <int LOG2L>
struct PCfg {
    static constexpr int L = 1 << LOG2L;
    static constexpr int T = L / 16;                 // threads
    static constexpr int NP16 = LOG2L / 4;
    static constexpr int RLAST = 1 << (LOG2L % 4);
    static constexpr bool PAD = (T % 16) == 0;
    static constexpr int ESTRIDE = T + (PAD ? T / 16 : 0);
    static constexpr int XELEMS = L + (PAD ? L / 16 : 0) + 1;       // exchange buffer, cd units
    static constexpr int SELEMS = 2 * (L / 2 + 1);                  // staging of (Ga, Gb) per frequency
    static constexpr size_t LDS_BYTES = (size_t)(XELEMS > SELEMS ? XELEMS : SELEMS) * sizeof(cd);
    static_assert(LOG2L >= 8 && LOG2L <= 12, "lag-domain lengths 256 .. 4096");
    __device__ static __forceinline__ int idx(int i) { return i + (PAD ? (i >> 4) : 0); }
};

__device__ __forceinline__ cd p_mul_mi(cd a) { return make_double2(a.y, -a.x); }     // * (-i)

__device__ __forceinline__ void p_dft4(cd& t0, cd& t1, cd& t2, cd& t3) {
    const cd a0 = cadd(t0, t2), a1 = csub(t0, t2), a2 = cadd(t1, t3), a3 = p_mul_mi(csub(t1, t3));
    t0 = cadd(a0, a2);
    t1 = cadd(a1, a3);
    t2 = csub(a0, a2);
    t3 = csub(a1, a3);
}

// in-place DFT of 16 values in natural order
__device__ __forceinline__ void p_dft16(cd (&t)[16]) {
    const double c1 = 0.92387953251128675613, s1 = 0.38268343236508977173, h = 0.70710678118654752440;
#pragma unroll
    for (int n2 = 0; n2 < 4; ++n2) p_dft4(t[n2], t[4 + n2], t[8 + n2], t[12 + n2]);
    // t[4 k1 + n2] = y[n2][k1]; twiddles W16^(n2 k1)
    t[5] = cmul(t[5], make_double2(c1, -s1));
    t[9] = make_double2((t[9].x + t[9].y) * h, (t[9].y - t[9].x) * h);
    t[13] = cmul(t[13], make_double2(s1, -c1));
    t[6] = make_double2((t[6].x + t[6].y) * h, (t[6].y - t[6].x) * h);
    t[10] = p_mul_mi(t[10]);
    t[14] = make_double2((t[14].y - t[14].x) * h, -(t[14].x + t[14].y) * h);
    t[7] = cmul(t[7], make_double2(s1, -c1));
    t[11] = make_double2((t[11].y - t[11].x) * h, -(t[11].x + t[11].y) * h);
    t[15] = cmul(t[15], make_double2(-c1, s1));
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1) p_dft4(t[4 * k1], t[4 * k1 + 1], t[4 * k1 + 2], t[4 * k1 + 3]);
#pragma unroll
    for (int k1 = 0; k1 < 4; ++k1)
#pragma unroll
        for (int k2 = k1 + 1; k2 < 4; ++k2) {
            const cd tmp = t[4 * k1 + k2];
            t[4 * k1 + k2] = t[4 * k2 + k1];
            t[4 * k2 + k1] = tmp;
        }
}

template <int R>
__device__ __forceinline__ void p_dftR(cd (&t)[R]) {
    if constexpr (R == 2) {
        const cd a = t[0], b = t[1];
        t[0] = cadd(a, b);
        t[1] = csub(a, b);
    } else if constexpr (R == 4) {
        p_dft4(t[0], t[1], t[2], t[3]);
    } else {
        const double h = 0.70710678118654752440;
        cd a[4] = {t[0], t[2], t[4], t[6]}, b[4] = {t[1], t[3], t[5], t[7]};
        p_dft4(a[0], a[1], a[2], a[3]);
        p_dft4(b[0], b[1], b[2], b[3]);
        b[1] = make_double2((b[1].x + b[1].y) * h, (b[1].y - b[1].x) * h);
        b[2] = p_mul_mi(b[2]);
        b[3] = make_double2((b[3].y - b[3].x) * h, -(b[3].x + b[3].y) * h);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            t[k] = cadd(a[k], b[k]);
            t[k + 4] = csub(a[k], b[k]);
        }
    }
}

// Forward FFT of length L: in v[e] = x[j + T e], out v[e] = X[j + T e].  tw[m] = exp(-2 pi i m / L).
// Every thread of the workgroup must call it; a barrier precedes every LDS write phase (the caller's earlier reads of
// `xb` are safe) and none follows the last reads.
template <int LOG2L>
__device__ __forceinline__ void p_fft(cd (&v)[16], cd* xb, int j, const cd* __restrict__ tw) {
    using C = PCfg<LOG2L>;
#pragma unroll
    for (int p = 0; p < C::NP16; ++p) {
        const int Ns = 1 << (4 * p);
        const int k = j & (Ns - 1);
        if (p > 0) {
            const int st = k * (C::L / (Ns * 16));
            const cd b1 = tw[st], b2 = tw[2 * st], b3 = tw[3 * st], a1 = tw[4 * st], a2 = tw[8 * st], a3 = tw[12 * st];
            const cd wb[4] = {make_double2(1.0, 0.0), b1, b2, b3}, wa[4] = {make_double2(1.0, 0.0), a1, a2, a3};
#pragma unroll
            for (int r = 1; r < 16; ++r) {
                const int hi = r >> 2, lo = r & 3;
                const cd w = (hi == 0) ? wb[lo] : (lo == 0 ? wa[hi] : cmul(wa[hi], wb[lo]));
                v[r] = cmul(v[r], w);
            }
        }
        p_dft16(v);
        const bool last = (p == C::NP16 - 1) && (C::RLAST == 1);
        if (!last) {
            const int B = ((j >> (4 * p)) << (4 * p + 4)) + k;
            const int wbase = C::idx(B);
            const int ws = (p == 0) ? 1 : Ns + (C::PAD ? Ns / 16 : 0);
            __syncthreads();
#pragma unroll
            for (int r = 0; r < 16; ++r) xb[wbase + r * ws] = v[r];
            __syncthreads();
            const int rb = C::idx(j);
#pragma unroll
            for (int e = 0; e < 16; ++e) v[e] = xb[rb + e * C::ESTRIDE];
        }
    }
    if constexpr (C::RLAST > 1) {
        constexpr int R = C::RLAST, M = 16 / R;
#pragma unroll
        for (int m = 0; m < M; ++m) {
            cd t[R];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                t[r] = v[m + r * M];
                if (r > 0) t[r] = cmul(t[r], tw[(j + C::T * m) * r]);
            }
            p_dftR<R>(t);
#pragma unroll
            for (int r = 0; r < R; ++r) v[m + r * M] = t[r];
        }
    }
}

// One pair of entries: (Ga, Gb) half spectra in registers (ga[e], gb[e] = value at f = j + T e, e < 8; thread 0 also
// holds the Nyquist bin in gn_a, gn_b) -> plus operator -> the same registers; returns this pair's g0 on thread 0.
template <int LOG2L>
__device__ __forceinline__ void p_pair(cd (&ga)[8], cd (&gb)[8], cd& gn_a, cd& gn_b, cd* lds, int j, const cd* __restrict__ tw,
                                       double& g0a, double& g0b) {
    using C = PCfg<LOG2L>;
    constexpr int T = C::T, L = C::L, half = L / 2;
    // ---- stage (Ga, Gb) so that every thread can fetch the mirror frequencies of its upper slots
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int f = j + T * e;
        lds[2 * f] = ga[e];
        lds[2 * f + 1] = gb[e];
    }
    if (j == 0) {
        lds[2 * half] = gn_a;
        lds[2 * half + 1] = gn_b;
    }
    __syncthreads();
    // Z(n) = Ga(n) + i Gb(n) for n <= L/2, conj(Ga(L-n)) + i conj(Gb(L-n)) above; DC and Nyquist: real parts only
    cd v[16];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int n = j + T * e;
        const bool up = n > half;
        const int f = up ? L - n : n;
        cd a = lds[2 * f], b = lds[2 * f + 1];
        if (f == 0 || f == half) { a.y = 0.0; b.y = 0.0; }
        if (up) { a.y = -a.y; b.y = -b.y; }
        // inverse transform = conj(FFT(conj(Z))): feed conj(Z) = (a.x - b.y) - i (a.y + b.x)
        v[e] = make_double2(a.x - b.y, -(a.y + b.x));
    }
    p_fft<LOG2L>(v, lds, j, tw);
    // z(t) = conj(v)/L; window: halve lags 0 and L/2, zero the negative lags (t > L/2)
    const double invL = 1.0 / (double)L;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int t = j + T * e;
        double w = t > half ? 0.0 : invL;
        if (t == 0 || t == half) w *= 0.5;
        v[e] = make_double2(v[e].x * w, -v[e].y * w);
    }
    if (j == 0) {
        g0a = v[0].x;
        g0b = v[0].y;
    }
    p_fft<LOG2L>(v, lds, j, tw);
    // ---- separate the two results: partner bin L - f lives in the upper slots
    __syncthreads();
#pragma unroll
    for (int e = 8; e < 16; ++e) lds[C::idx(j + T * e)] = v[e];
    __syncthreads();
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const int f = j + T * e;
        const cd y = v[e];
        cd yp = f == 0 ? y : lds[C::idx(L - f)];
        ga[e] = make_double2(0.5 * (y.x + yp.x), 0.5 * (y.y - yp.y));
        gb[e] = make_double2(0.5 * (y.y + yp.y), 0.5 * (yp.x - y.x));
    }
    if (j == 0) {                       // Nyquist: its own partner
        gn_a = make_double2(v[8].x, 0.0);
        gn_b = make_double2(v[8].y, 0.0);
    }
}

// grid = plus4_grid(nent) workgroups of T threads, ONE entry pair each (32 bytes of every frequency row); g, gp:
// (F, nent) complex128 (nent = n^2 for a whole matrix, fewer for an entry shard of the frequency-sharded factorisation);
// g0: (nent).  Four consecutive pairs share the 128-byte lines of every row, so they are dealt to four workgroups that
// the dispatcher places on the SAME XCD one after the other (block b -> XCD b % 8): the line is fetched from HBM once
// and found in that XCD's L2 by the other three.  64 data registers, two workgroups per CU.
__host__ __device__ inline long long plus4_grid(long long nent) { return (((nent + 1) / 2 + 31) / 32) * 32; }
template <int LOG2L>
__global__ void __launch_bounds__((PCfg<LOG2L>::T)) SPY_PLUS_KATTR plus4_kernel(const cd* g, int F, long long nent, const cd* tw, cd* gp, cd* g0) {
    using C = PCfg<LOG2L>;
    constexpr int T = C::T, half = C::L / 2;
    SPY_DYN_SMEM(cd, lds);
    const int j = threadIdx.x;
    const size_t nn = (size_t)nent;          // entries per frequency row
    const size_t b = blockIdx.x;
    const size_t e0 = 2 * ((b / 32) * 32 + (b % 8) * 4 + (b / 8) % 4);
    if (e0 >= nn) return;                                         // workgroup-uniform
    const bool two = e0 + 1 < nn;
    cd xa[8], xb[8], na = make_double2(0.0, 0.0), nb = na;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        const cd* row = g + (size_t)(j + T * e) * nn + e0;
        xa[e] = row[0];
        xb[e] = two ? row[1] : make_double2(0.0, 0.0);
    }
    if (j == 0) {
        const cd* row = g + (size_t)half * nn + e0;
        na = row[0];
        nb = two ? row[1] : make_double2(0.0, 0.0);
    }
    double z0a = 0.0, z0b = 0.0;
    p_pair<LOG2L>(xa, xb, na, nb, lds, j, tw, z0a, z0b);
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        cd* row = gp + (size_t)(j + T * e) * nn + e0;
        row[0] = xa[e];
        if (two) row[1] = xb[e];
    }
    if (j == 0) {
        cd* row = gp + (size_t)half * nn + e0;
        row[0] = na;
        g0[e0] = make_double2(z0a, 0.0);
        if (two) {
            row[1] = nb;
            g0[e0 + 1] = make_double2(z0b, 0.0);
        }
    }
    (void)F;
}

}  // namespace spywil
