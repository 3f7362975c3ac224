// Complex128 Stockham passes of ANY radix over length-L working arrays in LDS or global memory, 256 threads per
// transform: the plus operator of the Wilson factorisation (granger_kernels.h) and the reference-precision tapered FFT of
// lengths the radix-16 register kernel does not serve (mtmfft_f64_kernel.h).
#pragma once
#include "cd_math.h"

namespace spywil {

constexpr int PO_MAXFAC = 24;
struct PlusPlan {
    int L, nfac;
    int radix[PO_MAXFAC];
};

__device__ __forceinline__ void po_pass(const cd* in, cd* out, int L, int R, int Ns, const cd* tw, int sign, int tid) {
    // tw[m] = exp(-2 pi i m / L); sign = -1 forward, +1 inverse (conjugated twiddles)
    const int nb = L / R, tws = L / (Ns * R), wr = L / R;
    for (int jb = tid; jb < nb; jb += 256) {
        const int k = jb % Ns;
        const int base = (jb / Ns) * Ns * R + k;
        for (int q = 0; q < R; ++q) {
            cd s = make_double2(0.0, 0.0);
            for (int r = 0; r < R; ++r) {
                cd x = in[jb + r * nb];
                // twiddle exp(-+2 pi i r k / (Ns R)) and DFT kernel exp(-+2 pi i r q / R)
                long long idx = ((long long)r * k * tws + (long long)((r * q) % R) * wr) % L;
                cd w = tw[idx];
                if (sign > 0) w.y = -w.y;
                s = cadd(s, cmul(x, w));
            }
            out[base + q * Ns] = s;
        }
    }
}

// radix-2 / radix-4 Stockham passes with real butterflies (the lag-domain length is a power of two whenever the
// trial length is): one table twiddle per input instead of the R^2 table products of the generic pass
template <int R>
__device__ __forceinline__ void po_pass_r24(const cd* in, cd* out, int L, int Ns, const cd* tw, int sign, int tid) {
    const int nb = L / R, tws = L / (Ns * R);
    for (int jb = tid; jb < nb; jb += 256) {
        const int k = jb % Ns;
        const int base = (jb / Ns) * Ns * R + k;
        cd x[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            x[r] = in[jb + r * nb];
            if (r > 0 && Ns > 1) {
                cd w = tw[r * k * tws];
                if (sign > 0) w.y = -w.y;
                x[r] = cmul(x[r], w);
            }
        }
        if (R == 2) {
            out[base] = cadd(x[0], x[1]);
            out[base + Ns] = csub(x[0], x[1]);
        } else {
            const cd a0 = cadd(x[0], x[2]), a1 = csub(x[0], x[2]), a2 = cadd(x[1], x[3]), d = csub(x[1], x[3]);
            // forward: multiply d by -i; inverse: by +i
            const cd a3 = sign > 0 ? make_double2(-d.y, d.x) : make_double2(d.y, -d.x);
            out[base] = cadd(a0, a2);
            out[base + Ns] = cadd(a1, a3);
            out[base + 2 * Ns] = csub(a0, a2);
            out[base + 3 * Ns] = csub(a1, a3);
        }
    }
}

// radix-3 / 5 / 7 passes: one table twiddle per input and the R x R DFT kernel from compile-time constants (the
// generic pass fetches R^2 table products per butterfly through a 64-bit modulo) - the decimal trial lengths of the
// plus operator (5000 samples = 2^3 5^4) and of the reference-precision transform (2000, 5000 ...)
template <int R> struct DftConst;
template <> struct DftConst<3> {
    static constexpr double c[3] = {1.0, -0.5, -0.5};
    static constexpr double s[3] = {0.0, 0.86602540378443864676, -0.86602540378443864676};
};
template <> struct DftConst<5> {
    static constexpr double c[5] = {1.0, 0.30901699437494742410, -0.80901699437494742410, -0.80901699437494742410,
                                    0.30901699437494742410};
    static constexpr double s[5] = {0.0, 0.95105651629515357212, 0.58778525229247312917, -0.58778525229247312917,
                                    -0.95105651629515357212};
};
template <> struct DftConst<7> {
    static constexpr double c[7] = {1.0, 0.62348980185873353053, -0.22252093395631440429, -0.90096886790241912624,
                                    -0.90096886790241912624, -0.22252093395631440429, 0.62348980185873353053};
    static constexpr double s[7] = {0.0, 0.78183148246802980871, 0.97492791218182360702, 0.43388373911755812048,
                                    -0.43388373911755812048, -0.97492791218182360702, -0.78183148246802980871};
};

template <int R>
__device__ __forceinline__ void po_pass_small(const cd* in, cd* out, int L, int Ns, const cd* tw, int sign, int tid) {
    const int nb = L / R, tws = L / (Ns * R);
    const double sg = sign > 0 ? 1.0 : -1.0;            // exp(sign 2 pi i m / R) = c[m] + i sign s[m]
    for (int jb = tid; jb < nb; jb += 256) {
        const int k = jb % Ns;
        const int base = (jb / Ns) * Ns * R + k;
        cd x[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            x[r] = in[jb + r * nb];
            if (r > 0 && Ns > 1) {
                cd w = tw[r * k * tws];
                if (sign > 0) w.y = -w.y;
                x[r] = cmul(x[r], w);
            }
        }
#pragma unroll
        for (int q = 0; q < R; ++q) {
            cd acc = x[0];
#pragma unroll
            for (int r = 1; r < R; ++r) {
                constexpr int dummy = 0; (void)dummy;
                const int m = (r * q) % R;
                const double wr = DftConst<R>::c[m], wi = sg * DftConst<R>::s[m];
                acc.x += x[r].x * wr - x[r].y * wi;
                acc.y += x[r].x * wi + x[r].y * wr;
            }
            out[base + q * Ns] = acc;
        }
    }
}

__device__ __forceinline__ void po_pass_any(const cd* in, cd* out, int L, int R, int Ns, const cd* tw, int sign, int tid) {
    if (R == 4) po_pass_r24<4>(in, out, L, Ns, tw, sign, tid);
    else if (R == 2) po_pass_r24<2>(in, out, L, Ns, tw, sign, tid);
    else if (R == 5) po_pass_small<5>(in, out, L, Ns, tw, sign, tid);
    else if (R == 3) po_pass_small<3>(in, out, L, Ns, tw, sign, tid);
    else if (R == 7) po_pass_small<7>(in, out, L, Ns, tw, sign, tid);
    else po_pass(in, out, L, R, Ns, tw, sign, tid);
}

// factors of L in the order the passes take them: 4, 2, 3, 5, 7, 11, 13, then the remaining primes
inline bool plus_plan(int L, PlusPlan* pl) {
    pl->L = L;
    int k = 0, n = L;
    static const int cand[] = {4, 2, 3, 5, 7, 11, 13};
    for (int c : cand)
        while (n % c == 0 && n > 1) { if (k >= PO_MAXFAC) return false; pl->radix[k++] = c; n /= c; }
    for (int p = 17; n > 1; p += 2)
        while (n % p == 0) { if (k >= PO_MAXFAC) return false; pl->radix[k++] = p; n /= p; }
    pl->nfac = k;
    return true;
}

}  // namespace spywil
