// Any-length variant of the mtmfft/STFT kernel: mixed-radix Stockham in LDS
// (radices 16/8/4/2/3/5/7/11/13) and Bluestein's chirp-z for lengths with
// larger prime factors.  Same semantics and argument block as
// mtmfft_pow2_kernel; one workgroup (256 threads) = one channel pair of one
// segment.  This path serves trial lengths like 2000 (BASELINE config 1) and
// STFT windows like 500; the power-of-two kernel is the tuned one.
#pragma once
#include "mtmfft_kernel.h"

namespace spyfft {

constexpr int GEN_THREADS = 256;
constexpr int GEN_MAXFAC = 20;

struct GenPlan {
    int n;                     // transform length of the Stockham passes (nfft, or M for Bluestein)
    int nfac;
    int radix[GEN_MAXFAC];
    int nfft;                  // logical FFT length (output bins = nfft/2+1)
    int bluestein;             // 0/1
    const float2* chirp;       // nfft entries: exp(-i pi n^2 / nfft)          (Bluestein)
    const float2* bhat;        // n entries: FFT_M of the chirp filter / M     (Bluestein)
    int stage_x;               // segment staged in LDS (else re-read per taper)
};

template <int R>
__device__ __forceinline__ void dft_prime(float2 (&t)[R], const float2* __restrict__ tw, int nover_r) {
    // O(R^2) DFT with W_R^m = tw[m * (N/R)]
    float2 w[R];
#pragma unroll
    for (int m = 0; m < R; ++m) w[m] = tw[m * nover_r];
    float2 o[R];
#pragma unroll
    for (int q = 0; q < R; ++q) {
        float2 s = t[0];
#pragma unroll
        for (int r = 1; r < R; ++r) s = cadd(s, cmul(t[r], w[(r * q) % R]));
        o[q] = s;
    }
#pragma unroll
    for (int q = 0; q < R; ++q) t[q] = o[q];
}

template <int R>
__device__ __forceinline__ void stockham_pass(const float2* in, float2* out, int n, int Ns,
                                              const float2* __restrict__ tw, int tid) {
    const int nb = n / R;             // butterflies
    const int tws = n / (Ns * R);     // twiddle stride
    for (int jb = tid; jb < nb; jb += GEN_THREADS) {
        const int k = jb % Ns;
        float2 t[R];
#pragma unroll
        for (int r = 0; r < R; ++r) {
            t[r] = in[jb + r * nb];
            if (r > 0 && Ns > 1) t[r] = cmul(t[r], tw[r * k * tws]);
        }
        if constexpr (R == 2 || R == 4 || R == 8 || R == 16) {
            dft<R>(t);  // power-of-two radices share the register butterflies of fft_device.h
        } else {
            dft_prime<R>(t, tw, n / R);
        }
        const int base = (jb / Ns) * Ns * R + k;
#pragma unroll
        for (int r = 0; r < R; ++r) out[base + r * Ns] = t[r];
    }
}

// Forward FFT of length g.n on LDS buffer `a` (scratch `b`); returns the
// buffer that holds the result.  tw = exp(-2 pi i m / g.n).
__device__ __forceinline__ float2* stockham_fft(float2* a, float2* b, const GenPlan& g,
                                                const float2* __restrict__ tw, int tid) {
    int Ns = 1;
    for (int p = 0; p < g.nfac; ++p) {
        const int R = g.radix[p];
        switch (R) {
            case 16: stockham_pass<16>(a, b, g.n, Ns, tw, tid); break;
            case 8: stockham_pass<8>(a, b, g.n, Ns, tw, tid); break;
            case 4: stockham_pass<4>(a, b, g.n, Ns, tw, tid); break;
            case 2: stockham_pass<2>(a, b, g.n, Ns, tw, tid); break;
            case 3: stockham_pass<3>(a, b, g.n, Ns, tw, tid); break;
            case 5: stockham_pass<5>(a, b, g.n, Ns, tw, tid); break;
            case 7: stockham_pass<7>(a, b, g.n, Ns, tw, tid); break;
            case 11: stockham_pass<11>(a, b, g.n, Ns, tw, tid); break;
            default: stockham_pass<13>(a, b, g.n, Ns, tw, tid); break;
        }
        __syncthreads();
        Ns *= R;
        float2* t = a;
        a = b;
        b = t;
    }
    return a;
}

__device__ __forceinline__ void gen_block_sum4(double (&s)[4], double* scratch, int tid) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) s[i] += __shfl_xor(s[i], off);
    }
    const int lane = tid & 63, w = tid >> 6;
    if (lane == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) scratch[w * 4 + i] = s[i];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        double tot = 0.0;
        for (int ww = 0; ww < GEN_THREADS / 64; ++ww) tot += scratch[ww * 4 + i];
        s[i] = tot;
    }
    __syncthreads();
}

template <int OUTK, bool MEAN>
__global__ void __launch_bounds__(GEN_THREADS) mtmfft_generic_kernel(MtmArgs a, GenPlan g) {
    constexpr bool CPLX = (OUTK == 2);
    SPY_DYN_SMEM(float2, lds);
    // LDS carve: [xbuf: nsig (if staged)] [bufA: g.n] [bufB: g.n]
    float2* xbuf = lds;
    float2* bufA = lds + (g.stage_x ? a.nsig : 0);
    float2* bufB = bufA + g.n;

    const int tid = threadIdx.x;
    const int npairs = (a.nchan + 1) / 2;
    const long long id = blockIdx.x;
    const int b = (int)(id / npairs);
    const int c0 = 2 * (int)(id % npairs), c1 = c0 + 1;
    if (b >= a.nseg) return;
    const bool has1 = c1 < a.nchan;
    const long long col0 = a.chan_idx ? a.chan_idx[c0] : c0;
    const long long col1 = has1 ? (a.chan_idx ? a.chan_idx[c1] : c1) : 0;
    const long long start = a.seg_start[b], lo = a.seg_lo[b], hi = a.seg_hi[b];

    auto load = [&](int n) -> float2 {
        const long long row = start + n;
        float2 u = make_float2(0.f, 0.f);
        if (row >= lo && row < hi) {
            const float* p = a.data + row * a.ld;
            u.x = p[col0];
            if (has1) u.y = p[col1];
        }
        return u;
    };

    // ---- detrend coefficients
    const double mid = 0.5 * (a.nsig - 1);
    double m0 = 0.0, m1 = 0.0, b0 = 0.0, b1 = 0.0;
    if (a.detrend == 0 && a.means) {
        m0 = (double)a.means[(size_t)b * a.nchan + c0];            // the reference-order float32 means
        m1 = has1 ? (double)a.means[(size_t)b * a.nchan + c1] : 0.0;
    } else if (a.detrend >= 0) {
        double s[4] = {0.0, 0.0, 0.0, 0.0};
        for (int n = tid; n < a.nsig; n += GEN_THREADS) {
            const float2 u = load(n);
            s[0] += u.x;
            s[1] += u.y;
            if (a.detrend == 1) {
                s[2] += (n - mid) * u.x;
                s[3] += (n - mid) * u.y;
            }
        }
        gen_block_sum4(s, reinterpret_cast<double*>(bufA), tid);
        m0 = s[0] / a.nsig;
        m1 = s[1] / a.nsig;
        if (a.detrend == 1 && a.nsig > 1) {
            const double den = 12.0 / ((double)a.nsig * ((double)a.nsig * a.nsig - 1.0));
            b0 = s[2] * den;
            b1 = s[3] * den;
        }
    }
    auto sample = [&](int n) -> float2 {
        float2 u = load(n);
        if (a.detrend >= 0) {
            u.x -= (float)(m0 + b0 * (n - mid));
            u.y -= (float)(m1 + b1 * (n - mid));
        }
        return u;
    };
    if (g.stage_x) {
        for (int n = tid; n < a.nsig; n += GEN_THREADS) xbuf[n] = sample(n);
        __syncthreads();
    }

    const int nf = g.nfft / 2 + 1;
    const int kout = MEAN ? 1 : a.ntaper;
    const float hs = 0.5f * a.scale;
    // twiddles of the Stockham passes follow the chirp tables for Bluestein
    const float2* tw = a.tw;

    for (int k = 0; k < a.ntaper; ++k) {
        const float* w = a.tapers + (size_t)k * a.nsig;
        // ---- taper (and optional second demean)
        float dm0 = 0.f, dm1 = 0.f;
        if (a.demean_taper) {
            double s[4] = {0.0, 0.0, 0.0, 0.0};
            for (int n = tid; n < a.nsig; n += GEN_THREADS) {
                const float2 u = g.stage_x ? xbuf[n] : sample(n);
                s[0] += w[n] * u.x;
                s[1] += w[n] * u.y;
            }
            gen_block_sum4(s, reinterpret_cast<double*>(bufA), tid);
            dm0 = (float)(s[0] / a.nsig);
            dm1 = (float)(s[1] / a.nsig);
        }
        for (int n = tid; n < g.n; n += GEN_THREADS) {
            float2 z = make_float2(0.f, 0.f);
            if (n < a.nsig) {
                const float2 u = g.stage_x ? xbuf[n] : sample(n);
                z = make_float2(w[n] * u.x - dm0, w[n] * u.y - dm1);
                if (g.bluestein) z = cmul(z, g.chirp[n]);
            }
            bufA[n] = z;
        }
        __syncthreads();
        float2* Z = stockham_fft(bufA, bufB, g, tw, tid);
        if (g.bluestein) {
            // circular convolution with the chirp filter: Z *= bhat, inverse FFT, * chirp
            float2* other = (Z == bufA) ? bufB : bufA;
            for (int n = tid; n < g.n; n += GEN_THREADS) {
                const float2 t = cmul(Z[n], g.bhat[n]);
                Z[n] = make_float2(t.x, -t.y);  // conj for the inverse transform
            }
            __syncthreads();
            Z = stockham_fft(Z, other, g, tw, tid);
            for (int n = tid; n < g.nfft; n += GEN_THREADS) {
                const float2 t = make_float2(Z[n].x, -Z[n].y);
                Z[n] = cmul(t, g.chirp[n]);
            }
            __syncthreads();
        }

        // ---- separate channels, convert, store / accumulate
        for (int f = tid; f < nf; f += GEN_THREADS) {
            const int fi = a.fpos ? a.fpos[f] : f;
            if (fi < 0) continue;
            const float2 z = Z[f];
            const float2 zp = Z[(g.nfft - f) % g.nfft];
            const float2 xa = make_float2(hs * (z.x + zp.x), hs * (z.y - zp.y));
            const float2 xb = make_float2(hs * (z.y + zp.y), hs * (zp.x - z.x));
            const size_t o = (((size_t)b * kout + (MEAN ? 0 : k)) * a.nfsel + fi) * a.nchan + c0;
            if (CPLX) {
                float2* out = reinterpret_cast<float2*>(a.out);
                if (MEAN) {
                    float2 p0 = xa, p1 = xb;
                    if (k > 0) {
                        p0 = cadd(p0, out[o]);
                        if (has1) p1 = cadd(p1, out[o + 1]);
                    }
                    if (k == a.ntaper - 1) {
                        const float kk = (float)a.ntaper;
                        p0 = make_float2(p0.x / kk, p0.y / kk);
                        p1 = make_float2(p1.x / kk, p1.y / kk);
                    }
                    out[o] = p0;
                    if (has1) out[o + 1] = p1;
                } else {
                    out[o] = xa;
                    if (has1) out[o + 1] = xb;
                }
            } else {
                float* out = reinterpret_cast<float*>(a.out);
                float p0 = convert_real<OUTK>(xa, a.out_kind), p1 = convert_real<OUTK>(xb, a.out_kind);
                if (MEAN) {
                    if (k > 0) {
                        p0 += out[o];
                        if (has1) p1 += out[o + 1];
                    }
                    if (k == a.ntaper - 1) {
                        p0 /= (float)a.ntaper;
                        p1 /= (float)a.ntaper;
                    }
                }
                out[o] = p0;
                if (has1) out[o + 1] = p1;
            }
        }
        __syncthreads();
    }
}

}  // namespace spyfft
