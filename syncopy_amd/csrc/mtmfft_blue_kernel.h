// K1/K2 for lengths that are not a power of two (nfft <= 4096): Bluestein's chirp-z transform on the packed
// power-of-two engine.  Same semantics, arguments and channel packing as mtmfft_quad_kernel (mtmfft2_kernel.h);
// the length-nfft DFT of the tapered segment z[n] becomes a circular convolution of length M = 2^LOG2N >= 2 nfft - 1:
//     Z[k] = c[k] * IFFT_M( FFT_M(z c) * Bhat )[k],   c[n] = exp(-i pi n^2 / nfft),  Bhat = FFT_M(conj(c) wrapped) / M
// (chirp phases reduced exactly on the host: n^2 mod 2 nfft).  Two packed FFTs of length M and three pointwise
// products per taper replace the mixed-radix LDS kernel (mtmfft_generic.h), which stays for nfft > 4096.
// Reference semantics: specest/mtmfft.py:16-129, specest/compRoutines.py:169-189, specest/stft.py:101-154.
#pragma once
#include "mtmfft2_kernel.h"

namespace spyfft {

// LOG2N = log2(M); every thread still carries 16 packed values per transform; output bins f <= nfft/2 < M/4 + 1
// sit in the first NBIN register slots of the natural-order layout (f = j + T*e).
template <int LOG2N, int G, int OUTK, bool MEAN>
__global__ void __launch_bounds__((Cfg2<LOG2N, G>::NTHREADS)) SPYFFT_KATTR mtmfft_blue_kernel(MtmArgs a) {
    using C = Cfg2<LOG2N, G>;
    constexpr bool CPLX = (OUTK == 2);
    constexpr int T = C::T;
    SPY_DYN_SMEM(v2f, lds);
    v2f* const lre = lds;
    v2f* const lim = lds + C::PLANE;

    const int tid = threadIdx.x;
    const int h = tid % G, j0 = tid / G;

    // XCD-aware block -> (segment, quad group): the S workgroups that share 128-byte lines of the
    // (time x channel) rows get ids congruent mod 8 (same XCD / L2) and adjacent in dispatch order.
    const long long id = blockIdx.x;
    const int xcd = (int)(id & 7);
    const long long y = id >> 3;
    // each XCD walks a contiguous run of clusters (see mtmfft2_kernel.h): the rows of a spectrum meet in one L2
    const long long nclt = (long long)a.nseg * a.ncl, chunk = (nclt + 7) >> 3;
    const long long cidx = (long long)xcd * chunk + y / a.S;
    const int q = (int)(y % a.S);
    if (cidx >= nclt) return;
    const int b = (int)(cidx / a.ncl);
    const int pg = (int)(cidx % a.ncl) * a.S + q;
    if (pg >= a.npg) return;

    const int c0 = 4 * (pg * G + h);
    bool has[4];
    unsigned col[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        has[i] = c0 + i < a.nchan;
        col[i] = has[i] ? (unsigned)(a.chan_idx ? a.chan_idx[c0 + i] : c0 + i) : 0u;
    }
    const bool full = has[3];
    const long long start = a.seg_start[b];
    const long long rl = a.seg_lo[b] - start, rh = a.seg_hi[b] - start;
    const int rlo = (int)(rl < 0 ? 0 : (rl > a.nsig ? a.nsig : rl));
    const int rhi = (int)(rh < 0 ? 0 : (rh > a.nsig ? a.nsig : rh));
    const unsigned rowb = (unsigned)a.ld * 4u;        // bytes per row
    const float* seg = a.data + start * a.ld;         // wave-uniform; only rows in [rlo, rhi) are dereferenced

    // ---- load the segment once: x[e] = sample n = j + T*e; r = (c0, c1), i = (c2, c3)
    C2 x[16];
    if (rhi > rlo) {
        const bool vec4 = (a.chan_idx == nullptr) && full && ((a.ld & 3) == 0) &&
                          ((reinterpret_cast<size_t>(a.data) & 15) == 0);
        if (vec4) {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = j0 + T * e;
                const int nc = min(max(n, rlo), rhi - 1);
                const float4 t = ldg<float4>(seg, (unsigned)nc * rowb + col[0] * 4u);
                const bool ok = (n == nc);
                x[e].r = v2f{ok ? t.x : 0.f, ok ? t.y : 0.f};
                x[e].i = v2f{ok ? t.z : 0.f, ok ? t.w : 0.f};
            }
        } else {
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = j0 + T * e;
                const int nc = min(max(n, rlo), rhi - 1);
                float u[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float t = ldg<float>(seg, (unsigned)nc * rowb + col[i] * 4u);
                    u[i] = (n == nc && has[i]) ? t : 0.f;
                }
                x[e].r = v2f{u[0], u[1]};
                x[e].i = v2f{u[2], u[3]};
            }
        }
    } else {
#pragma unroll
        for (int e = 0; e < 16; ++e) x[e].r = x[e].i = splat(0.f);
    }

    // ---- polynomial removal over the nsig samples (float64 sums, branch-free; constant: the reference-order means)
    if (a.detrend == 0 && a.means) {
        const float* mp = a.means + (size_t)b * a.nchan + c0;
        float f[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) f[i] = has[i] ? mp[i] : 0.f;
        const v2f mr = v2f{f[0], f[1]}, mi = v2f{f[2], f[3]};
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const bool in = j0 + T * e < a.nsig;
            x[e].r -= in ? mr : splat(0.f);
            x[e].i -= in ? mi : splat(0.f);
        }
    } else if (a.detrend >= 0) {
        const float mid = 0.5f * (float)(a.nsig - 1);
        double s[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const int n = j0 + T * e;
            const float m = (n < a.nsig) ? 1.f : 0.f;
            const float u[4] = {m * x[e].r[0], m * x[e].r[1], m * x[e].i[0], m * x[e].i[1]};
#pragma unroll
            for (int i = 0; i < 4; ++i) s[i] += (double)u[i];
            if (a.detrend == 1) {
                const double dn = (double)(m * ((float)n - mid));   // exact: half-integers < 2^23
                s[4] += dn * x[e].r[0];
                s[5] += dn * x[e].r[1];
                s[6] += dn * x[e].i[0];
                s[7] += dn * x[e].i[1];
            }
        }
        block_sum<C::NTHREADS, G, 8>(s, reinterpret_cast<double*>(lds), tid, h);
        const double inv = 1.0 / a.nsig;
        if (a.detrend == 1 && a.nsig > 1) {
            const double den = 12.0 / ((double)a.nsig * ((double)a.nsig * a.nsig - 1.0));
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int n = j0 + T * e;
                const double dn = (double)((float)n - mid);
                const bool in = n < a.nsig;
                float t[4];
#pragma unroll
                for (int i = 0; i < 4; ++i) t[i] = in ? (float)(s[i] * inv + s[4 + i] * den * dn) : 0.f;
                x[e].r -= v2f{t[0], t[1]};
                x[e].i -= v2f{t[2], t[3]};
            }
        } else {
            const v2f mr = v2f{(float)(s[0] * inv), (float)(s[1] * inv)};
            const v2f mi = v2f{(float)(s[2] * inv), (float)(s[3] * inv)};
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const bool in = j0 + T * e < a.nsig;
                x[e].r -= in ? mr : splat(0.f);
                x[e].i -= in ? mi : splat(0.f);
            }
        }
    }

    // accumulators for the taper mean: real outputs ma.r = sum conv(X(c0,c1)), ma.i = sum conv(X(c2,c3));
    // complex: ma = X(c0,c1), mb = X(c2,c3)
    constexpr int NBIN = 5;
    C2 ma[MEAN ? NBIN : 1], mb[(MEAN && CPLX) ? NBIN : 1];
    if (MEAN) {
#pragma unroll
        for (int e = 0; e < NBIN; ++e) {
            ma[e].r = ma[e].i = splat(0.f);
            if (CPLX) mb[e].r = mb[e].i = splat(0.f);
        }
    }
    const int kout = MEAN ? 1 : a.ntaper;
    const float hs = 0.5f * a.scale;
    const unsigned nsig_m1 = (unsigned)(a.nsig - 1), nfft_m1 = (unsigned)(a.nfft - 1);
    const int nf = a.nfft / 2 + 1;
    constexpr unsigned OSZ = CPLX ? 8u : 4u;   // bytes per output element

    for (int k = 0; k < a.ntaper; ++k) {
        const int j = opaque(j0);
        C2 v[16];
        const float* w = a.tapers + (size_t)k * a.nsig;   // wave-uniform
#pragma unroll
        for (int e = 0; e < 16; ++e) {
            const unsigned n = (unsigned)(j + T * e);
            const float wl = ldg<float>(w, min(n, nsig_m1) * 4u);
            const float wn = (n <= nsig_m1) ? wl : 0.f;
            v[e].r = x[e].r * wn;
            v[e].i = x[e].i * wn;
        }
        if (a.demean_taper) {
            __syncthreads();          // block_sum writes its scratch into the planes other waves may still be reading
            double s[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                s[0] += v[e].r[0];
                s[1] += v[e].r[1];
                s[2] += v[e].i[0];
                s[3] += v[e].i[1];
            }
            block_sum<C::NTHREADS, G, 4>(s, reinterpret_cast<double*>(lds), tid, h);
            const v2f mr = v2f{(float)(s[0] / a.nsig), (float)(s[1] / a.nsig)};
            const v2f mi = v2f{(float)(s[2] / a.nsig), (float)(s[3] / a.nsig)};
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const bool in = j + T * e < a.nsig;
                v[e].r -= in ? mr : splat(0.f);
                v[e].i -= in ? mi : splat(0.f);
            }
        }
        // z[n] c[n]  (samples beyond nsig <= nfft are zero already)
#pragma unroll
        for (int e = 0; e < 16; ++e)
            v[e] = cmul_s(v[e], ldg<float2>(a.chirp, min((unsigned)(j + T * e), nfft_m1) * 8u));
        fft2_forward<LOG2N, G>(v, lds, j, h, a.tw);
#pragma unroll
        for (int e = 0; e < 16; ++e) v[e] = cmul_s(v[e], ldg<float2>(a.bhat, (unsigned)(j + T * e) * 8u));
        fft2_inverse<LOG2N, G>(v, lds, j, h, a.tw);
        // Z[k] = conv[k] c[k] for k < nfft, parked in LDS in natural order for the channel separation
        __syncthreads();              // the inverse FFT's last reads of the planes are done everywhere
        {
            const int wb = C::rbase(j, h);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const unsigned kk = (unsigned)(j + T * e);
                if (kk <= nfft_m1) {
                    const C2 z = cmul_s(v[e], ldg<float2>(a.chirp, kk * 8u));
                    lre[wb + e * C::ESTRIDE] = z.r;
                    lim[wb + e * C::ESTRIDE] = z.i;
                }
            }
        }
        __syncthreads();
        char* const slab = reinterpret_cast<char*>(a.out) +
                           ((size_t)b * kout + (MEAN ? 0 : k)) * (size_t)a.nfsel * (size_t)a.nchan * OSZ;
#pragma unroll
        for (int e = 0; e < NBIN; ++e) {
            const int f = j + T * e;
            if (f >= nf) break;
            const int p = (f == 0) ? 0 : a.nfft - f;            // partner bin (nfft - f) mod nfft
            C2 z, zp;
            z.r = lre[C::idx(f, h)];
            z.i = lim[C::idx(f, h)];
            zp.r = lre[C::idx(p, h)];
            zp.i = lim[C::idx(p, h)];
            C2 xa, xb;   // xa = X(c0, c1), xb = X(c2, c3)
            xa.r = (z.r + zp.r) * hs;
            xa.i = (z.i - zp.i) * hs;
            xb.r = (z.i + zp.i) * hs;
            xb.i = (zp.r - z.r) * hs;
            if (MEAN) {
                if (CPLX) {
                    ma[e] = cadd(ma[e], xa);
                    mb[e] = cadd(mb[e], xb);
                } else {
                    ma[e].r += v2f{convert_real<OUTK>(make_float2(xa.r[0], xa.i[0]), a.out_kind),
                                   convert_real<OUTK>(make_float2(xa.r[1], xa.i[1]), a.out_kind)};
                    ma[e].i += v2f{convert_real<OUTK>(make_float2(xb.r[0], xb.i[0]), a.out_kind),
                                   convert_real<OUTK>(make_float2(xb.r[1], xb.i[1]), a.out_kind)};
                }
                continue;
            }
            const int fi = a.fpos ? ldg<int>(a.fpos, (unsigned)f * 4u) : f;
            if (fi < 0) continue;
            const float2 X[4] = {make_float2(xa.r[0], xa.i[0]), make_float2(xa.r[1], xa.i[1]),
                                 make_float2(xb.r[0], xb.i[0]), make_float2(xb.r[1], xb.i[1])};
            const unsigned o = ((unsigned)fi * (unsigned)a.nchan + (unsigned)c0) * OSZ;
            if (!CPLX && full && (a.nchan & 3) == 0 && (reinterpret_cast<size_t>(a.out) & 15) == 0) {
                stg<float4>(slab, o, make_float4(convert_real<OUTK>(X[0], a.out_kind), convert_real<OUTK>(X[1], a.out_kind),
                                                 convert_real<OUTK>(X[2], a.out_kind), convert_real<OUTK>(X[3], a.out_kind)));
            } else {
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    if (!has[i]) continue;
                    if (CPLX) stg<float2>(slab, o + i * OSZ, X[i]);
                    else stg<float>(slab, o + i * OSZ, convert_real<OUTK>(X[i], a.out_kind));
                }
            }
        }
        // no barrier here: the next taper's first LDS write sits behind one (fft2_forward / block_sum)
    }

    if (MEAN) {
        char* const slab = reinterpret_cast<char*>(a.out) + (size_t)b * (size_t)a.nfsel * (size_t)a.nchan * OSZ;
        const float nt = (float)a.ntaper;
#pragma unroll
        for (int e = 0; e < NBIN; ++e) {
            const int f = j0 + T * e;
            if (f >= nf) break;
            const int fi = a.fpos ? ldg<int>(a.fpos, (unsigned)f * 4u) : f;
            if (fi < 0) continue;
            const unsigned o = ((unsigned)fi * (unsigned)a.nchan + (unsigned)c0) * OSZ;
            if (CPLX) {
                const float2 X[4] = {make_float2(ma[e].r[0] / nt, ma[e].i[0] / nt), make_float2(ma[e].r[1] / nt, ma[e].i[1] / nt),
                                     make_float2(mb[e].r[0] / nt, mb[e].i[0] / nt), make_float2(mb[e].r[1] / nt, mb[e].i[1] / nt)};
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (has[i]) stg<float2>(slab, o + i * OSZ, X[i]);
            } else {
                const float X[4] = {ma[e].r[0] / nt, ma[e].r[1] / nt, ma[e].i[0] / nt, ma[e].i[1] / nt};
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (has[i]) stg<float>(slab, o + i * OSZ, X[i]);
            }
        }
    }
}

}  // namespace spyfft
