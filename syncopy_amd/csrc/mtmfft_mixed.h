// K1m: the packed mtmfft / STFT kernel for 5-smooth lengths (N = 2^a 3^b 5^c, 16 <= N <= 10000): the trial lengths
// real recordings have (1 kHz x 0.5 / 1 / 2 / 3 / 5 s ...; BASELINE config 1 is N = 2000).  Same semantics, argument
// block and output layouts as mtmfft_quad_kernel (specest/mtmfft.py:16-129, specest/compRoutines.py:169-189).
//
// Round 1 served these lengths with Bluestein on the power-of-two engine (two length-M >= 2N transforms per
// spectrum, 4-7x the time of N = 2^k) or, above 4096, with the unpacked one-pair-per-workgroup Stockham kernel (17x).
// This engine keeps what makes the power-of-two kernel fast - four real channels per thread in packed register
// pairs (C2), G channel quads per workgroup sharing every twiddle, the segment read from HBM once and held in
// registers across the tapers, 16-byte coalesced row loads and stores - and replaces the radix-16 network by
// Stockham passes of radix 2/3/4/5/8/10 chosen on the host (fewest passes; decimal lengths are 10 x 10 x ...):
//   natural order in LDS -> pass_0 ... pass_{P-1} (in place: all reads, barrier, all writes, barrier) -> natural
//   order spectrum in LDS -> separation of the packed real channels, conversion, store / taper mean.
// TH threads serve one quad; thread j owns butterflies b = j + TH m of every pass (inputs b + (N/R) r) and samples /
// bins j + TH m outside the passes, at most MIX_V values at a time (the host picks TH accordingly).  Twiddles
// come from the length-N table: w^1..w^3 and w^4, w^8, w^12 are loaded, the others are one product each.
#pragma once
#include "fft2_device.h"
#include "mtmfft_kernel.h"
#include "mtmfft_mixed_plan.h"

namespace spyfft {

// ---- compile-time roots of unity for the composite butterflies
constexpr double mx_pi = 3.14159265358979323846264338327950288;
constexpr double mx_sin_q(double x) {      // |x| <= pi/4
    const double x2 = x * x;
    return x * (1 - x2 / 6 * (1 - x2 / 20 * (1 - x2 / 42 * (1 - x2 / 72 * (1 - x2 / 110 * (1 - x2 / 156 * (1 - x2 / 210)))))));
}
constexpr double mx_cos_q(double x) {
    const double x2 = x * x;
    return 1 - x2 / 2 * (1 - x2 / 12 * (1 - x2 / 30 * (1 - x2 / 56 * (1 - x2 / 90 * (1 - x2 / 132 * (1 - x2 / 182 * (1 - x2 / 240)))))));
}
constexpr double mx_cos_turn(int m, int R) {            // cos(2 pi m / R)
    m = ((m % R) + R) % R;
    double a = 2.0 * mx_pi * m / R;
    bool neg = false;
    if (a > mx_pi) a = 2.0 * mx_pi - a;
    if (a > mx_pi / 2) { a = mx_pi - a; neg = true; }
    const double c = (a > mx_pi / 4) ? mx_sin_q(mx_pi / 2 - a) : mx_cos_q(a);
    return neg ? -c : c;
}
constexpr double mx_sin_turn(int m, int R) {            // sin(2 pi m / R)
    m = ((m % R) + R) % R;
    double a = 2.0 * mx_pi * m / R;
    bool neg = false;
    if (a > mx_pi) { a = 2.0 * mx_pi - a; neg = true; }
    if (a > mx_pi / 2) a = mx_pi - a;
    const double s = (a > mx_pi / 4) ? mx_cos_q(mx_pi / 2 - a) : mx_sin_q(a);
    return neg ? -s : s;
}

__device__ __forceinline__ void dft3p(C2 (&t)[3]) {
    constexpr float s3 = (float)mx_sin_turn(1, 3);
    const C2 s = cadd(t[1], t[2]), d = csub(t[1], t[2]);
    const C2 m = C2{t[0].r - s.r * 0.5f, t[0].i - s.i * 0.5f};
    const C2 e = C2{d.i * s3, d.r * -s3};                 // -i sin(2 pi/3) d
    t[0] = cadd(t[0], s);
    t[1] = cadd(m, e);
    t[2] = csub(m, e);
}
__device__ __forceinline__ void dft5p(C2 (&t)[5]) {
    constexpr float c1 = (float)mx_cos_turn(1, 5), c2 = (float)mx_cos_turn(2, 5);
    constexpr float s1 = (float)mx_sin_turn(1, 5), s2 = (float)mx_sin_turn(2, 5);
    const C2 a1 = cadd(t[1], t[4]), a2 = cadd(t[2], t[3]), b1 = csub(t[1], t[4]), b2 = csub(t[2], t[3]);
    const C2 m1 = C2{t[0].r + a1.r * c1 + a2.r * c2, t[0].i + a1.i * c1 + a2.i * c2};
    const C2 m2 = C2{t[0].r + a1.r * c2 + a2.r * c1, t[0].i + a1.i * c2 + a2.i * c1};
    const C2 n1 = C2{b1.r * s1 + b2.r * s2, b1.i * s1 + b2.i * s2};
    const C2 n2 = C2{b1.r * s2 - b2.r * s1, b1.i * s2 - b2.i * s1};
    t[0] = cadd(t[0], cadd(a1, a2));
    t[1] = C2{m1.r + n1.i, m1.i - n1.r};                  // m1 - i n1
    t[4] = C2{m1.r - n1.i, m1.i + n1.r};
    t[2] = C2{m2.r + n2.i, m2.i - n2.r};
    t[3] = C2{m2.r - n2.i, m2.i + n2.r};
}

template <int R>
__device__ __forceinline__ void dftR(C2 (&t)[R]);

// R = P Q: n = Q n1 + n2, k = k1 + P k2 (Cooley-Tukey inside the registers of one thread)
template <int P, int Q>
__device__ __forceinline__ void dft_pq(C2 (&t)[P * Q]) {
    constexpr int R = P * Q;
    C2 y[Q][P];
#pragma unroll
    for (int n2 = 0; n2 < Q; ++n2) {
        C2 u[P];
#pragma unroll
        for (int n1 = 0; n1 < P; ++n1) u[n1] = t[Q * n1 + n2];
        dftR<P>(u);
#pragma unroll
        for (int k1 = 0; k1 < P; ++k1) {
            const float c = (float)mx_cos_turn(n2 * k1, R), s = (float)mx_sin_turn(n2 * k1, R);
            y[n2][k1] = (n2 * k1 == 0) ? u[k1] : cmul_s(u[k1], make_float2(c, -s));
        }
    }
#pragma unroll
    for (int k1 = 0; k1 < P; ++k1) {
        C2 u[Q];
#pragma unroll
        for (int n2 = 0; n2 < Q; ++n2) u[n2] = y[n2][k1];
        dftR<Q>(u);
#pragma unroll
        for (int k2 = 0; k2 < Q; ++k2) t[k1 + P * k2] = u[k2];
    }
}

template <int R>
__device__ __forceinline__ void dftR(C2 (&t)[R]) {
    if constexpr (R == 2) dft2p(t);
    else if constexpr (R == 3) dft3p(t);
    else if constexpr (R == 4) dft4p(t);
    else if constexpr (R == 5) dft5p(t);
    else if constexpr (R == 8) dft8p(t);
    else dft_pq<2, 5>(t);
}

// LDS element (i, h): one float4 {re of (c0, c1), im of (c2, c3)} at (i << lg) + h: a single 16-byte access per value
__device__ __forceinline__ C2 lds_get(const float4* z, int i) {
    const float4 t = z[i];
    return C2{v2f{t.x, t.y}, v2f{t.z, t.w}};
}
__device__ __forceinline__ void lds_put(float4* z, int i, C2 v) { z[i] = make_float4(v.r[0], v.r[1], v.i[0], v.i[1]); }

// One Stockham pass of radix R over the work buffer (in place).  Ns = product of the earlier radices.
// (A variant that requested the next pass's twiddles ahead of this pass's barrier, with the butterflies split into
// read / write halves around it, measured 15-20 % SLOWER: the flat register arrays it needs cost more than the L2
// latency it hides.)
template <int R>
__device__ __forceinline__ void mix_pass(float4* z, const MixPlan& g, int j, int h, bool active, int Ns, unsigned magic,
                                         const float2* __restrict__ tw) {
    constexpr int MB = MIX_V / R;                        // butterflies per thread at most (host: ceil(nb / th) <= MB)
    const int nb = g.n / R;
    const int lg = g.lg;
    C2 t[MB][R];
    int base[MB];
#pragma unroll
    for (int m = 0; m < MB; ++m) {
        const int b = j + g.th * m;
        base[m] = -1;
        if (active && b < nb) {
            int i = (b << lg) + h;
#pragma unroll
            for (int r = 0; r < R; ++r) {
                t[m][r] = lds_get(z, i);
                i += nb << lg;
            }
            int k = 0, q = b;
            if (Ns > 1) {
                q = (int)__umulhi((unsigned)b, magic);
                k = b - q * Ns;
                const unsigned kb = (unsigned)(k * (nb / Ns)) * 8u;            // byte offset of tw[k N / (Ns R)]
                float2 wb[4], wa[4];
                wb[0] = wa[0] = make_float2(1.f, 0.f);
#pragma unroll
                for (int s = 1; s < 4; ++s) {
                    if (s < R) wb[s] = ldg<float2>(tw, kb * (unsigned)s);
                    if (4 * s < R) wa[s] = ldg<float2>(tw, kb * (unsigned)(4 * s));
                }
#pragma unroll
                for (int r = 1; r < R; ++r) {
                    const int hi = r >> 2, lo = r & 3;
                    const float2 w = (hi == 0) ? wb[lo] : (lo == 0 ? wa[hi] : cmul(wa[hi], wb[lo]));
                    t[m][r] = cmul_s(t[m][r], w);
                }
            }
            dftR<R>(t[m]);
            base[m] = ((q * Ns * R + k) << lg) + h;
        }
    }
    __syncthreads();
    const int ws = Ns << lg;
#pragma unroll
    for (int m = 0; m < MB; ++m) {
        if (base[m] >= 0) {
            int i = base[m];
#pragma unroll
            for (int r = 0; r < R; ++r) {
                lds_put(z, i, t[m][r]);
                i += ws;
            }
        }
    }
    __syncthreads();
}

// Sum NS doubles over the threads that share `h` (G = 2^lg quads per workgroup); broadcast to all of them.
template <int NS>
__device__ __forceinline__ void mix_block_sum(double (&s)[NS], double* scratch, int tid, int h, int lg, int nwaves) {
    const int G = 1 << lg;
#pragma unroll
    for (int i = 0; i < NS; ++i) {
#pragma unroll
        for (int o = 0; o < 6; ++o) {
            const int off = 32 >> o;
            const double other = __shfl_xor(s[i], off);
            s[i] += (off >= G) ? other : 0.0;
        }
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
        for (int ww = 0; ww < nwaves; ++ww) tot += scratch[(ww * G + h) * NS + i];
        s[i] = tot;
    }
    __syncthreads();
}

// OUTK: 0 = power, 1 = any other real conversion, 2 = complex; MEAN: average over tapers; LB: largest workgroup.
// LB = 512 (N <= 5120) leaves the compiler 256 registers - with 128 it spills ~60 long-lived values to scratch, and
// the PMC counters showed those spills as 8x the algorithmic HBM writes; LB = 1024 serves 5120 < N <= 10000.
template <int OUTK, bool MEAN, int LB>
__global__ void __launch_bounds__(LB) mtmfft_mixed_kernel(MtmArgs a, MixPlan g) {
    constexpr bool CPLX = (OUTK == 2);
    constexpr int V = MIX_V, MF = MIX_V / 2 + 1;
    SPY_DYN_SMEM(float4, lds);
    const int N = g.n, TH = g.th, lg = g.lg, G = 1 << lg;
    float4* const z = lds;                               // work buffer: (N + 1) << lg elements
    float4* const xs = lds + ((N + 1) << lg);            // the detrended segment, if it is staged (g.stage)

    const int tid = threadIdx.x;
    const int h = tid & (G - 1), j = tid >> lg;
    const bool active = j < TH;                          // the workgroup is padded to whole waves
    const int nwaves = (int)(blockDim.x + 63) >> 6;

    // XCD-aware block -> (segment, quad group), as mtmfft_quad_kernel
    const long long id = blockIdx.x;
    const int xcd = (int)(id & 7);
    const long long y = id >> 3;
    const long long nclt = (long long)a.nseg * a.ncl, chunk = (nclt + 7) >> 3;
    const long long cidx = (long long)xcd * chunk + y / a.S;
    const int qq = (int)(y % a.S);
    if (cidx >= nclt) return;
    const int b = (int)(cidx / a.ncl);
    const int pg = (int)(cidx % a.ncl) * a.S + qq;
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
    const unsigned rowb = (unsigned)a.ld * 4u;
    const float* seg = a.data + start * a.ld;
    const bool vec4 = (a.chan_idx == nullptr) && full && ((a.ld & 3) == 0) && ((reinterpret_cast<size_t>(a.data) & 15) == 0);
    const bool any = rhi > rlo;

    // sample n of the four channels, zero outside the valid rows: r = (c0, c1), i = (c2, c3)
    auto raw = [&](int n) -> C2 {
        C2 o;
        o.r = o.i = splat(0.f);
        if (!any || !active) return o;
        const int nc = min(max(n, rlo), rhi - 1);
        const bool ok = (n == nc);
        if (vec4) {
            const float4 t = ldg<float4>(seg, (unsigned)nc * rowb + col[0] * 4u);
            o.r = v2f{ok ? t.x : 0.f, ok ? t.y : 0.f};
            o.i = v2f{ok ? t.z : 0.f, ok ? t.w : 0.f};
        } else {
            float u[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float t = ldg<float>(seg, (unsigned)nc * rowb + col[i] * 4u);
                u[i] = (ok && has[i]) ? t : 0.f;
            }
            o.r = v2f{u[0], u[1]};
            o.i = v2f{u[2], u[3]};
        }
        return o;
    };

    // ---- polynomial removal over the nsig samples: per channel  x[n] - (c[i] + d[i] (n - mid))
    const float mid = 0.5f * (float)(a.nsig - 1);
    double dc[4] = {0.0, 0.0, 0.0, 0.0}, dd[4] = {0.0, 0.0, 0.0, 0.0};
    float fc[4] = {0.f, 0.f, 0.f, 0.f};                  // constant detrending: the float32 value subtracted
    if (a.detrend == 0 && a.means) {
        const float* mp = a.means + (size_t)b * a.nchan + c0;
#pragma unroll
        for (int i = 0; i < 4; ++i) fc[i] = has[i] ? mp[i] : 0.f;
    } else if (a.detrend >= 0) {
        double s[8] = {0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0, 0.0};
#pragma unroll 2
        for (int m = 0; m < V; ++m) {
            const int n = j + TH * m;
            const C2 xv = raw(n);
            const float w = (active && n < a.nsig) ? 1.f : 0.f;
            const float u[4] = {w * xv.r[0], w * xv.r[1], w * xv.i[0], w * xv.i[1]};
#pragma unroll
            for (int i = 0; i < 4; ++i) s[i] += (double)u[i];
            if (a.detrend == 1) {
                const double dn = (double)(w * ((float)n - mid));
#pragma unroll
                for (int i = 0; i < 4; ++i) s[4 + i] += dn * u[i];
            }
        }
        mix_block_sum<8>(s, reinterpret_cast<double*>(lds), tid, h, lg, nwaves);
        const double inv = 1.0 / a.nsig;
        const double den = (a.detrend == 1 && a.nsig > 1) ? 12.0 / ((double)a.nsig * ((double)a.nsig * a.nsig - 1.0)) : 0.0;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            dc[i] = s[i] * inv;
            dd[i] = s[4 + i] * den;
            fc[i] = (float)dc[i];
        }
    }
    const bool linear = a.detrend == 1 && a.nsig > 1 && !(a.detrend == 0 && a.means);
    auto sample = [&](int n) -> C2 {                     // detrended sample (zero beyond nsig)
        C2 o = raw(n);
        if (a.detrend < 0 || !active || n >= a.nsig) return o;
        if (linear) {
            const double dn = (double)((float)n - mid);
            float t[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) t[i] = (float)(dc[i] + dd[i] * dn);
            o.r -= v2f{t[0], t[1]};
            o.i -= v2f{t[2], t[3]};
        } else {
            o.r -= v2f{fc[0], fc[1]};
            o.i -= v2f{fc[2], fc[3]};
        }
        return o;
    };
    if (g.stage) {
        // (block_sum ended with a barrier; nobody reads xs before the barrier at the top of the taper loop)
#pragma unroll 2
        for (int m = 0; m < V; ++m) {
            const int n = j + TH * m;
            if (active && n < N) lds_put(xs, (n << lg) + h, sample(n));
        }
    }

    // taper-mean accumulators for the bins f = j + TH m (real outputs: ma.r / ma.i = channels (c0, c1) / (c2, c3))
    C2 ma[MEAN ? MF : 1], mb[(MEAN && CPLX) ? MF : 1];
    if (MEAN) {
#pragma unroll
        for (int m = 0; m < MF; ++m) {
            ma[m].r = ma[m].i = splat(0.f);
            if (CPLX) mb[m].r = mb[m].i = splat(0.f);
        }
    }
    const int kout = MEAN ? 1 : a.ntaper;
    const float hs = 0.5f * a.scale;
    const unsigned nsig_m1 = (unsigned)(a.nsig - 1);
    constexpr unsigned OSZ = CPLX ? 8u : 4u;
    const bool fast = full && (a.fpos == nullptr) && ((reinterpret_cast<size_t>(a.out) & 15) == 0) &&
                      ((a.nchan & (CPLX ? 1 : 3)) == 0);
    const int nf = N / 2 + 1;

    for (int k = 0; k < a.ntaper; ++k) {
        const float* w = a.tapers + (size_t)k * a.nsig;       // wave-uniform
        __syncthreads();              // the previous taper's epilogue reads (block_sum's scratch, the staging writes) are done
        // tapered sample n of the quad (rolled loops: nothing here is worth 40 registers)
        auto tapered = [&](int n) -> C2 {
            const float wl = ldg<float>(w, min((unsigned)n, nsig_m1) * 4u);
            const float wn = (active && (unsigned)n <= nsig_m1) ? wl : 0.f;
            C2 xv;
            if (g.stage) {
                xv.r = xv.i = splat(0.f);
                if (active && n < N) xv = lds_get(xs, (n << lg) + h);
            } else {
                xv = sample(n);
            }
            xv.r *= wn;
            xv.i *= wn;
            return xv;
        };
        v2f mr = splat(0.f), mi = splat(0.f);
        if (a.demean_taper) {
            double s[4] = {0.0, 0.0, 0.0, 0.0};
#pragma unroll 2
            for (int m = 0; m < V; ++m) {
                const C2 t = tapered(j + TH * m);
                s[0] += t.r[0];
                s[1] += t.r[1];
                s[2] += t.i[0];
                s[3] += t.i[1];
            }
            mix_block_sum<4>(s, reinterpret_cast<double*>(z), tid, h, lg, nwaves);
            mr = v2f{(float)(s[0] / a.nsig), (float)(s[1] / a.nsig)};
            mi = v2f{(float)(s[2] / a.nsig), (float)(s[3] / a.nsig)};
        }
        // ---- tapered samples in natural order into the work buffer
#pragma unroll 2
        for (int m = 0; m < V; ++m) {
            const int n = j + TH * m;
            if (active && n < N) {
                C2 t = tapered(n);
                const bool in = n < a.nsig;
                t.r -= in ? mr : splat(0.f);
                t.i -= in ? mi : splat(0.f);
                lds_put(z, (n << lg) + h, t);
            }
        }
        __syncthreads();
        // ---- Stockham passes
        int Ns = 1;
        for (int p = 0; p < g.npass; ++p) {
            const int R = g.radix[p];
            const unsigned mg = g.magic[p];
            switch (R) {
                case 2: mix_pass<2>(z, g, j, h, active, Ns, mg, a.tw); break;
                case 3: mix_pass<3>(z, g, j, h, active, Ns, mg, a.tw); break;
                case 4: mix_pass<4>(z, g, j, h, active, Ns, mg, a.tw); break;
                case 5: mix_pass<5>(z, g, j, h, active, Ns, mg, a.tw); break;
                case 8: mix_pass<8>(z, g, j, h, active, Ns, mg, a.tw); break;
                default: mix_pass<10>(z, g, j, h, active, Ns, mg, a.tw); break;
            }
            Ns *= R;
        }
        // ---- separate the real channels, convert, store / accumulate: bins f = j + TH m
        char* const slab = reinterpret_cast<char*>(a.out) +
                           ((size_t)b * kout + (MEAN ? 0 : k)) * (size_t)a.nfsel * (size_t)a.nchan * OSZ;
#pragma unroll
        for (int m = 0; m < MF; ++m) {
            const int f = j + TH * m;
            if (!active || f >= nf) continue;
            const int fp = (f == 0) ? 0 : N - f;
            const C2 zf = lds_get(z, (f << lg) + h), zp = lds_get(z, (fp << lg) + h);
            C2 xa, xb;   // xa = X(c0, c1), xb = X(c2, c3)
            xa.r = (zf.r + zp.r) * hs;
            xa.i = (zf.i - zp.i) * hs;
            xb.r = (zf.i + zp.i) * hs;
            xb.i = (zp.r - zf.r) * hs;
            if (MEAN) {
                if (CPLX) {
                    ma[m] = cadd(ma[m], xa);
                    mb[m] = cadd(mb[m], xb);
                } else if (OUTK == 0) {
                    ma[m].r += xa.r * xa.r + xa.i * xa.i;
                    ma[m].i += xb.r * xb.r + xb.i * xb.i;
                } else {
                    ma[m].r += v2f{convert_real_slow(make_float2(xa.r[0], xa.i[0]), a.out_kind),
                                   convert_real_slow(make_float2(xa.r[1], xa.i[1]), a.out_kind)};
                    ma[m].i += v2f{convert_real_slow(make_float2(xb.r[0], xb.i[0]), a.out_kind),
                                   convert_real_slow(make_float2(xb.r[1], xb.i[1]), a.out_kind)};
                }
                continue;
            }
            if (fast) {
                const unsigned o = ((unsigned)f * (unsigned)a.nchan + (unsigned)c0) * OSZ;
                if (CPLX) {
                    stg<float4>(slab, o, make_float4(xa.r[0], xa.i[0], xa.r[1], xa.i[1]));
                    stg<float4>(slab, o + 16u, make_float4(xb.r[0], xb.i[0], xb.r[1], xb.i[1]));
                } else if (OUTK == 0) {
                    const v2f pa = xa.r * xa.r + xa.i * xa.i, pb2 = xb.r * xb.r + xb.i * xb.i;
                    stg<float4>(slab, o, make_float4(pa[0], pa[1], pb2[0], pb2[1]));
                } else {
                    stg<float4>(slab, o, make_float4(convert_real_slow(make_float2(xa.r[0], xa.i[0]), a.out_kind),
                                                     convert_real_slow(make_float2(xa.r[1], xa.i[1]), a.out_kind),
                                                     convert_real_slow(make_float2(xb.r[0], xb.i[0]), a.out_kind),
                                                     convert_real_slow(make_float2(xb.r[1], xb.i[1]), a.out_kind)));
                }
            } else {
                const int fi = a.fpos ? ldg<int>(a.fpos, (unsigned)f * 4u) : f;
                if (fi >= 0) {
                    const float2 X[4] = {make_float2(xa.r[0], xa.i[0]), make_float2(xa.r[1], xa.i[1]),
                                         make_float2(xb.r[0], xb.i[0]), make_float2(xb.r[1], xb.i[1])};
                    const unsigned o = ((unsigned)fi * (unsigned)a.nchan + (unsigned)c0) * OSZ;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        if (!has[i]) continue;
                        if (CPLX) stg<float2>(slab, o + i * OSZ, X[i]);
                        else stg<float>(slab, o + i * OSZ, convert_real<OUTK>(X[i], a.out_kind));
                    }
                }
            }
        }
        // the next taper's first LDS write sits behind a barrier (top of the loop)
    }

    if (MEAN) {
        char* const slab = reinterpret_cast<char*>(a.out) + (size_t)b * (size_t)a.nfsel * (size_t)a.nchan * OSZ;
        const float nt = (float)a.ntaper;
#pragma unroll
        for (int m = 0; m < MF; ++m) {
            const int f = j + TH * m;
            if (!active || f >= nf) continue;
            const int fi = a.fpos ? ldg<int>(a.fpos, (unsigned)f * 4u) : f;
            if (fi < 0) continue;
            const unsigned o = ((unsigned)fi * (unsigned)a.nchan + (unsigned)c0) * OSZ;
            if (CPLX) {
                const float2 X[4] = {make_float2(ma[m].r[0] / nt, ma[m].i[0] / nt), make_float2(ma[m].r[1] / nt, ma[m].i[1] / nt),
                                     make_float2(mb[m].r[0] / nt, mb[m].i[0] / nt), make_float2(mb[m].r[1] / nt, mb[m].i[1] / nt)};
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (has[i]) stg<float2>(slab, o + i * OSZ, X[i]);
            } else if (fast) {
                stg<float4>(slab, o, make_float4(ma[m].r[0] / nt, ma[m].r[1] / nt, ma[m].i[0] / nt, ma[m].i[1] / nt));
            } else {
                const float X[4] = {ma[m].r[0] / nt, ma[m].r[1] / nt, ma[m].i[0] / nt, ma[m].i[1] / nt};
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (has[i]) stg<float>(slab, o + i * OSZ, X[i]);
            }
        }
    }
}

}  // namespace spyfft
