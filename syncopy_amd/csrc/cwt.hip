// Host side of K3: Morlet kernel spectra (fp64 on the host, once per plan) and launch of the
// overlap-save CWT kernel (spyhip_cwt_plan_create / spyhip_cwt_exec).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <string>

#include "spy_common.h"
#include "host_fft.h"
#include "cwt_kernel.h"
#include "cwt64_kernel.h"

using spyfft::CwtArgs;

// scales whose (trimmed) kernel support needs the same block length share one launch: short kernels run on short
// blocks (less FFT work per output sample, two workgroups per CU) instead of on the block the longest one needs
struct CwtGroup {
    int log2n = 0, G = 1, V = 0, halo = 0, nblocks = 0, nscales = 0;
    int long_idx = -1, piece = 0;  // long_idx >= 0: piece `piece` of the long_idx-th scale whose kernel exceeds a block
    bool direct = false;           // 1024- and 2048-point blocks: cwt2d_kernel writes the output layout itself (no staging)
    spy::DevBuf<float2> tw, hspec;
    spy::DevBuf<int> cshift, sidx;
    spy::DevBuf<int> sidx_stage;   // with direct groups: scale s of this launch -> row of the (compact) staging buffer
    std::vector<int> scale_ids;    // host copy: the plan's scale index of every scale of this launch
};

struct spyhip_cwt_plan {
    spyhip_ctx* ctx = nullptr;
    int nsig = 0, nchan = 0, nscales = 0, detrend = -1, output = 0, ntime_out = 0;
    std::vector<CwtGroup*> groups;
    // trial sums (accumulate = 2) take longer blocks (build_groups): their own groups for the scales a block holds, built
    // at the first such call, followed by the (shared) piece groups of the long scales
    std::vector<CwtGroup*> groups_sum, groups_sum_owned;
    bool groups_sum_built = false;
    bool identity_time = true;
    spy::DevBuf<int> tpos, tfloor;
    spy::DevBuf<float> xt;        // channel-major copy of the chunk's pre-selected signals (cwt_stage_input_kernel)
    size_t xt_cap = 0;
    spy::DevBuf<double> trend, trend_part;
    size_t trend_cap = 0;
    spy::DevBuf<char> stage;      // time-contiguous staging of one chunk of segments
    size_t stage_cap = 0;         // its size in bytes
    int chunk = 0;                // segments per chunk the staging buffer holds
    bool direct = true;           // spyhip_cwt_plan_set_direct: groups flagged `direct` skip the staging buffer
    bool direct_ok = true;        // what plan creation decided (slots increasing, 32-bit row offsets)
    std::vector<int> staged;      // scales that still go through it (blocks of 4096 points and more), in staging-row order
    spy::DevBuf<int> smap;        // staging row -> scale index (device copy of `staged`)
    spy::DevBuf<int> lidx_stage;  // long scale -> staging row

    std::vector<int> long_scales; // scales whose trimmed kernel has more than CWT_PIECE - 1 taps: run piece by piece
    spy::DevBuf<int> lidx;        // their scale indices on the device
    spy::DevBuf<float2> stage_long;   // (chunk, long scale, channel, time) complex sums of the pieces (real outputs)
    int chunk_long = 0;
    // reference precision (spyhip_cwt_plan_set_precision, cwt64_kernel.h): the sampled kernels are kept on the host
    // so that their float64 spectra can be built when asked for
    std::vector<std::vector<double>> ker_re, ker_im;
    std::vector<int> ker_c;
    bool precision64 = false;
    int L64 = 0;
    spywil::PlusPlan plan64{};
    spy::DevBuf<double2> tw64, hspec64, work64;
    spy::DevBuf<int> centre64;
    long long chunk64 = 0;
    ~spyhip_cwt_plan() {
        for (auto* g : groups) delete g;
        for (auto* g : groups_sum_owned) delete g;
    }
};

namespace {
const double PI = 3.14159265358979323846264338327950288;
constexpr int CWT_PIECE = 8192;   // taps per piece of a long kernel: a 16384-point block then yields 8193 outputs

template <int LOG2N, int G, int OUTK>
int launch_cwt(spyhip_cwt_plan* p, const CwtArgs& a, unsigned grid) {
    using C = spyfft::Cfg<LOG2N, G>;
    auto kern = spyfft::cwt_kernel<LOG2N, G, OUTK>;
    SPY_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(C::NTHREADS), C::LDS_BYTES, p->ctx->stream, a);
    SPY_HIP_CHECK(hipGetLastError());
    return 0;
}

template <int LOG2N, int G, int OUTK, bool PAIRT>
int launch_cwt2(spyhip_cwt_plan* p, const CwtArgs& a, unsigned grid) {
    using C = spyfft::Cfg2<LOG2N, G>;
    auto kern = spyfft::cwt2_kernel<LOG2N, G, OUTK, PAIRT>;
    SPY_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(C::NTHREADS), C::LDS_BYTES, p->ctx->stream, a);
    SPY_HIP_CHECK(hipGetLastError());
    return 0;
}

template <int LOG2N, int G>
int launch_cwt2_out(spyhip_cwt_plan* p, const CwtArgs& a, unsigned grid, bool pairt = false) {
    if (pairt) {
        if (p->output == SPYHIP_OUT_FOURIER) return launch_cwt2<LOG2N, G, 2, true>(p, a, grid);
        if (p->output == SPYHIP_OUT_POW) return launch_cwt2<LOG2N, G, 0, true>(p, a, grid);
        return launch_cwt2<LOG2N, G, 1, true>(p, a, grid);
    }
    if (p->output == SPYHIP_OUT_FOURIER) return launch_cwt2<LOG2N, G, 2, false>(p, a, grid);
    if (p->output == SPYHIP_OUT_POW) return launch_cwt2<LOG2N, G, 0, false>(p, a, grid);
    return launch_cwt2<LOG2N, G, 1, false>(p, a, grid);
}

template <int LOG2N, int G, int OUTK>
int launch_cwt2d(spyhip_cwt_plan* p, const CwtArgs& a, unsigned grid) {
    using C = spyfft::Cfg2<LOG2N, G>;
    auto kern = spyfft::cwt2d_kernel<LOG2N, G, OUTK>;
    SPY_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(C::NTHREADS), C::LDS_BYTES, p->ctx->stream, a);
    SPY_HIP_CHECK(hipGetLastError());
    return 0;
}

template <int LOG2N, int G>
int launch_cwt2d_out(spyhip_cwt_plan* p, const CwtArgs& a, unsigned grid) {
    if (p->output == SPYHIP_OUT_FOURIER) return launch_cwt2d<LOG2N, G, 2>(p, a, grid);
    if (p->output == SPYHIP_OUT_POW) return launch_cwt2d<LOG2N, G, 0>(p, a, grid);
    return launch_cwt2d<LOG2N, G, 1>(p, a, grid);
}

constexpr int CWT_DIRECT_G10 = 8, CWT_DIRECT_G11 = 4;     // channel pairs per workgroup of the direct kernels

template <int LOG2N, int G>
int launch_cwt_out(spyhip_cwt_plan* p, const CwtArgs& a, unsigned grid) {
    if (p->output == SPYHIP_OUT_FOURIER) return launch_cwt<LOG2N, G, 2>(p, a, grid);
    if (p->output == SPYHIP_OUT_POW) return launch_cwt<LOG2N, G, 0>(p, a, grid);
    return launch_cwt<LOG2N, G, 1>(p, a, grid);
}
}  // namespace

// Groups of scales by the block length their (trimmed) kernel support needs: >= 4x the kernel (>= 75 % of a block is
// output) while that stays on the packed engine (<= 8192), else >= 2x, up to the 16384-point engine - and never below
// `nbmin`.  Which minimum pays depends on where the results go (measured at 128 ch x 16384 samples x 25 scales 4 ... 100 Hz,
// us/trial: trial sums 169 / 162 / 146 / 189 at 1024 / 2048 / 4096 / 8192 - longer blocks waste less on the halo and the
// staged kernels take them; per-trial outputs 245 / 250 / 265: the direct kernels exist for 1024 and 2048 points only).
// Built from the sampled kernels the plan keeps (ker_re / ker_im / ker_c); appends to `out`.
static int build_groups(spyhip_cwt_plan* p, int nbmin, std::vector<CwtGroup*>& out) {
    spyhip_ctx* ctx = p->ctx;
    const int nscales = p->nscales, nsig = p->nsig;
    std::vector<int> need(nscales);
    for (int s = 0; s < nscales; ++s) {
        const int Lt = (int)p->ker_re[s].size();
        int NB = nbmin;
        while (NB < 4 * (Lt + 1) && NB < 8192) NB <<= 1;
        while (NB < 2 * (Lt + 1) && NB < 16384) NB <<= 1;
        need[s] = 2 * (Lt + 1) > 16384 ? 0 : NB;    // (0: cut into pieces, spyhip_cwt_plan::long_scales)
    }
    for (int NB = 1024; NB <= 16384; NB <<= 1) {
        std::vector<int> ids;
        for (int s = 0; s < nscales; ++s)
            if (need[s] == NB) ids.push_back(s);
        if (ids.empty()) continue;
        int halo = 0, right = 0, lmax = 1;
        for (int s : ids) {
            const int Lt = (int)p->ker_re[s].size();
            lmax = std::max(lmax, Lt);
            halo = std::max(halo, Lt - 1 - p->ker_c[s]);              // reach to the left: L-1-c
            right = std::max(right, p->ker_c[s]);
        }
        const int V = NB - halo - right;
        if (V < 1) {
            spy::set_error("cwt_plan_create: kernel support of %d taps exceeds the %d-point block FFT "
                           "(scale too large for this signal length)", lmax, NB);
            return -3;
        }
        auto* g = new CwtGroup();
        out.push_back(g);
        g->log2n = spy::ilog2((unsigned)NB);
        // channel PAIRS per workgroup of the packed kernel (<= 2^13); channels per workgroup of the 2^14 kernel
        g->G = g->log2n == 10 ? 4 : (g->log2n == 11 ? 2 : 1);
        g->V = V; g->halo = halo; g->nblocks = (nsig + V - 1) / V; g->nscales = (int)ids.size();
        g->direct = g->log2n <= 11;
        g->scale_ids = ids;
        std::vector<float2> tw(NB), hs(ids.size() * (size_t)NB);
        for (int m = 0; m < NB; ++m) {
            const double ang = -2.0 * PI * m / NB;
            tw[m] = make_float2((float)std::cos(ang), (float)std::sin(ang));
        }
        std::vector<int> cshift(ids.size());
        for (size_t q = 0; q < ids.size(); ++q) {
            const int sc = ids[q];
            std::vector<double> re(NB, 0.0), im(NB, 0.0);
            for (size_t m = 0; m < p->ker_re[sc].size(); ++m) { re[m] = p->ker_re[sc][m]; im[m] = p->ker_im[sc][m]; }
            spy::fft_host(re, im);
            for (int k = 0; k < NB; ++k) hs[q * NB + k] = make_float2((float)(re[k] / NB), (float)(im[k] / NB));
            cshift[q] = halo + p->ker_c[sc];
        }
        if (g->tw.upload(tw, ctx->stream) || g->hspec.upload(hs, ctx->stream) || g->cshift.upload(cshift, ctx->stream) ||
            g->sidx.upload(ids, ctx->stream))
            return -2;
    }
    return 0;
}

// family 0: Morlet(w0 = p0) as Morlet.time / cwt_time sample it; family 1: the superlet formulation MorletSL with
// p0 = c_i cycles inside the Gaussian envelope of p1 = k_sd standard deviations (specest/superlet.py:268-363);
// family 2: Paul(m = p0), family 3: DOG(m = p0) - Ricker / Marr / Mexican_hat are DOG(2) - as Paul.time / DOG.time
// sample them (specest/wavelets/wavelets.py:140-223).  The transform convolves with a table of sampled taps, so every
// family runs on the same kernels.
static int cwt_plan_create_impl(spyhip_ctx* ctx, int nsig, int nchan, int nscales, const double* scales, double dt,
                                int family, double p0, double p1, int detrend, int output, const int32_t* tpos,
                                int ntime_out, spyhip_cwt_plan** out);

extern "C" int spyhip_cwt_plan_create(spyhip_ctx* ctx, int nsig, int nchan, int nscales, const double* scales,
                                      double dt, double w0, int detrend, int output, const int32_t* tpos,
                                      int ntime_out, spyhip_cwt_plan** out) {
    return cwt_plan_create_impl(ctx, nsig, nchan, nscales, scales, dt, 0, w0, 0.0, detrend, output, tpos, ntime_out, out);
}

extern "C" int spyhip_cwt_plan_create_family(spyhip_ctx* ctx, int nsig, int nchan, int nscales, const double* scales,
                                             double dt, int family, double p0, double p1, int detrend, int output,
                                             const int32_t* tpos, int ntime_out, spyhip_cwt_plan** out) {
    if (family < 0 || family > 3) { spy::set_error("cwt_plan_create_family: family %d (0 Morlet, 1 MorletSL, 2 Paul, 3 DOG)", family); return -1; }
    if (family == 1 && (!(p0 > 0) || !(p1 > 0))) { spy::set_error("cwt_plan_create_family: cycles and k_sd must be positive"); return -1; }
    if (family >= 2 && (p0 < 1 || p0 > 60 || p0 != std::floor(p0))) { spy::set_error("cwt_plan_create_family: order m = %g (integer 1 ... 60)", p0); return -1; }
    return cwt_plan_create_impl(ctx, nsig, nchan, nscales, scales, dt, family, p0, p1, detrend, output, tpos, ntime_out, out);
}

extern "C" int spyhip_cwt_plan_create_sl(spyhip_ctx* ctx, int nsig, int nchan, int nscales, const double* scales,
                                         double dt, double cycles, double k_sd, int detrend, int output,
                                         const int32_t* tpos, int ntime_out, spyhip_cwt_plan** out) {
    if (!(cycles > 0) || !(k_sd > 0)) { spy::set_error("cwt_plan_create_sl: cycles and k_sd must be positive"); return -1; }
    return cwt_plan_create_impl(ctx, nsig, nchan, nscales, scales, dt, 1, cycles, k_sd, detrend, output, tpos,
                                ntime_out, out);
}

static int cwt_plan_create_impl(spyhip_ctx* ctx, int nsig, int nchan, int nscales, const double* scales, double dt,
                                int family, double p0, double p1, int detrend, int output, const int32_t* tpos,
                                int ntime_out, spyhip_cwt_plan** out) {
    const double w0 = p0;
    if (!ctx || !scales || !out) { spy::set_error("cwt_plan_create: null argument"); return -1; }
    if (nsig < 1 || nchan < 1 || nscales < 1 || dt <= 0) { spy::set_error("cwt_plan_create: bad shape"); return -1; }
    if (output < SPYHIP_OUT_POW || output > SPYHIP_OUT_ABSIMAG) { spy::set_error("bad output kind %d", output); return -1; }
    SPY_HIP_CHECK(hipSetDevice(ctx->device));

    // ---- sampled kernels (transform.py:96-103), trimmed to the taps that can overlap the signal
    struct Ker { std::vector<double> re, im; int c; };
    std::vector<Ker> kers(nscales);
    for (int s = 0; s < nscales; ++s) {
        const double sc = scales[s];
        const double M = family == 1 ? 10.0 * sc * p0 / dt : 10.0 * sc / dt;       // superlet.py:366-375
        const double t0 = (-M + 1.0) / 2.0, t1 = (M + 1.0) / 2.0;
        long long L = (long long)std::ceil(t1 - t0);               // len(np.arange(t0, t1))
        if (L < 1) L = 1;
        const long long c = (L - 1) / 2;                            // fftconvolve mode="same" offset
        // y[n] = sum_m h[m] x[n + c - m], 0 <= n + c - m < nsig  =>  m in [c - (nsig-1), c + (nsig-1)]
        const long long m0 = std::max<long long>(0, c - (nsig - 1));
        const long long m1 = std::min<long long>(L, c + nsig);      // exclusive
        Ker& k = kers[s];
        k.c = (int)(c - m0);
        const double norm = std::sqrt(dt) / (sc * 8.0 * PI) * std::pow(PI, -0.25);
        const double corr = std::exp(-0.5 * w0 * w0);
        // MorletSL: sqrt(dt)/(4 pi) * k_sd / (s c (2 pi)^1.5) * exp(i t/s) * exp(-(k_sd t/s / (2 pi c))^2 / 2)
        const double norm_sl = std::sqrt(dt) / (4.0 * PI) * p1 / (sc * p0 * std::pow(2.0 * PI, 1.5));
        // Paul(m): 2^m i^m m! / sqrt(pi (2m)!) (1 - i x)^-(m+1); DOG(m): (-1)^(m+1) / sqrt(Gamma(m + 1/2)) He_m(x) exp(-x^2/2);
        // both with cwt_time's amplitude normalisation sqrt(dt) / (8 pi s)  (wavelets.py:140-223, transform.py:96-103)
        const int mo = family >= 2 ? (int)p0 : 0;
        const double norm_t = std::sqrt(dt) / (sc * 8.0 * PI);
        const double paul_c = family == 2 ? std::exp(mo * std::log(2.0) + std::lgamma(mo + 1.0) - 0.5 * (std::log(PI) + std::lgamma(2.0 * mo + 1.0))) : 0.0;
        const double dog_c = family == 3 ? ((mo + 1) % 2 ? -1.0 : 1.0) * std::exp(-0.5 * std::lgamma(mo + 0.5)) : 0.0;
        k.re.resize(m1 - m0);
        k.im.resize(m1 - m0);
        for (long long m = m0; m < m1; ++m) {
            const double x = (t0 + (double)m) * dt / sc;            // t / s
            if (family == 2) {
                // (1 - i x)^-(m+1) = r^-(m+1) exp(i (m+1) atan(x)),  times i^m
                const double r = std::sqrt(1.0 + x * x), ph = (mo + 1) * std::atan(x) + 0.5 * PI * mo;
                const double a = norm_t * paul_c * std::pow(r, -(double)(mo + 1));
                k.re[m - m0] = a * std::cos(ph);
                k.im[m - m0] = a * std::sin(ph);
                continue;
            }
            if (family == 3) {
                double h0 = 1.0, h1 = x;                            // probabilists' Hermite: He_{n+1} = x He_n - n He_{n-1}
                for (int q = 1; q < mo; ++q) { const double h2 = x * h1 - q * h0; h0 = h1; h1 = h2; }
                k.re[m - m0] = norm_t * dog_c * (mo == 0 ? 1.0 : h1) * std::exp(-0.5 * x * x);
                k.im[m - m0] = 0.0;
                continue;
            }
            if (family == 1) {
                const double u = p1 * x / (2.0 * PI * p0);
                const double g = norm_sl * std::exp(-0.5 * u * u);
                k.re[m - m0] = g * std::cos(x);
                k.im[m - m0] = g * std::sin(x);
                continue;
            }
            const double g = norm * std::exp(-0.5 * x * x);
            k.re[m - m0] = g * (std::cos(w0 * x) - corr);
            k.im[m - m0] = g * std::sin(w0 * x);
        }
    }
    auto* p = new spyhip_cwt_plan();
    p->ctx = ctx; p->nsig = nsig; p->nchan = nchan; p->nscales = nscales;
    p->detrend = detrend; p->output = output;
    for (int s = 0; s < nscales; ++s) {
        p->ker_re.push_back(kers[s].re);
        p->ker_im.push_back(kers[s].im);
        p->ker_c.push_back(kers[s].c);
    }
    // block length per scale and the groups of scales that share one (build_groups); scales whose kernel no block holds
    for (int s = 0; s < nscales; ++s)
        if (2 * ((int)kers[s].re.size() + 1) > 16384) p->long_scales.push_back(s);
    if (int rc = build_groups(p, 1024, p->groups)) { delete p; return rc; }
    // ---- kernels longer than a block: h = sum_p h_p (pieces of CWT_PIECE taps), y = sum_p h_p * x.  Piece p is an
    // overlap-save convolution of its own: taps [p PL, p PL + Lp), centre c_p = c - p PL (may be negative or beyond the
    // piece), input window from o0 - halo_p with halo_p = Lp - 1 - c_p, output n of block o0 at q = n - o0 + Lp - 1.
    for (size_t li = 0; li < p->long_scales.size(); ++li) {
        const int sc = p->long_scales[li];
        const int Lt = (int)kers[sc].re.size();
        const int NB = 16384;
        for (int pc = 0; pc * CWT_PIECE < Lt; ++pc) {
            const int m0 = pc * CWT_PIECE, Lp = std::min(CWT_PIECE, Lt - m0);
            auto* g = new CwtGroup();
            p->groups.push_back(g);
            g->log2n = 14; g->G = 1; g->nscales = 1; g->long_idx = (int)li; g->piece = pc;
            g->V = NB - (Lp - 1);
            g->halo = Lp - 1 - (kers[sc].c - m0);
            g->nblocks = (nsig + g->V - 1) / g->V;
            std::vector<float2> tw(NB), hs(NB);
            for (int m = 0; m < NB; ++m) {
                const double ang = -2.0 * PI * m / NB;
                tw[m] = make_float2((float)std::cos(ang), (float)std::sin(ang));
            }
            std::vector<double> re(NB, 0.0), im(NB, 0.0);
            for (int m = 0; m < Lp; ++m) { re[m] = kers[sc].re[m0 + m]; im[m] = kers[sc].im[m0 + m]; }
            spy::fft_host(re, im);
            for (int k = 0; k < NB; ++k) hs[k] = make_float2((float)(re[k] / NB), (float)(im[k] / NB));
            // complex outputs add up in the staging rows of the scale itself, real ones in the complex side buffer
            std::vector<int> cshift{Lp - 1}, ids{output == SPYHIP_OUT_FOURIER ? sc : (int)li};
            if (g->tw.upload(tw, ctx->stream) || g->hspec.upload(hs, ctx->stream) || g->cshift.upload(cshift, ctx->stream) ||
                g->sidx.upload(ids, ctx->stream)) {
                delete p;
                return -2;
            }
        }
    }
    if (!p->long_scales.empty() && p->lidx.upload(p->long_scales, ctx->stream)) { delete p; return -2; }
    // compact staging rows for the scales the direct kernels do not serve
    {
        std::vector<int> row(nscales, -1);
        auto stage_row = [&](int sc) {
            if (row[sc] < 0) { row[sc] = (int)p->staged.size(); p->staged.push_back(sc); }
            return row[sc];
        };
        std::vector<std::vector<int>> ids(p->groups.size());
        for (size_t gi = 0; gi < p->groups.size(); ++gi) {
            CwtGroup* g = p->groups[gi];
            if (g->direct) continue;
            if (g->long_idx >= 0) {
                const int sc = p->long_scales[g->long_idx];
                // complex outputs: the pieces add up in the scale's own staging row; real ones in the side buffer (row = long index)
                ids[gi] = {output == SPYHIP_OUT_FOURIER ? stage_row(sc) : g->long_idx};
                if (output != SPYHIP_OUT_FOURIER) stage_row(sc);
            }
        }
        for (size_t gi = 0; gi < p->groups.size(); ++gi) {
            CwtGroup* g = p->groups[gi];
            if (g->direct || g->long_idx >= 0) continue;
            for (int sc : g->scale_ids) ids[gi].push_back(stage_row(sc));
        }
        for (size_t gi = 0; gi < p->groups.size(); ++gi)
            if (!ids[gi].empty() && p->groups[gi]->sidx_stage.upload(ids[gi], ctx->stream)) { delete p; return -2; }
        if (!p->staged.empty() && p->smap.upload(p->staged, ctx->stream)) { delete p; return -2; }
        if (!p->long_scales.empty()) {
            std::vector<int> lrow;
            for (int sc : p->long_scales) lrow.push_back(row[sc]);
            if (p->lidx_stage.upload(lrow, ctx->stream)) { delete p; return -2; }
        }
    }
    p->identity_time = (tpos == nullptr);
    p->ntime_out = tpos ? ntime_out : nsig;
    if (tpos) {
        std::vector<int> tp(tpos, tpos + nsig);
        for (int v : tp)
            if (v >= ntime_out) { spy::set_error("cwt_plan_create: tpos entry %d >= ntime_out %d", v, ntime_out); delete p; return -1; }
        if (p->tpos.upload(tp, ctx->stream)) { delete p; return -2; }
        // the direct kernels address a tile relative to the slot reached before it: slots must increase with the samples
        std::vector<int> fl(nsig);
        int last = -1;
        for (int n = 0; n < nsig; ++n) {
            if (tp[n] >= 0) {
                if (tp[n] <= last) p->direct = false;
                last = tp[n];
            }
            fl[n] = std::max(last, 0);
        }
        if (p->tfloor.upload(fl, ctx->stream)) { delete p; return -2; }
    }
    // ... and a block's rows must stay within 32-bit byte offsets
    if ((double)p->nscales * p->nchan * 8.0 * 2048.0 >= 4294967296.0) p->direct = false;
    p->direct_ok = p->direct;
    *out = p;
    return 0;
}

extern "C" int spyhip_cwt_plan_set_precision(spyhip_cwt_plan* p, int reference) {
    if (!p) { spy::set_error("cwt_plan_set_precision: null plan"); return -1; }
    if (!reference) { p->precision64 = false; return 0; }
    if (!p->hspec64.p) {
        SPY_HIP_CHECK(hipSetDevice(p->ctx->device));
        size_t lmax = 1;
        for (const auto& k : p->ker_re) lmax = std::max(lmax, k.size());
        long long L = 16;
        while (L < (long long)p->nsig + (long long)lmax - 1) L <<= 1;
        if (L > (1 << 22)) { spy::set_error("cwt_plan_set_precision: convolution length %lld beyond 2^22", L); return -3; }
        p->L64 = (int)L;
        if (!spywil::plus_plan(p->L64, &p->plan64)) { spy::set_error("cwt_plan_set_precision: no radix schedule"); return -3; }
        std::vector<double2> tw(L), hs((size_t)p->nscales * L);
        for (long long m = 0; m < L; ++m) {
            const double ang = -2.0 * PI * (double)m / (double)L;
            tw[m] = make_double2(std::cos(ang), std::sin(ang));
        }
        for (int sc = 0; sc < p->nscales; ++sc) {
            std::vector<double> re(L, 0.0), im(L, 0.0);
            for (size_t m = 0; m < p->ker_re[sc].size(); ++m) { re[m] = p->ker_re[sc][m]; im[m] = p->ker_im[sc][m]; }
            spy::fft_host(re, im);
            for (long long k = 0; k < L; ++k) hs[(size_t)sc * L + k] = make_double2(re[k] / (double)L, im[k] / (double)L);
        }
        if (p->tw64.upload(tw, p->ctx->stream) || p->hspec64.upload(hs, p->ctx->stream) ||
            p->centre64.upload(p->ker_c, p->ctx->stream)) return -2;
    }
    p->precision64 = true;
    return 0;
}

template <int OUTK>
static int launch_cwt64(spyhip_cwt_plan* p, const spyfft::Cwt64Args& a, unsigned grid) {
    hipLaunchKernelGGL(spyfft::cwt64_kernel<OUTK>, dim3(grid), dim3(256), 0, p->ctx->stream, a);
    SPY_HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" int spyhip_cwt_plan_destroy(spyhip_cwt_plan* p) {
    delete p;
    return 0;
}

extern "C" int spyhip_cwt_exec(spyhip_cwt_plan* p, const float* data_d, int64_t ld, const int32_t* chan_idx_d,
                               const int64_t* seg_start_d, const int64_t* trial_lo_d, const int64_t* trial_hi_d,
                               int nseg, void* out_d, int accumulate) {
    if (!p || !data_d || !seg_start_d || !trial_lo_d || !trial_hi_d || !out_d) { spy::set_error("cwt_exec: null argument"); return -1; }
    if (nseg <= 0) return 0;
    SPY_HIP_CHECK(hipSetDevice(p->ctx->device));
    CwtArgs a{};
    a.data = data_d; a.ld = ld; a.chan_idx = chan_idx_d;
    a.seg_start = reinterpret_cast<const long long*>(seg_start_d);
    a.trial_lo = reinterpret_cast<const long long*>(trial_lo_d);
    a.trial_hi = reinterpret_cast<const long long*>(trial_hi_d);
    a.nseg = nseg; a.nsig = p->nsig; a.nchan = p->nchan; a.nscales = p->nscales;
    a.nscales_total = p->nscales;
    a.detrend = p->detrend; a.out_kind = p->output;
    a.tpos = p->identity_time ? nullptr : p->tpos.p;
    a.tfloor = p->identity_time ? nullptr : p->tfloor.p;
    a.ntime_out = p->ntime_out; a.out = out_d; a.accumulate = accumulate;
    if (p->detrend >= 0) {
        const size_t need = (size_t)nseg * p->nchan * 2;
        if (need > p->trend_cap) {
            if (p->trend.p) { (void)hipFree(p->trend.p); p->trend.p = nullptr; }
            if (p->trend_part.p) { (void)hipFree(p->trend_part.p); p->trend_part.p = nullptr; }
            if (p->trend.alloc(need) || p->trend_part.alloc(need * spyfft::CWT_TREND_SPLITS)) return -2;
            p->trend_cap = need;
        }
        a.trend = p->trend.p;
        if (nseg > 65535) { spy::set_error("cwt_exec: more than 65535 segments per call"); return -1; }
        if (p->detrend == 0) {
            // the reference's float32 mean in its own summation order (one thread per segment and channel)
            hipLaunchKernelGGL(spyfft::cwt_mean_np_kernel, dim3((p->nchan + 63) / 64, nseg), dim3(64), 0, p->ctx->stream, a,
                               p->trend.p);
        } else {
            hipLaunchKernelGGL(spyfft::cwt_trend_partial_kernel, dim3((p->nchan + 63) / 64, spyfft::CWT_TREND_SPLITS, nseg),
                               dim3(256), 0, p->ctx->stream, a, p->trend_part.p);
            hipLaunchKernelGGL(spyfft::cwt_trend_final_kernel, dim3((unsigned)(((size_t)nseg * p->nchan + 255) / 256)), dim3(256),
                               0, p->ctx->stream, a, p->trend_part.p, p->trend.p);
        }
        SPY_HIP_CHECK(hipGetLastError());
    }
    // Per-segment outputs (accumulate 0 / 1): scales on 1024- / 2048-point blocks leave their kernel in the output layout
    // (cwt2d_kernel); the others (and every scale of a plan with spyhip_cwt_plan_set_direct(plan, 0) or float64 precision)
    // go through the time-contiguous staging buffer.  Trial sums (accumulate 2): every scale staged, the packed kernels
    // carrying one channel of TWO consecutive segments per thread and storing the sum (cwt_kernel.h, PAIRT) - a staging
    // row set then holds a pair of segments.  As many row sets per chunk as fit ~4 GiB (at least one).
    const bool use_direct = p->direct && !p->precision64 && accumulate != 2;
    bool pairt = accumulate == 2 && !p->precision64;
    const std::vector<CwtGroup*>* groups = &p->groups;
    if (pairt && p->nsig >= 4096) {                  // trial sums of long signals: blocks of at least 4096 points
        if (!p->groups_sum_built) {
            if (int rc = build_groups(p, 4096, p->groups_sum_owned)) {
                for (auto* g : p->groups_sum_owned) delete g;
                p->groups_sum_owned.clear();
                return rc;
            }
            p->groups_sum = p->groups_sum_owned;
            for (CwtGroup* gr : p->groups)
                if (gr->long_idx >= 0) p->groups_sum.push_back(gr);
            p->groups_sum_built = true;
        }
        groups = &p->groups_sum;
    }
    for (const CwtGroup* gr : *groups) pairt = pairt && gr->log2n <= 13;       // (the 16384-point kernel is not packed)
    if (!pairt) groups = &p->groups;
    const int nst = use_direct ? (int)p->staged.size() : p->nscales;          // staging rows per row set
    const size_t esz = (p->output == SPYHIP_OUT_FOURIER) ? 8 : 4;
    const size_t per_seg = (size_t)nst * p->nchan * p->nsig * esz;
    const int nsets = pairt ? (nseg + 1) / 2 : nseg;
    int chunk = nst ? (int)std::max<size_t>(1, std::min<size_t>((size_t)nsets, ((size_t)4 << 30) / std::max<size_t>(per_seg, 1)))
                    : std::min(nseg, 65535);
    if (per_seg * chunk > p->stage_cap) {
        if (p->stage.p) { SPY_HIP_CHECK(hipStreamSynchronize(p->ctx->stream)); (void)hipFree(p->stage.p); p->stage.p = nullptr; p->stage_cap = 0; }
        if (p->stage.alloc(per_seg * chunk)) return -2;
        p->stage_cap = per_seg * chunk;
    }
    a.stage = p->stage.p;
    const int nlong = (int)p->long_scales.size();
    const bool long_side = nlong > 0 && esz == 4;      // real outputs: the pieces are summed as complex numbers first
    if (long_side && chunk > p->chunk_long) {
        if (p->stage_long.p) { SPY_HIP_CHECK(hipStreamSynchronize(p->ctx->stream)); (void)hipFree(p->stage_long.p); p->stage_long.p = nullptr; }
        if (p->stage_long.alloc((size_t)chunk * nlong * p->nchan * p->nsig)) return -2;
        p->chunk_long = chunk;
    }
    if (pairt) chunk *= 2;                             // from here on: segments per chunk
    // channel-major copy of the chunk's signals for the float32 kernels (several channels per row: a gather otherwise)
    const bool use_xt = !p->precision64 && p->nchan > 1;
    if (use_xt) {
        const size_t need = (size_t)std::min(chunk, nseg) * p->nchan * p->nsig;
        if (need > p->xt_cap) {
            if (p->xt.p) { SPY_HIP_CHECK(hipStreamSynchronize(p->ctx->stream)); (void)hipFree(p->xt.p); p->xt.p = nullptr; p->xt_cap = 0; }
            if (p->xt.alloc(need)) return -2;
            p->xt_cap = need;
        }
    }
    for (int s0 = 0; s0 < nseg; s0 += chunk) {
        const int ns = std::min(chunk, nseg - s0);
        CwtArgs c = a;
        c.seg0 = s0;
        c.seg_start = a.seg_start + s0;
        c.trial_lo = a.trial_lo + s0;
        c.trial_hi = a.trial_hi + s0;
        if (a.trend) c.trend = a.trend + (size_t)s0 * p->nchan * 2;
        c.nseg = ns;
        if (p->nscales > 65535 || ns > 65535) { spy::set_error("cwt_exec: grid too large"); return -1; }
        if (use_xt) {
            hipLaunchKernelGGL(spyfft::cwt_stage_input_kernel, dim3((p->nsig + 63) / 64, (p->nchan + 63) / 64, ns), dim3(256), 0,
                               p->ctx->stream, c, p->xt.p);
            SPY_HIP_CHECK(hipGetLastError());
            c.xt = p->xt.p;
        }
        if (p->precision64) {
            // float64 convolutions, one workgroup per (segment, channel), three length-L work arrays each: launches of
            // at most ~2 GiB of them
            spyfft::Cwt64Args fa{};
            fa.c = c;
            fa.L = p->L64; fa.plan = p->plan64; fa.tw64 = p->tw64.p; fa.hspec64 = p->hspec64.p; fa.centre = p->centre64.p;
            const long long items = (long long)ns * p->nchan;
            const size_t per = (size_t)3 * p->L64 * sizeof(double2);
            long long cw = std::max<long long>(2LL * p->ctx->num_cu, (long long)(((size_t)2 << 30) / per));
            if (cw > items) cw = items;
            if (cw > p->chunk64) {
                if (p->work64.p) { SPY_HIP_CHECK(hipStreamSynchronize(p->ctx->stream)); (void)hipFree(p->work64.p); p->work64.p = nullptr; }
                if (p->work64.alloc((size_t)cw * 3 * p->L64)) return -2;
                p->chunk64 = cw;
            }
            fa.work = p->work64.p;
            for (long long w0 = 0; w0 < items; w0 += p->chunk64) {
                fa.wg0 = w0;
                const unsigned g = (unsigned)std::min<long long>(p->chunk64, items - w0);
                int rc = p->output == SPYHIP_OUT_FOURIER ? launch_cwt64<2>(p, fa, g)
                         : (p->output == SPYHIP_OUT_POW ? launch_cwt64<0>(p, fa, g) : launch_cwt64<1>(p, fa, g));
                if (rc) return rc;
            }
        }
        for (size_t gi = 0; gi < groups->size(); ++gi) {              // one launch per block length
            const CwtGroup* gr = (*groups)[gi];
            if (p->precision64) break;
            CwtArgs k = c;
            k.nscales = gr->nscales;
            k.sidx = gr->sidx.p;
            k.tw = gr->tw.p; k.hspec = gr->hspec.p; k.cshift = gr->cshift.p;
            k.V = gr->V; k.halo = gr->halo; k.nblocks = gr->nblocks;
            if (use_direct && gr->direct) {
                // (G = 4 / 2 - the staged kernels' workgroup shape, 32- / 16-byte runs - measured 284 against 236 us/trial at c4)
                const int G = gr->log2n == 10 ? CWT_DIRECT_G10 : CWT_DIRECT_G11;
                const long long ngrp = ((p->nchan + 1) / 2 + G - 1) / G;
                const long long grid = (long long)ns * ngrp * gr->nblocks;
                if (grid > 0x7fffffffLL) { spy::set_error("cwt_exec: grid too large"); return -1; }
                int rc = gr->log2n == 10 ? launch_cwt2d_out<10, CWT_DIRECT_G10>(p, k, (unsigned)grid)
                                         : launch_cwt2d_out<11, CWT_DIRECT_G11>(p, k, (unsigned)grid);
                if (rc) return rc;
                continue;
            }
            if (use_direct) { k.sidx = gr->sidx_stage.p; k.nscales_total = nst; }     // compact staging rows
            if (gr->long_idx >= 0) {
                k.stage_add = gr->piece > 0;
                if (long_side) { k.stage = p->stage_long.p; k.nscales_total = nlong; k.sidx = gr->sidx.p; }
            }
            // work units per row set: channel pairs of a segment; (PAIRT) channels of a segment pair; channels (2^14 blocks)
            const long long nunit = pairt ? p->nchan : (gr->log2n <= 13 ? (p->nchan + 1) / 2 : p->nchan);
            const long long ngrp = (nunit + gr->G - 1) / gr->G;
            const long long grid = (long long)(pairt ? (ns + 1) / 2 : ns) * ngrp * gr->nblocks;
            if (grid > 0x7fffffffLL) { spy::set_error("cwt_exec: grid too large"); return -1; }
            const unsigned g = (unsigned)grid;
            int rc;
            switch (gr->log2n) {
                case 10: rc = launch_cwt2_out<10, 4>(p, k, g, pairt); break;
                case 11: rc = launch_cwt2_out<11, 2>(p, k, g, pairt); break;
                case 12: rc = launch_cwt2_out<12, 1>(p, k, g, pairt); break;
                case 13: rc = launch_cwt2_out<13, 1>(p, k, g, pairt); break;
                case 14: rc = gr->long_idx >= 0 ? launch_cwt<14, 1, 2>(p, k, g) : launch_cwt_out<14, 1>(p, k, g); break;
                default: spy::set_error("cwt_exec: unsupported block length 2^%d", gr->log2n); return -1;
            }
            if (rc) return rc;
        }
        if (nst == 0) continue;                       // every scale left its kernel in the output layout
        if (long_side && !p->precision64) {
            const long long tot = (long long)ns * nlong * p->nchan * p->nsig;
            if ((tot + 255) / 256 > 0x7fffffffLL) { spy::set_error("cwt_exec: grid too large"); return -1; }
            hipLaunchKernelGGL(spyfft::cwt_long_convert_kernel, dim3((unsigned)((tot + 255) / 256)), dim3(256), 0,
                               p->ctx->stream, p->stage_long.p, use_direct ? p->lidx_stage.p : p->lidx.p, nlong, ns, nst, p->nchan,
                               p->nsig, p->output, reinterpret_cast<float*>(p->stage.p));
        }
        if (pairt) c.nseg = (ns + 1) / 2;             // scatter kernels: row sets to add up ...
        c.nscales = nst;                              // ... staging rows per row set ...
        if (use_direct) { c.smap = p->smap.p; c.nscales_out = p->nscales; }     // ... and where they go in the output
        const dim3 sg((p->nsig + 63) / 64, nst, accumulate == 2 ? 1 : ns);
        // real outputs of long trials: tiles of 256 samples x 16 channels (1-KiB reads of the staging rows: 51 -> 46 us/trial at c4)
        const bool wide = esz == 4 && (p->nsig & 3) == 0 && p->nsig >= 1024;
        if (esz == 8) hipLaunchKernelGGL(spyfft::cwt_scatter_kernel<float2>, sg, dim3(256), 0, p->ctx->stream, c);
        else if (wide) hipLaunchKernelGGL(spyfft::cwt_scatter_wide_kernel, dim3((p->nsig + 255) / 256, sg.y, sg.z), dim3(256), 0, p->ctx->stream, c);
        else hipLaunchKernelGGL(spyfft::cwt_scatter_kernel<float>, sg, dim3(256), 0, p->ctx->stream, c);
        SPY_HIP_CHECK(hipGetLastError());
    }
    return 0;
}

extern "C" int spyhip_cwt_plan_set_direct(spyhip_cwt_plan* p, int on) {
    if (!p) { spy::set_error("cwt_plan_set_direct: null plan"); return -1; }
    if (on && !p->direct_ok) {                   // (slots not increasing / rows beyond 32-bit offsets: staging only)
        spy::set_error("cwt_plan_set_direct: this plan's outputs cannot be written by the transform kernels (time slots not "
                       "increasing with the samples, or rows beyond 32-bit offsets)");
        return -3;
    }
    p->direct = on != 0;
    return 0;
}

extern "C" int spyhip_slt_combine(spyhip_ctx* ctx, void* acc_d, const void* spec_d, int64_t nrows, int nscales,
                                  int nsub, int s0, int nchan, const double* expo, int init, int modulus_only) {
    if (!ctx || !acc_d || !spec_d || !expo) { spy::set_error("slt_combine: null argument"); return -1; }
    if (nrows < 0 || nscales < 1 || nsub < 1 || s0 < 0 || s0 + nsub > nscales || nchan < 1) {
        spy::set_error("slt_combine: bad shape");
        return -1;
    }
    if (nrows == 0) return 0;
    SPY_HIP_CHECK(hipSetDevice(ctx->device));
    // the kernel takes <= SLT_MAX_SCALES exponents by value: wider scale sets go in slices
    for (int q0 = 0; q0 < nsub; q0 += spyfft::SLT_MAX_SCALES) {
        const int nq = std::min(spyfft::SLT_MAX_SCALES, nsub - q0);
        spyfft::SltArgs a{};
        a.acc = reinterpret_cast<float2*>(acc_d);
        a.spec = reinterpret_cast<const float2*>(spec_d);
        a.nrows = nrows; a.nscales = nscales; a.nsub = nq; a.s0 = s0 + q0; a.nchan = nchan; a.init = init;
        a.nsub_total = nsub; a.q0 = q0; a.modulus_only = modulus_only & 3; a.square = (modulus_only >> 2) & 1;
        for (int q = 0; q < nq; ++q) a.expo[q] = expo[q0 + q];
        const long long blocks = ((long long)nrows * nq * nchan + 255) / 256;
        if (blocks > 0x7fffffffLL) { spy::set_error("slt_combine: grid too large"); return -1; }
        hipLaunchKernelGGL(spyfft::slt_combine_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream, a);
    }
    SPY_HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" int spyhip_spec_convert(spyhip_ctx* ctx, const void* in_d, int64_t n, int output, void* out_d) {
    if (!ctx || !in_d || !out_d) { spy::set_error("spec_convert: null argument"); return -1; }
    if (output < SPYHIP_OUT_POW || output > SPYHIP_OUT_ABSIMAG || output == SPYHIP_OUT_FOURIER) {
        spy::set_error("spec_convert: %d is not a real output kind", output);
        return -1;
    }
    if (n <= 0) return 0;
    SPY_HIP_CHECK(hipSetDevice(ctx->device));
    const long long blocks = (n + 255) / 256;
    if (blocks > 0x7fffffffLL) { spy::set_error("spec_convert: grid too large"); return -1; }
    hipLaunchKernelGGL(spyfft::spec_convert_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream,
                       reinterpret_cast<const float2*>(in_d), (long long)n, output, reinterpret_cast<float*>(out_d));
    SPY_HIP_CHECK(hipGetLastError());
    return 0;
}
