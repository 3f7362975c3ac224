// Host side of K1/K2: plan construction and kernel dispatch for the tapered-FFT
// kernels (spyhip_fft_plan_create / spyhip_fft_exec of include/spyhip.h).
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <string>

#include "spy_common.h"
#include "host_fft.h"
#include "mtmfft_kernel.h"
#include "mtmfft2_kernel.h"
#include "mtmfft_blue_kernel.h"
#include "mtmfft_long.h"
#include "mtmfft_generic.h"
#include "mtmfft_mixed_plan.h"

namespace spyfft {
}  // namespace spyfft
#include "f64_stockham.h"     // PlusPlan + plus_plan (the factor schedule of the any-length reference-precision kernel)
#include "mtmfft_f64_args.h"  // F64Args (the kernels themselves stay out of this translation unit: they pull in the Wilson kernels)
namespace spyfft {
int dec64_launch_a(hipStream_t stream, const F64Args& a, int nfft, int npairs, int outk, bool mean);
int dec64_launch_b(hipStream_t stream, const F64Args& a, int nfft, int npairs, int outk, bool mean);
int dec64_launch_c(hipStream_t stream, const F64Args& a, int nfft, int npairs, int outk, bool mean);
int dec64_launch_d(hipStream_t stream, const F64Args& a, int nfft, int npairs, int outk, bool mean);
int dec64_launch_e(hipStream_t stream, const F64Args& a, int nfft, int npairs, int outk, bool mean);
int dec64_launch_f(hipStream_t stream, const F64Args& a, int nfft, int npairs, int outk, bool mean);
int dec64_launch_g(hipStream_t stream, const F64Args& a, int nfft, int npairs, int outk, bool mean);
int dec64_launch_h(hipStream_t stream, const F64Args& a, int nfft, int npairs, int outk, bool mean);
int dec64_launch_i(hipStream_t stream, const F64Args& a, int nfft, int npairs, int outk, bool mean);
int dec64_launch_j(hipStream_t stream, const F64Args& a, int nfft, int npairs, int outk, bool mean);
int dec64_launch_k(hipStream_t stream, const F64Args& a, int nfft, int npairs, int outk, bool mean);
int dec64_launch_l(hipStream_t stream, const F64Args& a, int nfft, int npairs, int outk, bool mean);
int dec64_launch_m(hipStream_t stream, const F64Args& a, int nfft, int npairs, int outk, bool mean);
int dec64_launch_n(hipStream_t stream, const F64Args& a, int nfft, int npairs, int outk, bool mean);
int dec64_launch_half_a(hipStream_t stream, const F64Args& a, int nfft, int nchan, int outk, bool mean);
int dec64_launch_half_b(hipStream_t stream, const F64Args& a, int nfft, int nchan, int outk, bool mean);
struct Long64Args {           // mtmfft_declong64.h (kept out of this translation unit, as F64Args)
    MtmArgs m;
    const double* tapers64;
    const double2* twM;
    const double2* twN;
    const double2* twP;
    double2* scratch;
    const double* stats;
    const double* wsum;
    int seg0, nsegc;
    int npair;
};
int declong64_group(int M);
int declong64_launch_sub_a(hipStream_t stream, const Long64Args& a, int M, int P, long long nblocks);
int declong64_launch_sub_b(hipStream_t stream, const Long64Args& a, int M, int P, long long nblocks);
int declong64_launch_post(hipStream_t stream, const Long64Args& a, int P, int M, int outk, bool mean);
int f64_any_launch(hipStream_t stream, F64Args a, long long grid, long long chunk, int outk, bool mean);
int dec_launch_a(hipStream_t stream, const MtmArgs& a, int nfft, int nquads, int outk, bool mean);
int dec_launch_b(hipStream_t stream, const MtmArgs& a, int nfft, int nquads, int outk, bool mean);
int dec_launch_c(hipStream_t stream, const MtmArgs& a, int nfft, int nquads, int outk, bool mean);
int dec_launch_d(hipStream_t stream, const MtmArgs& a, int nfft, int nquads, int outk, bool mean);
int dec_launch_e(hipStream_t stream, const MtmArgs& a, int nfft, int nquads, int outk, bool mean);
int dec_launch_f(hipStream_t stream, const MtmArgs& a, int nfft, int nquads, int outk, bool mean);
int dec_launch_g(hipStream_t stream, const MtmArgs& a, int nfft, int nquads, int outk, bool mean);
int dec_launch_h(hipStream_t stream, const MtmArgs& a, int nfft, int nquads, int outk, bool mean);
int dec_launch_i(hipStream_t stream, const MtmArgs& a, int nfft, int nquads, int outk, bool mean);
int dec_launch_j(hipStream_t stream, const MtmArgs& a, int nfft, int nquads, int outk, bool mean);
int dec_launch_k(hipStream_t stream, const MtmArgs& a, int nfft, int nquads, int outk, bool mean);
int dec_launch_l(hipStream_t stream, const MtmArgs& a, int nfft, int nquads, int outk, bool mean);
int dec_launch_c2(hipStream_t stream, const MtmArgs& a, int nquads);
int dec_launch_half_a(hipStream_t stream, const MtmArgs& a, int nfft, int npairs, int outk, bool mean);
int dec_launch_half_b(hipStream_t stream, const MtmArgs& a, int nfft, int npairs, int outk, bool mean);
int dec_launch_half_c(hipStream_t stream, const MtmArgs& a, int nfft, int npairs, int outk, bool mean);
int quad_half_launch(hipStream_t stream, const MtmArgs& a, int npairs, int outk, bool mean);
// mtmfft_declong_{a,b}.hip: N = P M through HBM (mtmfft_declong.h)
int declong_group(int M);
int declong_launch_sub_a(hipStream_t stream, const LongArgs& a, int M, int P, long long nblocks);
int declong_launch_sub_b(hipStream_t stream, const LongArgs& a, int M, int P, long long nblocks);
int declong_launch_post(hipStream_t stream, const LongArgs& a, int P, int M, int outk, bool mean);
int mixed_launch(hipStream_t stream, const MtmArgs& a, const MixPlan& g, int threads, size_t lds, unsigned grid, int outk,
                 bool mean);
}

using spyfft::GenPlan;
using spyfft::MtmArgs;

struct spyhip_fft_plan {
    spyhip_ctx* ctx = nullptr;
    int nsig = 0, nfft = 0, nchan = 0, ntaper = 0, nfsel = 0, output = 0, keeptapers = 1;
    int detrend = -1, demean_taper = 0;
    float scale = 1.f;
    bool pow2 = false;
    bool dec = false;           // compile-time radix schedules for decimal lengths (mtmfft_dec_kernel.h)
    bool half = false;          // ... in HALF form: channel pairs, the real transform through the schedule of nfft / 2 (10240 < nfft <= 20480)
    spy::DevBuf<float2> twh;    // exp(-2 pi i f / nfft), f <= nfft / 4
    bool mixed = false;         // packed mixed-radix engine for 5-smooth lengths (mtmfft_mixed.h)
    spyfft::MixPlan mix{};
    int mix_threads = 0;
    bool blue = false;          // Bluestein on the packed power-of-two engine (nfft <= 4096, not a power of two)
    bool longp = false;         // Bluestein with four-step length-M transforms through HBM (mtmfft_long.h)
    bool long_direct = false;   // ... or, for power-of-two nfft, one plain four-step transform
    int dl_P = 0, dl_M = 0;     // N = P M: scheduled sub-transforms of length M + one radix-P pass through HBM (mtmfft_declong.h)
    int l1 = 0, l2 = 0;         // M1 = 2^l1, M2 = 2^l2
    spy::DevBuf<float2> tw1, tw2, twM;
    spy::DevBuf<double> wsum, stats, stats_part;
    spy::DevBuf<float4> scratch;
    size_t stats_cap = 0, scratch_cap = 0;
    int log2n = 0, G = 1;
    GenPlan gen{};
    size_t lds_bytes = 0;
    spy::DevBuf<float> tapers;
    spy::DevBuf<float> tapers_half;   // tapers * scale / 2 (mtmfft_quad_kernel: no scaling left in its epilogue)
    spy::DevBuf<float2> tw, chirp, bhat;
    spy::DevBuf<int> fpos;
    bool identity_freq = true;
    bool blocked = false;
    unsigned* absmax = nullptr;  // spyhip_fft_plan_set_absmax: where the exec calls leave the range of the spectra
    float wnorm = 0.f;           // max_k || w_k scale ||_2
    bool precision64 = false;   // float64 taper product + FFT, complex64 rounding where the reference rounds (mtmfft_f64_kernel.h)
    bool f64_any = false;       // ... through the any-length kernel (work arrays in global memory)
    bool f64_dec = false;       // ... through the compile-time-schedule kernel (mtmfft_dec64_kernel.h)
    int f64_blue = 0;           // any-length kernel in its Bluestein form: the length M = 2^m >= 2 nfft - 1
    bool f64_dl = false;        // ... N = dl_P x dl_M through HBM (mtmfft_declong64.h)
    bool f64_half = false;      // ... single channels through the schedule of nfft / 2 (CfgD64::HALF: 10240 < nfft <= 20480)
    spy::DevBuf<double2> tw64h; // exp(-2 pi i m / (nfft / 2)) for it (tw64 then serves as the half-step table)
    spy::DevBuf<double2> tw64_sub, tw64_P, scratch64;
    size_t scratch64_cap = 0;
    spy::DevBuf<double2> chirp64, bhat64;
    spywil::PlusPlan f64_plan{};
    spy::DevBuf<double2> f64_work;
    long long f64_chunk = 0;
    spy::DevBuf<double> tapers64;
    spy::DevBuf<double2> tw64;
    std::string fp32_kernel_name;
    bool ref_mean = false;      // constant detrending with the reference's float32 row-order means (seq_mean_kernel)
    bool seg_f64 = false;       // the reference holds the segments as float64 arrays (padded sliding windows)
    spy::DevBuf<float> means;
    size_t means_cap = 0;
    spy::DevBuf<float2> xpair;  // pair-major copy of the segments of a launch (pair forms of trials beyond 10240 samples)
    size_t xpair_cap = 0;
    std::string kernel_name;
};

namespace {

const double PI = 3.14159265358979323846264338327950288;

std::vector<float2> twiddle_table(int n) {
    std::vector<float2> t(n);
    for (int m = 0; m < n; ++m) {
        const double ang = -2.0 * PI * (double)m / (double)n;
        t[m] = make_float2((float)std::cos(ang), (float)std::sin(ang));
    }
    return t;
}

// radix schedule of the generic Stockham passes; false if n has a prime factor > 13
bool factorize(int n, int* radix, int* nfac) {
    static const int cand[] = {16, 8, 4, 2, 3, 5, 7, 11, 13};
    int k = 0;
    for (int c : cand) {
        while (n % c == 0 && n > 1) {
            if (k >= spyfft::GEN_MAXFAC) return false;
            radix[k++] = c;
            n /= c;
        }
    }
    *nfac = k;
    return n == 1;
}

template <int LOG2N, int G, int OUTK, bool MEAN>
int launch_quad(const spyhip_fft_plan* p, const MtmArgs& a, unsigned grid) {
    using C = spyfft::Cfg2<LOG2N, G>;
    auto kern = spyfft::mtmfft_quad_kernel<LOG2N, G, OUTK, MEAN>;
    // (per device, cheap: set at every launch)
    SPY_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES));
    MtmArgs b = a;
    b.tapers = p->tapers_half.p;          // this kernel expects the scale / 2 folded into the window
    if (!b.tapers) { spy::set_error("fft_exec: plan without the pre-scaled taper table"); return -1; }
    hipLaunchKernelGGL(kern, dim3(grid), dim3(C::NTHREADS), C::LDS_BYTES, p->ctx->stream, b);
    SPY_HIP_CHECK(hipGetLastError());
    return 0;
}

template <int LOG2N, int G>
int launch_quad_mode(const spyhip_fft_plan* p, const MtmArgs& a, unsigned grid) {
    const bool mean = !p->keeptapers;
    const int outk = p->output == SPYHIP_OUT_FOURIER ? 2 : (p->output == SPYHIP_OUT_POW ? 0 : 1);
    switch (outk * 2 + (mean ? 1 : 0)) {
        case 0: return launch_quad<LOG2N, G, 0, false>(p, a, grid);
        case 1: return launch_quad<LOG2N, G, 0, true>(p, a, grid);
        case 2: return launch_quad<LOG2N, G, 1, false>(p, a, grid);
        case 3: return launch_quad<LOG2N, G, 1, true>(p, a, grid);
        case 4: return launch_quad<LOG2N, G, 2, false>(p, a, grid);
        default: return launch_quad<LOG2N, G, 2, true>(p, a, grid);
    }
}

template <int LOG2N, int G, int OUTK, bool MEAN>
int launch_blue(const spyhip_fft_plan* p, const MtmArgs& a, unsigned grid) {
    using C = spyfft::Cfg2<LOG2N, G>;
    auto kern = spyfft::mtmfft_blue_kernel<LOG2N, G, OUTK, MEAN>;
    // (per device, cheap: set at every launch)
    SPY_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(C::NTHREADS), C::LDS_BYTES, p->ctx->stream, a);
    SPY_HIP_CHECK(hipGetLastError());
    return 0;
}

template <int LOG2N, int G>
int launch_blue_mode(const spyhip_fft_plan* p, const MtmArgs& a, unsigned grid) {
    const bool mean = !p->keeptapers;
    const int outk = p->output == SPYHIP_OUT_FOURIER ? 2 : (p->output == SPYHIP_OUT_POW ? 0 : 1);
    switch (outk * 2 + (mean ? 1 : 0)) {
        case 0: return launch_blue<LOG2N, G, 0, false>(p, a, grid);
        case 1: return launch_blue<LOG2N, G, 0, true>(p, a, grid);
        case 2: return launch_blue<LOG2N, G, 1, false>(p, a, grid);
        case 3: return launch_blue<LOG2N, G, 1, true>(p, a, grid);
        case 4: return launch_blue<LOG2N, G, 2, false>(p, a, grid);
        default: return launch_blue<LOG2N, G, 2, true>(p, a, grid);
    }
}

// ---- long transforms (mtmfft_long.h): one instantiation per factor length
// interleave per factor length: 64 -> 64 ... 1024 -> 4 (256 threads per workgroup, length/16 threads per FFT)
template <int L>
int launch_long_stage(spyhip_ctx* ctx, const spyfft::LongArgs& a, int stage, long long items) {
    constexpr int G = 4096 >> L;
    using C = spyfft::Cfg2<L, G>;
    const long long grid = items * ((stage == 1 ? a.M1 : a.M2) / G);
    if (grid > 0x7fffffffLL) { spy::set_error("fft_exec: grid too large"); return -1; }
    if (stage == 0) {
        auto kern = spyfft::long_cols_kernel<L, G>;
        SPY_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES));
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(C::NTHREADS), C::LDS_BYTES, ctx->stream, a);
    } else if (stage == 1) {
        auto kern = spyfft::long_rows_kernel<L, G>;
        SPY_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES));
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(C::NTHREADS), C::LDS_BYTES, ctx->stream, a);
    } else {
        auto kern = spyfft::long_cols_inv_kernel<L, G>;
        SPY_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)C::LDS_BYTES));
        hipLaunchKernelGGL(kern, dim3((unsigned)grid), dim3(C::NTHREADS), C::LDS_BYTES, ctx->stream, a);
    }
    SPY_HIP_CHECK(hipGetLastError());
    return 0;
}

int launch_long(spyhip_ctx* ctx, const spyfft::LongArgs& a, int l, int stage, long long items) {
    switch (l) {
        case 6: return launch_long_stage<6>(ctx, a, stage, items);
        case 7: return launch_long_stage<7>(ctx, a, stage, items);
        case 8: return launch_long_stage<8>(ctx, a, stage, items);
        case 9: return launch_long_stage<9>(ctx, a, stage, items);
        case 10: return launch_long_stage<10>(ctx, a, stage, items);
        default: spy::set_error("fft_exec: no long-transform stage for 2^%d", l); return -1;
    }
}

template <int OUTK, bool MEAN>
int launch_long_post(spyhip_ctx* ctx, const spyfft::LongArgs& a) {
    const long long tot = (long long)a.nsegc * a.nquad * (a.m.nfft / 2 + 1);
    hipLaunchKernelGGL((spyfft::long_post_kernel<OUTK, MEAN>), dim3((unsigned)((tot + 255) / 256)), dim3(256), 0, ctx->stream, a);
    SPY_HIP_CHECK(hipGetLastError());
    return 0;
}

template <int OUTK, bool MEAN>
int launch_generic(const spyhip_fft_plan* p, const MtmArgs& a, unsigned grid) {
    auto kern = spyfft::mtmfft_generic_kernel<OUTK, MEAN>;
    SPY_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)p->lds_bytes));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(spyfft::GEN_THREADS), p->lds_bytes, p->ctx->stream, a, p->gen);
    SPY_HIP_CHECK(hipGetLastError());
    return 0;
}

// N = P M with M a sub-transform length of mtmfft_declong.h (in order of preference: cost per point of the schedule,
// then the fewest radix-P terms) and P in {2, 3, 4, 5, 6, 8}
bool declong_split(int nfft, int* P, int* M) {
    static const int subs[] = {4096, 2000, 5000, 4000, 10000, 8000};
    for (int m : subs) {
        if (nfft % m) continue;
        const int q = nfft / m;
        if (q == 2 || q == 3 || q == 4 || q == 5 || q == 6 || q == 8) { *P = q; *M = m; return true; }
    }
    return false;
}

// trial lengths beyond one workgroup's LDS in quad form whose HALF has a compile-time schedule (mtmfft_dec_{m,n}.hip)
bool half_length(int nfft) {
    return nfft == 12000 || nfft == 12288 || nfft == 15000 || nfft == 16000 || nfft == 16384 || nfft == 20000;
}

std::vector<float2> half_step_table(int nfft) {
    std::vector<float2> t((size_t)nfft / 4 + 1);
    for (int f = 0; f <= nfft / 4; ++f) {
        const double ang = -2.0 * PI * (double)f / (double)nfft;
        t[f] = make_float2((float)std::cos(ang), (float)std::sin(ang));
    }
    return t;
}

// channel quads interleaved per workgroup of the packed kernel (256 threads up to N = 4096)
int default_G(int log2n) {
    switch (log2n) {
        case 8: return 16;
        case 9: return 8;
        case 10: return 4;
        case 11: return 2;
        default: return 1;
    }
}

}  // namespace

extern "C" int spyhip_fft_plan_create(spyhip_ctx* ctx, int nsig, int nfft, int nchan, int ntaper,
                                      const double* tapers, double scale, int detrend, int demean_taper,
                                      const int32_t* freq_idx, int nfsel, int output, int keeptapers,
                                      spyhip_fft_plan** out) {
    if (!ctx || !out || !tapers) { spy::set_error("fft_plan_create: null argument"); return -1; }
    if (nsig < 1 || nfft < nsig || nchan < 1 || ntaper < 1) {
        spy::set_error("fft_plan_create: need 1 <= nsig <= nfft, nchan >= 1, ntaper >= 1 (got %d %d %d %d)",
                       nsig, nfft, nchan, ntaper);
        return -1;
    }
    if (output < SPYHIP_OUT_POW || output > SPYHIP_OUT_ABSIMAG) { spy::set_error("bad output kind %d", output); return -1; }
    if (detrend < -1 || detrend > 1) { spy::set_error("bad detrend %d", detrend); return -1; }
    const int nf = nfft / 2 + 1;
    auto* p = new spyhip_fft_plan();
    p->ctx = ctx;
    p->nsig = nsig; p->nfft = nfft; p->nchan = nchan; p->ntaper = ntaper;
    p->output = output; p->keeptapers = keeptapers ? 1 : 0;
    p->detrend = detrend; p->demean_taper = demean_taper ? 1 : 0;
    p->scale = (float)scale;
    SPY_HIP_CHECK(hipSetDevice(ctx->device));

    {
        double wmax = 0.0;
        for (int k = 0; k < ntaper; ++k) {
            double q = 0.0;
            for (int n = 0; n < nsig; ++n) q += tapers[(size_t)k * nsig + n] * tapers[(size_t)k * nsig + n];
            wmax = std::max(wmax, q);
        }
        p->wnorm = (float)(std::sqrt(wmax) * std::fabs(scale) * (1.0 + 1e-6));
    }
    std::vector<float> tf((size_t)ntaper * nsig);
    for (size_t i = 0; i < tf.size(); ++i) tf[i] = (float)tapers[i];
    if (p->tapers.upload(tf, ctx->stream)) { delete p; return -2; }
    if (spy::is_pow2((unsigned)nfft) && nfft >= 256 && nfft <= 16384) {
        for (size_t i = 0; i < tf.size(); ++i) tf[i] = (float)(tapers[i] * (0.5 * scale));
        if (p->tapers_half.upload(tf, ctx->stream)) { delete p; return -2; }
    }
    {   // the windows as the reference holds them (float64), for spyhip_fft_plan_set_precision
        std::vector<double> td(tapers, tapers + (size_t)ntaper * nsig);
        if (p->tapers64.upload(td, ctx->stream)) { delete p; return -2; }
    }

    // frequency selection -> inverse map bin -> output slot
    p->identity_freq = (freq_idx == nullptr);
    p->nfsel = freq_idx ? nfsel : nf;
    if (freq_idx) {
        std::vector<int> fpos(nf, -1);
        bool ident = (nfsel == nf);
        for (int i = 0; i < nfsel; ++i) {
            const int f = freq_idx[i];
            if (f < 0 || f >= nf) { spy::set_error("freq_idx[%d]=%d outside [0,%d)", i, f, nf); delete p; return -1; }
            if (fpos[f] >= 0) { spy::set_error("freq_idx holds duplicate bin %d", f); delete p; return -1; }
            fpos[f] = i;
            ident = ident && (f == i);
        }
        p->identity_freq = ident;
        if (!ident && p->fpos.upload(fpos, ctx->stream)) { delete p; return -2; }
    }

    const int outk = output == SPYHIP_OUT_FOURIER ? 2 : (output == SPYHIP_OUT_POW ? 0 : 1);
    char mode[32];
    std::snprintf(mode, sizeof mode, "%d, %s", outk, p->keeptapers ? "false" : "true");
    p->pow2 = spy::is_pow2((unsigned)nfft) && nfft >= 256 && nfft <= 16384 && !std::getenv("SPYHIP_FORCE_GENERIC");
    if (p->pow2) {
        p->log2n = spy::ilog2((unsigned)nfft);
        p->G = default_G(p->log2n);
        // complex spectra of every taper at N = 4096 are store-bound: two quads per workgroup (one workgroup per
        // CU) write 64 contiguous bytes per bin row and are 13 % faster; everything else prefers two independent
        // 256-thread workgroups per CU
        if (p->log2n == 12 && output == SPYHIP_OUT_FOURIER && p->keeptapers) p->G = 2;
        p->half = p->log2n == 14;          // 2^14: channel pairs through the 8192-point schedule (mtmfft_dec_n.hip)
        if (p->tw.upload(twiddle_table(p->half ? nfft / 2 : nfft), ctx->stream)) { delete p; return -2; }
        if (p->half && p->twh.upload(half_step_table(nfft), ctx->stream)) { delete p; return -2; }
        char buf[128];
        if (p->half) std::snprintf(buf, sizeof buf, "mtmfft_quad_kernel<13, 1, %s, HALF of N = %d>", mode, nfft);
        else std::snprintf(buf, sizeof buf, "mtmfft_quad_kernel<%d, %d, %s>", p->log2n, p->G, mode);
        p->kernel_name = buf;
    } else if ((nfft == 400 || nfft == 800 || nfft == 1200 || nfft == 1600 || nfft == 2400 || nfft == 3200 || nfft == 4800 || nfft == 8000 ||
                nfft == 100 || nfft == 200 || nfft == 500 || nfft == 1000 || nfft == 2000 || nfft == 2500 || nfft == 4000 || nfft == 5000 || nfft == 10000 ||
                nfft == 300 || nfft == 600 || nfft == 1500 || nfft == 3000 || nfft == 6000 || nfft == 7500 || nfft == 768 || nfft == 1536 || nfft == 3072 ||
                nfft == 6144) && !std::getenv("SPYHIP_FORCE_GENERIC")) {
        // decimal trial lengths (1 kHz x 0.2 ... 5 s): radix schedules fixed at compile time, 10 values per thread
        p->dec = true;
        // HALF form where it measured faster than the quad form (tools/half_probe.py): 5000 (88 KB of LDS per quad: one
        // workgroup per CU; pairs 12.8 vs 16.0 us/trial at 256 channels) and 10000 with the taper mean (split exchanges
        // in quad form: 39.3 vs 43.9, complex 42.6 vs 61.7; with every taper kept the 8-byte stores of a pair cost more)
        p->half = nfft == 5000 || (nfft == 10000 && !p->keeptapers);
        if (p->tw.upload(twiddle_table(p->half ? nfft / 2 : nfft), ctx->stream)) { delete p; return -2; }
        if (p->half && p->twh.upload(half_step_table(nfft), ctx->stream)) { delete p; return -2; }
        char buf[128];
        std::snprintf(buf, sizeof buf, p->half ? "mtmfft_dec_kernel<HALF of N = %d, %s>" : "mtmfft_dec_kernel<N = %d, %s>", nfft, mode);
        p->kernel_name = buf;
    } else if (!std::getenv("SPYHIP_FORCE_GENERIC") &&
               // (5-smooth lengths take the mixed-radix engine whatever the taper count: the chirp-z kernel's two length-M
               // transforms and three pointwise products leave ~4x the float32 error of a direct transform)
               spyfft::mix_schedule(nfft, (nchan + 3) / 4, &p->mix, &p->mix_threads, &p->lds_bytes) &&
               p->lds_bytes <= ctx->lds_per_block) {
        // 5-smooth lengths (2000, 3000, 5000, 500 ...): the packed mixed-radix engine
        p->mixed = true;
        p->G = 1 << p->mix.lg;
        if (p->tw.upload(twiddle_table(nfft), ctx->stream)) { delete p; return -2; }
        std::string sched;
        for (int i = 0; i < p->mix.npass; ++i) sched += (i ? "x" : "") + std::to_string(p->mix.radix[i]);
        char buf[160];
        std::snprintf(buf, sizeof buf, "mtmfft_mixed_kernel<%s> N=%d (%s) %d threads x %d quads", mode, nfft, sched.c_str(),
                      p->mix.th, p->G);
        p->kernel_name = buf;
    } else if (nfft >= 2 && 2 * nfft - 1 <= 8192 && !std::getenv("SPYHIP_FORCE_GENERIC")) {
        // Bluestein on the packed power-of-two engine: M = 2^log2n >= 2 nfft - 1 (at least 256)
        int M = 256;
        while (M < 2 * nfft - 1) M <<= 1;
        p->blue = true;
        p->log2n = spy::ilog2((unsigned)M);
        p->G = default_G(p->log2n);
        std::vector<float2> chirp(nfft);
        std::vector<double> br(M, 0.0), bi(M, 0.0);
        for (long long n = 0; n < nfft; ++n) {
            const long long m = (n * n) % (2LL * nfft);  // exact phase reduction
            const double ang = PI * (double)m / (double)nfft;
            chirp[n] = make_float2((float)std::cos(ang), (float)-std::sin(ang));
            br[n] = std::cos(ang);
            bi[n] = std::sin(ang);
            if (n > 0) { br[M - n] = br[n]; bi[M - n] = bi[n]; }
        }
        spy::fft_host(br, bi);
        std::vector<float2> bhat(M);
        for (int i = 0; i < M; ++i) bhat[i] = make_float2((float)(br[i] / M), (float)(bi[i] / M));
        if (p->chirp.upload(chirp, ctx->stream) || p->bhat.upload(bhat, ctx->stream) ||
            p->tw.upload(twiddle_table(M), ctx->stream)) { delete p; return -2; }
        char buf[128];
        std::snprintf(buf, sizeof buf, "mtmfft_blue_kernel<%d, %d, %s>", p->log2n, p->G, mode);
        p->kernel_name = buf;
    } else if (nfft > 10240 && !std::getenv("SPYHIP_FORCE_GENERIC") &&
               declong_split(nfft, &p->dl_P, &p->dl_M)) {
        // longer than one workgroup's LDS, N = P M with M a scheduled length: decimation in time through HBM
        std::vector<double> ws((size_t)2 * ntaper);
        const double mid = 0.5 * (nsig - 1);
        for (int k = 0; k < ntaper; ++k) {
            double s0 = 0.0, s1 = 0.0;
            for (int n = 0; n < nsig; ++n) { const double w = tf[(size_t)k * nsig + n]; s0 += w; s1 += w * (n - mid); }
            ws[2 * k] = s0;
            ws[2 * k + 1] = s1;
        }
        if (p->tw1.upload(twiddle_table(p->dl_M), ctx->stream) || p->tw2.upload(twiddle_table(p->dl_P), ctx->stream) ||
            p->twM.upload(twiddle_table(nfft), ctx->stream) || p->wsum.upload(ws, ctx->stream)) { delete p; return -2; }
        char buf[128];
        std::snprintf(buf, sizeof buf, "declong<%d x %d, %s>", p->dl_P, p->dl_M, mode);
        p->kernel_name = buf;
        if (half_length(nfft)) {
            // ... but up to 20480 samples a channel PAIR still fits one workgroup's LDS: the real transform through the
            // schedule of nfft / 2 (CfgD::HALF).  The tables above stay for the reference-precision twin (declong64).
            p->half = true;
            if (p->tw.upload(twiddle_table(nfft / 2), ctx->stream) || p->twh.upload(half_step_table(nfft), ctx->stream)) { delete p; return -2; }
            std::snprintf(buf, sizeof buf, "mtmfft_dec_kernel<HALF of N = %d, %s>", nfft, mode);
            p->kernel_name = buf;
        }
    } else if (nfft <= (1 << 19) && !std::getenv("SPYHIP_FORCE_GENERIC") &&
               !(nfft <= 10240 && [&] { int r[spyfft::GEN_MAXFAC], nf2 = 0; return factorize(nfft, r, &nf2); }())) {
        // (lengths up to 10240 with prime factors <= 13 stay on the mixed-radix LDS kernel below: measured 10-20 %
        // faster than the HBM round trips of this path; everything longer, and awkward lengths, come here)
        // Bluestein with four-step transforms through HBM: M = 2^m >= 2 nfft - 1 (>= 4096), M1 = 2^ceil(m/2), M2 = M / M1
        int m = 12;
        p->long_direct = spy::is_pow2((unsigned)nfft) && nfft >= 4096;
        while ((1LL << m) < (p->long_direct ? (long long)nfft : 2LL * nfft - 1)) ++m;
        const int M = 1 << m;
        p->longp = true;
        p->l1 = (m + 1) / 2;
        p->l2 = m / 2;
        const int M1 = 1 << p->l1, M2 = 1 << p->l2;
        std::vector<float2> chirp(nfft);
        std::vector<double> br(M, 0.0), bi(M, 0.0);
        for (long long n = 0; n < nfft; ++n) {
            const long long q = (n * n) % (2LL * nfft);  // exact phase reduction
            const double ang = PI * (double)q / (double)nfft;
            chirp[n] = make_float2((float)std::cos(ang), (float)-std::sin(ang));
            br[n] = std::cos(ang);
            bi[n] = std::sin(ang);
            if (n > 0) { br[M - n] = br[n]; bi[M - n] = bi[n]; }
        }
        spy::fft_host(br, bi);
        std::vector<float2> bhat((size_t)M);          // [k1][k2] order, 1/M folded in
        for (int k1 = 0; k1 < M1; ++k1)
            for (int k2 = 0; k2 < M2; ++k2) {
                const size_t k = (size_t)k1 + (size_t)M1 * k2;
                bhat[(size_t)k1 * M2 + k2] = make_float2((float)(br[k] / M), (float)(bi[k] / M));
            }
        std::vector<double> ws((size_t)2 * ntaper);
        const double mid = 0.5 * (nsig - 1);
        for (int k = 0; k < ntaper; ++k) {
            double s0 = 0.0, s1 = 0.0;
            for (int n = 0; n < nsig; ++n) { const double w = tf[(size_t)k * nsig + n]; s0 += w; s1 += w * (n - mid); }
            ws[2 * k] = s0;
            ws[2 * k + 1] = s1;
        }
        if (p->chirp.upload(chirp, ctx->stream) || p->bhat.upload(bhat, ctx->stream) ||
            p->tw1.upload(twiddle_table(M1), ctx->stream) || p->tw2.upload(twiddle_table(M2), ctx->stream) ||
            p->twM.upload(twiddle_table(M), ctx->stream) || p->wsum.upload(ws, ctx->stream)) { delete p; return -2; }
        char buf[128];
        std::snprintf(buf, sizeof buf, "mtmfft_long<%d x %d, %s>", M1, M2, mode);
        p->kernel_name = buf;
    } else {
        GenPlan& g = p->gen;
        g.nfft = nfft;
        g.bluestein = 0;
        g.n = nfft;
        if (nfft < 16) { spy::set_error("nfft=%d too short (need >= 16)", nfft); delete p; return -1; }
        if (!factorize(nfft, g.radix, &g.nfac)) {
            // Bluestein: circular convolution of length M = pow2 >= 2*nfft-1
            int M = 16;
            while (M < 2 * nfft - 1) M <<= 1;
            g.bluestein = 1;
            g.n = M;
            factorize(M, g.radix, &g.nfac);
            std::vector<float2> chirp(nfft);
            std::vector<double> br(M, 0.0), bi(M, 0.0);
            for (long long n = 0; n < nfft; ++n) {
                const long long m = (n * n) % (2LL * nfft);  // exact phase reduction
                const double ang = PI * (double)m / (double)nfft;
                chirp[n] = make_float2((float)std::cos(ang), (float)-std::sin(ang));
                br[n] = std::cos(ang);
                bi[n] = std::sin(ang);
                if (n > 0) { br[M - n] = br[n]; bi[M - n] = bi[n]; }
            }
            spy::fft_host(br, bi);
            std::vector<float2> bhat(M);
            for (int i = 0; i < M; ++i) bhat[i] = make_float2((float)(br[i] / M), (float)(bi[i] / M));
            if (p->chirp.upload(chirp, ctx->stream) || p->bhat.upload(bhat, ctx->stream)) { delete p; return -2; }
            g.chirp = p->chirp.p;
            g.bhat = p->bhat.p;
        }
        if (p->tw.upload(twiddle_table(g.n), ctx->stream)) { delete p; return -2; }
        const size_t work = (size_t)2 * g.n * sizeof(float2);
        const size_t staged = work + (size_t)nsig * sizeof(float2);
        g.stage_x = staged <= ctx->lds_per_block ? 1 : 0;
        p->lds_bytes = g.stage_x ? staged : work;
        if (p->lds_bytes > ctx->lds_per_block) {
            spy::set_error("nfft=%d needs %zu bytes of LDS (> %zu): unsupported length", nfft, p->lds_bytes, ctx->lds_per_block);
            delete p;
            return -3;
        }
        char buf[128];
        std::snprintf(buf, sizeof buf, "mtmfft_generic_kernel<%s>", mode);
        p->kernel_name = buf;
    }
    *out = p;
    return 0;
}

extern "C" int spyhip_fft_plan_destroy(spyhip_fft_plan* p) {
    delete p;
    return 0;
}

extern "C" int spyhip_fft_plan_set_blocked(spyhip_fft_plan* p, int on) {
    if (!p) { spy::set_error("fft_plan_set_blocked: null plan"); return -1; }
    if (on && p->precision64) { spy::set_error("fft_plan_set_blocked: not with the reference-precision kernel"); return -3; }
    if (on && !(p->pow2 && p->log2n <= 13 && p->output == SPYHIP_OUT_FOURIER && p->keeptapers)) {
        spy::set_error("fft_plan_set_blocked: the channel-blocked layout needs output=FOURIER, keeptapers=1 and a "
                       "power-of-two nfft in 256..8192");
        return -3;
    }
    p->blocked = on != 0;
    return 0;
}

extern "C" int spyhip_fft_plan_set_absmax(spyhip_fft_plan* p, float* absmax_d) {
    if (!p) { spy::set_error("fft_plan_set_absmax: null plan"); return -1; }
    if (!absmax_d) { p->absmax = nullptr; return 0; }
    // the packed power-of-two kernel (mtmfft2_kernel.h) bounds its spectra from the samples it holds; every other family
    // (other lengths, float64 transforms) gets a pass over the segments ahead of the transform (seg_range_kernel)
    const bool ok = !p->blocked && p->output == SPYHIP_OUT_FOURIER && p->keeptapers;
    if (!ok) {                 // a documented answer ("not tracked"), not a failure: the error string stays as it is
        p->absmax = nullptr;
        return -3;
    }
    p->absmax = reinterpret_cast<unsigned*>(absmax_d);
    return 0;
}

extern "C" int spyhip_fft_plan_set_precision(spyhip_fft_plan* p, int reference) {
    if (!p) { spy::set_error("fft_plan_set_precision: null plan"); return -1; }
    if (!reference) {
        if (p->precision64) p->kernel_name = p->fp32_kernel_name;
        p->precision64 = false;
        return 0;
    }
    if (p->blocked) {
        spy::set_error("fft_plan_set_precision: the reference-precision kernels write the standard layout");
        return -3;
    }
    SPY_HIP_CHECK(hipSetDevice(p->ctx->device));
    // compile-time radix schedules (mtmfft_dec64_launch.h): the powers of two 256 ... 16384 and the decimal lengths
    static const int dec64_lengths[] = {256, 512, 1024, 2048, 4096, 8192, 16384, 200, 500, 1000, 2000, 2500, 4000, 5000, 10000,
                                        600, 1500, 3000, 6000, 7500, 768, 1536, 3072, 6144,
                                        100, 400, 800, 1600, 3200, 8000, 300, 1200, 2400, 4800};
    p->f64_dec = false;
    for (int n : dec64_lengths) p->f64_dec = p->f64_dec || (n == p->nfft);
    p->f64_half = half_length(p->nfft);
    if (p->f64_half) p->f64_dec = false;
    p->f64_dl = !p->f64_dec && !p->f64_half && p->dl_P > 0;
    p->f64_any = !p->f64_dec && !p->f64_dl && !p->f64_half;
    if (p->f64_half && !p->tw64h.p) {
        const int nh = p->nfft / 2;
        std::vector<double2> t(nh);
        for (int m = 0; m < nh; ++m) {
            const double ang = -2.0 * PI * (double)m / (double)nh;
            t[m] = make_double2(std::cos(ang), std::sin(ang));
        }
        if (p->tw64h.upload(t, p->ctx->stream)) return -2;
    }
    if (p->f64_dl && !p->tw64_sub.p) {
        auto table = [](int n) {
            std::vector<double2> t(n);
            for (int m = 0; m < n; ++m) {
                const double ang = -2.0 * PI * (double)m / (double)n;
                t[m] = make_double2(std::cos(ang), std::sin(ang));
            }
            return t;
        };
        if (p->tw64_sub.upload(table(p->dl_M), p->ctx->stream) || p->tw64_P.upload(table(p->dl_P), p->ctx->stream)) return -2;
    }
    p->f64_blue = 0;
    int twlen = p->nfft;
    if (p->f64_any) {
        // any other length: generic Stockham passes over work arrays in LDS / global memory; the O(R^2) pass of a prime
        // factor R is only reasonable for small R - beyond 61 the transform takes Bluestein's form on M = 2^m >= 2 nfft - 1
        if (p->nfft < 2 || p->nfft > (1 << 20)) {
            spy::set_error("fft_plan_set_precision: the reference-precision kernels serve transform lengths 2 ... 2^20 (nfft = %d)",
                           p->nfft);
            p->f64_any = false;
            return -3;
        }
        int big = 1;
        if (!spywil::plus_plan(p->nfft, &p->f64_plan)) big = 1 << 30;
        else for (int i = 0; i < p->f64_plan.nfac; ++i) big = std::max(big, p->f64_plan.radix[i]);
        if (big > 61) {
            int M = 16;
            while (M < 2 * p->nfft - 1) M <<= 1;
            spywil::plus_plan(M, &p->f64_plan);
            p->f64_blue = M;
            twlen = M;
            if (!p->chirp64.p) {
                const int nfft = p->nfft;
                std::vector<double2> chirp(nfft);
                std::vector<double> br(M, 0.0), bi(M, 0.0);
                for (long long n = 0; n < nfft; ++n) {
                    const long long q = (n * n) % (2LL * nfft);              // exact phase reduction
                    const double ang = PI * (double)q / (double)nfft;
                    chirp[n] = make_double2(std::cos(ang), -std::sin(ang));
                    br[n] = std::cos(ang);
                    bi[n] = std::sin(ang);
                    if (n > 0) { br[M - n] = br[n]; bi[M - n] = bi[n]; }
                }
                spy::fft_host(br, bi);
                std::vector<double2> bhat(M);
                for (int i = 0; i < M; ++i) bhat[i] = make_double2(br[i] / M, bi[i] / M);
                if (p->chirp64.upload(chirp, p->ctx->stream) || p->bhat64.upload(bhat, p->ctx->stream)) return -2;
            }
        }
    }
    if (!p->tw64.p) {
        std::vector<double2> t(twlen);
        for (int m = 0; m < twlen; ++m) {
            const double ang = -2.0 * PI * (double)m / (double)twlen;
            t[m] = make_double2(std::cos(ang), std::sin(ang));
        }
        if (p->tw64.upload(t, p->ctx->stream)) return -2;
    }
    if (!p->precision64) p->fp32_kernel_name = p->kernel_name;
    p->precision64 = true;
    char buf[128];
    if (p->f64_half)
        std::snprintf(buf, sizeof buf, "mtmfft_dec64_kernel<HALF of N = %d, %d, %s>", p->nfft,
                      p->output == SPYHIP_OUT_FOURIER ? 2 : (p->output == SPYHIP_OUT_POW ? 0 : 1), p->keeptapers ? "false" : "true");
    else if (p->f64_dl)
        std::snprintf(buf, sizeof buf, "declong64_kernel<%d x %d, %d, %s>", p->dl_P, p->dl_M,
                      p->output == SPYHIP_OUT_FOURIER ? 2 : (p->output == SPYHIP_OUT_POW ? 0 : 1), p->keeptapers ? "false" : "true");
    else if (p->f64_dec)
        std::snprintf(buf, sizeof buf, "mtmfft_dec64_kernel<N = %d, %d, %s>", p->nfft,
                      p->output == SPYHIP_OUT_FOURIER ? 2 : (p->output == SPYHIP_OUT_POW ? 0 : 1), p->keeptapers ? "false" : "true");
    else if (p->f64_blue)
        std::snprintf(buf, sizeof buf, "mtmfft_f64_any_kernel<%d, %s> N=%d (Bluestein, M = %d)",
                      p->output == SPYHIP_OUT_FOURIER ? 2 : (p->output == SPYHIP_OUT_POW ? 0 : 1), p->keeptapers ? "false" : "true",
                      p->nfft, p->f64_blue);
    else
        std::snprintf(buf, sizeof buf, "mtmfft_f64_any_kernel<%d, %s> N=%d",
                      p->output == SPYHIP_OUT_FOURIER ? 2 : (p->output == SPYHIP_OUT_POW ? 0 : 1), p->keeptapers ? "false" : "true",
                      p->nfft);
    p->kernel_name = buf;
    return 0;
}

extern "C" int spyhip_fft_plan_set_reference_mean(spyhip_fft_plan* p, int on) {
    if (!p) { spy::set_error("fft_plan_set_reference_mean: null plan"); return -1; }
    p->ref_mean = on == 1;
    p->seg_f64 = on == 2;
    return 0;
}

extern "C" const char* spyhip_fft_plan_kernel_name(const spyhip_fft_plan* p) {
    return p ? p->kernel_name.c_str() : "";
}

extern "C" int spyhip_fft_exec(spyhip_fft_plan* p, const float* data_d, int64_t ld, const int32_t* chan_idx_d,
                               const int64_t* seg_start_d, const int64_t* seg_lo_d, const int64_t* seg_hi_d,
                               int nseg, void* out_d) {
    if (!p || !data_d || !seg_start_d || !seg_lo_d || !seg_hi_d || !out_d) { spy::set_error("fft_exec: null argument"); return -1; }
    if (nseg <= 0) return 0;
    SPY_HIP_CHECK(hipSetDevice(p->ctx->device));
    MtmArgs a{};
    a.data = data_d; a.ld = ld; a.chan_idx = chan_idx_d;
    a.seg_start = reinterpret_cast<const long long*>(seg_start_d);
    a.seg_lo = reinterpret_cast<const long long*>(seg_lo_d);
    a.seg_hi = reinterpret_cast<const long long*>(seg_hi_d);
    a.nseg = nseg; a.nsig = p->nsig; a.nchan = p->nchan; a.ntaper = p->ntaper;
    a.tapers = p->tapers.p; a.tw = p->tw.p; a.scale = p->scale;
    a.detrend = p->detrend; a.demean_taper = p->demean_taper;
    a.fpos = p->identity_freq ? nullptr : p->fpos.p;
    a.nfsel = p->nfsel; a.out_kind = p->output; a.out = out_d;
    a.blocked = p->blocked ? 1 : 0;
    a.means = nullptr;
    a.seg_f64 = p->seg_f64 ? 1 : 0;
    a.absmax = (p->absmax && !p->blocked) ? p->absmax : nullptr;
    a.wnorm = p->wnorm;
    if (a.absmax && !(p->pow2 && p->log2n <= 13 && !p->precision64)) {
        // every family but the packed power-of-two kernel (which bounds its spectra from the samples it holds): a pass
        // over the segments ahead of the transform
        const int bt = std::min(256, ((p->nchan + 63) / 64) * 64);
        const int ncb = (p->nchan + bt - 1) / bt;
        for (int s0 = 0; s0 < nseg; s0 += 65535) {
            MtmArgs m = a;
            m.seg_start += s0; m.seg_lo += s0; m.seg_hi += s0;
            hipLaunchKernelGGL(spyfft::seg_range_kernel, dim3(ncb, std::min(65535, nseg - s0)), dim3(bt), 0, p->ctx->stream, m);
        }
        SPY_HIP_CHECK(hipGetLastError());
        a.absmax = nullptr;                      // (the transform kernels of these families do not look at it)
    }
    if (p->ref_mean && p->detrend == 0) {
        // the per-channel means of every segment in the reference's summation order, ahead of the transform
        const size_t need = (size_t)nseg * p->nchan;
        if (need > p->means_cap) {
            if (p->means.p) { SPY_HIP_CHECK(hipStreamSynchronize(p->ctx->stream)); (void)hipFree(p->means.p); p->means.p = nullptr; }
            if (p->means.alloc(need)) return -2;
            p->means_cap = need;
        }
        const int bt = std::min(256, ((p->nchan + 63) / 64) * 64);      // threads per workgroup: whole waves, up to four
        const int ncb = (p->nchan + bt - 1) / bt;
        for (int s0 = 0; s0 < nseg; s0 += 65535) {
            MtmArgs m = a;
            m.seg_start += s0; m.seg_lo += s0; m.seg_hi += s0;
            const int ns = std::min(65535, nseg - s0);
            hipLaunchKernelGGL(spyfft::seq_mean_kernel, dim3(ncb, ns), dim3(bt), 0, p->ctx->stream, m,
                               p->means.p + (size_t)s0 * p->nchan);
        }
        SPY_HIP_CHECK(hipGetLastError());
        a.means = p->means.p;
    }
    const int npairs = (p->nchan + 1) / 2;
    if (p->precision64) {
        spyfft::F64Args fa{};
        fa.m = a;
        fa.tapers64 = p->tapers64.p;
        fa.tw64 = p->tw64.p;
        fa.scale64 = (double)p->scale;
        const long long grid = (long long)nseg * npairs;
        const int outk64 = p->output == SPYHIP_OUT_FOURIER ? 2 : (p->output == SPYHIP_OUT_POW ? 0 : 1);
        if (p->f64_half) {
            fa.tw64 = p->tw64h.p;
            fa.tw64_full = p->tw64.p;
            int rc;
            if ((rc = spyfft::dec64_launch_half_a(p->ctx->stream, fa, p->nfft, p->nchan, outk64, !p->keeptapers)) != -100) return rc;
            if ((rc = spyfft::dec64_launch_half_b(p->ctx->stream, fa, p->nfft, p->nchan, outk64, !p->keeptapers)) != -100) return rc;
            spy::set_error("fft_exec: no half-length reference-precision schedule for nfft = %d", p->nfft);
            return -1;
        }
        if (p->f64_dl) {
            // N = P M through HBM (mtmfft_declong64.h); trend and post-taper mean from the float64 sums of long_stats_kernel.
            // (float64 segments of padded sliding windows, seg_f64, are treated as float32 trials here: the nuance is one
            // rounding of the trend and such windows are not this long)
            spyfft::Long64Args L{};
            a.nfft = p->nfft;
            L.m = a;
            L.tapers64 = p->tapers64.p;
            L.twM = p->tw64_sub.p; L.twN = p->tw64.p; L.twP = p->tw64_P.p;
            L.wsum = p->wsum.p;
            L.npair = npairs;
            const size_t N = (size_t)p->nfft;
            const size_t nstat = (size_t)nseg * p->nchan * (2 + p->ntaper);
            const int nz = p->demean_taper ? p->ntaper + 1 : 1;
            if (nstat > p->stats_cap) {
                if (p->stats.p) { (void)hipFree(p->stats.p); p->stats.p = nullptr; }
                if (p->stats_part.p) { (void)hipFree(p->stats_part.p); p->stats_part.p = nullptr; }
                if (p->stats.alloc(nstat) ||
                    p->stats_part.alloc((size_t)nseg * (p->ntaper + 1) * spyfft::LONG_SPLITS * p->nchan * 2)) return -2;
                p->stats_cap = nstat;
            }
            L.stats = p->stats.p;
            if ((p->detrend >= 0 && !(p->detrend == 0 && a.means)) || p->demean_taper) {
                if (nseg > 65535 || nz * spyfft::LONG_SPLITS > 65535) { spy::set_error("fft_exec: too many segments / tapers per call"); return -1; }
                hipLaunchKernelGGL(spyfft::long_stats_kernel, dim3((p->nchan + 63) / 64, nseg, nz * spyfft::LONG_SPLITS),
                                   dim3(256), 0, p->ctx->stream, a, p->stats_part.p, nz);
                hipLaunchKernelGGL(spyfft::long_stats_final_kernel, dim3((unsigned)(((size_t)nseg * p->nchan + 255) / 256)), dim3(256),
                                   0, p->ctx->stream, a, p->stats_part.p, nz, p->stats.p);
                SPY_HIP_CHECK(hipGetLastError());
            }
            const size_t per_seg = (size_t)npairs * p->ntaper * N;           // double2 elements
            size_t chunk = std::max<size_t>(1, std::min<size_t>((size_t)nseg, (((size_t)2 << 30) / sizeof(double2)) / std::max<size_t>(per_seg, 1)));
            if (chunk * per_seg > p->scratch64_cap) {
                if (p->scratch64.p) { (void)hipFree(p->scratch64.p); p->scratch64.p = nullptr; }
                if (p->scratch64.alloc(chunk * per_seg)) return -2;
                p->scratch64_cap = chunk * per_seg;
            }
            L.scratch = p->scratch64.p;
            const int G = spyfft::declong64_group(p->dl_M);
            const long long ngrp = (npairs + G - 1) / G;
            for (int s0 = 0; s0 < nseg; s0 += (int)chunk) {
                L.seg0 = s0;
                L.nsegc = std::min<int>((int)chunk, nseg - s0);
                const long long nblocks = (long long)L.nsegc * p->dl_P * ngrp;
                int rc = spyfft::declong64_launch_sub_a(p->ctx->stream, L, p->dl_M, p->dl_P, nblocks);
                if (rc == -100) rc = spyfft::declong64_launch_sub_b(p->ctx->stream, L, p->dl_M, p->dl_P, nblocks);
                if (rc == -100) { spy::set_error("fft_exec: no float64 sub-transform of length %d", p->dl_M); rc = -1; }
                if (!rc) rc = spyfft::declong64_launch_post(p->ctx->stream, L, p->dl_P, p->dl_M, outk64, !p->keeptapers);
                if (rc) return rc;
            }
            return 0;
        }
        if (p->f64_dec) {
            int rc;
            if ((rc = spyfft::dec64_launch_a(p->ctx->stream, fa, p->nfft, npairs, outk64, !p->keeptapers)) != -100) return rc;
            if ((rc = spyfft::dec64_launch_b(p->ctx->stream, fa, p->nfft, npairs, outk64, !p->keeptapers)) != -100) return rc;
            if ((rc = spyfft::dec64_launch_c(p->ctx->stream, fa, p->nfft, npairs, outk64, !p->keeptapers)) != -100) return rc;
            if ((rc = spyfft::dec64_launch_d(p->ctx->stream, fa, p->nfft, npairs, outk64, !p->keeptapers)) != -100) return rc;
            if ((rc = spyfft::dec64_launch_e(p->ctx->stream, fa, p->nfft, npairs, outk64, !p->keeptapers)) != -100) return rc;
            if ((rc = spyfft::dec64_launch_f(p->ctx->stream, fa, p->nfft, npairs, outk64, !p->keeptapers)) != -100) return rc;
            if ((rc = spyfft::dec64_launch_g(p->ctx->stream, fa, p->nfft, npairs, outk64, !p->keeptapers)) != -100) return rc;
            if ((rc = spyfft::dec64_launch_h(p->ctx->stream, fa, p->nfft, npairs, outk64, !p->keeptapers)) != -100) return rc;
            if ((rc = spyfft::dec64_launch_i(p->ctx->stream, fa, p->nfft, npairs, outk64, !p->keeptapers)) != -100) return rc;
            if ((rc = spyfft::dec64_launch_j(p->ctx->stream, fa, p->nfft, npairs, outk64, !p->keeptapers)) != -100) return rc;
            if ((rc = spyfft::dec64_launch_k(p->ctx->stream, fa, p->nfft, npairs, outk64, !p->keeptapers)) != -100) return rc;
            if ((rc = spyfft::dec64_launch_l(p->ctx->stream, fa, p->nfft, npairs, outk64, !p->keeptapers)) != -100) return rc;
            if ((rc = spyfft::dec64_launch_m(p->ctx->stream, fa, p->nfft, npairs, outk64, !p->keeptapers)) != -100) return rc;
            if ((rc = spyfft::dec64_launch_n(p->ctx->stream, fa, p->nfft, npairs, outk64, !p->keeptapers)) != -100) return rc;
            spy::set_error("fft_exec: no reference-precision schedule for nfft = %d", p->nfft);
            return -1;
        }
        {   // f64_any (spyhip_fft_plan_set_precision: the complement of f64_half / f64_dl / f64_dec)
            // two complex128 work arrays (length nfft, or the Bluestein length M) per workgroup: in LDS while they fit
            // (leaving room for the static reduction scratch), else in global memory, launches of at most 1 GiB of them
            const size_t wlen = p->f64_blue ? (size_t)p->f64_blue : (size_t)p->nfft;
            const size_t per = (size_t)2 * wlen * sizeof(double2);
            fa.plan = p->f64_plan;
            fa.blue_n = p->f64_blue ? p->nfft : 0;
            fa.chirp64 = p->chirp64.p;
            fa.bhat64 = p->bhat64.p;
            if (per + 1024 <= (size_t)p->ctx->lds_per_block) {
                fa.work = nullptr;
                if (grid > 0x7fffffffLL) { spy::set_error("fft_exec: grid too large (%lld blocks)", grid); return -1; }
                return spyfft::f64_any_launch(p->ctx->stream, fa, grid, grid,
                                              p->output == SPYHIP_OUT_FOURIER ? 2 : (p->output == SPYHIP_OUT_POW ? 0 : 1), !p->keeptapers);
            }
            long long chunk = std::max<long long>(p->ctx->num_cu, ((size_t)1 << 30) / per);
            if (chunk > grid) chunk = grid;
            if (chunk > p->f64_chunk) {
                if (p->f64_work.p) { SPY_HIP_CHECK(hipStreamSynchronize(p->ctx->stream)); (void)hipFree(p->f64_work.p); p->f64_work.p = nullptr; }
                if (p->f64_work.alloc((size_t)chunk * 2 * wlen)) return -2;
                p->f64_chunk = chunk;
            }
            fa.work = p->f64_work.p;
            return spyfft::f64_any_launch(p->ctx->stream, fa, grid, p->f64_chunk,
                                          p->output == SPYHIP_OUT_FOURIER ? 2 : (p->output == SPYHIP_OUT_POW ? 0 : 1), !p->keeptapers);
        }
    }
    if (p->half) {
        const bool mean = !p->keeptapers;
        const int outk = p->output == SPYHIP_OUT_FOURIER ? 2 : (p->output == SPYHIP_OUT_POW ? 0 : 1);
        a.twh = p->twh.p;
        int rc;
        if (p->nfft > 10240) {
            // long trials: a pair workgroup's 8 bytes per row come from a pair-major copy of the segments (pair_stage_kernel)
            // instead of one L2 request per row and lane; launches of at most 4 GiB of it.  256 ch x 7 tapers incl. the
            // copy: 12000 51.2 -> 46.4, 16384 49.8 -> 46.6, 20000 102.1 -> 94.9 us/trial (what is left per segment is
            // what a quad workgroup of the same engine pays as well)
            const long long xstride = ((long long)p->nsig + 1) & ~1LL;
            const size_t per_seg = (size_t)npairs * (size_t)xstride;                     // float2 elements
            const int chunk = (int)std::max<size_t>(1, std::min<size_t>((size_t)nseg, (((size_t)4 << 30) / sizeof(float2)) / per_seg));
            if ((size_t)chunk * per_seg > p->xpair_cap) {
                if (p->xpair.p) { SPY_HIP_CHECK(hipStreamSynchronize(p->ctx->stream)); (void)hipFree(p->xpair.p); p->xpair.p = nullptr; }
                if (p->xpair.alloc((size_t)chunk * per_seg)) return -2;
                p->xpair_cap = (size_t)chunk * per_seg;
            }
            const size_t oelem = (size_t)(mean ? 1 : p->ntaper) * p->nfsel * p->nchan * (outk == 2 ? 8 : 4);
            for (int s0 = 0; s0 < nseg; s0 += chunk) {
                MtmArgs m = a;
                m.seg_start += s0; m.seg_lo += s0; m.seg_hi += s0;
                m.nseg = std::min(chunk, nseg - s0);
                m.out = reinterpret_cast<char*>(a.out) + (size_t)s0 * oelem;
                if (m.means) m.means += (size_t)s0 * p->nchan;
                for (int z0 = 0; z0 < m.nseg; z0 += 65535) {
                    MtmArgs z = m;
                    z.seg_start += z0; z.seg_lo += z0; z.seg_hi += z0;
                    const int nz = std::min(65535, m.nseg - z0);
                    hipLaunchKernelGGL(spyfft::pair_stage_kernel, dim3((unsigned)((xstride + 63) / 64), (p->nchan + 63) / 64, nz), dim3(256), 0,
                                       p->ctx->stream, z, p->xpair.p + (size_t)z0 * per_seg, xstride, npairs);
                }
                SPY_HIP_CHECK(hipGetLastError());
                m.xpair = p->xpair.p;
                m.xstride = xstride;
                m.chan_idx = nullptr;                    // (the copy is in selected-channel order already)
                if (p->nfft == 16384) {
                    m.tapers = p->tapers_half.p;
                    rc = spyfft::quad_half_launch(p->ctx->stream, m, npairs, outk, mean);
                } else {
                    rc = spyfft::dec_launch_half_a(p->ctx->stream, m, p->nfft, npairs, outk, mean);
                    if (rc == -100) rc = spyfft::dec_launch_half_b(p->ctx->stream, m, p->nfft, npairs, outk, mean);
                    if (rc == -100) { spy::set_error("fft_exec: no half-length schedule for nfft = %d", p->nfft); rc = -1; }
                }
                if (rc) return rc;
            }
            return 0;
        }
        // (up to 10240 samples: 5000 and 10000 in HALF form, rows gathered straight from the trial queue)
        if ((rc = spyfft::dec_launch_half_a(p->ctx->stream, a, p->nfft, npairs, outk, mean)) != -100) return rc;
        if ((rc = spyfft::dec_launch_half_b(p->ctx->stream, a, p->nfft, npairs, outk, mean)) != -100) return rc;
        if ((rc = spyfft::dec_launch_half_c(p->ctx->stream, a, p->nfft, npairs, outk, mean)) != -100) return rc;
        spy::set_error("fft_exec: no half-length schedule for nfft = %d", p->nfft);
        return -1;
    }
    if (p->pow2) {
        // work items per segment: channel quads (2^14 went above: channel pairs through the 8192-point schedule)
        const int G = p->G;
        const int nitem = (p->nchan + 3) / 4;
        a.npg = (nitem + G - 1) / G;
        int S = 8 / G; if (S < 1) S = 1; if (S > a.npg) S = a.npg;   // workgroups sharing 128-byte rows
        a.S = S;
        a.ncl = (a.npg + S - 1) / S;
        const long long nclusters = (long long)nseg * a.ncl;
        const long long grid = ((nclusters + 7) / 8) * S * 8;
        if (grid > 0x7fffffffLL) { spy::set_error("fft_exec: grid too large (%lld blocks)", grid); return -1; }
        const unsigned g = (unsigned)grid;
        switch (p->log2n) {
            case 8: return launch_quad_mode<8, 16>(p, a, g);
            case 9: return launch_quad_mode<9, 8>(p, a, g);
            case 10: return launch_quad_mode<10, 4>(p, a, g);
            case 11: return launch_quad_mode<11, 2>(p, a, g);
            case 12: return G == 2 ? launch_quad<12, 2, 2, false>(p, a, g) : launch_quad_mode<12, 1>(p, a, g);
            case 13: return launch_quad_mode<13, 1>(p, a, g);
            default: spy::set_error("no kernel for log2n=%d", p->log2n); return -1;
        }
    }
    if (p->dec) {
        const int nquads = (p->nchan + 3) / 4;
        const bool mean = !p->keeptapers;
        const int outk = p->output == SPYHIP_OUT_FOURIER ? 2 : (p->output == SPYHIP_OUT_POW ? 0 : 1);
        if (p->nfft == 2000 && outk == 2 && !mean && nquads >= 2) return spyfft::dec_launch_c2(p->ctx->stream, a, nquads);
        int rc;
        if ((rc = spyfft::dec_launch_h(p->ctx->stream, a, p->nfft, nquads, outk, mean)) != -100) return rc;
        if ((rc = spyfft::dec_launch_a(p->ctx->stream, a, p->nfft, nquads, outk, mean)) != -100) return rc;
        if ((rc = spyfft::dec_launch_b(p->ctx->stream, a, p->nfft, nquads, outk, mean)) != -100) return rc;
        if ((rc = spyfft::dec_launch_c(p->ctx->stream, a, p->nfft, nquads, outk, mean)) != -100) return rc;
        if ((rc = spyfft::dec_launch_d(p->ctx->stream, a, p->nfft, nquads, outk, mean)) != -100) return rc;
        if ((rc = spyfft::dec_launch_e(p->ctx->stream, a, p->nfft, nquads, outk, mean)) != -100) return rc;
        if ((rc = spyfft::dec_launch_f(p->ctx->stream, a, p->nfft, nquads, outk, mean)) != -100) return rc;
        if ((rc = spyfft::dec_launch_g(p->ctx->stream, a, p->nfft, nquads, outk, mean)) != -100) return rc;
        if ((rc = spyfft::dec_launch_i(p->ctx->stream, a, p->nfft, nquads, outk, mean)) != -100) return rc;
        if ((rc = spyfft::dec_launch_j(p->ctx->stream, a, p->nfft, nquads, outk, mean)) != -100) return rc;
        if ((rc = spyfft::dec_launch_k(p->ctx->stream, a, p->nfft, nquads, outk, mean)) != -100) return rc;
        if ((rc = spyfft::dec_launch_l(p->ctx->stream, a, p->nfft, nquads, outk, mean)) != -100) return rc;
        spy::set_error("fft_exec: no decimal-length kernel for nfft = %d", p->nfft);
        return -1;
    }
    if (p->mixed) {
        const int G = p->G;
        const int nitem = (p->nchan + 3) / 4;
        a.npg = (nitem + G - 1) / G;
        int S = 8 / G; if (S < 1) S = 1; if (S > a.npg) S = a.npg;
        a.S = S;
        a.ncl = (a.npg + S - 1) / S;
        const long long nclusters = (long long)nseg * a.ncl;
        const long long grid = ((nclusters + 7) / 8) * S * 8;
        if (grid > 0x7fffffffLL) { spy::set_error("fft_exec: grid too large (%lld blocks)", grid); return -1; }
        const bool mean = !p->keeptapers;
        const int outk = p->output == SPYHIP_OUT_FOURIER ? 2 : (p->output == SPYHIP_OUT_POW ? 0 : 1);
        return spyfft::mixed_launch(p->ctx->stream, a, p->mix, p->mix_threads, p->lds_bytes, (unsigned)grid, outk, mean);
    }
    if (p->dl_P) {
        spyfft::LongArgs L{};
        a.nfft = p->nfft;
        L.m = a;
        L.M1 = p->nfft; L.M2 = 1;                  // long_post_kernel: natural-order spectrum of length N
        L.tw1 = p->tw1.p; L.tw2 = p->tw2.p; L.twM = p->twM.p;
        L.wsum = p->wsum.p;
        L.direct = 0;
        L.nquad = (p->nchan + 3) / 4;
        const size_t N = (size_t)p->nfft;
        const size_t nstat = (size_t)nseg * p->nchan * (2 + p->ntaper);
        const int nz = p->demean_taper ? p->ntaper + 1 : 1;
        if (nstat > p->stats_cap) {
            if (p->stats.p) { (void)hipFree(p->stats.p); p->stats.p = nullptr; }
            if (p->stats_part.p) { (void)hipFree(p->stats_part.p); p->stats_part.p = nullptr; }
            if (p->stats.alloc(nstat) ||
                p->stats_part.alloc((size_t)nseg * (p->ntaper + 1) * spyfft::LONG_SPLITS * p->nchan * 2)) return -2;
            p->stats_cap = nstat;
        }
        L.stats = p->stats.p;
        // (constant detrending with the reference-order means of seq_mean_kernel needs no sums of its own)
        if ((p->detrend >= 0 && !(p->detrend == 0 && a.means)) || p->demean_taper) {
            if (nseg > 65535 || nz * spyfft::LONG_SPLITS > 65535) { spy::set_error("fft_exec: too many segments / tapers per call"); return -1; }
            hipLaunchKernelGGL(spyfft::long_stats_kernel, dim3((p->nchan + 63) / 64, nseg, nz * spyfft::LONG_SPLITS),
                               dim3(256), 0, p->ctx->stream, a, p->stats_part.p, nz);
            hipLaunchKernelGGL(spyfft::long_stats_final_kernel, dim3((unsigned)(((size_t)nseg * p->nchan + 255) / 256)), dim3(256),
                               0, p->ctx->stream, a, p->stats_part.p, nz, p->stats.p);
            SPY_HIP_CHECK(hipGetLastError());
        }
        const size_t per_seg = (size_t)L.nquad * p->ntaper * N;          // float4 elements
        size_t chunk = std::max<size_t>(1, std::min<size_t>((size_t)nseg, (((size_t)2 << 30) / sizeof(float4)) / std::max<size_t>(per_seg, 1)));
        if (chunk * per_seg > p->scratch_cap) {
            if (p->scratch.p) { (void)hipFree(p->scratch.p); p->scratch.p = nullptr; }
            if (p->scratch.alloc(chunk * per_seg)) return -2;
            p->scratch_cap = chunk * per_seg;
        }
        L.scratch = p->scratch.p;
        const bool mean = !p->keeptapers;
        const int outk = p->output == SPYHIP_OUT_FOURIER ? 2 : (p->output == SPYHIP_OUT_POW ? 0 : 1);
        const int G = spyfft::declong_group(p->dl_M);
        const long long ngrp = (L.nquad + G - 1) / G;
        for (int s0 = 0; s0 < nseg; s0 += (int)chunk) {
            L.seg0 = s0;
            L.nsegc = std::min<int>((int)chunk, nseg - s0);
            const long long nblocks = (long long)L.nsegc * p->dl_P * ngrp;
            int rc = spyfft::declong_launch_sub_a(p->ctx->stream, L, p->dl_M, p->dl_P, nblocks);
            if (rc == -100) rc = spyfft::declong_launch_sub_b(p->ctx->stream, L, p->dl_M, p->dl_P, nblocks);
            if (rc == -100) { spy::set_error("fft_exec: no sub-transform of length %d", p->dl_M); rc = -1; }
            if (!rc) rc = spyfft::declong_launch_post(p->ctx->stream, L, p->dl_P, p->dl_M, outk, mean);
            if (rc) return rc;
        }
        return 0;
    }
    if (p->longp) {
        spyfft::LongArgs L{};
        a.nfft = p->nfft;
        L.m = a;
        L.M1 = 1 << p->l1; L.M2 = 1 << p->l2;
        L.tw1 = p->tw1.p; L.tw2 = p->tw2.p; L.twM = p->twM.p; L.chirp = p->chirp.p; L.bhat = p->bhat.p;
        L.wsum = p->wsum.p;
        L.direct = p->long_direct ? 1 : 0;
        L.nquad = (p->nchan + 3) / 4;
        const size_t M = (size_t)L.M1 * L.M2;
        const size_t nstat = (size_t)nseg * p->nchan * (2 + p->ntaper);
        const int nz = p->demean_taper ? p->ntaper + 1 : 1;
        if (nstat > p->stats_cap) {
            if (p->stats.p) { (void)hipFree(p->stats.p); p->stats.p = nullptr; }
            if (p->stats_part.p) { (void)hipFree(p->stats_part.p); p->stats_part.p = nullptr; }
            if (p->stats.alloc(nstat) ||
                p->stats_part.alloc((size_t)nseg * (p->ntaper + 1) * spyfft::LONG_SPLITS * p->nchan * 2)) return -2;
            p->stats_cap = nstat;
        }
        L.stats = p->stats.p;
        // (constant detrending with the reference-order means of seq_mean_kernel needs no sums of its own)
        if ((p->detrend >= 0 && !(p->detrend == 0 && a.means)) || p->demean_taper) {
            if (nseg > 65535 || nz * spyfft::LONG_SPLITS > 65535) { spy::set_error("fft_exec: too many segments / tapers per call"); return -1; }
            hipLaunchKernelGGL(spyfft::long_stats_kernel, dim3((p->nchan + 63) / 64, nseg, nz * spyfft::LONG_SPLITS),
                               dim3(256), 0, p->ctx->stream, a, p->stats_part.p, nz);
            hipLaunchKernelGGL(spyfft::long_stats_final_kernel, dim3((unsigned)(((size_t)nseg * p->nchan + 255) / 256)), dim3(256),
                               0, p->ctx->stream, a, p->stats_part.p, nz, p->stats.p);
            SPY_HIP_CHECK(hipGetLastError());
        }
        // segments per chunk: scratch of ~2 GiB (at least one segment)
        const size_t per_seg = (size_t)L.nquad * p->ntaper * M;          // float4 elements
        size_t chunk = std::max<size_t>(1, std::min<size_t>((size_t)nseg, (((size_t)2 << 30) / sizeof(float4)) / std::max<size_t>(per_seg, 1)));
        if (chunk * per_seg > p->scratch_cap) {
            if (p->scratch.p) { (void)hipFree(p->scratch.p); p->scratch.p = nullptr; }
            if (p->scratch.alloc(chunk * per_seg)) return -2;
            p->scratch_cap = chunk * per_seg;
        }
        L.scratch = p->scratch.p;
        const bool mean = !p->keeptapers;
        const int outk = p->output == SPYHIP_OUT_FOURIER ? 2 : (p->output == SPYHIP_OUT_POW ? 0 : 1);
        for (int s0 = 0; s0 < nseg; s0 += (int)chunk) {
            L.seg0 = s0;
            L.nsegc = std::min<int>((int)chunk, nseg - s0);
            const long long items = (long long)L.nsegc * L.nquad * p->ntaper;
            int rc = launch_long(p->ctx, L, p->l1, 0, items);
            if (!rc) rc = launch_long(p->ctx, L, p->l2, 1, items);
            if (!rc && !L.direct) rc = launch_long(p->ctx, L, p->l1, 2, items);
            if (rc) return rc;
            switch (outk * 2 + (mean ? 1 : 0)) {
                case 0: rc = launch_long_post<0, false>(p->ctx, L); break;
                case 1: rc = launch_long_post<0, true>(p->ctx, L); break;
                case 2: rc = launch_long_post<1, false>(p->ctx, L); break;
                case 3: rc = launch_long_post<1, true>(p->ctx, L); break;
                case 4: rc = launch_long_post<2, false>(p->ctx, L); break;
                default: rc = launch_long_post<2, true>(p->ctx, L); break;
            }
            if (rc) return rc;
        }
        return 0;
    }
    if (p->blue) {
        a.nfft = p->nfft; a.chirp = p->chirp.p; a.bhat = p->bhat.p;
        const int G = p->G;
        const int nitem = (p->nchan + 3) / 4;
        a.npg = (nitem + G - 1) / G;
        int S = 8 / G; if (S < 1) S = 1; if (S > a.npg) S = a.npg;
        a.S = S;
        a.ncl = (a.npg + S - 1) / S;
        const long long nclusters = (long long)nseg * a.ncl;
        const long long grid = ((nclusters + 7) / 8) * S * 8;
        if (grid > 0x7fffffffLL) { spy::set_error("fft_exec: grid too large (%lld blocks)", grid); return -1; }
        const unsigned g = (unsigned)grid;
        switch (p->log2n) {
            case 8: return launch_blue_mode<8, 16>(p, a, g);
            case 9: return launch_blue_mode<9, 8>(p, a, g);
            case 10: return launch_blue_mode<10, 4>(p, a, g);
            case 11: return launch_blue_mode<11, 2>(p, a, g);
            case 12: return launch_blue_mode<12, 1>(p, a, g);
            case 13: return launch_blue_mode<13, 1>(p, a, g);
            default: spy::set_error("no Bluestein kernel for M=2^%d", p->log2n); return -1;
        }
    }
    const long long grid = (long long)nseg * npairs;
    if (grid > 0x7fffffffLL) { spy::set_error("fft_exec: grid too large (%lld blocks)", grid); return -1; }
    const bool mean = !p->keeptapers;
    const int outk = p->output == SPYHIP_OUT_FOURIER ? 2 : (p->output == SPYHIP_OUT_POW ? 0 : 1);
    switch (outk * 2 + (mean ? 1 : 0)) {
        case 0: return launch_generic<0, false>(p, a, (unsigned)grid);
        case 1: return launch_generic<0, true>(p, a, (unsigned)grid);
        case 2: return launch_generic<1, false>(p, a, (unsigned)grid);
        case 3: return launch_generic<1, true>(p, a, (unsigned)grid);
        case 4: return launch_generic<2, false>(p, a, (unsigned)grid);
        default: return launch_generic<2, true>(p, a, (unsigned)grid);
    }
}
