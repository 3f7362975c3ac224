// CPU emulation driver for the HIP kernels (TEST INFRASTRUCTURE ONLY, see hip_emu.h).
// Compiles the kernel headers of syncopy_amd/csrc unchanged and runs them
// workgroup by workgroup on OS threads.  Built by tests/emu/build_emu.py into
// tests/emu/_build/libspyemu.so; only tests/ load it.
#include "hip_emu.h"
#include <algorithm>
using std::max;
using std::min;

namespace emu {
thread_local dim3 t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;
thread_local BlockCtx* t_ctx = nullptr;
}  // namespace emu

#include "../../include/spyhip.h"
#include "../../syncopy_amd/csrc/mtmfft_kernel.h"
#include "../../syncopy_amd/csrc/mtmfft_generic.h"
#include "../../syncopy_amd/csrc/csd_kernel.h"
#include "../../syncopy_amd/csrc/csd3m_kernel.h"
#include "../../syncopy_amd/csrc/ppc_kernel.h"
#include "../../syncopy_amd/csrc/ccov_kernel.h"
#include "../../syncopy_amd/csrc/jack_kernel.h"
#include "../../syncopy_amd/csrc/mtmfft2_kernel.h"
#include "../../syncopy_amd/csrc/mtmfft_dec_kernel.h"
#include "../../syncopy_amd/csrc/mtmfft_blue_kernel.h"
#include "../../syncopy_amd/csrc/mtmfft_mixed.h"
#include "../../syncopy_amd/csrc/mtmfft_long.h"
#include "../../syncopy_amd/csrc/mtmfft_declong.h"
#include "../../syncopy_amd/csrc/cwt_kernel.h"
#include "../../syncopy_amd/csrc/granger_kernels.h"
#include "../../syncopy_amd/csrc/wilson_plus_kernel.h"
#include "../../syncopy_amd/csrc/mtmfft_dec64_cfg.h"

namespace spy {
void set_error(const char*, ...) {}
}  // namespace spy

using spyfft::GenPlan;
using spyfft::MtmArgs;

namespace {

template <int LOG2N, int G, int OUTK, bool MEAN>
void run_quad(const MtmArgs& a, unsigned grid, long only_block) {
    using C = spyfft::Cfg2<LOG2N, G>;
    emu::launch(dim3(grid), dim3(C::NTHREADS), C::LDS_BYTES,
                [&] { spyfft::mtmfft_quad_kernel<LOG2N, G, OUTK, MEAN>(a); }, only_block);
}

template <int LOG2N, int G>
void run_quad_mode(const MtmArgs& a, unsigned grid, int outk, int mean, long only_block) {
    switch (outk * 2 + mean) {
        case 0: run_quad<LOG2N, G, 0, false>(a, grid, only_block); break;
        case 1: run_quad<LOG2N, G, 0, true>(a, grid, only_block); break;
        case 2: run_quad<LOG2N, G, 1, false>(a, grid, only_block); break;
        case 3: run_quad<LOG2N, G, 1, true>(a, grid, only_block); break;
        case 4: run_quad<LOG2N, G, 2, false>(a, grid, only_block); break;
        default: run_quad<LOG2N, G, 2, true>(a, grid, only_block); break;
    }
}

template <int LOG2N, int G>
void run_blue_mode(const MtmArgs& a, unsigned grid, int outk, int mean) {
    using C = spyfft::Cfg2<LOG2N, G>;
    auto go = [&](auto fn) { emu::launch(dim3(grid), dim3(C::NTHREADS), C::LDS_BYTES, fn); };
    switch (outk * 2 + mean) {
        case 0: go([&] { spyfft::mtmfft_blue_kernel<LOG2N, G, 0, false>(a); }); break;
        case 1: go([&] { spyfft::mtmfft_blue_kernel<LOG2N, G, 0, true>(a); }); break;
        case 2: go([&] { spyfft::mtmfft_blue_kernel<LOG2N, G, 1, false>(a); }); break;
        case 3: go([&] { spyfft::mtmfft_blue_kernel<LOG2N, G, 1, true>(a); }); break;
        case 4: go([&] { spyfft::mtmfft_blue_kernel<LOG2N, G, 2, false>(a); }); break;
        default: go([&] { spyfft::mtmfft_blue_kernel<LOG2N, G, 2, true>(a); }); break;
    }
}

template <int L>
void run_long_stage(const spyfft::LongArgs& a, int stage, long long items) {
    constexpr int G = 4096 >> L;
    using C = spyfft::Cfg2<L, G>;
    const unsigned grid = (unsigned)(items * ((stage == 1 ? a.M1 : a.M2) / G));
    if (stage == 0) emu::launch(dim3(grid), dim3(C::NTHREADS), C::LDS_BYTES, [&] { spyfft::long_cols_kernel<L, G>(a); });
    else if (stage == 1) emu::launch(dim3(grid), dim3(C::NTHREADS), C::LDS_BYTES, [&] { spyfft::long_rows_kernel<L, G>(a); });
    else emu::launch(dim3(grid), dim3(C::NTHREADS), C::LDS_BYTES, [&] { spyfft::long_cols_inv_kernel<L, G>(a); });
}
int run_long(const spyfft::LongArgs& a, int l, int stage, long long items) {
    switch (l) {
        case 6: run_long_stage<6>(a, stage, items); return 0;
        case 7: run_long_stage<7>(a, stage, items); return 0;
        case 8: run_long_stage<8>(a, stage, items); return 0;
        default: return -1;          // longer factors are exercised on the GPU only (emulation time)
    }
}

}  // namespace

// mtmfft_dec_kernel (compile-time radix schedules): id = 1000 / 2000 / 5000 (V = 10), 2001 (20 x 10 x 10), 4096 / 512 (V = 8)
template <class Cf>
static void run_dec_mode(const MtmArgs& a0, int nseg, int nchan, int outk, int mean) {
    MtmArgs a = a0;
    const int G = Cf::G;
    const int nitem = Cf::HALF ? (nchan + 1) / 2 : (nchan + 3) / 4;      // channel pairs / quads (as mtmfft_dec_launch.h)
    a.npg = (nitem + G - 1) / G;
    int S = (Cf::HALF ? 16 : 8) / G; if (S < 1) S = 1; if (S > a.npg) S = a.npg;
    a.S = S;
    a.ncl = (a.npg + S - 1) / S;
    const long long nclusters = (long long)nseg * a.ncl;
    const unsigned grid = (unsigned)(((nclusters + 7) / 8) * S * 8);
    auto go = [&](auto fn) { emu::launch(dim3(grid), dim3(Cf::NTHREADS), Cf::LDS_BYTES, fn); };
    switch (outk * 2 + mean) {
        case 0: go([&] { spyfft::mtmfft_dec_kernel<Cf, 0, false>(a); }); break;
        case 1: go([&] { spyfft::mtmfft_dec_kernel<Cf, 0, true>(a); }); break;
        case 2: go([&] { spyfft::mtmfft_dec_kernel<Cf, 1, false>(a); }); break;
        case 3: go([&] { spyfft::mtmfft_dec_kernel<Cf, 1, true>(a); }); break;
        case 4: go([&] { spyfft::mtmfft_dec_kernel<Cf, 2, false>(a); }); break;
        default: go([&] { spyfft::mtmfft_dec_kernel<Cf, 2, true>(a); }); break;
    }
}

static int g_blocked = 0;   // hand-over layout toggle shared by the FFT and CSD entry points
static int g_force_4m = 0;  // SPYHIP_CSD_4M: 256 channels on the 4-multiplication kernel
static const float* g_means = nullptr;   // (nseg x nchan) reference-order means for the next FFT call, or none
static const float* g_twh = nullptr;     // exp(-2 pi i f / nfft), f <= nfft / 4: the table of the HALF-form schedules

template <int LOG2N, int G>
static void emu_launch_ccov(const spyfft::CcovArgs& a) {
    using C = spyfft::Cfg2<LOG2N, G>;
    const long long grid = 8 * (((a.npairs + 2 * G - 1) / (2 * G) + 7) / 8);
    emu::launch(dim3((unsigned)grid), dim3(C::NTHREADS), C::LDS_BYTES, [&] { spyfft::ccov_lags_kernel<LOG2N, G>(a); });
}

// plus4_kernel (power-of-two lag-domain lengths 256 .. 4096); returns 0, or 1 if there is no such kernel for F
template <int LOG2L>
void run_plus4(const double* g, int F, int n, const double* tw, double* gp, double* g0) {
    using C = spywil::PCfg<LOG2L>;
    emu::launch(dim3((unsigned)spywil::plus4_grid((long long)n * n)), dim3(C::T), C::LDS_BYTES, [&] {
        spywil::plus4_kernel<LOG2L>(reinterpret_cast<const spywil::cd*>(g), F, (long long)n * n, reinterpret_cast<const spywil::cd*>(tw),
                                    reinterpret_cast<spywil::cd*>(gp), reinterpret_cast<spywil::cd*>(g0)); });
}

// K1L2 (mtmfft_declong.h): N = P M through scratch memory - statistics, scheduled sub-transforms of the decimated samples,
// radix-P step + separation + conversion.  (P, M) = (6, 2000), (3, 4096), (4, 5000), (2, 10000)
template <class C, int P>
static void run_declong(spyfft::LongArgs& L, int outk, bool mean) {
    constexpr int M = C::N, G = C::G;
    const int ngrp = (L.nquad + G - 1) / G;
    emu::launch(dim3((unsigned)((long long)L.nsegc * P * ngrp)), dim3(C::NTHREADS), C::LDS_BYTES,
                [&] { spyfft::declong_sub_kernel<C>(L, P); });
    const unsigned grid = (unsigned)(((long long)L.nsegc * L.nquad * (M / 2 + 1) + 255) / 256);
    switch (outk * 2 + (mean ? 1 : 0)) {
        case 0: emu::launch(dim3(grid), dim3(256), 0, [&] { spyfft::declong_post_kernel<P, 0, false>(L, M); }); break;
        case 1: emu::launch(dim3(grid), dim3(256), 0, [&] { spyfft::declong_post_kernel<P, 0, true>(L, M); }); break;
        case 2: emu::launch(dim3(grid), dim3(256), 0, [&] { spyfft::declong_post_kernel<P, 1, false>(L, M); }); break;
        case 3: emu::launch(dim3(grid), dim3(256), 0, [&] { spyfft::declong_post_kernel<P, 1, true>(L, M); }); break;
        case 4: emu::launch(dim3(grid), dim3(256), 0, [&] { spyfft::declong_post_kernel<P, 2, false>(L, M); }); break;
        default: emu::launch(dim3(grid), dim3(256), 0, [&] { spyfft::declong_post_kernel<P, 2, true>(L, M); }); break;
    }
}

extern "C" {

void emu_set_blocked(int on) { g_blocked = on; }
void emu_set_force_4m(int on) { g_force_4m = on; }
void emu_set_means(const float* m) { g_means = m; }
void emu_set_twh(const float* t) { g_twh = t; }

// spyfft::seq_mean_kernel as spyhip_fft_exec launches it (plan option spyhip_fft_plan_set_reference_mean)
void emu_seq_mean(const float* data, long long ld, const int* chan_idx, const long long* seg_start,
                  const long long* seg_lo, const long long* seg_hi, int nseg, int nsig, int nchan, float* means) {
    MtmArgs a{};
    a.data = data; a.ld = ld; a.chan_idx = chan_idx;
    a.seg_start = seg_start; a.seg_lo = seg_lo; a.seg_hi = seg_hi;
    a.nseg = nseg; a.nsig = nsig; a.nchan = nchan;
    emu::launch(dim3((nchan + 63) / 64, nseg), dim3(64), 0, [&] { spyfft::seq_mean_kernel(a, means); }, -1);
}

// Mirrors the argument marshalling of spyhip_fft_exec for the power-of-two kernels
// (packed quad kernel up to 2^13, pair kernel for 2^14).
// All pointers are host pointers.  Returns 0, or -1 for an unsupported (log2n, G).
int emu_mtmfft_pow2(int log2n, int G, const float* data, long long ld, const int* chan_idx,
                    const long long* seg_start, const long long* seg_lo, const long long* seg_hi, int nseg,
                    int nsig, int nchan, int ntaper, const float* tapers, const float* tw, float scale,
                    int detrend, int demean_taper, const int* fpos, int nfsel, int out_kind, int keeptapers,
                    void* out) {
    MtmArgs a{};
    a.data = data; a.ld = ld; a.chan_idx = chan_idx;
    a.seg_start = seg_start; a.seg_lo = seg_lo; a.seg_hi = seg_hi;
    a.nseg = nseg; a.nsig = nsig; a.nchan = nchan; a.ntaper = ntaper;
    a.tapers = tapers; a.tw = reinterpret_cast<const float2*>(tw); a.scale = scale;
    a.detrend = detrend; a.demean_taper = demean_taper; a.fpos = fpos; a.nfsel = nfsel;
    a.out_kind = out_kind; a.out = out;
    a.means = g_means;
    a.blocked = g_blocked;
    const bool quad = log2n <= 13;
    // mtmfft_quad_kernel expects the window times scale / 2 (the plan uploads that table, mtmfft.hip)
    std::vector<float> th;
    {
        th.resize((size_t)ntaper * nsig);
        for (size_t i = 0; i < th.size(); ++i) th[i] = (float)((double)tapers[i] * (0.5 * (double)scale));
        a.tapers = th.data();
    }
    const int nitem = quad ? (nchan + 3) / 4 : (nchan + 1) / 2;
    a.npg = (nitem + G - 1) / G;
    int S = (quad ? 8 : 16) / G; if (S < 1) S = 1; if (S > a.npg) S = a.npg;
    a.S = S;
    a.ncl = (a.npg + S - 1) / S;
    const long long nclusters = (long long)nseg * a.ncl;
    const unsigned grid = (unsigned)(((nclusters + 7) / 8) * S * 8);
    const int outk = out_kind == SPYHIP_OUT_FOURIER ? 2 : (out_kind == SPYHIP_OUT_POW ? 0 : 1);
    const int mean = keeptapers ? 0 : 1;
    if (log2n == 14) {
        // 2^14: channel pairs through the 8192-point engine (HALF form; `tw` belongs to 8192, the half-step table was set)
        a.twh = reinterpret_cast<const float2*>(g_twh);
        using C = spyfft::Cfg2<13, 1>;
        auto go = [&](auto fn) { emu::launch(dim3(grid), dim3(C::NTHREADS), C::LDS_BYTES, fn); };
        switch (outk * 2 + mean) {
            case 0: go([&] { spyfft::mtmfft_quad_kernel<13, 1, 0, false, true>(a); }); break;
            case 1: go([&] { spyfft::mtmfft_quad_kernel<13, 1, 0, true, true>(a); }); break;
            case 2: go([&] { spyfft::mtmfft_quad_kernel<13, 1, 1, false, true>(a); }); break;
            case 3: go([&] { spyfft::mtmfft_quad_kernel<13, 1, 1, true, true>(a); }); break;
            case 4: go([&] { spyfft::mtmfft_quad_kernel<13, 1, 2, false, true>(a); }); break;
            default: go([&] { spyfft::mtmfft_quad_kernel<13, 1, 2, true, true>(a); }); break;
        }
        return 0;
    }
    switch (log2n * 100 + G) {
        case 816: run_quad_mode<8, 16>(a, grid, outk, mean, -1); break;
        case 908: run_quad_mode<9, 8>(a, grid, outk, mean, -1); break;
        case 1004: run_quad_mode<10, 4>(a, grid, outk, mean, -1); break;
        case 1102: run_quad_mode<11, 2>(a, grid, outk, mean, -1); break;
        case 1201: run_quad_mode<12, 1>(a, grid, outk, mean, -1); break;
        case 1202: if (outk != 2 || mean) return -1; run_quad<12, 2, 2, false>(a, grid, -1); break;
        case 1301: run_quad_mode<13, 1>(a, grid, outk, mean, -1); break;
        default: return -1;
    }
    return 0;
}

int emu_mtmfft_dec(int id, const float* data, long long ld, const int* chan_idx,
                   const long long* seg_start, const long long* seg_lo, const long long* seg_hi, int nseg,
                   int nsig, int nchan, int ntaper, const float* tapers, const float* tw, float scale,
                   int detrend, int demean_taper, const int* fpos, int nfsel, int out_kind, int keeptapers,
                   void* out) {
    MtmArgs a{};
    a.data = data; a.ld = ld; a.chan_idx = chan_idx;
    a.seg_start = seg_start; a.seg_lo = seg_lo; a.seg_hi = seg_hi;
    a.nseg = nseg; a.nsig = nsig; a.nchan = nchan; a.ntaper = ntaper;
    a.tapers = tapers; a.tw = reinterpret_cast<const float2*>(tw); a.scale = scale;
    a.detrend = detrend; a.demean_taper = demean_taper; a.fpos = fpos; a.nfsel = nfsel;
    a.out_kind = out_kind; a.out = out;
    a.means = g_means;
    a.twh = reinterpret_cast<const float2*>(g_twh);
    const int outk = out_kind == SPYHIP_OUT_FOURIER ? 2 : (out_kind == SPYHIP_OUT_POW ? 0 : 1);
    const int mean = keeptapers ? 0 : 1;
    switch (id) {
        // HALF form (id = -nfft): channel pairs, the real transform of nfft samples through the schedule of nfft / 2
        case -2000: run_dec_mode<spyfft::CfgD<10, 10, 10, 1, 2, 1, false, true>>(a, nseg, nchan, outk, mean); break;
        case -2002: run_dec_mode<spyfft::CfgD<10, 10, 10, 1, 2, 1, true, true>>(a, nseg, nchan, outk, mean); break;     // split exchanges
        case -1200: run_dec_mode<spyfft::CfgD<10, 10, 2, 1, 4, 3, false, true>>(a, nseg, nchan, outk, mean); break;     // 3 x 200
        case -5000: run_dec_mode<spyfft::CfgD<10, 10, 5, 5, 1, 1, false, true>>(a, nseg, nchan, outk, mean); break;
        case -12000: run_dec_mode<spyfft::CfgD<10, 10, 10, 2, 1, 3, false, true>>(a, nseg, nchan, outk, mean); break;
        case -1024: run_dec_mode<spyfft::CfgD<16, 16, 2, 1, 2, 1, false, true>>(a, nseg, nchan, outk, mean); break;
        case 1000: run_dec_mode<spyfft::CfgD<10, 10, 10, 1, 2>>(a, nseg, nchan, outk, mean); break;
        case 2000: run_dec_mode<spyfft::CfgD<10, 10, 10, 2, 1>>(a, nseg, nchan, outk, mean); break;
        case 2001: run_dec_mode<spyfft::CfgD<20, 10, 10, 1, 2>>(a, nseg, nchan, outk, mean); break;
        case 5000: run_dec_mode<spyfft::CfgD<10, 10, 10, 5, 1>>(a, nseg, nchan, outk, mean); break;
        case 600: run_dec_mode<spyfft::CfgD<10, 10, 2, 1, 4, 3>>(a, nseg, nchan, outk, mean); break;
        case 100: run_dec_mode<spyfft::CfgD<10, 10, 1, 1, 16>>(a, nseg, nchan, outk, mean); break;
        case 400: run_dec_mode<spyfft::CfgD<10, 10, 2, 2, 8>>(a, nseg, nchan, outk, mean); break;
        case 2400: run_dec_mode<spyfft::CfgD<20, 20, 2, 1, 1, 3>>(a, nseg, nchan, outk, mean); break;
        case 3200: run_dec_mode<spyfft::CfgD<20, 20, 4, 2, 1>>(a, nseg, nchan, outk, mean); break;
        case 300: run_dec_mode<spyfft::CfgD<10, 10, 1, 1, 8, 3>>(a, nseg, nchan, outk, mean); break;
        case 768: run_dec_mode<spyfft::CfgD<16, 16, 1, 1, 4, 3>>(a, nseg, nchan, outk, mean); break;
        case 3072: run_dec_mode<spyfft::CfgD<16, 16, 4, 1, 1, 3>>(a, nseg, nchan, outk, mean); break;
        case 10000: run_dec_mode<spyfft::CfgD<20, 20, 5, 5, 1, 1, true>>(a, nseg, nchan, outk, mean); break;
        case 1001: run_dec_mode<spyfft::CfgD<10, 10, 10, 1, 2, 1, true>>(a, nseg, nchan, outk, mean); break;
        case 1500: run_dec_mode<spyfft::CfgD<10, 10, 5, 1, 2, 3>>(a, nseg, nchan, outk, mean); break;
        case 3000: run_dec_mode<spyfft::CfgD<10, 10, 10, 1, 1, 3>>(a, nseg, nchan, outk, mean); break;
        case 6000: run_dec_mode<spyfft::CfgD<10, 10, 10, 2, 1, 3>>(a, nseg, nchan, outk, mean); break;
        case 512: run_dec_mode<spyfft::CfgD<8, 8, 8, 1, 4>>(a, nseg, nchan, outk, mean); break;
        case 4096: run_dec_mode<spyfft::CfgD<8, 8, 8, 8, 1>>(a, nseg, nchan, outk, mean); break;
        default: return -1;
    }
    return 0;
}

// Mirrors spyhip_fft_exec for the Bluestein kernel (tables built by the Python mirror of mtmfft.hip).
int emu_mtmfft_blue(int log2m, int G, int nfft, const float* chirp, const float* bhat, const float* data, long long ld,
                    const int* chan_idx, const long long* seg_start, const long long* seg_lo, const long long* seg_hi,
                    int nseg, int nsig, int nchan, int ntaper, const float* tapers, const float* tw, float scale,
                    int detrend, int demean_taper, const int* fpos, int nfsel, int out_kind, int keeptapers, void* out) {
    MtmArgs a{};
    a.data = data; a.ld = ld; a.chan_idx = chan_idx;
    a.seg_start = seg_start; a.seg_lo = seg_lo; a.seg_hi = seg_hi;
    a.nseg = nseg; a.nsig = nsig; a.nchan = nchan; a.ntaper = ntaper;
    a.tapers = tapers; a.tw = reinterpret_cast<const float2*>(tw); a.scale = scale;
    a.detrend = detrend; a.demean_taper = demean_taper; a.fpos = fpos; a.nfsel = nfsel;
    a.out_kind = out_kind; a.out = out;
    a.means = g_means;
    a.nfft = nfft; a.chirp = reinterpret_cast<const float2*>(chirp); a.bhat = reinterpret_cast<const float2*>(bhat);
    const int nitem = (nchan + 3) / 4;
    a.npg = (nitem + G - 1) / G;
    int S = 8 / G; if (S < 1) S = 1; if (S > a.npg) S = a.npg;
    a.S = S;
    a.ncl = (a.npg + S - 1) / S;
    const long long nclusters = (long long)nseg * a.ncl;
    const unsigned grid = (unsigned)(((nclusters + 7) / 8) * S * 8);
    const int outk = out_kind == SPYHIP_OUT_FOURIER ? 2 : (out_kind == SPYHIP_OUT_POW ? 0 : 1);
    const int mean = keeptapers ? 0 : 1;
    switch (log2m * 100 + G) {
        case 816: run_blue_mode<8, 16>(a, grid, outk, mean); break;
        case 908: run_blue_mode<9, 8>(a, grid, outk, mean); break;
        case 1004: run_blue_mode<10, 4>(a, grid, outk, mean); break;
        case 1102: run_blue_mode<11, 2>(a, grid, outk, mean); break;
        case 1201: run_blue_mode<12, 1>(a, grid, outk, mean); break;
        case 1301: run_blue_mode<13, 1>(a, grid, outk, mean); break;
        default: return -1;
    }
    return 0;
}

// Mirrors the long-transform branch of spyhip_fft_exec (one chunk holding every segment).
int emu_mtmfft_long(int l1, int l2, int nfft, const float* chirp, const float* bhat, const float* tw1, const float* tw2,
                    const float* twM, const double* wsum, const float* data, long long ld, const int* chan_idx,
                    const long long* seg_start, const long long* seg_lo, const long long* seg_hi, int nseg, int nsig,
                    int nchan, int ntaper, const float* tapers, float scale, int detrend, int demean_taper,
                    const int* fpos, int nfsel, int out_kind, int keeptapers, void* out) {
    MtmArgs a{};
    a.data = data; a.ld = ld; a.chan_idx = chan_idx;
    a.seg_start = seg_start; a.seg_lo = seg_lo; a.seg_hi = seg_hi;
    a.nseg = nseg; a.nsig = nsig; a.nchan = nchan; a.ntaper = ntaper;
    a.tapers = tapers; a.scale = scale; a.detrend = detrend; a.demean_taper = demean_taper;
    a.fpos = fpos; a.nfsel = nfsel; a.out_kind = out_kind; a.out = out; a.nfft = nfft;
    a.means = g_means;
    spyfft::LongArgs L{};
    L.m = a;
    L.M1 = 1 << l1; L.M2 = 1 << l2;
    L.tw1 = reinterpret_cast<const float2*>(tw1); L.tw2 = reinterpret_cast<const float2*>(tw2);
    L.twM = reinterpret_cast<const float2*>(twM); L.chirp = reinterpret_cast<const float2*>(chirp);
    L.bhat = reinterpret_cast<const float2*>(bhat); L.wsum = wsum;
    L.nquad = (nchan + 3) / 4;
    std::vector<double> stats((size_t)nseg * nchan * (2 + ntaper), 0.0);
    L.stats = stats.data();
    if (detrend >= 0 || demean_taper) {
        const int nz = demean_taper ? ntaper + 1 : 1;
        std::vector<double> part((size_t)nseg * nz * spyfft::LONG_SPLITS * nchan * 2, 0.0);
        emu::launch(dim3((nchan + 63) / 64, nseg, nz * spyfft::LONG_SPLITS), dim3(256), 0,
                    [&] { spyfft::long_stats_kernel(a, part.data(), nz); });
        emu::launch(dim3((unsigned)(((size_t)nseg * nchan + 255) / 256)), dim3(256), 0,
                    [&] { spyfft::long_stats_final_kernel(a, part.data(), nz, stats.data()); });
    }
    const size_t M = (size_t)L.M1 * L.M2;
    const long long items = (long long)nseg * L.nquad * ntaper;
    std::vector<float4> scratch((size_t)items * M);
    L.scratch = scratch.data();
    L.seg0 = 0; L.nsegc = nseg;
    if (run_long(L, l1, 0, items) || run_long(L, l2, 1, items) || run_long(L, l1, 2, items)) return -1;
    const int outk = out_kind == SPYHIP_OUT_FOURIER ? 2 : (out_kind == SPYHIP_OUT_POW ? 0 : 1);
    const bool mean = !keeptapers;
    const unsigned grid = (unsigned)(((long long)nseg * L.nquad * (nfft / 2 + 1) + 255) / 256);
    switch (outk * 2 + (mean ? 1 : 0)) {
        case 0: emu::launch(dim3(grid), dim3(256), 0, [&] { spyfft::long_post_kernel<0, false>(L); }); break;
        case 1: emu::launch(dim3(grid), dim3(256), 0, [&] { spyfft::long_post_kernel<0, true>(L); }); break;
        case 2: emu::launch(dim3(grid), dim3(256), 0, [&] { spyfft::long_post_kernel<1, false>(L); }); break;
        case 3: emu::launch(dim3(grid), dim3(256), 0, [&] { spyfft::long_post_kernel<1, true>(L); }); break;
        case 4: emu::launch(dim3(grid), dim3(256), 0, [&] { spyfft::long_post_kernel<2, false>(L); }); break;
        default: emu::launch(dim3(grid), dim3(256), 0, [&] { spyfft::long_post_kernel<2, true>(L); }); break;
    }
    return 0;
}

int emu_mtmfft_declong(int P, int M, const float* twsub, const float* twN, const float* twP, const double* wsum,
                       const float* data, long long ld, const int* chan_idx, const long long* seg_start,
                       const long long* seg_lo, const long long* seg_hi, int nseg, int nsig, int nchan, int ntaper,
                       const float* tapers, float scale, int detrend, int demean_taper, const int* fpos, int nfsel,
                       int out_kind, int keeptapers, void* out) {
    MtmArgs a{};
    a.data = data; a.ld = ld; a.chan_idx = chan_idx;
    a.seg_start = seg_start; a.seg_lo = seg_lo; a.seg_hi = seg_hi;
    a.nseg = nseg; a.nsig = nsig; a.nchan = nchan; a.ntaper = ntaper;
    a.tapers = tapers; a.scale = scale; a.detrend = detrend; a.demean_taper = demean_taper;
    a.fpos = fpos; a.nfsel = nfsel; a.out_kind = out_kind; a.out = out; a.nfft = P * M;
    a.means = g_means;
    spyfft::LongArgs L{};
    L.m = a;
    L.M1 = P * M; L.M2 = 1;
    L.tw1 = reinterpret_cast<const float2*>(twsub); L.tw2 = reinterpret_cast<const float2*>(twP);
    L.twM = reinterpret_cast<const float2*>(twN); L.wsum = wsum;
    L.nquad = (nchan + 3) / 4;
    std::vector<double> stats((size_t)nseg * nchan * (2 + ntaper), 0.0);
    L.stats = stats.data();
    if ((detrend >= 0 && !(detrend == 0 && a.means)) || demean_taper) {
        const int nz = demean_taper ? ntaper + 1 : 1;
        std::vector<double> part((size_t)nseg * nz * spyfft::LONG_SPLITS * nchan * 2, 0.0);
        emu::launch(dim3((nchan + 63) / 64, nseg, nz * spyfft::LONG_SPLITS), dim3(256), 0,
                    [&] { spyfft::long_stats_kernel(a, part.data(), nz); });
        emu::launch(dim3((unsigned)(((size_t)nseg * nchan + 255) / 256)), dim3(256), 0,
                    [&] { spyfft::long_stats_final_kernel(a, part.data(), nz, stats.data()); });
    }
    std::vector<float4> scratch((size_t)nseg * L.nquad * ntaper * (size_t)P * M);
    L.scratch = scratch.data();
    L.seg0 = 0; L.nsegc = nseg;
    const int outk = out_kind == SPYHIP_OUT_FOURIER ? 2 : (out_kind == SPYHIP_OUT_POW ? 0 : 1);
    const bool mean = !keeptapers;
    if (P == 6 && M == 2000) run_declong<spyfft::CfgD<10, 10, 10, 2, 1>, 6>(L, outk, mean);
    else if (P == 3 && M == 4096) run_declong<spyfft::CfgD<16, 16, 16, 1, 1>, 3>(L, outk, mean);
    else if (P == 4 && M == 5000) run_declong<spyfft::CfgD<10, 10, 10, 5, 1>, 4>(L, outk, mean);
    else if (P == 2 && M == 10000) run_declong<spyfft::CfgD<20, 20, 5, 5, 1, 1, true>, 2>(L, outk, mean);
    else return -1;
    return 0;
}

int emu_mtmfft_generic(int n, int nfac, const int* radix, int nfft, int bluestein, const float* chirp,
                       const float* bhat, int stage_x, const float* data, long long ld, const int* chan_idx,
                       const long long* seg_start, const long long* seg_lo, const long long* seg_hi, int nseg,
                       int nsig, int nchan, int ntaper, const float* tapers, const float* tw, float scale,
                       int detrend, int demean_taper, const int* fpos, int nfsel, int out_kind, int keeptapers,
                       void* out) {
    MtmArgs a{};
    a.data = data; a.ld = ld; a.chan_idx = chan_idx;
    a.seg_start = seg_start; a.seg_lo = seg_lo; a.seg_hi = seg_hi;
    a.nseg = nseg; a.nsig = nsig; a.nchan = nchan; a.ntaper = ntaper;
    a.tapers = tapers; a.tw = reinterpret_cast<const float2*>(tw); a.scale = scale;
    a.detrend = detrend; a.demean_taper = demean_taper; a.fpos = fpos; a.nfsel = nfsel;
    a.out_kind = out_kind; a.out = out;
    a.means = g_means;
    GenPlan g{};
    g.n = n; g.nfac = nfac; g.nfft = nfft; g.bluestein = bluestein; g.stage_x = stage_x;
    for (int i = 0; i < nfac; ++i) g.radix[i] = radix[i];
    g.chirp = reinterpret_cast<const float2*>(chirp);
    g.bhat = reinterpret_cast<const float2*>(bhat);
    const size_t lds = ((size_t)2 * n + (stage_x ? nsig : 0)) * sizeof(float2);
    const unsigned grid = (unsigned)nseg * (unsigned)((nchan + 1) / 2);
    const int outk = out_kind == SPYHIP_OUT_FOURIER ? 2 : (out_kind == SPYHIP_OUT_POW ? 0 : 1);
    const bool mean = !keeptapers;
    auto go = [&](auto fn) { emu::launch(dim3(grid), dim3(spyfft::GEN_THREADS), lds, fn); };
    switch (outk * 2 + (mean ? 1 : 0)) {
        case 0: go([&] { spyfft::mtmfft_generic_kernel<0, false>(a, g); }); break;
        case 1: go([&] { spyfft::mtmfft_generic_kernel<0, true>(a, g); }); break;
        case 2: go([&] { spyfft::mtmfft_generic_kernel<1, false>(a, g); }); break;
        case 3: go([&] { spyfft::mtmfft_generic_kernel<1, true>(a, g); }); break;
        case 4: go([&] { spyfft::mtmfft_generic_kernel<2, false>(a, g); }); break;
        default: go([&] { spyfft::mtmfft_generic_kernel<2, true>(a, g); }); break;
    }
    return 0;
}

// The packed mixed-radix engine (mtmfft_mixed.h) with the schedule spyhip_fft_plan_create computes (mix_schedule).
// Returns 1 if the length is not served by it.  force_nostage: take the re-read-per-taper path although the segment
// would fit into LDS; info (7 ints): th, G, npass, stage, threads, radix[0], radix[last].
int emu_mtmfft_mixed(int nfft, int force_nostage, int* info, const float* data, long long ld, const int* chan_idx,
                     const long long* seg_start, const long long* seg_lo, const long long* seg_hi, int nseg,
                     int nsig, int nchan, int ntaper, const float* tapers, const float* tw, float scale,
                     int detrend, int demean_taper, const int* fpos, int nfsel, int out_kind, int keeptapers,
                     void* out) {
    spyfft::MixPlan g{};
    int threads = 0;
    size_t lds = 0;
    if (!spyfft::mix_schedule(nfft, (nchan + 3) / 4, &g, &threads, &lds)) return 1;
    if (force_nostage) g.stage = 0;
    if (info) {
        info[0] = g.th; info[1] = 1 << g.lg; info[2] = g.npass; info[3] = g.stage; info[4] = threads;
        info[5] = g.radix[0]; info[6] = g.radix[g.npass - 1];
    }
    MtmArgs a{};
    a.data = data; a.ld = ld; a.chan_idx = chan_idx;
    a.seg_start = seg_start; a.seg_lo = seg_lo; a.seg_hi = seg_hi;
    a.nseg = nseg; a.nsig = nsig; a.nchan = nchan; a.ntaper = ntaper;
    a.tapers = tapers; a.tw = reinterpret_cast<const float2*>(tw); a.scale = scale;
    a.detrend = detrend; a.demean_taper = demean_taper; a.fpos = fpos; a.nfsel = nfsel;
    a.out_kind = out_kind; a.out = out;
    a.means = g_means;
    const int G = 1 << g.lg;
    const int nitem = (nchan + 3) / 4;
    a.npg = (nitem + G - 1) / G;
    int S = 8 / G; if (S < 1) S = 1; if (S > a.npg) S = a.npg;
    a.S = S;
    a.ncl = (a.npg + S - 1) / S;
    const long long nclusters = (long long)nseg * a.ncl;
    const unsigned grid = (unsigned)(((nclusters + 7) / 8) * S * 8);
    const int outk = out_kind == SPYHIP_OUT_FOURIER ? 2 : (out_kind == SPYHIP_OUT_POW ? 0 : 1);
    const bool mean = !keeptapers;
    auto go = [&](auto fn) { emu::launch(dim3(grid), dim3(threads), lds, fn); };
    switch (outk * 2 + (mean ? 1 : 0)) {
        case 0: go([&] { spyfft::mtmfft_mixed_kernel<0, false, 1024>(a, g); }); break;
        case 1: go([&] { spyfft::mtmfft_mixed_kernel<0, true, 1024>(a, g); }); break;
        case 2: go([&] { spyfft::mtmfft_mixed_kernel<1, false, 1024>(a, g); }); break;
        case 3: go([&] { spyfft::mtmfft_mixed_kernel<1, true, 1024>(a, g); }); break;
        case 4: go([&] { spyfft::mtmfft_mixed_kernel<2, false, 1024>(a, g); }); break;
        default: go([&] { spyfft::mtmfft_mixed_kernel<2, true, 1024>(a, g); }); break;
    }
    return 0;
}

// Mirrors spyhip_csd_accumulate (host logic of csd.hip) for the emulated MFMA kernel.
int emu_csd_accumulate(const float* spec, long long nrows, int F, int C, float* acc, int force_tpw) {
    spycsd::CsdArgs a{};
    a.spec = reinterpret_cast<const float2*>(spec);
    a.nrows = nrows; a.F = F; a.C = C;
    a.acc = reinterpret_cast<float2*>(acc);
    a.nt = (C + 31) / 32;
    a.ntiles = a.nt * (a.nt + 1) / 2;
    a.nitems = (long long)F * a.ntiles;
    a.cpad = a.nt * 32;
    a.blocked = g_blocked;
    // as csd.hip: even C <= 256, row-major spectra -> the instruction-lean path (force_tpw != 0 picks a generic kernel)
    const bool fast = force_tpw == 0 && !g_blocked && C <= 256;
    int ta = 1, tb = 1;
    if (fast || a.ntiles >= 21) { ta = 5; tb = 4; }
    else if (a.ntiles >= 6) { ta = 3; tb = 2; }
    if (force_tpw == 1) { ta = 1; tb = 1; }
    if (force_tpw == 3) { ta = 3; tb = 2; }
    if (force_tpw == 5) { ta = 5; tb = 4; }
    int per = 4 * (ta + tb);
    int nfb = (per + a.ntiles - 1) / a.ntiles;
    if (per % a.ntiles != 0 && a.ntiles > 1) nfb += 1;
    if (nfb > F) nfb = F;
    const size_t rowbytes = (size_t)nfb * a.cpad * sizeof(float2);
    int kb = 32;
    while (kb > 4 && (size_t)kb * rowbytes > (size_t)spycsd::CSD_THREADS * spycsd::CSD_PF * sizeof(float2)) kb -= 4;
    if (kb > nrows && !fast) kb = (int)((nrows + 3) & ~3LL);
    size_t lds = 3 * (size_t)kb * rowbytes;
    if (fast) {
        int nf = 256 / C;
        while (nf > 1 && nf * a.ntiles > 40) --nf;
        if (nf > F) nf = F;
        a.fast_per = nf * a.ntiles;
        per = a.fast_per;
        kb = 16;
        lds = 3 * (size_t)16 * 256 * sizeof(float2) + 512;
    }
    a.kb = kb;
    a.item_base = 0; a.item_end = a.nitems;
    unsigned grid = (unsigned)((a.nitems + per - 1) / per);
    const unsigned T = spycsd::CSD_THREADS;
    const int chp_w = (C + 15) & ~15;
    const bool wide3m = !g_force_4m && (chp_w == 304 || chp_w == 320 || chp_w == 384);   // (csd.hip: every count up to 512)
    if (force_tpw == 0 && !g_blocked && C > 256 && C <= 512 && !wide3m) {      // as csd.hip: the wide variant
        a.fast_nwgf = (a.ntiles + 39) / 40;
        a.fast_per = (a.ntiles + a.fast_nwgf - 1) / a.fast_nwgf;
        a.kb = 8;
        grid = (unsigned)(F * a.fast_nwgf);
        emu::launch(dim3(grid), dim3(T), 3 * (size_t)16 * 256 * sizeof(float2) + 512,
                    [&] { spycsd::csd_accum_kernel<5, 4, 3>(a); });
        return 7;
    }
    if (force_tpw == 0 && C == 256 && !g_force_4m) {
        // as csd.hip: 256 channels take the 3-multiplication kernel, one workgroup of 8 waves per frequency
        a.item_end = (long long)F * spycsd::M3_TILES_PER_F;
        emu::launch(dim3((unsigned)F), dim3(512), spycsd::M3_LDS_BYTES, [&] { spycsd::csd3m_kernel<256, 8>(a); });
        return 8;
    }
    if (force_tpw == 0 && C > 512 && !g_blocked) {
        // as csd.hip: blocks of 256 channels - Hermitian product per block (instances 16 / 256 here), rectangle per pair
        const long long nrows3 = (C & 1) ? nrows - 1 : nrows;
        const int nb = (C + 255) / 256;
        spycsd::CsdArgs b = a;
        b.nrows = nrows3;
        b.ctot = C;
        for (int I = 0; I < nb && nrows3 > 0; ++I) {
            const int nI = std::min(256, C - 256 * I);
            b.ch0 = 256 * I; b.n0 = nI; b.ch1 = 0; b.n1 = 0;
            const int chp = g_force_4m ? 256 : (nI + 15) & ~15, fpr = chp < 256 ? 256 / chp : 1;
            const long long nprow = (F + fpr - 1) / fpr;
            b.item_base = 0; b.item_end = nprow * spycsd::M3_TILES_PER_F;
            if (g_force_4m) emu::launch(dim3((unsigned)nprow), dim3(512), spycsd::M3_LDS_BYTES, [&] { spycsd::csd3m_kernel<256, 8, false, false, true>(b); });
            else if (chp == 256) emu::launch(dim3((unsigned)nprow), dim3(512), spycsd::M3_LDS_BYTES, [&] { spycsd::csd3m_kernel<256, 8, false>(b); });
            else if (chp == 16) emu::launch(dim3((unsigned)nprow), dim3(512), spycsd::M3_LDS_BYTES, [&] { spycsd::csd3m_kernel<16, 8, false>(b); });
            else return -1;
            for (int J = 0; J < I; ++J) {
                b.ch0 = 256 * J; b.n0 = 256; b.ch1 = 256 * I; b.n1 = nI;
                b.item_base = 0; b.item_end = (long long)F * spycsd::M3_TILES_PER_F;
                constexpr int NPR = spycsd::M3Tab<512, true>::NP;
                if (g_force_4m)
                    emu::launch(dim3((unsigned)(((F + 7) / 8) * 8 * NPR)), dim3(512), spycsd::M3_LDS_BYTES,
                                [&] { spycsd::csd3m_kernel<512, 8, false, true, true>(b); });
                else
                    emu::launch(dim3((unsigned)(((F + 7) / 8) * 8 * NPR)), dim3(512), spycsd::M3_LDS_BYTES,
                                [&] { spycsd::csd3m_kernel<512, 8, false, true>(b); });
            }
        }
        if (nrows3 < nrows) {                    // csd_rank1_kernel of csd.hip, on the host
            const float2* x = a.spec + (size_t)nrows3 * F * C;
            for (int f = 0; f < F; ++f)
                for (int i = 0; i < C; ++i)
                    for (int j = 0; j <= i; ++j) {
                        const float2 u = x[(size_t)f * C + i], v = x[(size_t)f * C + j];
                        float2& o = a.acc[((size_t)f * C + i) * C + j];
                        o.x += u.x * v.x + u.y * v.y;
                        o.y += u.y * v.x - u.x * v.y;
                    }
        }
        return g_force_4m ? 11 : 10;
    }
    // (the emulator instantiates a sample of the 3M channel counts; the others take the 4-multiplication kernels here)
    const int chp_emu = (C + 15) & ~15;
    const bool have3m = chp_emu == 16 || chp_emu == 32 || chp_emu == 48 || chp_emu == 64 || chp_emu == 96 || chp_emu == 128 ||
                        chp_emu == 192 || chp_emu == 240 || chp_emu == 256 || chp_emu == 304 || chp_emu == 320 || chp_emu == 384;
    if (force_tpw == 0 && C != 256 && C <= 512 && !g_blocked && !g_force_4m && have3m) {
        // as csd.hip: the 3-multiplication kernel instance of the next multiple of 16 with narrower rows padded inside
        // its LDS image; floor(256 / CHp) frequencies per workgroup, or (above 256 channels) NP workgroups per
        // frequency in XCD-aware order; odd channel counts: the last row through the 4-multiplication kernels
        const int chp = (C + 15) & ~15;
        const int fpr = chp < 256 ? 256 / chp : 1;
        const long long nprow = (F + fpr - 1) / fpr;
        const long long nrows3 = (C & 1) ? nrows - 1 : nrows;
        a.item_end = nprow * spycsd::M3_TILES_PER_F;
        a.nrows = nrows3;
        const dim3 b(512);
#define EMU_M3(CH)                                                                                                      \
    case CH: {                                                                                                          \
        const int np = spycsd::M3Tab<CH>::NP;                                                                           \
        const dim3 g((unsigned)(np == 1 ? nprow : ((nprow + 7) / 8) * 8 * np));                                          \
        if (nrows3 > 0) emu::launch(g, b, spycsd::M3_LDS_BYTES, [&] { spycsd::csd3m_kernel<CH, 8, false>(a); });          \
        break;                                                                                                          \
    }
        switch (chp) {
            EMU_M3(16) EMU_M3(32) EMU_M3(48) EMU_M3(64) EMU_M3(96) EMU_M3(128) EMU_M3(192) EMU_M3(240) EMU_M3(256) EMU_M3(304) EMU_M3(320) EMU_M3(384)
            default: return -1;            // (the emulator instantiates a sample of the channel counts)
        }
#undef EMU_M3
        if (nrows3 < nrows) {
            const int keep = g_force_4m;
            g_force_4m = 1;
            emu_csd_accumulate(spec + (size_t)nrows3 * F * C * 2, 1, F, C, acc, 0);
            g_force_4m = keep;
        }
        return 9;
    }
    if (fast && C == 256) emu::launch(dim3(grid), dim3(T), lds, [&] { spycsd::csd_accum_kernel<5, 4, 1>(a); });
    else if (fast) emu::launch(dim3(grid), dim3(T), lds, [&] { spycsd::csd_accum_kernel<5, 4, 2>(a); });
    else if (ta == 5) emu::launch(dim3(grid), dim3(T), lds, [&] { spycsd::csd_accum_kernel<5, 4>(a); });
    else if (ta == 3) emu::launch(dim3(grid), dim3(T), lds, [&] { spycsd::csd_accum_kernel<3, 2>(a); });
    else emu::launch(dim3(grid), dim3(T), lds, [&] { spycsd::csd_accum_kernel<1, 1>(a); });
    if (fast) return 6;
    return ta;
}

void emu_csd_finalize(float* acc, int F, int C, float scale) {
    const int nt = (C + 31) / 32;
    emu::launch(dim3((unsigned)(F * (nt * (nt + 1) / 2))), dim3(256), 0,
                [&] { spycsd::csd_finalize_kernel(reinterpret_cast<float2*>(acc), F, C, scale); });
}

void emu_coh_from_accumulator(const float* acc, int F, int C, float scale, int kind, void* out) {
    const int nt = (C + 31) / 32;
    const dim3 grid((unsigned)(F * (nt * (nt + 1) / 2)));
    if (kind == SPYHIP_OUT_FOURIER)
        emu::launch(grid, dim3(256), 0, [&] { spycsd::coh_from_acc_kernel<true>(reinterpret_cast<const float2*>(acc), F, C, scale, kind, out); });
    else
        emu::launch(grid, dim3(256), 0, [&] { spycsd::coh_from_acc_kernel<false>(reinterpret_cast<const float2*>(acc), F, C, scale, kind, out); });
}

void emu_coh_normalize(const float* csd, int F, int C, int kind, void* out) {
    if (kind == SPYHIP_OUT_FOURIER)
        emu::launch(dim3(4), dim3(256), 0, [&] { spycsd::coh_normalize_kernel<true>(reinterpret_cast<const float2*>(csd), F, C, kind, out); });
    else
        emu::launch(dim3(4), dim3(256), 0, [&] { spycsd::coh_normalize_kernel<false>(reinterpret_cast<const float2*>(csd), F, C, kind, out); });
}

// K9 (mirrors spyhip_jack_coh_accumulate)
void emu_jack_coh(const float* spec, int ntrials, int K, int F, int C, const float* S, const void* direct, int kind,
                  long long T, double* sum_d, double* sum_d2) {
    spycsd::JackArgs a{};
    a.spec = reinterpret_cast<const float2*>(spec);
    a.S = reinterpret_cast<const float2*>(S);
    a.direct = direct;
    a.ntrials = ntrials; a.K = K; a.F = F; a.C = C; a.kind = kind;
    a.T = (float)T;
    a.sum_d = sum_d; a.sum_d2 = sum_d2;
    const int nt = (C + 31) / 32;
    const size_t lds = 2 * (size_t)2 * K * 32 * sizeof(float2);
    const dim3 grid((unsigned)(8 * ((F + 7) / 8) * (nt * (nt + 1) / 2)));
    if (kind == SPYHIP_OUT_FOURIER) emu::launch(grid, dim3(256), lds, [&] { spycsd::jack_coh_kernel<true>(a); });
    else emu::launch(grid, dim3(256), lds, [&] { spycsd::jack_coh_kernel<false>(a); });
}

// K7 (mirrors ppc.hip)
void emu_ppc_accumulate(const float* spec, int ntrials, int ntaper, int F, int C, float* acc) {
    spyppc::PpcArgs a{};
    a.spec = reinterpret_cast<const float2*>(spec);
    a.ntrials = ntrials; a.ntaper = ntaper; a.F = F; a.C = C;
    a.acc = reinterpret_cast<float2*>(acc);
    const int nt = (C + 31) / 32;
    const size_t lds = 2 * (size_t)2 * ntaper * 32 * sizeof(float2);
    emu::launch(dim3((unsigned)(8 * ((F + 7) / 8) * (nt * (nt + 1) / 2))), dim3(256), lds, [&] { spyppc::ppc_accum_kernel(a); });
}

void emu_ppc_accumulate_csd(const float* csd, int ntrials, long long n, float* acc) {
    emu::launch(dim3((unsigned)((n + 255) / 256)), dim3(256), 0, [&] {
        spyppc::ppc_accum_csd_kernel(reinterpret_cast<const float2*>(csd), n, ntrials, reinterpret_cast<float2*>(acc));
    });
}

void emu_ppc_finalize(const float* acc, int F, int ni, int nj, int lower_only, long long T, float* out) {
    const long long n = (long long)F * ni * nj;
    emu::launch(dim3((unsigned)((n + 255) / 256)), dim3(256), 0, [&] {
        spyppc::ppc_finalize_kernel(reinterpret_cast<const float2*>(acc), F, ni, nj, lower_only, (double)T, out);
    });
}

// K8 (mirrors ccov.hip): tw = exp(-2 pi i m / L) from the caller; norm as spyhip_ccov_from_accumulator
int emu_ccov(const float* acc, const float* tw, int nfft, int nchan, int nsamples, double scale, int norm, float* out) {
    spyfft::CcovArgs a{};
    a.acc = reinterpret_cast<const float2*>(acc);
    a.tw = reinterpret_cast<const float2*>(tw);
    a.C = nchan; a.nsamples = nsamples;
    a.nlag = nsamples / 2 + (nsamples & 1);
    a.q = (nsamples & 1) ? 0 : 1;
    a.npairs = (long long)nchan * (nchan + 1) / 2;
    a.scale = (float)(scale / (double)nfft);
    a.out = out;
    switch (nfft) {
        case 1024: emu_launch_ccov<10, 4>(a); break;
        case 2048: emu_launch_ccov<11, 2>(a); break;
        case 4096: emu_launch_ccov<12, 1>(a); break;
        case 8192: emu_launch_ccov<13, 1>(a); break;
        default: return -1;
    }
    if (norm) {
        std::vector<float> d(nchan);
        const float dc = (float)(scale / ((double)nsamples * (double)nsamples));
        emu::launch(dim3((nchan + 255) / 256), dim3(256), 0,
                    [&] { spyfft::ccov_diag_kernel(out, a.acc, nchan, norm, dc, d.data()); });
        const long long n = (long long)a.nlag * nchan * nchan;
        emu::launch(dim3((unsigned)((n + 255) / 256)), dim3(256), 0,
                    [&] { spyfft::ccov_normalize_kernel(out, n, nchan, d.data()); });
    }
    return 0;
}

// CWT: plan tables (kernel spectra, shifts) are built by the Python mirror of cwt.hip.
int emu_cwt(int log2n, int G, const float* data, long long ld, const int* chan_idx, const long long* seg_start,
            const long long* trial_lo, const long long* trial_hi, int nseg, int nsig, int nchan, int nscales,
            const float* tw, const float* hspec, const int* cshift, int V, int halo, int nblocks, int detrend,
            int out_kind, const int* tpos, int ntime_out, void* out, int accumulate, int mode) {
    // mode bit 0: trial sums on PAIRS of segments (cwt2_kernel<..., PAIRT>; accumulate = 2); bit 1: the direct kernels
    // (cwt2d_kernel, 1024- / 2048-point blocks with 8 / 4 channel pairs per workgroup; accumulate 0 / 1); bit 2: the
    // channel-major input copy (cwt_stage_input_kernel) - as cwt.hip combines them
    spyfft::CwtArgs a{};
    a.data = data; a.ld = ld; a.chan_idx = chan_idx; a.seg_start = seg_start; a.trial_lo = trial_lo; a.trial_hi = trial_hi;
    a.nseg = nseg; a.nsig = nsig; a.nchan = nchan; a.nscales = nscales;
    a.tw = reinterpret_cast<const float2*>(tw); a.hspec = reinterpret_cast<const float2*>(hspec); a.cshift = cshift;
    a.V = V; a.halo = halo; a.nblocks = nblocks; a.detrend = detrend; a.out_kind = out_kind; a.tpos = tpos;
    a.ntime_out = ntime_out; a.out = out; a.accumulate = accumulate;
    std::vector<double> trend((size_t)nseg * nchan * 2, 0.0);
    a.trend = trend.data();
    if (detrend == 0) {                      // as cwt.hip: the reference-order float32 mean
        emu::launch(dim3((nchan + 63) / 64, nseg), dim3(64), 0, [&] { spyfft::cwt_mean_np_kernel(a, trend.data()); });
    } else if (detrend > 0) {
        std::vector<double> part(trend.size() * spyfft::CWT_TREND_SPLITS, 0.0);
        emu::launch(dim3((nchan + 63) / 64, spyfft::CWT_TREND_SPLITS, nseg), dim3(256), 0,
                    [&] { spyfft::cwt_trend_partial_kernel(a, part.data()); });
        emu::launch(dim3((unsigned)(((size_t)nseg * nchan + 255) / 256)), dim3(256), 0,
                    [&] { spyfft::cwt_trend_final_kernel(a, part.data(), trend.data()); });
    }
    const bool pairt = (mode & 1) != 0, direct = (mode & 2) != 0;
    std::vector<float> xt;
    if (mode & 4) {
        xt.assign((size_t)nseg * nchan * nsig, 0.f);
        emu::launch(dim3((nsig + 63) / 64, (nchan + 63) / 64, nseg), dim3(256), 0, [&] { spyfft::cwt_stage_input_kernel(a, xt.data()); });
        a.xt = xt.data();
    }
    const int outk = out_kind == SPYHIP_OUT_FOURIER ? 2 : (out_kind == SPYHIP_OUT_POW ? 0 : 1);
    if (direct) {
        std::vector<int> fl(nsig, 0);
        if (tpos) {
            int last = -1;
            for (int n = 0; n < nsig; ++n) { if (tpos[n] >= 0) last = tpos[n]; fl[n] = last > 0 ? last : 0; }
            a.tfloor = fl.data();
        }
        const unsigned dgrid = (unsigned)nseg * (unsigned)(((nchan + 1) / 2 + G - 1) / G) * (unsigned)nblocks;
#define CWT2D_CASE(L, GG) \
        if (log2n == L && G == GG) { \
            using C = spyfft::Cfg2<L, GG>; \
            if (outk == 2) emu::launch(dim3(dgrid), dim3(C::NTHREADS), C::LDS_BYTES, [&] { spyfft::cwt2d_kernel<L, GG, 2>(a); }); \
            else if (outk == 0) emu::launch(dim3(dgrid), dim3(C::NTHREADS), C::LDS_BYTES, [&] { spyfft::cwt2d_kernel<L, GG, 0>(a); }); \
            else emu::launch(dim3(dgrid), dim3(C::NTHREADS), C::LDS_BYTES, [&] { spyfft::cwt2d_kernel<L, GG, 1>(a); }); \
            return 0; \
        }
        CWT2D_CASE(10, 8) CWT2D_CASE(11, 4)
#undef CWT2D_CASE
        return -1;
    }
    const int nsets = pairt ? (nseg + 1) / 2 : nseg;              // staging row sets
    const int nunit = pairt ? nchan : (log2n <= 13 ? (nchan + 1) / 2 : nchan);      // packed kernel: channel pairs / (PAIRT) channels
    const unsigned grid = (unsigned)nsets * (unsigned)((nunit + G - 1) / G) * (unsigned)nblocks;
    // one chunk holding every segment (cwt.hip sizes chunks by memory; the kernels are the same)
    std::vector<float> stage((size_t)nsets * nscales * nchan * nsig * (outk == 2 ? 2 : 1), 0.f);
    a.stage = stage.data();
    a.seg0 = 0;
    const dim3 sgrid((nsig + 63) / 64, nscales, accumulate == 2 ? 1 : nseg);
    if (pairt) {
        if (accumulate != 2 || log2n > 13) return -1;
        spyfft::CwtArgs sc = a;
        sc.nseg = nsets;                                           // the transposition pass adds row SETS
#define CWT2P_CASE(L, GG) \
        if (log2n == L && G == GG) { \
            using C = spyfft::Cfg2<L, GG>; \
            if (outk == 2) emu::launch(dim3(grid), dim3(C::NTHREADS), C::LDS_BYTES, [&] { spyfft::cwt2_kernel<L, GG, 2, true>(a); }); \
            else if (outk == 0) emu::launch(dim3(grid), dim3(C::NTHREADS), C::LDS_BYTES, [&] { spyfft::cwt2_kernel<L, GG, 0, true>(a); }); \
            else emu::launch(dim3(grid), dim3(C::NTHREADS), C::LDS_BYTES, [&] { spyfft::cwt2_kernel<L, GG, 1, true>(a); }); \
            if (outk == 2) emu::launch(sgrid, dim3(256), 0, [&] { spyfft::cwt_scatter_kernel<float2>(sc); }); \
            else emu::launch(sgrid, dim3(256), 0, [&] { spyfft::cwt_scatter_kernel<float>(sc); }); \
            return 0; \
        }
        CWT2P_CASE(10, 4) CWT2P_CASE(11, 2) CWT2P_CASE(12, 1) CWT2P_CASE(13, 1)
#undef CWT2P_CASE
        return -1;
    }
#define CWT_CASE(L, GG) \
    if (log2n == L && G == GG) { \
        using C = spyfft::Cfg<L, GG>; \
        if (outk == 2) emu::launch(dim3(grid), dim3(C::NTHREADS), C::LDS_BYTES, [&] { spyfft::cwt_kernel<L, GG, 2>(a); }); \
        else if (outk == 0) emu::launch(dim3(grid), dim3(C::NTHREADS), C::LDS_BYTES, [&] { spyfft::cwt_kernel<L, GG, 0>(a); }); \
        else emu::launch(dim3(grid), dim3(C::NTHREADS), C::LDS_BYTES, [&] { spyfft::cwt_kernel<L, GG, 1>(a); }); \
        if (outk == 2) emu::launch(sgrid, dim3(256), 0, [&] { spyfft::cwt_scatter_kernel<float2>(a); }); \
        else emu::launch(sgrid, dim3(256), 0, [&] { spyfft::cwt_scatter_kernel<float>(a); }); \
        return 0; \
    }
    CWT_CASE(14, 1)
#undef CWT_CASE
#define CWT2_CASE(L, GG) \
    if (log2n == L && G == GG) { \
        using C = spyfft::Cfg2<L, GG>; \
        if (outk == 2) emu::launch(dim3(grid), dim3(C::NTHREADS), C::LDS_BYTES, [&] { spyfft::cwt2_kernel<L, GG, 2>(a); }); \
        else if (outk == 0) emu::launch(dim3(grid), dim3(C::NTHREADS), C::LDS_BYTES, [&] { spyfft::cwt2_kernel<L, GG, 0>(a); }); \
        else emu::launch(dim3(grid), dim3(C::NTHREADS), C::LDS_BYTES, [&] { spyfft::cwt2_kernel<L, GG, 1>(a); }); \
        if (outk == 2) emu::launch(sgrid, dim3(256), 0, [&] { spyfft::cwt_scatter_kernel<float2>(a); }); \
        else emu::launch(sgrid, dim3(256), 0, [&] { spyfft::cwt_scatter_kernel<float>(a); }); \
        return 0; \
    }
    CWT2_CASE(10, 4) CWT2_CASE(11, 2) CWT2_CASE(12, 1) CWT2_CASE(13, 1)
#undef CWT2_CASE
    return -1;
}

// ---- Wilson / Granger kernels, one entry per kernel (the Python test re-creates the host loop of granger.hip)
using spywil::cd;
void emu_w_widen(const float* in, double* out, int C, long long n, double eps) {
    emu::launch(dim3(4), dim3(256), 0, [&] { spywil::widen_kernel(reinterpret_cast<const float2*>(in), reinterpret_cast<cd*>(out), C, n, eps); });
}
void emu_w_gemm(const double* A, const double* B, double* Cm, int n, int batch, long long sA, long long sB, long long sC, int opB, int addI) {
    if (n >= 48) {      // as granger.hip: fp64 MFMA tiles
        const bool herm = opB == 1 && A == B && sA == sB;
        dim3 g(spywil::zgemm_groups(n, herm ? 3 : 0) * ((batch + 7) / 8) * 8);
        if (herm)     // as granger.hip: X X^H takes the Hermitian instance (lower-triangle tiles only)
            emu::launch(g, dim3(256), 0, [&] { spywil::zgemm_mfma_kernel<3>(reinterpret_cast<const cd*>(A), reinterpret_cast<const cd*>(B), reinterpret_cast<cd*>(Cm), n, sA, sB, sC, opB, addI, nullptr, nullptr, nullptr, batch); });
        else
            emu::launch(g, dim3(256), 0, [&] { spywil::zgemm_mfma_kernel<0>(reinterpret_cast<const cd*>(A), reinterpret_cast<const cd*>(B), reinterpret_cast<cd*>(Cm), n, sA, sB, sC, opB, addI, nullptr, nullptr, nullptr, batch); });
        return;
    }
    dim3 grid((n + 31) / 32, (n + 31) / 32, batch);
    emu::launch(grid, dim3(256), 0, [&] { spywil::zgemm_kernel(reinterpret_cast<const cd*>(A), reinterpret_cast<const cd*>(B), reinterpret_cast<cd*>(Cm), n, sA, sB, sC, opB, addI); });
}
// the fused forms of the matrix-core gemm: op(B) + Badd, and max |Ref - A op(B)| / |Ref| instead of the product
double emu_w_gemm_fused(const double* A, const double* B, double* Cm, int n, int batch, long long sB, int opB, const double* Badd,
                        const double* Ref) {
    const int ntile = spywil::zgemm_tiles(n, Ref != nullptr);
    dim3 g(spywil::zgemm_groups(n, Ref ? 2 : 1) * ((batch + 7) / 8) * 8);
    std::vector<double> part((size_t)ntile * batch, -1.0);
    emu::launch(g, dim3(256), 0, [&] {
        if (Ref)
            spywil::zgemm_mfma_kernel<2>(reinterpret_cast<const cd*>(A), reinterpret_cast<const cd*>(B), reinterpret_cast<cd*>(Cm), n,
                                         (long long)n * n, sB, (long long)n * n, opB, 0, nullptr, reinterpret_cast<const cd*>(Ref), part.data(), batch);
        else
            spywil::zgemm_mfma_kernel<1>(reinterpret_cast<const cd*>(A), reinterpret_cast<const cd*>(B), reinterpret_cast<cd*>(Cm), n,
                                         (long long)n * n, sB, (long long)n * n, opB, 0, reinterpret_cast<const cd*>(Badd), nullptr, nullptr, batch); });
    if (!Ref) return 0.0;
    double out = -1.0;
    emu::launch(dim3(1), dim3(256), 0, [&] { spywil::maxred_kernel(part.data(), (int)part.size(), &out); });
    return out;
}
void emu_w_skew(const double* g0, double* S, double* g0S, int n) {
    emu::launch(dim3((n * n + 255) / 256), dim3(256), 0, [&] { spywil::skew_kernel(reinterpret_cast<const cd*>(g0), reinterpret_cast<cd*>(S), reinterpret_cast<cd*>(g0S), n); });
}
void emu_w_inv(double* M, int n, int batch, int* info) {
    emu::launch(dim3(batch), dim3(256), (size_t)n * 36, [&] { spywil::zinv_kernel(reinterpret_cast<cd*>(M), n, info); });
}
void emu_w_inv_blocked(double* M, int n, int batch, int* info) {
    const int npad = ((n + spywil::ZB - 1) / spywil::ZB) * spywil::ZB;
    emu::launch(dim3(batch), dim3(256), ((size_t)spywil::ZB * npad + spywil::ZB * spywil::ZB) * 16,
                [&] { spywil::zinv_blocked_kernel(reinterpret_cast<cd*>(M), n, info); });
}
void emu_w_inv_mfma(double* M, int n, int batch, int* info) {
    const int npad = ((n + spywil::ZM - 1) / spywil::ZM) * spywil::ZM;
    // out of place from a copy of the input, as the Wilson iteration calls it
    std::vector<double> src(M, M + (size_t)batch * n * n * 2);
    std::fill(M, M + (size_t)batch * n * n * 2, -777.0);
    emu::launch(dim3(batch), dim3(spywil::ZT), ((size_t)spywil::ZM * (npad + 1) + spywil::ZM * (spywil::ZM + 1)) * 16,
                [&] { spywil::zinv_mfma_kernel(reinterpret_cast<cd*>(M), reinterpret_cast<const cd*>(src.data()), n, info); });
}
void emu_w_inv_mfma64(double* M, int n, int batch, int* info) {
    std::vector<double> src(M, M + (size_t)batch * n * n * 2);
    std::fill(M, M + (size_t)batch * n * n * 2, -777.0);
    emu::launch(dim3(batch), dim3(spywil::ZT), (size_t)2 * spywil::ZW * (spywil::ZW + 1) * 16,
                [&] { spywil::zinv64_mfma_kernel(reinterpret_cast<cd*>(M), reinterpret_cast<const cd*>(src.data()), n, info); });
}
void emu_w_chol(double* M, int n, int batch, int* info) {
    if (n <= 256 && n >= 2 * spywil::CHP) {       // as granger.hip: the panel kernel
        const size_t plds = ((size_t)n * (spywil::CHP + 1) + spywil::CHP * (spywil::CHP + 1)) * 16;
        emu::launch(dim3(batch), dim3(256), plds, [&] { spywil::zchol_panel_kernel(reinterpret_cast<cd*>(M), n, info); });
        return;
    }
    emu::launch(dim3(batch), dim3(256), (size_t)n * 16, [&] { spywil::zchol_kernel(reinterpret_cast<cd*>(M), n, info); });
}
void emu_w_gamma0(const double* A, int F, int n, double* out) {
    emu::launch(dim3((n * n + 255) / 256), dim3(256), 0, [&] { spywil::gamma0_kernel(reinterpret_cast<const cd*>(A), F, n, reinterpret_cast<cd*>(out), 0, F); });
}
int emu_w_plus(const double* g, int F, int n, const double* tw, double* gp, double* g0) {
    spywil::PlusPlan pl{};
    const int L = 2 * (F - 1);
    pl.L = L;
    int k = 0, m = L;
    const int cand[] = {4, 2, 3, 5, 7, 11, 13};
    for (int c : cand) while (m % c == 0 && m > 1) { pl.radix[k++] = c; m /= c; }
    for (int p = 17; m > 1; p += 2) while (m % p == 0) { pl.radix[k++] = p; m /= p; }
    pl.nfac = k;
    emu::launch(dim3(n * n), dim3(256), (size_t)2 * L * 16, [&] { spywil::plus_kernel(reinterpret_cast<const cd*>(g), F, (long long)n * n, pl, reinterpret_cast<const cd*>(tw), reinterpret_cast<cd*>(gp), reinterpret_cast<cd*>(g0)); });
    return k;
}
int emu_w_plus4(const double* g, int F, int n, const double* tw, double* gp, double* g0) {
    switch (2 * (F - 1)) {
        case 256: run_plus4<8>(g, F, n, tw, gp, g0); return 0;
        case 512: run_plus4<9>(g, F, n, tw, gp, g0); return 0;
        case 1024: run_plus4<10>(g, F, n, tw, gp, g0); return 0;
        case 2048: run_plus4<11>(g, F, n, tw, gp, g0); return 0;
        case 4096: run_plus4<12>(g, F, n, tw, gp, g0); return 0;
        default: return 1;
    }
}
void emu_w_addS(double* gp, const double* g0, double* out0, int F, int n) {
    emu::launch(dim3(4), dim3(256), 0, [&] { spywil::add_S_kernel(reinterpret_cast<cd*>(gp), reinterpret_cast<const cd*>(g0), reinterpret_cast<cd*>(out0), F, n); });
}
double emu_w_relerr(const double* A, const double* B, long long n) {
    std::vector<double> part(8);
    emu::launch(dim3(8), dim3(256), 0, [&] { spywil::relerr_kernel(reinterpret_cast<const cd*>(A), reinterpret_cast<const cd*>(B), n, part.data()); });
    double m = 0; for (double v : part) if (v > m || v != v) m = v;
    return m;
}
void emu_w_power(const double* M, int n, int batch, int iters, double* lam) {
    emu::launch(dim3(batch), dim3(256), (size_t)2 * n * 16, [&] { spywil::power_kernel(reinterpret_cast<const cd*>(M), n, iters, lam); });
}
void emu_w_granger(const double* CSD, const double* H, const double* Sigma, int F, int n, float* out) {
    emu::launch(dim3(4), dim3(256), 0, [&] { spywil::granger_kernel(reinterpret_cast<const cd*>(CSD), reinterpret_cast<const cd*>(H), reinterpret_cast<const cd*>(Sigma), F, n, out); });
}

}  // extern "C"

// ---- reference-precision tapered FFT: compile-time schedules (mtmfft_dec64_kernel.h) and the any-length kernel
// (mtmfft_f64_kernel.h, incl. its Bluestein form)
template <class Cf>
static void run_dec64_mode(spyfft::F64Args fa, int nseg, int nchan, int outk, int mean) {
    MtmArgs& a = fa.m;
    const int G = Cf::G;
    const int npairs = Cf::HALF ? nchan : (nchan + 1) / 2;              // (HALF: single channels, as mtmfft_dec64_launch.h)
    a.npg = (npairs + G - 1) / G;
    int S = (Cf::HALF ? 32 : 16) / G; if (S < 1) S = 1; if (S > a.npg) S = a.npg;
    a.S = S;
    a.ncl = (a.npg + S - 1) / S;
    const long long nclusters = (long long)nseg * a.ncl;
    const unsigned grid = (unsigned)(((nclusters + 7) / 8) * S * 8);
    auto go = [&](auto fn) { emu::launch(dim3(grid), dim3(Cf::NTHREADS), Cf::LDS_BYTES, fn); };
    switch (outk * 2 + mean) {
        case 0: go([&] { spyfft::mtmfft_dec64_kernel<Cf, 0, false>(fa); }); break;
        case 1: go([&] { spyfft::mtmfft_dec64_kernel<Cf, 0, true>(fa); }); break;
        case 2: go([&] { spyfft::mtmfft_dec64_kernel<Cf, 1, false>(fa); }); break;
        case 3: go([&] { spyfft::mtmfft_dec64_kernel<Cf, 1, true>(fa); }); break;
        case 4: go([&] { spyfft::mtmfft_dec64_kernel<Cf, 2, false>(fa); }); break;
        default: go([&] { spyfft::mtmfft_dec64_kernel<Cf, 2, true>(fa); }); break;
    }
}

extern "C" int emu_mtmfft_f64(int nfft, int blue_m, const float* data, long long ld, const int* chan_idx,
                              const long long* seg_start, const long long* seg_lo, const long long* seg_hi, int nseg,
                              int nsig, int nchan, int ntaper, const double* tapers64, const double* tw64,
                              const double* chirp64, const double* bhat64, float scale, int detrend, int demean_taper,
                              int seg_f64, const int* fpos, int nfsel, int out_kind, int keeptapers, int use_dec, void* out) {
    spyfft::F64Args fa{};
    MtmArgs& a = fa.m;
    a.data = data; a.ld = ld; a.chan_idx = chan_idx;
    a.seg_start = seg_start; a.seg_lo = seg_lo; a.seg_hi = seg_hi;
    a.nseg = nseg; a.nsig = nsig; a.nchan = nchan; a.ntaper = ntaper;
    a.scale = scale; a.detrend = detrend; a.demean_taper = demean_taper; a.fpos = fpos; a.nfsel = nfsel;
    a.out_kind = out_kind; a.out = out; a.means = g_means; a.seg_f64 = seg_f64;
    fa.tapers64 = tapers64;
    fa.tw64 = reinterpret_cast<const double2*>(tw64);
    const int outk = out_kind == SPYHIP_OUT_FOURIER ? 2 : (out_kind == SPYHIP_OUT_POW ? 0 : 1);
    const int mean = keeptapers ? 0 : 1;
    if (use_dec == 2) {
        // HALF form (CfgD64::HALF): single channels through the schedule of nfft / 2; tables as spyhip_fft_plan_set_precision
        // builds them: tw64 of nfft / 2 for the passes, the length-nfft table for the half step
        const int rn = nfft == 2002 ? 2000 : nfft;       // (id 2002 = nfft 2000 with split exchanges)
        std::vector<double2> th((size_t)rn / 2);
        for (int m = 0; m < rn / 2; ++m) th[m] = reinterpret_cast<const double2*>(tw64)[2 * m];
        fa.tw64_full = fa.tw64;
        fa.tw64 = th.data();
        using spyfft::CfgD64;
        switch (nfft) {
            case 2000: run_dec64_mode<CfgD64<10, 10, 10, 1, 2, false, true, true, 1, true>>(fa, nseg, nchan, outk, mean); break;
            case 2002: run_dec64_mode<CfgD64<10, 10, 10, 1, 2, true, true, true, 1, true>>(fa, nseg, nchan, outk, mean); break;   // split exchanges
            case 1200: run_dec64_mode<CfgD64<10, 10, 2, 1, 4, false, true, true, 3, true>>(fa, nseg, nchan, outk, mean); break;   // 3 x 200
            case 1024: run_dec64_mode<CfgD64<16, 16, 2, 1, 2, false, true, false, 1, true>>(fa, nseg, nchan, outk, mean); break;  // powers from the table
            case 12000: run_dec64_mode<spyfft::D64H_12000>(fa, nseg, nchan, outk, mean); break;
            default: return -1;
        }
        return 0;
    }
    if (use_dec) {
        switch (nfft) {          // (512 / 2048 / 500 differ from 1024 / 4096 / 1000 in one radix only: left to the GPU tests)
            case 256: run_dec64_mode<spyfft::D64_256>(fa, nseg, nchan, outk, mean); break;
            case 1024: run_dec64_mode<spyfft::D64_1024>(fa, nseg, nchan, outk, mean); break;
            case 4096: run_dec64_mode<spyfft::D64_4096>(fa, nseg, nchan, outk, mean); break;
            case 8192: run_dec64_mode<spyfft::D64_8192>(fa, nseg, nchan, outk, mean); break;
            case 16384: run_dec64_mode<spyfft::D64_16384>(fa, nseg, nchan, outk, mean); break;
            case 200: run_dec64_mode<spyfft::D64_200>(fa, nseg, nchan, outk, mean); break;
            case 1000: run_dec64_mode<spyfft::D64_1000>(fa, nseg, nchan, outk, mean); break;
            case 2000: run_dec64_mode<spyfft::D64_2000>(fa, nseg, nchan, outk, mean); break;
            case 2500: run_dec64_mode<spyfft::D64_2500>(fa, nseg, nchan, outk, mean); break;
            case 4000: run_dec64_mode<spyfft::D64_4000>(fa, nseg, nchan, outk, mean); break;
            case 5000: run_dec64_mode<spyfft::D64_5000>(fa, nseg, nchan, outk, mean); break;
            case 10000: run_dec64_mode<spyfft::D64_10000>(fa, nseg, nchan, outk, mean); break;
            case 600: run_dec64_mode<spyfft::D64_600>(fa, nseg, nchan, outk, mean); break;
            case 100: run_dec64_mode<spyfft::D64_100>(fa, nseg, nchan, outk, mean); break;
            case 300: run_dec64_mode<spyfft::D64_300>(fa, nseg, nchan, outk, mean); break;
            case 400: run_dec64_mode<spyfft::D64_400>(fa, nseg, nchan, outk, mean); break;
            case 2400: run_dec64_mode<spyfft::D64_2400>(fa, nseg, nchan, outk, mean); break;
            case 3200: run_dec64_mode<spyfft::D64_3200>(fa, nseg, nchan, outk, mean); break;
            case 768: run_dec64_mode<spyfft::D64_768>(fa, nseg, nchan, outk, mean); break;
            case 1500: run_dec64_mode<spyfft::D64_1500>(fa, nseg, nchan, outk, mean); break;
            case 3000: run_dec64_mode<spyfft::D64_3000>(fa, nseg, nchan, outk, mean); break;
            case 3072: run_dec64_mode<spyfft::D64_3072>(fa, nseg, nchan, outk, mean); break;
            case 6000: run_dec64_mode<spyfft::D64_6000>(fa, nseg, nchan, outk, mean); break;
            case 7500: run_dec64_mode<spyfft::D64_7500>(fa, nseg, nchan, outk, mean); break;
            default: return -1;
        }
        return 0;
    }
    // any-length kernel; work arrays in "LDS" (the emulator's dynamic buffer has no size limit)
    const int L = blue_m ? blue_m : nfft;
    if (!spywil::plus_plan(L, &fa.plan)) return -2;
    fa.work = nullptr;
    fa.wg0 = 0;
    fa.blue_n = blue_m ? nfft : 0;
    fa.chirp64 = reinterpret_cast<const double2*>(chirp64);
    fa.bhat64 = reinterpret_cast<const double2*>(bhat64);
    const unsigned grid = (unsigned)((long long)nseg * ((nchan + 1) / 2));
    const size_t lds = (size_t)2 * L * sizeof(double2);
    auto go = [&](auto fn) { emu::launch(dim3(grid), dim3(256), lds, fn); };
    switch (outk * 2 + mean) {
        case 0: go([&] { spyfft::mtmfft_f64_any_kernel<0, false>(fa); }); break;
        case 1: go([&] { spyfft::mtmfft_f64_any_kernel<0, true>(fa); }); break;
        case 2: go([&] { spyfft::mtmfft_f64_any_kernel<1, false>(fa); }); break;
        case 3: go([&] { spyfft::mtmfft_f64_any_kernel<1, true>(fa); }); break;
        case 4: go([&] { spyfft::mtmfft_f64_any_kernel<2, false>(fa); }); break;
        default: go([&] { spyfft::mtmfft_f64_any_kernel<2, true>(fa); }); break;
    }
    return 0;
}
