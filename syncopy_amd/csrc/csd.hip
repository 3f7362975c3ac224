// Host side of K4/K5: cross-spectral accumulation, finalisation and coherence
// normalisation (spyhip_csd_accumulate / spyhip_csd_finalize / spyhip_coh_normalize).
#include <algorithm>

#include <cstdlib>

#include "spy_common.h"
#include "csd_kernel.h"
#include "csd3m_launch.h"
#include "csdh_launch.h"

using spycsd::CsdArgs;

namespace {

template <int TA, int TB, int FAST = 0>
int launch_accum(spyhip_ctx* ctx, CsdArgs a, long long item_base, long long item_end, int nsplit = 1) {
    auto kern = spycsd::csd_accum_kernel<TA, TB, FAST>;
    const int per = 4 * (TA + TB);
    // frequencies a workgroup can touch: items [i0, i0+per) span at most this many f
    int nfb = (per + a.ntiles - 1) / a.ntiles;
    if (per % a.ntiles != 0 && a.ntiles > 1) nfb += 1;
    if (nfb > a.F) nfb = a.F;
    const size_t rowbytes = (size_t)nfb * a.cpad * sizeof(float2);
    // a chunk holds at most 512 threads x CSD_PF staged elements; LDS holds three chunks
    const size_t chunk_max = (size_t)spycsd::CSD_THREADS * spycsd::CSD_PF * sizeof(float2);
    int kb = 32;
    while (kb > 4 && (size_t)kb * rowbytes > chunk_max) kb -= 4;
    if (!FAST && ((size_t)kb * rowbytes > chunk_max || 3 * (size_t)kb * rowbytes > ctx->lds_per_block)) {
        spy::set_error("csd_accumulate: %d channels do not fit the LDS staging buffer", a.C);
        return -3;
    }
    const long long rows_wg = nsplit > 1 ? a.rows_per_split : a.nrows;
    if (kb > rows_wg && !FAST) kb = (int)((rows_wg + 3) & ~3LL);
    if (FAST) kb = 16;                                   // three 32 KiB buffers of 16 rows x 256 elements
    a.kb = kb;
    a.item_base = item_base;
    a.item_end = item_end;
    // FAST: + one row of slack - a tile's columns past the last frequency of a row are read (and never stored)
    const size_t lds = FAST ? 3 * (size_t)16 * 256 * sizeof(float2) + 512 : 3 * (size_t)kb * rowbytes;
    SPY_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern),
                                      hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    const long long wg_items = FAST ? a.fast_per : per;
    long long grid = (item_end - item_base + wg_items - 1) / wg_items;
    if (FAST == 3) grid = ((item_end - item_base) / a.ntiles) * a.fast_nwgf;     // whole frequencies x workgroups each
    if (grid <= 0) return 0;
    if (grid > 0x7fffffffLL) { spy::set_error("csd_accumulate: grid too large"); return -1; }
    hipLaunchKernelGGL(kern, dim3((unsigned)grid, (unsigned)nsplit), dim3(spycsd::CSD_THREADS), lds, ctx->stream, a);
    SPY_HIP_CHECK(hipGetLastError());
    return 0;
}

// The workgroups beyond the last full round of a launch, re-cut: 1 tile per wave AND the rows split over blockIdx.y, so
// that the short workgroups fill the chip once; splits > 0 leave partial sums in library scratch that a fixed-order
// reduction adds afterwards (deterministic, no atomics).  `first` = first item of the tail.
int launch_tail(spyhip_ctx* ctx, CsdArgs a, long long first, int64_t nrows, int nfreq, int nchan) {
    const long long tail_wg = (a.nitems - first + 7) / 8;
    long long nsplit = ctx->num_cu / tail_wg;
    const long long max_split = (nrows + 63) / 64;          // at least 64 rows per split
    if (nsplit > max_split) nsplit = max_split;
    // the partial sums of the splits are reduced per whole frequency from f0 on: the tail must start on a
    // frequency boundary (always true on the fast paths, whose items per workgroup are a multiple of ntiles; the
    // (5,4) path of the blocked layout has 36 whatever ntiles is)
    if (nsplit < 2 || first % a.ntiles != 0) return launch_accum<1, 1>(ctx, a, first, a.nitems);
    const int f0 = (int)(first / a.ntiles), nf = nfreq - f0;
    const size_t need = (size_t)(nsplit - 1) * nf * nchan * nchan * sizeof(float2);
    if (need > ctx->scratch_bytes) {
        if (ctx->scratch) { (void)hipFree(ctx->scratch); ctx->scratch = nullptr; ctx->scratch_bytes = 0; }
        SPY_HIP_CHECK(hipMalloc(&ctx->scratch, need));
        ctx->scratch_bytes = need;
    }
    a.rows_per_split = ((nrows + nsplit - 1) / nsplit + 3) & ~3LL;
    nsplit = (nrows + a.rows_per_split - 1) / a.rows_per_split;
    a.part = reinterpret_cast<float2*>(ctx->scratch);
    a.part_f0 = f0;
    a.part_nf = nf;
    int rc = launch_accum<1, 1>(ctx, a, first, a.nitems, (int)nsplit);
    if (rc) return rc;
    if (nsplit > 1) {
        const long long n = (long long)nf * nchan * nchan;
        hipLaunchKernelGGL(spycsd::csd_reduce_parts_kernel, dim3((unsigned)std::min<long long>((n + 255) / 256, 4096)),
                           dim3(256), 0, ctx->stream, a.acc, a.part, (int)nsplit - 1, f0, nf, nchan);
        SPY_HIP_CHECK(hipGetLastError());
    }
    return 0;
}

// acc[f, i, j] += x[f, i] conj(x[f, j]) (i >= j) for ONE row of spectra: the last row of an odd channel count above 512
// (the 3M kernels' 16-byte copies would read 8 bytes past the end of the spectra there)
__global__ void __launch_bounds__(256) csd_rank1_kernel(const float2* __restrict__ x, int F, int C, float2* __restrict__ acc) {
    const long long per = (long long)C * C, tot = (long long)F * per, stride = (long long)gridDim.x * blockDim.x;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += stride) {
        const long long f = e / per, r = e - f * per;
        const int i = (int)(r / C), j = (int)(r - (long long)i * C);
        if (j > i) continue;
        const float2 a = x[f * C + i], b = x[f * C + j];
        float2 o = acc[e];
        o.x += a.x * b.x + a.y * b.y;
        o.y += a.y * b.x - a.x * b.y;
        acc[e] = o;
    }
}

// packed rows to give the 3M kernel when the workgroups beyond the last full round of the chip go to the re-cut tail
long long m3_main_rows(spyhip_ctx* ctx, long long nprow, int np) {
    const long long nwg = nprow * np, rem = nwg % ctx->num_cu;
    if (nwg > ctx->num_cu && rem > 0 && rem * 4 <= ctx->num_cu) return nprow - (rem + np - 1) / np;
    return nprow;
}

}  // namespace

static int csd_accumulate_impl(spyhip_ctx* ctx, const void* spec_d, int64_t nrows, int nfreq, int nchan, void* acc_d,
                               int blocked, bool only_4m = false);

extern "C" int spyhip_csd_set_phase_exact(spyhip_ctx* ctx, int on) {
    if (!ctx) { spy::set_error("csd_set_phase_exact: null context"); return -1; }
    ctx->csd_phase_exact = on ? 1 : 0;
    return 0;
}

extern "C" int spyhip_csd_accumulate(spyhip_ctx* ctx, const void* spec_d, int64_t nrows, int nfreq, int nchan,
                                     void* acc_d) {
    if (ctx) ctx->k4h_nf = 0;
    return csd_accumulate_impl(ctx, spec_d, nrows, nfreq, nchan, acc_d, 0);
}

// K4h (csdh_kernel.h): 256 channels on the half-precision matrix cores with split float32 operands.  The frequencies
// beyond the last full round of workgroups go to the re-cut float32 tail like on the other paths.  [f0, f0 + nf): the
// frequencies of this call - a caller that wants the results of a range while the next is still being accumulated (the
// coherence pipeline: normalisation and host copy of range r under the products of range r + 1) launches range by range.
static int csd_accumulate_split_range(spyhip_ctx* ctx, const void* spec_d, int64_t nrows, int nfreq, int nchan, void* acc_d,
                                      const float* absmax_d, int f0, int nf) {
    static const bool env_f32 = std::getenv("SPYHIP_CSD_F32") != nullptr;
    if (!spec_d || !acc_d || nfreq < 1 || f0 < 0 || nf < 0 || f0 + nf > nfreq) {
        spy::set_error("csd_accumulate_split: null argument / bad shape");
        return -1;
    }
    if (nf == 0) return 0;
    SPY_HIP_CHECK(hipSetDevice(ctx->device));
    (void)env_f32;
    if (!ctx->k4h_done) SPY_HIP_CHECK(hipEventCreateWithFlags(&ctx->k4h_done, hipEventDisableTiming));
    else SPY_HIP_CHECK(hipStreamWaitEvent(ctx->stream, ctx->k4h_done, 0));      // the previous call's flag readers are through
    const size_t need = (size_t)nfreq * sizeof(int) + 256 * sizeof(float);
    if (need > ctx->k4h_bytes) {
        if (ctx->k4h_buf) { SPY_HIP_CHECK(hipStreamSynchronize(ctx->stream)); (void)hipFree(ctx->k4h_buf); ctx->k4h_buf = nullptr; ctx->k4h_bytes = 0; }
        SPY_HIP_CHECK(hipMalloc(&ctx->k4h_buf, need));
        SPY_HIP_CHECK(hipMemsetAsync(ctx->k4h_buf, 0, need, ctx->stream));      // (ranges not launched yet read as "not flagged")
        ctx->k4h_bytes = need;
    }
    float* const own_max = reinterpret_cast<float*>(ctx->k4h_buf);
    int* const flags = reinterpret_cast<int*>(own_max + 256);
    const float2* spec = reinterpret_cast<const float2*>(spec_d);
    if (!absmax_d) {
        SPY_HIP_CHECK(hipMemsetAsync(own_max, 0, 256 * sizeof(float), ctx->stream));
        int rc = spycsd::csdh_absmax(ctx->stream, spec, (long long)nrows * nfreq * 256, 256, own_max);
        if (rc) return rc;
        absmax_d = own_max;
    }
    const long long rem = nfreq % ctx->num_cu;
    int f_main = nfreq;
    if (nfreq > ctx->num_cu && rem > 0 && rem * 4 <= ctx->num_cu) f_main = nfreq - (int)rem;
    const int f1 = f0 + nf, h1 = std::min(f1, f_main);
    if (f1 > f_main && f1 != nfreq) {          // the re-cut float32 tail [f_main, nfreq) goes in one piece
        spy::set_error("csd_accumulate_split_range: a range beyond frequency %d must end at nfreq = %d", f_main, nfreq);
        return -1;
    }
    int rc = 0;
    if (h1 > f0)
        rc = spycsd::csdh_run(ctx->stream, spec, nrows, nfreq, reinterpret_cast<float2*>(acc_d), absmax_d, flags, f0, h1 - f0,
                              ctx->csd_phase_exact != 0);
    ctx->k4h_nf = f_main;
    if (!rc) SPY_HIP_CHECK(hipEventRecord(ctx->k4h_done, ctx->stream));         // (behind csdh_kernel and its only_flagged stand-in)
    if (rc || f1 <= f_main) return rc;
    CsdArgs a{};
    a.spec = spec;
    a.nrows = nrows; a.F = nfreq; a.C = 256;
    a.acc = reinterpret_cast<float2*>(acc_d);
    a.nt = 8; a.ntiles = 36; a.nitems = (long long)nfreq * 36; a.cpad = 256;
    a.fast_per = 36;
    return launch_tail(ctx, a, (long long)std::max(f0, f_main) * 36, nrows, nfreq, 256);
}

extern "C" int spyhip_csd_accumulate_split(spyhip_ctx* ctx, const void* spec_d, int64_t nrows, int nfreq, int nchan,
                                           void* acc_d, const float* absmax_d) {
    static const bool env_f32 = std::getenv("SPYHIP_CSD_F32") != nullptr;
    if (!ctx || nchan != 256 || nrows < 1 || env_f32) {
        if (ctx) ctx->k4h_nf = 0;           // spyhip_csd_split_fallbacks reports THIS call: nothing went to the half-precision kernel
        return csd_accumulate_impl(ctx, spec_d, nrows, nfreq, nchan, acc_d, 0);
    }
    return csd_accumulate_split_range(ctx, spec_d, nrows, nfreq, nchan, acc_d, absmax_d, 0, nfreq);
}

extern "C" int spyhip_csd_accumulate_split_range(spyhip_ctx* ctx, const void* spec_d, int64_t nrows, int nfreq, int nchan,
                                                 void* acc_d, const float* absmax_d, int f0, int nf) {
    static const bool env_f32 = std::getenv("SPYHIP_CSD_F32") != nullptr;
    if (!ctx || nchan != 256 || nrows < 1 || env_f32 || !absmax_d) {
        spy::set_error("csd_accumulate_split_range: 256 channels, at least one row and the range of the spectra (absmax_d) are required");
        return -1;
    }
    return csd_accumulate_split_range(ctx, spec_d, nrows, nfreq, nchan, acc_d, absmax_d, f0, nf);
}

extern "C" int spyhip_csd_split_fallbacks(spyhip_ctx* ctx, int* count) {
    if (!ctx || !count) { spy::set_error("csd_split_fallbacks: null argument"); return -1; }
    *count = 0;
    if (!ctx->k4h_buf || ctx->k4h_nf <= 0) return 0;
    SPY_HIP_CHECK(hipSetDevice(ctx->device));
    std::vector<int> h((size_t)ctx->k4h_nf);
    SPY_HIP_CHECK(hipMemcpyAsync(h.data(), reinterpret_cast<const char*>(ctx->k4h_buf) + 256 * sizeof(float), h.size() * sizeof(int),
                                 hipMemcpyDeviceToHost, ctx->stream));
    SPY_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    for (int v : h) *count += v != 0;
    return 0;
}

extern "C" int spyhip_csd_accumulate_blocked(spyhip_ctx* ctx, const void* spec_d, int64_t nrows, int nfreq, int nchan,
                                             void* acc_d) {
    return csd_accumulate_impl(ctx, spec_d, nrows, nfreq, nchan, acc_d, 1);
}

static int csd_accumulate_impl(spyhip_ctx* ctx, const void* spec_d, int64_t nrows, int nfreq, int nchan, void* acc_d,
                               int blocked, bool only_4m) {
    if (!ctx || !spec_d || !acc_d) { spy::set_error("csd_accumulate: null argument"); return -1; }
    if (nrows < 0 || nfreq < 1 || nchan < 1) { spy::set_error("csd_accumulate: bad shape"); return -1; }
    if (nrows == 0) return 0;
    SPY_HIP_CHECK(hipSetDevice(ctx->device));
    CsdArgs a{};
    a.spec = reinterpret_cast<const float2*>(spec_d);
    a.nrows = nrows; a.F = nfreq; a.C = nchan;
    a.acc = reinterpret_cast<float2*>(acc_d);
    a.nt = (nchan + 31) / 32;
    a.ntiles = a.nt * (a.nt + 1) / 2;
    a.nitems = (long long)nfreq * a.ntiles;
    a.cpad = a.nt * 32;
    a.blocked = blocked;
    // The instruction-lean path: C <= 256, row-major spectra.  A 256-element LDS row holds nfb consecutive
    // frequencies (1 for C > 128, 2 for C = 128, 4 for C = 64, ...) and a workgroup owns their nfb * ntiles <= 40 tiles.
    const bool fast = !blocked && nchan <= 256;            // (odd C: 8-byte staging loads instead of 16-byte ones)
    if (fast) {
        int nfb = 256 / nchan;
        while (nfb > 1 && nfb * a.ntiles > 40) --nfb;
        if (nfb > nfreq) nfb = nfreq;
        a.fast_per = nfb * a.ntiles;
    }
    // C in (256, 512]: the same path with 512-element rows; the tiles of a frequency are shared by
    // ceil(ntiles / 40) workgroups (512 channels: 4 x 34 tiles), each staging the whole row.
    // spyhip_csd_set_phase_exact selects them per context (imag / angle outputs, see include/spyhip.h).
    const bool force_4m = ctx->csd_phase_exact != 0 || only_4m;
    const bool wide3m = !force_4m && nchan > 256 && spycsd::m3_available(nchan);
    if (!blocked && nchan > 256 && nchan <= 512 && !wide3m) {
        a.fast_nwgf = (a.ntiles + 39) / 40;
        a.fast_per = (a.ntiles + a.fast_nwgf - 1) / a.fast_nwgf;
        const long long nwg = (long long)nfreq * a.fast_nwgf;
        long long f_main = nfreq;
        const long long rem = nwg % ctx->num_cu;
        if (nwg > ctx->num_cu && rem > 0 && rem * 4 <= ctx->num_cu) f_main = ((nwg - rem) / a.fast_nwgf);
        int rc = launch_accum<5, 4, 3>(ctx, a, 0, f_main * a.ntiles);
        if (rc || f_main == nfreq) return rc;
        return launch_accum<1, 1>(ctx, a, f_main * a.ntiles, a.nitems);     // the last partial round, re-cut
    }
    // 256 channels, row-major spectra: the 3-multiplication kernel, one workgroup per frequency; the workgroups
    // beyond the last full round (F = 2049 on 256 CUs: one frequency) go to the re-cut tail like on the other paths.
    if (nchan == 256 && !force_4m) {         // either hand-over layout: the kernel's LDS copies gather
        const long long nwg = nfreq, rem = nwg % ctx->num_cu;
        long long f_main = nfreq;
        if (nwg > ctx->num_cu && rem > 0 && rem * 4 <= ctx->num_cu) f_main = nwg - rem;
        int rc = spycsd::m3_launch(256, ctx->stream, a, f_main);
        if (rc == -100) { spy::set_error("csd_accumulate: no 3M kernel for 256 channels"); return -1; }
        if (rc || f_main == nfreq) return rc;
        return launch_tail(ctx, a, f_main * a.ntiles, nrows, nfreq, nchan);
    }
    // every other channel count up to 512, row-major spectra: the 3M kernel instance of the next multiple of 16 with
    // the narrower rows padded inside its LDS image (csd3m_kernel.h, EXACT = false).  Below 256 (padded) channels
    // floor(256 / CHp) frequencies per workgroup (the last packed row may be partial); above, 512-element LDS rows and
    // several workgroups per frequency.  Odd channel counts: the last row of spectra goes to the 4-multiplication
    // kernels (the 16-byte copy of the last channel reaches 8 bytes beyond its frequency).
    if (nchan != 256 && !blocked && !force_4m && spycsd::m3_available(nchan)) {
        const int chp = spycsd::m3_padded(nchan);
        const int64_t nrows3 = (nchan & 1) ? nrows - 1 : nrows;
        if (nrows3 > 0) {
            CsdArgs b = a;
            b.nrows = nrows3;
            const int fpr = chp < 256 ? 256 / chp : 1;
            const long long nprow = (nfreq + fpr - 1) / fpr;
            const long long p_main = m3_main_rows(ctx, nprow, spycsd::m3_parts(nchan));
            int rc = spycsd::m3_launch(nchan, ctx->stream, b, p_main);
            if (rc == -100) { spy::set_error("csd_accumulate: no 3M kernel for %d channels", nchan); return -1; }
            if (rc) return rc;
            if (p_main < nprow && (rc = launch_tail(ctx, b, fpr * p_main * a.ntiles, nrows3, nfreq, nchan))) return rc;
        }
        if (nrows3 == nrows) return 0;
        return csd_accumulate_impl(ctx, reinterpret_cast<const float2*>(spec_d) + (size_t)nrows3 * nfreq * nchan, 1, nfreq, nchan,
                                   acc_d, 0, true);
    }
    // more than 512 channels, row-major spectra: the lower triangle in blocks of 256 channels - the Hermitian product of
    // every block with itself (the 3M instance of its width, reading its channel range out of the wide rows) and the
    // rectangle of every pair of blocks (csd3m_kernel<512, 8, false, true>: the two ranges side by side in one
    // 512-element LDS image, the 256 sub-tiles of the off-diagonal quadrant shared by three workgroups per frequency).
    // No channel count is too wide for LDS any more: a launch never stages more than 512 channels.  Phase-exact
    // accumulation (force_4m) takes the same walk with the 4-multiplication instances of the tiled kernel (the 32 x 32-tile
    // kernels below cannot stage rows this wide).
    if (nchan > 512 && !blocked) {
        const int64_t nrows3 = (nchan & 1) ? nrows - 1 : nrows;
        const int nb = (nchan + 255) / 256;
        if (nrows3 > 0) {
            CsdArgs b = a;
            b.nrows = nrows3;
            b.ctot = nchan;
            for (int I = 0; I < nb; ++I) {
                const int nI = std::min(256, nchan - 256 * I);
                b.ch0 = 256 * I; b.n0 = nI; b.ch1 = 0; b.n1 = 0;
                const int chp = force_4m ? 256 : spycsd::m3_padded(nI);
                const int fpr = chp < 256 ? 256 / chp : 1;
                int rc = force_4m ? spycsd::m4_launch_block(ctx->stream, b, nfreq)
                                  : spycsd::m3_launch_padded(chp, ctx->stream, b, (nfreq + fpr - 1) / fpr);
                if (rc == -100) { spy::set_error("csd_accumulate: no 3M kernel for a block of %d channels", nI); return -1; }
                if (rc) return rc;
                for (int J = 0; J < I; ++J) {
                    b.ch0 = 256 * J; b.n0 = 256; b.ch1 = 256 * I; b.n1 = nI;
                    if ((rc = force_4m ? spycsd::m4_launch_rect(ctx->stream, b, nfreq) : spycsd::m3_launch_rect(ctx->stream, b, nfreq)))
                        return rc;
                }
            }
        }
        if (nrows3 < nrows) {
            const long long tot = (long long)nfreq * nchan * nchan;
            hipLaunchKernelGGL(csd_rank1_kernel, dim3((unsigned)std::min<long long>((tot + 255) / 256, 65535)), dim3(256), 0,
                               ctx->stream, reinterpret_cast<const float2*>(spec_d) + (size_t)nrows3 * nfreq * nchan, nfreq, nchan,
                               reinterpret_cast<float2*>(acc_d));
            SPY_HIP_CHECK(hipGetLastError());
        }
        return 0;
    }
    // tiles per wave (waves 0-3, waves 4-7): (5,4) packs the 36 tiles of C=256 into one workgroup per frequency
    if (fast || a.ntiles >= 21) {
        // One workgroup per CU: F = 2049 frequencies on 256 CUs would leave a 9th, almost empty round.
        // The workgroups beyond the last full round are re-cut into 1-tile-per-wave workgroups
        // (4.5x more, each 5x shorter), so the tail costs ~1/5 of a round and stays deterministic.
        const long long per = fast ? a.fast_per : 36, nwg = (a.nitems + per - 1) / per;
        const long long full = (nwg / ctx->num_cu) * ctx->num_cu, rem = nwg - full;
        if (full > 0 && rem > 0 && rem * 4 <= ctx->num_cu) {
            // 36 tiles in every workgroup (C = 256): the variant without per-tile guards
            int rc = !fast ? launch_accum<5, 4>(ctx, a, 0, full * per)
                           : (nchan == 256 ? launch_accum<5, 4, 1>(ctx, a, 0, full * per)
                                             : launch_accum<5, 4, 2>(ctx, a, 0, full * per));
            if (rc) return rc;
            return launch_tail(ctx, a, full * per, nrows, nfreq, nchan);
        }
        if (!fast) return launch_accum<5, 4>(ctx, a, 0, a.nitems);
        return nchan == 256 ? launch_accum<5, 4, 1>(ctx, a, 0, a.nitems) : launch_accum<5, 4, 2>(ctx, a, 0, a.nitems);
    }
    if (a.ntiles >= 6) return launch_accum<3, 2>(ctx, a, 0, a.nitems);
    return launch_accum<1, 1>(ctx, a, 0, a.nitems);
}

extern "C" int spyhip_csd_finalize(spyhip_ctx* ctx, void* acc_d, int nfreq, int nchan, double scale) {
    if (!ctx || !acc_d) { spy::set_error("csd_finalize: null argument"); return -1; }
    SPY_HIP_CHECK(hipSetDevice(ctx->device));
    const long long nt = (nchan + 31) / 32, blocks = (long long)nfreq * (nt * (nt + 1) / 2);
    if (blocks > 0x7fffffffLL) { spy::set_error("csd_finalize: grid too large"); return -1; }
    hipLaunchKernelGGL(spycsd::csd_finalize_kernel, dim3((unsigned)blocks), dim3(256), 0, ctx->stream,
                       reinterpret_cast<float2*>(acc_d), nfreq, nchan, (float)scale);
    SPY_HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" int spyhip_coh_from_accumulator(spyhip_ctx* ctx, const void* acc_d, int nfreq, int nchan, double scale,
                                           int output, void* out_d) {
    if (!ctx || !acc_d || !out_d) { spy::set_error("coh_from_accumulator: null argument"); return -1; }
    if (output < SPYHIP_OUT_POW || output > SPYHIP_OUT_ABSIMAG) { spy::set_error("coh_from_accumulator: bad output %d", output); return -1; }
    SPY_HIP_CHECK(hipSetDevice(ctx->device));
    const long long nt = (nchan + 31) / 32, blocks = (long long)nfreq * (nt * (nt + 1) / 2);
    if (blocks > 0x7fffffffLL) { spy::set_error("coh_from_accumulator: grid too large"); return -1; }
    if (output == SPYHIP_OUT_FOURIER)
        hipLaunchKernelGGL(spycsd::coh_from_acc_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream,
                           reinterpret_cast<const float2*>(acc_d), nfreq, nchan, (float)scale, output, out_d);
    else
        hipLaunchKernelGGL(spycsd::coh_from_acc_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream,
                           reinterpret_cast<const float2*>(acc_d), nfreq, nchan, (float)scale, output, out_d);
    SPY_HIP_CHECK(hipGetLastError());
    return 0;
}

static int tril_move(spyhip_ctx* ctx, void* acc_d, int nfreq, int nchan, void* packed_d, bool unpack) {
    if (!ctx || !acc_d || !packed_d) { spy::set_error("csd_tril: null argument"); return -1; }
    SPY_HIP_CHECK(hipSetDevice(ctx->device));
    const long long n = (long long)nfreq * nchan * nchan;
    long long blocks = (n + 255) / 256;
    if (blocks > 16384) blocks = 16384;
    if (unpack)
        hipLaunchKernelGGL(spycsd::csd_tril_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream,
                           reinterpret_cast<float2*>(acc_d), reinterpret_cast<float2*>(packed_d), nfreq, nchan);
    else
        hipLaunchKernelGGL(spycsd::csd_tril_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream,
                           reinterpret_cast<float2*>(acc_d), reinterpret_cast<float2*>(packed_d), nfreq, nchan);
    SPY_HIP_CHECK(hipGetLastError());
    return 0;
}

extern "C" int spyhip_csd_tril_pack(spyhip_ctx* ctx, const void* acc_d, int nfreq, int nchan, void* packed_d) {
    return tril_move(ctx, const_cast<void*>(acc_d), nfreq, nchan, packed_d, false);
}

extern "C" int spyhip_csd_tril_unpack(spyhip_ctx* ctx, const void* packed_d, int nfreq, int nchan, void* acc_d) {
    return tril_move(ctx, acc_d, nfreq, nchan, const_cast<void*>(packed_d), true);
}

extern "C" int spyhip_coh_normalize(spyhip_ctx* ctx, const void* csd_d, int nfreq, int nchan, int output,
                                    void* out_d) {
    if (!ctx || !csd_d || !out_d) { spy::set_error("coh_normalize: null argument"); return -1; }
    if (output < SPYHIP_OUT_POW || output > SPYHIP_OUT_ABSIMAG) { spy::set_error("coh_normalize: bad output %d", output); return -1; }
    SPY_HIP_CHECK(hipSetDevice(ctx->device));
    const long long n = (long long)nfreq * nchan * nchan;
    long long blocks = (n + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    if (output == SPYHIP_OUT_FOURIER)
        hipLaunchKernelGGL(spycsd::coh_normalize_kernel<true>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream,
                           reinterpret_cast<const float2*>(csd_d), nfreq, nchan, output, out_d);
    else
        hipLaunchKernelGGL(spycsd::coh_normalize_kernel<false>, dim3((unsigned)blocks), dim3(256), 0, ctx->stream,
                           reinterpret_cast<const float2*>(csd_d), nfreq, nchan, output, out_d);
    SPY_HIP_CHECK(hipGetLastError());
    return 0;
}

#include "jack_kernel.h"

extern "C" int spyhip_jack_coh_accumulate(spyhip_ctx* ctx, const void* spec_d, int ntrials, int ntaper, int nfreq,
                                          int nchan, const void* csd_d, const void* direct_d, int output,
                                          int64_t ntrials_total, void* sum_d, void* sum_d2) {
    if (!ctx || !spec_d || !csd_d || !direct_d || !sum_d || !sum_d2) { spy::set_error("jack_coh_accumulate: null argument"); return -1; }
    if (ntrials < 0 || ntaper < 1 || nfreq < 1 || nchan < 1 || ntrials_total < 2) { spy::set_error("jack_coh_accumulate: bad shape"); return -1; }
    if (output < SPYHIP_OUT_POW || output > SPYHIP_OUT_ABSIMAG) { spy::set_error("bad output kind %d", output); return -1; }
    if (ntrials == 0) return 0;
    SPY_HIP_CHECK(hipSetDevice(ctx->device));
    spycsd::JackArgs a{};
    a.spec = reinterpret_cast<const float2*>(spec_d);
    a.S = reinterpret_cast<const float2*>(csd_d);
    a.direct = direct_d;
    a.ntrials = ntrials; a.K = ntaper; a.F = nfreq; a.C = nchan; a.kind = output;
    a.T = (float)ntrials_total;
    a.sum_d = reinterpret_cast<double*>(sum_d);
    a.sum_d2 = reinterpret_cast<double*>(sum_d2);
    const long long nt = (nchan + 31) / 32, blocks = 8LL * ((nfreq + 7) / 8) * (nt * (nt + 1) / 2);
    if (blocks > 0x7fffffffLL) { spy::set_error("jack_coh_accumulate: grid too large"); return -1; }
    const size_t lds = 2 * (size_t)2 * ntaper * 32 * sizeof(float2);
    if (lds > ctx->lds_per_block) { spy::set_error("jack_coh_accumulate: %d tapers do not fit the LDS staging buffer", ntaper); return -3; }
    if (output == SPYHIP_OUT_FOURIER) {
        SPY_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(spycsd::jack_coh_kernel<true>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(spycsd::jack_coh_kernel<true>, dim3((unsigned)blocks), dim3(256), lds, ctx->stream, a);
    } else {
        SPY_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(spycsd::jack_coh_kernel<false>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
        hipLaunchKernelGGL(spycsd::jack_coh_kernel<false>, dim3((unsigned)blocks), dim3(256), lds, ctx->stream, a);
    }
    SPY_HIP_CHECK(hipGetLastError());
    return 0;
}
