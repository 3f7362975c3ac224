// Plan of the packed mixed-radix engine (mtmfft_mixed.h): what the host computes and hands to the kernel.
#pragma once
#include <cstddef>

namespace spyfft {

constexpr int MIX_MAXPASS = 8;
constexpr int MIX_V = 10;                  // values (of 4 channels each) a thread holds at most: 40 registers,
                                           // so that 1024-thread workgroups (N = 10000) fit and short lengths run 4 waves / SIMD

struct MixPlan {
    int n;                                 // FFT length
    int th;                                // threads per channel quad
    int lg;                                // log2 of the channel quads per workgroup
    int npass;
    int radix[MIX_MAXPASS];
    int stage;                             // the detrended segment is staged in LDS (else re-read per taper)
    unsigned magic[MIX_MAXPASS];           // floor(2^32 / Ns) + 1 for the pass's stride Ns > 1: b / Ns = umulhi(b, magic)
};

// workgroup shape for 2^lg quads per workgroup, with / without the staged segment
inline void mix_layout(MixPlan* g, int lg, int stage, int* nthreads, size_t* lds_bytes) {
    g->lg = lg;
    g->stage = stage;
    *nthreads = (((g->th << lg) + 63) / 64) * 64;
    size_t lds = (size_t)16 * (g->n + 1) * (1u << lg) * (stage ? 2 : 1);
    if (lds < 16384) lds = 16384;                              // room for the block sums' scratch
    *lds_bytes = lds;
}

// ---- host side: radix schedule, threads per quad, quads per workgroup.  false: n is not 5-smooth / out of range.
inline bool mix_schedule(int n, int nquads, MixPlan* g, int* nthreads, size_t* lds_bytes) {
    if (n < 16 || n > 10000) return false;
    int a = 0, b = 0, c = 0, m = n;
    while (m % 2 == 0) { m /= 2; ++a; }
    while (m % 3 == 0) { m /= 3; ++b; }
    while (m % 5 == 0) { m /= 5; ++c; }
    if (m != 1) return false;
    // pair twos with fives (radix 10) where that saves a pass; among the schedules with the fewest passes take the
    // one that needs the fewest threads per quad (every phase holds at most MIX_V values per thread)
    const int mf = MIX_V / 2 + 1, nf = n / 2 + 1;
    int best_np = 1 << 30, best_th = 1 << 30;
    for (int u = 0; u <= (a < c ? a : c); ++u) {
        int rad[MIX_MAXPASS + 8], k = 0;
        for (int i = 0; i < c - u; ++i) rad[k++] = 5;          // an odd radix first: its stride-R scatter is conflict free
        for (int i = 0; i < b; ++i) rad[k++] = 3;
        for (int i = 0; i < u; ++i) rad[k++] = 10;
        int twos = a - u;
        while (twos >= 3) { rad[k++] = 8; twos -= 3; }
        if (twos) rad[k++] = 1 << twos;
        if (k > MIX_MAXPASS) continue;
        int th = (n + MIX_V - 1) / MIX_V;
        if ((nf + mf - 1) / mf > th) th = (nf + mf - 1) / mf;
        for (int p = 0; p < k; ++p) {
            const int nb = n / rad[p], mb = MIX_V / rad[p];
            if ((nb + mb - 1) / mb > th) th = (nb + mb - 1) / mb;
        }
        if (k < best_np || (k == best_np && th < best_th)) {
            best_np = k; best_th = th;
            g->npass = k;
            for (int p = 0; p < k; ++p) g->radix[p] = rad[p];
        }
    }
    if (best_np > MIX_MAXPASS) return false;
    const int th = best_th;
    if (th > 1024) return false;
    g->n = n;
    g->th = th;
    long long Ns = 1;
    for (int p = 0; p < g->npass; ++p) {
        g->magic[p] = Ns > 1 ? (unsigned)((1ULL << 32) / (unsigned long long)Ns + 1ULL) : 0u;
        Ns *= g->radix[p];
    }
    // quads per workgroup: <= 512 threads, and only while staged segment + work buffer stay within half a CU's LDS
    const size_t per_quad = (size_t)16 * (n + 1);
    int lg = 0;
    while (lg < 4 && (th << (lg + 1)) <= 512 && (1 << (lg + 1)) <= 2 * nquads - 1 &&
           2 * per_quad * (2u << lg) <= (size_t)80 * 1024) ++lg;
    // the detrended segment is staged in LDS whenever it fits next to the work buffer (N <= 5000): measured 12-15 %
    // faster than re-reading it from L2 for every taper even where it halves the workgroups per CU
    mix_layout(g, lg, 2 * per_quad * (1u << lg) <= (size_t)160 * 1024 ? 1 : 0, nthreads, lds_bytes);
    return true;
}

}  // namespace spyfft
