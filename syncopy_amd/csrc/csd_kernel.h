// K4: taper- and trial-accumulated Hermitian rank-K update on the fp32 matrix cores
//
//     acc[f,i,j] += sum_r X[r,f,i] * conj(X[r,f,j])         (i-tile >= j-tile)
//
// Reference semantics: connectivity/csd.py:94-102 (outer product + taper mean) and
// the trial sum of computational_routine.py:1022-1032; the reference materialises a
// (K,F,C,C) temporary per trial, here the product only ever exists as MFMA
// accumulators.
//
// Work item = (frequency f, 32x32 channel tile (ti,tj), ti >= tj).  A workgroup of
// 4 waves takes 4*TPW consecutive items (for C=256: the 36 lower-triangle tiles of
// ONE frequency, 9 per wave); the rows X[r, f, :] of the frequencies it touches are
// staged through LDS in chunks of KB rows, interleaved complex exactly as in HBM, so
// one ds_read_b64 yields (re, im) of an operand.  Per tile and pair of rows:
//     re += Ar*Br ; re += Ai*Bi ; im += Ai*Br ; im += (-Ar)*Bi
// = 4 v_mfma_f32_32x32x2_f32 (exact fp32 FMA chains, 8 real flop per complex MAC).
#pragma once

#include "csd_args.h"

namespace spycsd {

__device__ __forceinline__ void tile_of(int tt, int& ti, int& tj) {
    // inverse of tt = ti*(ti+1)/2 + tj, tj <= ti
    int i = (int)((sqrtf(8.0f * (float)tt + 1.0f) - 1.0f) * 0.5f);
    while ((i + 1) * (i + 2) / 2 <= tt) ++i;
    while (i * (i + 1) / 2 > tt) --i;
    ti = i;
    tj = tt - i * (i + 1) / 2;
}

__device__ __forceinline__ int opaque_i(int v) { return spy_opaque(v); }

__device__ __forceinline__ void sched_fence_csd() { spy_sched_fence(); }

constexpr int CSD_PF = 8;    // staged float2 elements per thread and chunk (chunk <= 32 KiB of LDS)

// Four waves own TA tiles each, four own TB tiles each (TA >= TB): with (5,4) a workgroup
// covers the 36 lower-triangle tiles of one frequency at C=256 and every SIMD (waves w and
// w+4) carries 9 tiles.  5 x 32 accumulator registers per wave fit the AGPR file, so the
// MFMA chain never leaves it (9 tiles on ONE wave need 288 and made hipcc rotate accumulators
// through VGPRs inside the loop).
// One chunk of kb rows is fetched into registers (all loads in flight together) while the
// previous chunk is being multiplied, then written to LDS.
//
// FAST (host-selected: even C <= 256, row-major spectra, kb = 16; a 256-element LDS row holds nfb = 256 / C
// consecutive frequencies - one for C in (128, 256], two for C = 128, four for C = 64 ... - and the workgroup owns
// their nfb * ntiles <= 40 tiles, dealt round-robin to the waves): the matrix pipe
// and every other instruction of a SIMD's two waves share one issue port, so what is not an MFMA costs matrix
// time (ablations: staging instructions -12 %, loop overhead -12 %).  The fast path keeps the non-MFMA count
// minimal: 16-byte staging loads/stores addressed by a scalar base + one lane offset (4 + 4 instructions per
// chunk instead of ~250), the row-pair loop fully unrolled with every LDS fragment address = one register per
// tile + an immediate, the three LDS buffers reached by bumping those registers once per chunk.
// FAST == 1 tile ownership (C = 256: 8 channel blocks, 36 lower-triangle tiles per frequency): every wave owns ONE
// diagonal tile, always in slot 3, plus three off-diagonal tiles (slots 0-2) and - waves 0, 2, 5, 7 - a fifth one
// (slot 4).  The diagonal tile is Hermitian: only its 16 x 16 sub-blocks (0,0), (1,0), (1,1) are computed, with
// v_mfma_f32_16x16x4_f32 on four rows at a time (3/4 of the matrix work of a full 32 x 32 tile; the consumers of the
// accumulator only ever read elements with i >= j).  Entry = five (ti, tj) pairs, 3 bits each, slot t at bits 6t.
constexpr unsigned fast_pack(int a0, int b0, int a1, int b1, int a2, int b2, int a3, int b3, int a4, int b4) {
    return (unsigned)(a0 | (b0 << 3)) | ((unsigned)(a1 | (b1 << 3)) << 6) | ((unsigned)(a2 | (b2 << 3)) << 12) |
           ((unsigned)(a3 | (b3 << 3)) << 18) | ((unsigned)(a4 | (b4 << 3)) << 24);
}
__device__ __forceinline__ void fast_tile(int wave, int t, int& ti, int& tj) {
    unsigned e;
    switch (wave) {
        case 0: e = fast_pack(4, 2, 4, 0, 2, 0, 0, 0, 1, 0); break;
        case 1: e = fast_pack(6, 2, 6, 1, 2, 1, 1, 1, 0, 0); break;
        case 2: e = fast_pack(7, 3, 7, 0, 3, 0, 3, 3, 3, 2); break;
        case 3: e = fast_pack(5, 3, 5, 1, 3, 1, 5, 5, 0, 0); break;
        case 4: e = fast_pack(7, 4, 7, 1, 4, 1, 7, 7, 0, 0); break;
        case 5: e = fast_pack(6, 4, 6, 3, 4, 3, 4, 4, 5, 4); break;
        case 6: e = fast_pack(6, 5, 6, 0, 5, 0, 6, 6, 0, 0); break;
        default: e = fast_pack(7, 5, 7, 2, 5, 2, 2, 2, 7, 6); break;
    }
    e >>= 6 * t;
    ti = (int)(e & 7u);
    tj = (int)((e >> 3) & 7u);
}

// FAST: 0 generic path, 1 instruction-lean path with 36 tiles in every workgroup (C = 256: every wave has 4 or 5
// tiles, no per-tile guards in the loop), 2 instruction-lean path with any tile count per wave, 3 the same for even
// C in (256, 512]: 512-element LDS rows (8 rows per chunk), fast_nwgf workgroups share the tiles of one frequency
template <int TA, int TB, int FAST = 0>
__global__ void __launch_bounds__(CSD_THREADS) csd_accum_kernel(CsdArgs a) {
    SPY_DYN_SMEM(float2, X);   // 2 x [kb][rowlen]
    constexpr int PER = 4 * (TA + TB);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = spy_wave_index(tid);   // scalar: tile-count tests become s_cbranch, not exec masks
    const int l31 = lane & 31, lhi = lane >> 5;
    // waves {0,2,5,7} own TA tiles, {1,3,4,6} own TB: every SIMD carries TA+TB tiles whether the hardware places
    // waves w and w+4 or waves 2s and 2s+1 of a workgroup on the same SIMD
    constexpr unsigned BIG = FAST ? 0xA5u : 0x0Fu;   // (generic path: waves 0-3, one live register less)
    int ntile_w = ((BIG >> wave) & 1u) ? TA : TB;                            // wave-uniform
    const int first_w = wave * TB + __builtin_popcount(BIG & ((1u << wave) - 1u)) * (TA - TB);
    // FAST: tile v of the workgroup goes to the wave at position v % 8 of the order (0, 2, 5, 7, 1, 3, 4, 6) - the
    // waves that take one tile more when the count is not a multiple of 8 are spread over the four SIMDs
    const int wpos = ((BIG >> wave) & 1u) ? __builtin_popcount(BIG & ((1u << wave) - 1u))
                                          : 4 + __builtin_popcount(~BIG & ((1u << wave) - 1u));

    const int split = a.rows_per_split > 0 ? (int)blockIdx.y : 0;
    const long long row_lo = (long long)split * a.rows_per_split;
    if (row_lo >= a.nrows && split > 0) return;
    const long long nrows = (a.rows_per_split > 0 && row_lo + a.rows_per_split < a.nrows) ? a.rows_per_split
                                                                                          : a.nrows - row_lo;
    // Blocked spectra keep 4 neighbouring frequencies of a channel quad in one 128-byte line: give those 4
    // workgroups block ids that are congruent mod 8 (same XCD, same L2) and adjacent in dispatch order, so the
    // line is fetched from HBM once instead of by four L2s.
    long long wg = blockIdx.x;
    if (a.blocked) {
        const long long g32 = ((long long)gridDim.x >> 5) << 5;
        if (wg < g32) wg = (wg & ~31LL) + 4 * (wg & 7) + ((wg & 31) >> 3);
    }
    const int per_wg = FAST ? a.fast_per : PER;
    long long item0 = a.item_base + wg * per_wg;
    long long last = item0 + per_wg;
    if constexpr (FAST == 3) {
        // workgroup (f, k): tiles [k * per, (k + 1) * per) of frequency f, never across a frequency boundary
        const long long fq = a.item_base / a.ntiles + wg / a.fast_nwgf;
        item0 = fq * a.ntiles + (wg % a.fast_nwgf) * per_wg;
        last = item0 + per_wg;
        if (last > (fq + 1) * a.ntiles) last = (fq + 1) * a.ntiles;
    }
    if (last > a.item_end) last = a.item_end;
    if (item0 >= last) return;
    if constexpr (FAST) {
        const int nv = (int)(last - item0);
        ntile_w = nv > wpos ? (nv - wpos + 7) >> 3 : 0;
    }
    // item index of this wave's tile t
    auto item_of = [&](int t) { return FAST ? item0 + wpos + 8 * t : item0 + first_w + t; };
    const int f_lo = (int)(item0 / a.ntiles);
    const int nfb = (int)((last - 1) / a.ntiles) - f_lo + 1;
    const int rowlen = nfb * a.cpad;

    int aoff[TA], boff[TA];
    f32x16 accr[TA], acci[TA];
#pragma unroll
    for (int t = 0; t < TA; ++t) {
        const long long item = item_of(t);
        int f = f_lo, ti = 0, tj = 0;
        if constexpr (FAST == 1) {
            fast_tile(wave, t, ti, tj);
        } else if (t < ntile_w && item < a.item_end) {
            f = (int)(item / a.ntiles);
            tile_of((int)(item % a.ntiles), ti, tj);
        }
        // FAST packs the frequencies of a row at their distance in memory (C, not the padded tile width): a tile's
        // columns beyond C belong to the next frequency and only feed accumulator entries that are never stored
        const int fstride = FAST ? a.C : a.cpad;
        aoff[t] = (f - f_lo) * fstride + ti * 32 + l31;
        boff[t] = (f - f_lo) * fstride + tj * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            accr[t][r] = 0.f;
            acci[t][r] = 0.f;
        }
    }

    if constexpr (FAST) {
        constexpr int ROWLEN = FAST == 3 ? 512 : 256, KB = 4096 / ROWLEN, CHUNK = KB * ROWLEN;   // 32 KiB per LDS buffer
        constexpr int TPR = ROWLEN / 2, RSTEP = CSD_THREADS / TPR;  // threads per row (16 bytes each), rows per pass
        constexpr int NPRE = 2;
        float4* const X4 = reinterpret_cast<float4*>(X);
        const size_t rowstride = (size_t)a.F * a.C;                 // float2 elements between rows
        const unsigned rowbytes = (unsigned)rowstride * 8u;
        const char* const fb = reinterpret_cast<const char*>(a.spec + (size_t)row_lo * rowstride + (size_t)f_lo * a.C);
        const int srow = tid / TPR;                                 // this thread stages rows srow + RSTEP v, v < 4
        // ... columns 2*(tid % TPR), +1 (16 bytes) of the row = the (up to ROWLEN) elements of this workgroup's
        // frequencies; columns past them are zero-filled (their load is redirected to column 0)
        const int nf_wg = (int)((last - 1) / a.ntiles) - f_lo + 1;
        const bool colok = 2 * (tid % TPR) < nf_wg * a.C;
        const unsigned coff = colok ? (unsigned)(tid % TPR) * 16u : 0u;
        // odd C: rows are 8-byte aligned only and the last pair of an odd-length row has no second element - the
        // pair is fetched as two 8-byte loads, the missing element redirected to column 0 and zeroed
        const bool odd = FAST != 1 && (a.C & 1);
        const bool hiok = 2 * (tid % TPR) + 1 < nf_wg * a.C;
        const unsigned coff_hi = hiok ? coff + 8u : 0u;
        const long long nchunk = (nrows + KB - 1) / KB;
        float4 pf[4];
        unsigned okmask = 0;
        auto fetch = [&](long long c) {
            const long long r0 = c * KB;
            const long long left = nrows - r0;
            const int rleft = left < KB ? (int)left : KB;           // >= 1 valid rows in this chunk
            const char* base = fb + (size_t)r0 * rowbytes;          // wave-uniform
            okmask = 0;
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const int row = srow + RSTEP * v;
                const int rc = row < rleft ? row : rleft - 1;       // clamped: the load is unconditional
                if (odd) {
                    const float2 lo = *reinterpret_cast<const float2*>(base + ((unsigned)rc * rowbytes + coff));
                    float2 hi = *reinterpret_cast<const float2*>(base + ((unsigned)rc * rowbytes + coff_hi));
                    if (!hiok) hi = make_float2(0.f, 0.f);
                    pf[v] = make_float4(lo.x, lo.y, hi.x, hi.y);
                } else {
                    pf[v] = *reinterpret_cast<const float4*>(base + ((unsigned)rc * rowbytes + coff));
                }
                okmask |= (row < rleft && colok) ? (1u << v) : 0u;
            }
        };
        auto put = [&](int buf) {
#pragma unroll
            for (int v = 0; v < 4; ++v)
                X4[buf * (CHUNK / 2) + tid + CSD_THREADS * v] = ((okmask >> v) & 1u) ? pf[v] : make_float4(0.f, 0.f, 0.f, 0.f);
        };
        fetch(0);
        put(0);
        if (nchunk > 1) {
            fetch(1);
            put(1);
        }
        if (nchunk > 2) fetch(2);
        __syncthreads();

        // byte address (inside X) of this lane's A / B fragment of row pair 0 in the CURRENT buffer
        unsigned aA[TA], aB[TA];
#pragma unroll
        for (int t = 0; t < TA; ++t) {
            aA[t] = (unsigned)(lhi * ROWLEN + aoff[t]) * 8u;
            aB[t] = (unsigned)(lhi * ROWLEN + boff[t]) * 8u;
        }
        const char* const Xb = reinterpret_cast<const char*>(X);
        auto lds2 = [&](unsigned addr, int imm) { return *reinterpret_cast<const float2*>(Xb + addr + imm); };
        // FAST == 1: the diagonal tile (slot 3) in 16 x 16 x 4 fragments - lane l holds channel (l & 15) of sub-block
        // 0 / 1 of the block and row (l >> 4) of a group of four rows
        constexpr int DG = 3;
        unsigned dA = (unsigned)((lane >> 4) * ROWLEN + (aoff[DG] - l31) + (lane & 15)) * 8u;
        f32x4 dre[3], dim[3];                                        // sub-blocks (0,0), (1,0), (1,1)
#pragma unroll
        for (int q = 0; q < 3; ++q)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                dre[q][r] = 0.f;
                dim[q][r] = 0.f;
            }
        auto mfma4d = [&](int q, float2 av, float2 bv) {
            dre[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.x, bv.x, dre[q], 0, 0, 0);
            dim[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.x, dim[q], 0, 0, 0);
            dre[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(av.y, bv.y, dre[q], 0, 0, 0);
            dim[q] = __builtin_amdgcn_mfma_f32_16x16x4f32(-av.x, bv.y, dim[q], 0, 0, 0);
        };
        auto mfma4 = [&](int t, float2 av, float2 bv) {
            accr[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, accr[t], 0, 0, 0);
            acci[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.x, acci[t], 0, 0, 0);
            accr[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, accr[t], 0, 0, 0);
            acci[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(-av.x, bv.y, acci[t], 0, 0, 0);
        };
        float2 pa[NPRE], pb[NPRE];
#pragma unroll
        for (int t = 0; t < NPRE; ++t) {
            pa[t] = lds2(aA[t], 0);
            pb[t] = lds2(aB[t], 0);
        }
        int b0 = 0;
        for (long long c = 0; c < nchunk; ++c) {
            const int b2 = b0 == 0 ? 2 : b0 - 1;                    // (b0 + 2) % 3
            if (c + 2 < nchunk) {
                put(b2);
                if (c + 3 < nchunk) fetch(c + 3);
            }
#pragma unroll
            for (int st = 0; st < KB / 2; ++st) {
                constexpr int RP = 2 * ROWLEN * 8;                  // bytes per row pair
                float2 av[TA], bv[TA];
#pragma unroll
                for (int t = 0; t < NPRE; ++t) {
                    av[t] = pa[t];
                    bv[t] = pb[t];
                }
#pragma unroll
                for (int t = NPRE; t < TA; ++t) {
                    if (FAST == 1 && t == DG) continue;
                    av[t] = lds2(aA[t], st * RP);
                    bv[t] = lds2(aB[t], st * RP);
                }
                float2 d0 = make_float2(0.f, 0.f), d1 = d0;
                if (FAST == 1 && (st & 1)) {                        // rows 2 (st - 1) .. 2 st + 1 of the chunk
                    d0 = lds2(dA, (st - 1) * RP);
                    d1 = lds2(dA, (st - 1) * RP + 128);
                }
                sched_fence_csd();
#pragma unroll
                for (int t = 0; t < NPRE; ++t)
                    if (FAST == 1 || t < ntile_w) mfma4(t, av[t], bv[t]);
                sched_fence_csd();
                if (st + 1 < KB / 2) {
#pragma unroll
                    for (int t = 0; t < NPRE; ++t) {
                        pa[t] = lds2(aA[t], (st + 1) * RP);
                        pb[t] = lds2(aB[t], (st + 1) * RP);
                    }
                } else {
                    // last row pair of the chunk: move every fragment address to the next buffer (+32 KiB, or
                    // back by 64 KiB), then fetch the first fragments of the next chunk from there
                    const int delta = (b0 == 2) ? -2 * CHUNK * 8 : CHUNK * 8;
#pragma unroll
                    for (int t = 0; t < TA; ++t) {
                        aA[t] += (unsigned)delta;
                        aB[t] += (unsigned)delta;
                    }
                    dA += (unsigned)delta;
                    if (c + 1 < nchunk) {
#pragma unroll
                        for (int t = 0; t < NPRE; ++t) {
                            pa[t] = lds2(aA[t], 0);
                            pb[t] = lds2(aB[t], 0);
                        }
                    }
                }
                sched_fence_csd();
#pragma unroll
                for (int t = NPRE; t < TA; ++t) {
                    if ((FAST != 1 || t >= TB) && t >= ntile_w) break;
                    if (FAST == 1 && t == DG) continue;
                    mfma4(t, av[t], bv[t]);
                }
                if (FAST == 1 && (st & 1)) {
                    mfma4d(0, d0, d0);
                    mfma4d(1, d1, d0);
                    mfma4d(2, d1, d1);
                }
            }
            __syncthreads();
            b0 = b0 == 2 ? 0 : b0 + 1;
        }
        if constexpr (FAST == 1) {
            // diagonal tile: lane l holds column (l & 15) and rows 4 (l >> 4) + r of each 16 x 16 sub-block
            float2* const base = a.acc + (size_t)f_lo * a.C * a.C;
            const int g0 = aoff[DG] - l31;                           // first channel of the diagonal block
#pragma unroll
            for (int q = 0; q < 3; ++q) {
                const int ia = (q >= 1) ? 16 : 0, jb = (q == 2) ? 16 : 0;
                float2* const pb = base + (size_t)(g0 + ia + 4 * (lane >> 4)) * a.C + (g0 + jb + (lane & 15));
                float2 old[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) old[r] = pb[(size_t)r * a.C];
#pragma unroll
                for (int r = 0; r < 4; ++r) pb[(size_t)r * a.C] = make_float2(old[r].x + dre[q][r], old[r].y + dim[q][r]);
            }
        }
    } else {
    // element i of this thread = LDS slot u = tid + 512*i = (row kr, column cc) of the chunk
    const int total = a.kb * rowlen;                       // <= 512 * CSD_PF
    const int step_q = CSD_THREADS / rowlen, step_r = CSD_THREADS % rowlen;
    const int kr0 = tid / rowlen, cc0 = tid % rowlen;
    // float2 elements between rows r and r+1 (blocked: quads are padded to 4 channels)
    const size_t rowstride = a.blocked ? (size_t)a.F * (size_t)(((a.C + 3) >> 2) << 2) : (size_t)a.F * a.C;
    const unsigned rowbytes = (unsigned)rowstride * 8u;    // a chunk spans < 4 GiB: 32-bit lane offsets
    const float2* fbase = a.spec + (size_t)row_lo * rowstride + (a.blocked ? (size_t)f_lo * 4 : (size_t)f_lo * a.C);

    float2 pf[CSD_PF];
    unsigned okmask = 0;
    auto fetch = [&](long long r0) {
        okmask = 0;
        // opaque: the (row, column) walk is recomputed per chunk instead of being hoisted out of the
        // row loop into ~100 long-lived VGPRs (offsets + predicates of all CSD_PF elements)
        int kr = opaque_i(kr0), cc = opaque_i(cc0);
        const char* base = reinterpret_cast<const char*>(fbase + (size_t)r0 * rowstride);   // wave-uniform
        const long long rleft = nrows - r0;
#pragma unroll
        for (int i = 0; i < CSD_PF; ++i) {
            // branch-free: clamp to a valid element, load unconditionally, select afterwards - a load
            // inside a conditional block gets its own s_waitcnt vmcnt(0) and the CSD_PF round trips to
            // HBM would run back to back instead of together
            int fb = 0, c = cc;
            if (nfb > 1) {
                fb = cc / a.cpad;
                c = cc - fb * a.cpad;
            }
            const bool ok = (tid + CSD_THREADS * i < total) && (kr < rleft) && (c < a.C);
            const int krc = (int)(kr < rleft ? kr : rleft - 1);
            const int fbc = fb < nfb ? fb : nfb - 1;
            const int cc_c = c < a.C ? c : a.C - 1;
            // standard: row-major (r, f, c); blocked: (r, c/4, f, c%4) with 32-bit offsets from the chunk base
            const unsigned eoff = a.blocked ? ((unsigned)(cc_c >> 2) * (unsigned)a.F + (unsigned)fbc) * 32u + (unsigned)(cc_c & 3) * 8u
                                            : (unsigned)(fbc * a.C + cc_c) * 8u;
            pf[i] = *reinterpret_cast<const float2*>(base + ((unsigned)krc * rowbytes + eoff));
            okmask |= ok ? (1u << i) : 0u;      // the zero-fill select happens at LDS-write time: no early wait
            cc += step_r;
            kr += step_q;
            if (cc >= rowlen) {
                cc -= rowlen;
                kr += 1;
            }
        }
    };

    // LDS holds THREE chunks.  Iteration c multiplies chunk c out of buffer c%3 while chunk c+2 (fetched
    // into registers during iteration c-1) is written into buffer (c+2)%3 and chunk c+3 is being fetched:
    //  - write-after-read: buffer (c+2)%3 was last read in iteration c-1, behind that iteration's barrier;
    //  - read-after-write: buffer (c+1)%3 was written at the top of iteration c-1, also behind that barrier,
    // so the operand fragments of the NEXT row pair - including the first pair of chunk c+1 - are read
    // while the MFMA chain of the current pair runs and no wave ever waits on LDS latency; one barrier
    // per chunk remains.  (buffers addressed as X[b*total + ...]: a pointer array decays to flat addressing)
    const long long nchunk = (nrows + a.kb - 1) / a.kb;
    fetch(0);
#pragma unroll
    for (int i = 0; i < CSD_PF; ++i)
        if (tid + CSD_THREADS * i < total)
            X[tid + CSD_THREADS * i] = ((okmask >> i) & 1u) ? pf[i] : make_float2(0.f, 0.f);
    if (nchunk > 1) {
        fetch(a.kb);
#pragma unroll
        for (int i = 0; i < CSD_PF; ++i)
            if (tid + CSD_THREADS * i < total)
                X[total + tid + CSD_THREADS * i] = ((okmask >> i) & 1u) ? pf[i] : make_float2(0.f, 0.f);
    }
    if (nchunk > 2) fetch(2LL * a.kb);
    __syncthreads();

    // Operand fragments of the first NPRE tiles of the NEXT row pair are read while the current pair's
    // MFMA chain runs (registers for all TA tiles of two row pairs do not fit beside the accumulators):
    // their MFMAs (>= 512 cycles) then cover the LDS latency of the remaining tiles' reads.
    constexpr int NPRE = TA >= 4 ? 2 : 1;
    float2 pa[NPRE], pb[NPRE];
    auto prefetch = [&](int base, int ks) {
        const float2* xr = X + (base + (ks + lhi) * rowlen);
#pragma unroll
        for (int t = 0; t < NPRE; ++t) {
            pa[t] = xr[aoff[t]];
            pb[t] = xr[boff[t]];
        }
    };
    auto mfma4 = [&](int t, float2 av, float2 bv) {
        accr[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, accr[t], 0, 0, 0);
        acci[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.x, acci[t], 0, 0, 0);
        accr[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, accr[t], 0, 0, 0);
        acci[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(-av.x, bv.y, acci[t], 0, 0, 0);
    };

    prefetch(0, 0);
    int b0 = 0;                                            // buffer of the current chunk (c % 3)
    for (long long c = 0; c < nchunk; ++c) {
        const int b1 = (b0 == 2) ? 0 : b0 + 1, b2 = (b1 == 2) ? 0 : b1 + 1;
        bool stage = c + 2 < nchunk;
#ifdef CSD_DBG_NOSTAGE
        stage = false;
#endif
        const int cur = b0 * total, nx1 = b1 * total;
        for (int ks = 0; ks < a.kb; ks += 2) {
            if (stage && ks == 0) {
                const int nxt = b2 * total;
#pragma unroll
                for (int i = 0; i < CSD_PF; ++i)
                    if (tid + CSD_THREADS * i < total)
                        X[nxt + tid + CSD_THREADS * i] = ((okmask >> i) & 1u) ? pf[i] : make_float2(0.f, 0.f);
                if (c + 3 < nchunk) fetch((c + 3) * a.kb);
            }
            const float2* xr = X + (cur + (ks + lhi) * rowlen);
            float2 av[TA], bv[TA];
#pragma unroll
            for (int t = 0; t < NPRE; ++t) {
                av[t] = pa[t];
                bv[t] = pb[t];
            }
#pragma unroll
            for (int t = NPRE; t < TA; ++t) {
                av[t] = xr[aoff[t]];
                bv[t] = xr[boff[t]];
            }
            sched_fence_csd();      // keep the reads ahead of the MFMA chain (hipcc otherwise re-sinks them)
#pragma unroll
            for (int t = 0; t < NPRE; ++t) mfma4(t, av[t], bv[t]);
            sched_fence_csd();
            if (ks + 2 < a.kb) prefetch(cur, ks + 2);
            else if (c + 1 < nchunk) prefetch(nx1, 0);
            sched_fence_csd();
#pragma unroll
            for (int t = NPRE; t < TA; ++t) {
                if (t >= TB && t >= ntile_w) break;           // half of the waves own TB tiles only
                mfma4(t, av[t], bv[t]);
            }
        }
#ifndef CSD_DBG_NOBARRIER
        __syncthreads();
#endif
        b0 = b1;
    }

    }

    // ---- acc += tile (each (f, tile) is owned by exactly one wave: plain read-modify-write)
#pragma unroll
    for (int t = 0; t < TA; ++t) {
        const long long item = item_of(t);
        if (t >= ntile_w || (FAST != 1 && item >= a.item_end)) continue;
        if (FAST == 1 && t == 3) continue;                          // the diagonal tile was written above
        int f = f_lo, ti, tj;
        if constexpr (FAST == 1) {
            fast_tile(wave, t, ti, tj);
        } else {
            f = (int)(item / a.ntiles);
            tile_of((int)(item % a.ntiles), ti, tj);
        }
        const int j = tj * 32 + l31;
        // read-modify-write of the 16 rows this lane holds: all loads first (clamped, branch-free), then stores
        const int jc = j < a.C ? j : a.C - 1;
        float2* const pbase = (split == 0 ? a.acc + (size_t)f * a.C * a.C
                                          : a.part + ((size_t)(split - 1) * a.part_nf + (f - a.part_f0)) * a.C * a.C) + jc;
        if (split == 0) {
            float2 old[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                old[r] = pbase[(size_t)(i < a.C ? i : a.C - 1) * a.C];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                if (i < a.C && j < a.C) pbase[(size_t)i * a.C] = make_float2(old[r].x + accr[t][r], old[r].y + acci[t][r]);
            }
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int i = ti * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
                if (i < a.C && j < a.C) pbase[(size_t)i * a.C] = make_float2(accr[t][r], acci[t][r]);
            }
        }
    }
}

// acc[f, i, j] += sum_s part[s][f - f0][i][j] over the lower-triangle tiles (fixed order: deterministic)
__global__ void __launch_bounds__(256) csd_reduce_parts_kernel(float2* acc, const float2* part, int nparts, int f0, int nf,
                                                               int C) {
    const long long n = (long long)nf * C * C;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) {
        const int j = (int)(e % C);
        const int i = (int)((e / C) % C);
        if ((i >> 5) < (j >> 5)) continue;
        float2 v = acc[(long long)f0 * C * C + e];
        for (int s = 0; s < nparts; ++s) {
            const float2 p = part[(long long)s * n + e];
            v.x += p.x;
            v.y += p.y;
        }
        acc[(long long)f0 * C * C + e] = v;
    }
}

// lower triangle (i >= j) of acc (F, C, C) <-> packed (F, C(C+1)/2): the accumulator only carries the lower
// triangle before csd_finalize, so the multi-GPU all-reduce ships 0.54 GB instead of 1.07 GB at C = 256
template <bool UNPACK>
__global__ void __launch_bounds__(256) csd_tril_kernel(float2* acc, float2* packed, int F, int C) {
    const long long n = (long long)F * C * C;
    const long long ntri = (long long)C * (C + 1) / 2;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) {
        const int j = (int)(e % C);
        const long long fi = e / C;
        const int i = (int)(fi % C);
        if (j > i) continue;
        const long long p = (fi / C) * ntri + (long long)i * (i + 1) / 2 + j;
        if (UNPACK) acc[e] = packed[p];
        else packed[p] = acc[e];
    }
}

// acc[f,i,j] *= scale (i >= j), zero the diagonal's imaginary part, mirror to the upper triangle.
// One workgroup per (frequency, lower-triangle 32 x 32 tile); the mirror goes through an LDS transpose so that
// both triangles are written row-wise.
__global__ void __launch_bounds__(256) csd_finalize_kernel(float2* acc, int F, int C, float scale) {
    __shared__ float2 tile[32][33];
    const int nt = (C + 31) / 32, ntiles = nt * (nt + 1) / 2;
    const long long item = blockIdx.x;
    const int f = (int)(item / ntiles);
    int ti, tj;
    tile_of((int)(item % ntiles), ti, tj);
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    float2* A = acc + (size_t)f * C * C;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = ty + 8 * k, c = tx;
        const int i = ti * 32 + r, j = tj * 32 + c;
        float2 up = make_float2(0.f, 0.f);
        if (i < C && j < C && j <= i) {
            float2 v = A[(size_t)i * C + j];
            v.x *= scale;
            v.y = (i == j) ? 0.f : v.y * scale;
            A[(size_t)i * C + j] = v;
            up = make_float2(v.x, -v.y);
        }
        tile[c][r] = up;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = ty + 8 * k, c = tx;                        // element (r, c) of the mirrored tile (tj, ti)
        const int i = tj * 32 + r, j = ti * 32 + c;
        if (i < C && j < C && j > i) A[(size_t)i * C + j] = tile[r][c];
    }
}

// K5: coherency = csd / sqrt(S_ii * S_jj), then the output conversion (csd.py:118-172)
template <bool CPLX>
__global__ void __launch_bounds__(256) coh_normalize_kernel(const float2* csd, int F, int C, int kind, void* out) {
    const long long n = (long long)F * C * C;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) {
        const int j = (int)(e % C);
        const long long fi = e / C;
        const int i = (int)(fi % C);
        const long long fbase = (fi - i) * C;   // f*C*C
        const float di = csd[fbase + (long long)i * C + i].x;
        const float dj = csd[fbase + (long long)j * C + j].x;
        const float s = sqrtf(di * dj);
        const float2 v = csd[e];
        const float2 c = make_float2(v.x / s, v.y / s);
        if (CPLX) {
            reinterpret_cast<float2*>(out)[e] = c;
        } else {
            float o;
            switch (kind) {
                case SPYHIP_OUT_POW: o = c.x * c.x + c.y * c.y; break;
                case SPYHIP_OUT_ABS: o = sqrtf(c.x * c.x + c.y * c.y); break;
                case SPYHIP_OUT_REAL: o = c.x; break;
                case SPYHIP_OUT_IMAG: o = c.y; break;
                case SPYHIP_OUT_ANGLE: o = atan2f(c.y, c.x); break;
                case SPYHIP_OUT_ABSREAL: o = fabsf(c.x); break;
                default: o = fabsf(c.y); break;
            }
            reinterpret_cast<float*>(out)[e] = o;
        }
    }
}

// K5 fused: coherence straight from the raw lower-triangle accumulator - scale, normalise, convert and mirror in
// one pass (reads 0.54 GB + writes 0.54 GB at C = 256 instead of the 3.2 GB of csd_finalize + coh_normalize).
// Same float operations in the same order as the two-step path: bit-identical results.
__device__ __forceinline__ float coh_convert(float2 c, int kind) {
    switch (kind) {
        case SPYHIP_OUT_POW: return c.x * c.x + c.y * c.y;
        case SPYHIP_OUT_ABS: return sqrtf(c.x * c.x + c.y * c.y);
        case SPYHIP_OUT_REAL: return c.x;
        case SPYHIP_OUT_IMAG: return c.y;
        case SPYHIP_OUT_ANGLE: return atan2f(c.y, c.x);
        case SPYHIP_OUT_ABSREAL: return fabsf(c.x);
        default: return fabsf(c.y);
    }
}
// One workgroup per (frequency, lower-triangle 32 x 32 tile): the tile is read row-wise, its coherence written
// row-wise, and the Hermitian mirror goes through an LDS transpose so that it is written row-wise too (a direct
// mirror store scatters 4-byte writes over C rows and costs more than the whole rest of the pass).
template <bool CPLX>
__global__ void __launch_bounds__(256) coh_from_acc_kernel(const float2* acc, int F, int C, float scale, int kind, void* out) {
    __shared__ float2 tile[32][33];
    __shared__ float drow[32], dcol[32];
    const int nt = (C + 31) / 32, ntiles = nt * (nt + 1) / 2;
    const long long item = blockIdx.x;
    const int f = (int)(item / ntiles);
    int ti, tj;
    tile_of((int)(item % ntiles), ti, tj);
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const float2* A = acc + (size_t)f * C * C;
    if (threadIdx.x < 32) {
        const int i = ti * 32 + threadIdx.x;
        drow[threadIdx.x] = i < C ? A[(size_t)i * C + i].x * scale : 1.f;
    } else if (threadIdx.x < 64) {
        const int j = tj * 32 + threadIdx.x - 32;
        dcol[threadIdx.x - 32] = j < C ? A[(size_t)j * C + j].x * scale : 1.f;
    }
    __syncthreads();
    const size_t obase = (size_t)f * C * C;
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = ty + 8 * k, c = tx;
        const int i = ti * 32 + r, j = tj * 32 + c;
        float2 up = make_float2(0.f, 0.f);
        if (i < C && j < C && j <= i) {                          // the accumulator's lower triangle
            float2 v = A[(size_t)i * C + j];
            v.x *= scale;
            v.y = (i == j) ? 0.f : v.y * scale;
            const float s = sqrtf(drow[r] * dcol[c]);
            const float2 lo = make_float2(v.x / s, v.y / s);
            up = make_float2(v.x / s, -v.y / s);                 // (f, j, i) = conjugate
            if (CPLX) reinterpret_cast<float2*>(out)[obase + (size_t)i * C + j] = lo;
            else reinterpret_cast<float*>(out)[obase + (size_t)i * C + j] = coh_convert(lo, kind);
        }
        tile[c][r] = up;
    }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int r = ty + 8 * k, c = tx;                        // element (r, c) of the mirrored tile (tj, ti)
        const int i = tj * 32 + r, j = ti * 32 + c;
        if (i < C && j < C && j > i) {                           // strictly upper part only
            const float2 u = tile[r][c];
            if (CPLX) reinterpret_cast<float2*>(out)[obase + (size_t)i * C + j] = u;
            else reinterpret_cast<float*>(out)[obase + (size_t)i * C + j] = coh_convert(u, kind);
        }
    }
}

}  // namespace spycsd
