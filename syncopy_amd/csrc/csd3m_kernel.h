// K4 for 256 channels with the 3-multiplication complex product (Karatsuba / "3M"):
//
//     acc[f,i,j] += sum_r X[r,f,i] * conj(X[r,f,j])                                  (i >= j at 16-channel granularity)
//     P1 = sum Ar Br,  P2 = sum Ai Bi,  P3 = sum (Ar + Ai)(Br - Bi)     (A = X[.,i], B = X[.,j])
//     re = P1 + P2,    im = P3 - P1 + P2
//
// = 3 matrix instructions per sub-tile and group of rows instead of the 4 of csd_kernel.h: a quarter less matrix work
// for the same 8 algorithmic flop per complex MAC.  Reference semantics as csd_kernel.h
// (connectivity/csd.py:94-102 + the trial sum of shared/computational_routine.py:1022-1032).
//
// One 512-thread workgroup per frequency, two waves per SIMD.  The unit of work is the 16 x 16 sub-tile on
// v_mfma_f32_16x16x4_f32: the lower triangle of the 16 x 16 block matrix has 136 sub-tiles = 8 waves x 17, so every
// wave carries exactly 51 accumulators (204 registers) and the same matrix work (the 16 diagonal sub-tiles are
// computed in full: 6 % of the work lands above the diagonal, where nobody reads it).  Every wave's 17 sub-tiles touch
// 8 of the 16 channel blocks (M3_BLK): 8 LDS fragment reads + <= 13 additions feed 51 MFMAs of 32 cycles -
// everything that is not a matrix instruction issues in their shadow.  The workgroup stages the whole rows
// X[r, f, :] (2 KiB each) global -> LDS directly (global_load_lds_dwordx4: no staging registers, no ds_write pass)
// into three 16-row buffers: iteration c multiplies chunk c while chunk c+2 lands.  Rows past the end of a ragged
// last chunk are zero-filled with plain LDS stores.
//
// Since round 5 the 256-channel, row-major case runs csdh_kernel.h (half-precision matrix cores, split operands); this
// kernel serves every other channel count, the channel-blocked layout, and the frequencies csdh_kernel declines
// (CsdArgs::only_flagged).
//
// The imaginary part of a diagonal element is P3 - P1 + P2 of rounded sums, i.e. rounding noise instead of the
// exact zero of X conj(X): csd_finalize_kernel / coh_from_acc_kernel (the readers of the diagonal) set it to zero,
// exactly as they do for the other kernels.
#pragma once
#include <type_traits>

#include "spy_intrinsics.h"

#define SPY_M3_KATTR(WPG) SPY_WAVES_PER_EU((WPG) / 4, (WPG) / 4)

namespace spycsd {

// compile-time loop: f(std::integral_constant<int, I>) for I in [I0, N) - tile / block indices stay constants
template <int I, int N, typename F>
__device__ __forceinline__ void m3_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        m3_for<I + 1, N>(f);
    }
}

__device__ __forceinline__ void m3_sched_fence() { spy_sched_fence(); }

constexpr int M3_CHUNK_BYTES = 32 * 1024;            // per buffer (16 rows of 256 channels or 8 rows of 512), three buffers
constexpr int M3_LDS_BYTES = 3 * M3_CHUNK_BYTES;
constexpr int M3_TILES_PER_F = 36;                   // 32 x 32 tiles per frequency in CsdArgs' item units (256 channels)

// ---- ownership of the lower-triangle 16 x 16 sub-tiles.  CH channels per frequency: one 256-element LDS row holds
// 256 / CH consecutive frequencies of a row of spectra (they are contiguous in memory: (rows, F, C) with C = CH), so
// "block" b = 0 ... 15 is channel block b % (CH / 16) of frequency (packed row) * (256 / CH) + b / (CH / 16).
// Wave g's sub-tile t is (row block BLK[g][TA[g][t]], column block BLK[g][TB[g][t]]), row block >= column block.
// Primary template: tables generated at compile time for any CH = 16 nb <= 256 (FPR = 256 / CH frequencies per
// row): the sub-tiles are listed frequency by frequency, 4 x 4 super-tile by super-tile (so that consecutive ones share
// channel blocks) and cut into 8 runs of NT; wave g's run may be shorter (CNT[g]).  Hand-made tables below for the
// counts that matter most.
// RECT (CH = 512 only): the 16 x 16 sub-tiles of the off-diagonal RECTANGLE blocks 16 ... 31 (rows) x 0 ... 15 (columns)
// of a 512-element image whose two halves hold two different 256-channel ranges of a wider recording: the piece of the
// lower triangle between two channel blocks (more than 512 channels, csd.hip).
template <int CH, bool RECT = false>
struct M3Tab {
    static_assert(CH % 16 == 0 && CH >= 16 && CH <= 512, "channel count of the generated 3M tables");
    static_assert(!RECT || CH == 512, "the rectangle tables are those of the 512-element image");
    // up to 256 channels: 256-element LDS rows holding 256 / CH frequencies, 16-row chunks, one workgroup per packed
    // row; 272 ... 512 channels: 512-element rows (one frequency), 8-row chunks (the same 32 KiB per buffer), and the
    // sub-tiles of a frequency shared by NP workgroups that each stage the whole rows (<= 14 sub-tiles per wave there:
    // the fragments of up to 16 distinct channel blocks need registers too)
    static constexpr int ROWLEN = CH <= 256 ? 256 : 512, KB = CH <= 256 ? 16 : 8;
    static constexpr int BPF = CH / 16, FPR = ROWLEN / CH, NSUB = RECT ? 256 : BPF * (BPF + 1) / 2, NTOT = FPR * NSUB;
    static constexpr int NP = CH <= 256 ? 1 : (NTOT + 111) / 112;
    static constexpr int NW = 8 * NP;
    static constexpr int NT = (NTOT + NW - 1) / NW;
    struct Gen {
        int blk[NW][32];
        int nb[NW];
        int ta[NW][NT];
        int tb[NW][NT];
        int cnt[NW];
        int nbmax;
    };
    static constexpr Gen make() {
        Gen g{};
        int seq_r[NTOT > 0 ? NTOT : 1] = {}, seq_c[NTOT > 0 ? NTOT : 1] = {};
        int n = 0;
        for (int d = 0; d < FPR; ++d)
            for (int I = (RECT ? 4 : 0); I < (BPF + 3) / 4; ++I)
                for (int J = 0; J <= (RECT ? 3 : I); ++J)
                    for (int r = 4 * I; r < 4 * I + 4 && r < BPF; ++r)
                        for (int c = 4 * J; c < 4 * J + 4 && c <= r; ++c) {
                            seq_r[n] = d * BPF + r;
                            seq_c[n] = d * BPF + c;
                            ++n;
                        }
        g.nbmax = 1;
        for (int w = 0; w < NW; ++w) {
            g.nb[w] = 0;
            g.cnt[w] = 0;
            for (int i = 0; i < 32; ++i) g.blk[w][i] = 0;
            for (int t = 0; t < NT; ++t) { g.ta[w][t] = 0; g.tb[w][t] = 0; }
            for (int t = 0; t < NT; ++t) {
                const int k = w * NT + t;
                if (k >= NTOT) break;
                int ia = -1, ib = -1;
                for (int i = 0; i < g.nb[w]; ++i) {
                    if (g.blk[w][i] == seq_r[k]) ia = i;
                    if (g.blk[w][i] == seq_c[k]) ib = i;
                }
                if (ia < 0) { ia = g.nb[w]; g.blk[w][g.nb[w]++] = seq_r[k]; }
                if (seq_c[k] == seq_r[k]) ib = ia;
                if (ib < 0) { ib = g.nb[w]; g.blk[w][g.nb[w]++] = seq_c[k]; }
                g.ta[w][t] = ia;
                g.tb[w][t] = ib;
                g.cnt[w] = t + 1;
            }
            if (g.nb[w] > g.nbmax) g.nbmax = g.nb[w];
        }
        return g;
    }
    static constexpr Gen T = make();
    static constexpr int NB = T.nbmax;
    static constexpr int blk(int g, int i) { return T.blk[g][i]; }
    static constexpr int ta(int g, int t) { return T.ta[g][t]; }
    static constexpr int tb(int g, int t) { return T.tb[g][t]; }
    static constexpr int cnt(int g) { return T.cnt[g]; }
};

// 256 channels: 136 sub-tiles = 8 waves x 17, every wave touches 8 blocks
//   g 0-3: the four 4 x 4 squares of rows 8-15 x columns 0-7, plus one sub-tile of a triangle each
//   g 4, 5: the squares rows 4-7 x columns 0-3 and rows 12-15 x columns 8-11, plus one
//   g 6, 7: the lower triangles of blocks {0-3}, {4-7} and of {8-11}, {12-15}, minus the six given away
template <>
struct M3Tab<256> {
    static constexpr int ROWLEN = 256, KB = 16, NP = 1, NW = 8;
    static constexpr int NT = 17, NB = 8;
    static constexpr int BLK[8][8] = {
        { 0,  1,  2,  3,  8,  9, 10, 11},
        { 4,  5,  6,  7,  8,  9, 10, 11},
        { 0,  1,  2,  3, 12, 13, 14, 15},
        { 4,  5,  6,  7, 12, 13, 14, 15},
        { 0,  1,  2,  3,  4,  5,  6,  7},
        { 8,  9, 10, 11, 12, 13, 14, 15},
        { 0,  1,  2,  3,  4,  5,  6,  7},
        { 8,  9, 10, 11, 12, 13, 14, 15},
    };
    static constexpr int TA[8][17] = {
        { 4,  4,  4,  4,  5,  5,  5,  5,  6,  6,  6,  6,  7,  7,  7,  7,  3},
        { 4,  4,  4,  4,  5,  5,  5,  5,  6,  6,  6,  6,  7,  7,  7,  7,  7},
        { 4,  4,  4,  4,  5,  5,  5,  5,  6,  6,  6,  6,  7,  7,  7,  7,  3},
        { 4,  4,  4,  4,  5,  5,  5,  5,  6,  6,  6,  6,  7,  7,  7,  7,  7},
        { 4,  4,  4,  4,  5,  5,  5,  5,  6,  6,  6,  6,  7,  7,  7,  7,  3},
        { 4,  4,  4,  4,  5,  5,  5,  5,  6,  6,  6,  6,  7,  7,  7,  7,  3},
        { 0,  1,  1,  2,  2,  2,  3,  4,  5,  5,  6,  6,  6,  7,  7,  7,  7},
        { 0,  1,  1,  2,  2,  2,  3,  3,  4,  5,  5,  6,  6,  6,  7,  7,  7},
    };
    static constexpr int TB[8][17] = {
        { 0,  1,  2,  3,  0,  1,  2,  3,  0,  1,  2,  3,  0,  1,  2,  3,  0},
        { 0,  1,  2,  3,  0,  1,  2,  3,  0,  1,  2,  3,  0,  1,  2,  3,  4},
        { 0,  1,  2,  3,  0,  1,  2,  3,  0,  1,  2,  3,  0,  1,  2,  3,  1},
        { 0,  1,  2,  3,  0,  1,  2,  3,  0,  1,  2,  3,  0,  1,  2,  3,  4},
        { 0,  1,  2,  3,  0,  1,  2,  3,  0,  1,  2,  3,  0,  1,  2,  3,  2},
        { 0,  1,  2,  3,  0,  1,  2,  3,  0,  1,  2,  3,  0,  1,  2,  3,  1},
        { 0,  0,  1,  0,  1,  2,  3,  4,  4,  5,  4,  5,  6,  4,  5,  6,  7},
        { 0,  0,  1,  0,  1,  2,  2,  3,  4,  4,  5,  4,  5,  6,  5,  6,  7},
    };
    static constexpr int blk(int g, int i) { return BLK[g][i]; }
    static constexpr int ta(int g, int t) { return TA[g][t]; }
    static constexpr int tb(int g, int t) { return TB[g][t]; }
    static constexpr int cnt(int) { return NT; }
};

// 128 channels: two frequencies per LDS row, 2 x 36 sub-tiles = 8 waves x 9; waves 0-3 own the first frequency
// (blocks 0-7), waves 4-7 the second (blocks 8-15).  Within a frequency:
//   wave 0: rows 4-5 x columns 0-3 + (3,0)      wave 1: rows 6-7 x columns 0-3 + (7,6)
//   wave 2: the triangle of blocks 0-3 minus (3,0)   wave 3: the triangle of blocks 4-7 minus (7,6)
template <>
struct M3Tab<128> {
    static constexpr int ROWLEN = 256, KB = 16, NP = 1, NW = 8;
    static constexpr int NT = 9, NB = 6;
    static constexpr int BLK[8][6] = {
        { 0,  1,  2,  3,  4,  5},
        { 0,  1,  2,  3,  6,  7},
        { 0,  1,  2,  3,  0,  0},
        { 4,  5,  6,  7,  4,  4},
        { 8,  9, 10, 11, 12, 13},
        { 8,  9, 10, 11, 14, 15},
        { 8,  9, 10, 11,  8,  8},
        {12, 13, 14, 15, 12, 12},
    };
    static constexpr int TA[8][9] = {
        { 4,  4,  4,  4,  5,  5,  5,  5,  3},
        { 4,  4,  4,  4,  5,  5,  5,  5,  5},
        { 0,  1,  1,  2,  2,  2,  3,  3,  3},
        { 0,  1,  1,  2,  2,  2,  3,  3,  3},
        { 4,  4,  4,  4,  5,  5,  5,  5,  3},
        { 4,  4,  4,  4,  5,  5,  5,  5,  5},
        { 0,  1,  1,  2,  2,  2,  3,  3,  3},
        { 0,  1,  1,  2,  2,  2,  3,  3,  3},
    };
    static constexpr int TB[8][9] = {
        { 0,  1,  2,  3,  0,  1,  2,  3,  0},
        { 0,  1,  2,  3,  0,  1,  2,  3,  4},
        { 0,  0,  1,  0,  1,  2,  1,  2,  3},
        { 0,  0,  1,  0,  1,  2,  0,  1,  3},
        { 0,  1,  2,  3,  0,  1,  2,  3,  0},
        { 0,  1,  2,  3,  0,  1,  2,  3,  4},
        { 0,  0,  1,  0,  1,  2,  1,  2,  3},
        { 0,  0,  1,  0,  1,  2,  0,  1,  3},
    };
    static constexpr int blk(int g, int i) { return BLK[g][i]; }
    static constexpr int ta(int g, int t) { return TA[g][t]; }
    static constexpr int tb(int g, int t) { return TB[g][t]; }
    static constexpr int cnt(int) { return NT; }
};

// 64 channels: four frequencies per LDS row, 4 x 10 sub-tiles = 8 waves x 5; waves 2d, 2d + 1 own frequency d (blocks 4d...)
template <>
struct M3Tab<64> {
    static constexpr int ROWLEN = 256, KB = 16, NP = 1, NW = 8;
    static constexpr int NT = 5, NB = 4;
    static constexpr int BLK[8][4] = {
        { 0,  1,  2,  0}, { 0,  1,  2,  3}, { 4,  5,  6,  4}, { 4,  5,  6,  7},
        { 8,  9, 10,  8}, { 8,  9, 10, 11}, {12, 13, 14, 12}, {12, 13, 14, 15},
    };
    static constexpr int TA[8][5] = {
        {0, 1, 1, 2, 2}, {2, 3, 3, 3, 3}, {0, 1, 1, 2, 2}, {2, 3, 3, 3, 3},
        {0, 1, 1, 2, 2}, {2, 3, 3, 3, 3}, {0, 1, 1, 2, 2}, {2, 3, 3, 3, 3},
    };
    static constexpr int TB[8][5] = {
        {0, 0, 1, 0, 1}, {2, 0, 1, 2, 3}, {0, 0, 1, 0, 1}, {2, 0, 1, 2, 3},
        {0, 0, 1, 0, 1}, {2, 0, 1, 2, 3}, {0, 0, 1, 0, 1}, {2, 0, 1, 2, 3},
    };
    static constexpr int blk(int g, int i) { return BLK[g][i]; }
    static constexpr int ta(int g, int t) { return TA[g][t]; }
    static constexpr int tb(int g, int t) { return TB[g][t]; }
    static constexpr int cnt(int) { return NT; }
};

// 32 channels: eight frequencies per LDS row, wave g owns frequency g (blocks 2g, 2g + 1): 3 sub-tiles
template <>
struct M3Tab<32> {
    static constexpr int ROWLEN = 256, KB = 16, NP = 1, NW = 8;
    static constexpr int NT = 3, NB = 2;
    static constexpr int BLK[8][2] = {{0, 1}, {2, 3}, {4, 5}, {6, 7}, {8, 9}, {10, 11}, {12, 13}, {14, 15}};
    static constexpr int TA[8][3] = {{0, 1, 1}, {0, 1, 1}, {0, 1, 1}, {0, 1, 1}, {0, 1, 1}, {0, 1, 1}, {0, 1, 1}, {0, 1, 1}};
    static constexpr int TB[8][3] = {{0, 0, 1}, {0, 0, 1}, {0, 0, 1}, {0, 0, 1}, {0, 0, 1}, {0, 0, 1}, {0, 0, 1}, {0, 0, 1}};
    static constexpr int blk(int g, int i) { return BLK[g][i]; }
    static constexpr int ta(int g, int t) { return TA[g][t]; }
    static constexpr int tb(int g, int t) { return TB[g][t]; }
    static constexpr int cnt(int) { return NT; }
};

template <class TAB>
__host__ __device__ constexpr bool m3_is_row(int g, int i) {      // block i of wave g is the row block of some sub-tile
    for (int t = 0; t < TAB::cnt(g); ++t)
        if (TAB::ta(g, t) == i) return true;
    return false;
}
template <class TAB>
__host__ __device__ constexpr bool m3_is_col(int g, int i) {
    for (int t = 0; t < TAB::cnt(g); ++t)
        if (TAB::tb(g, t) == i) return true;
    return false;
}

// 16 bytes per lane global -> LDS (spy_intrinsics.h: spy_glds16).  The copies are ordered by hand: one spy_wait_vmem()
// before the barrier that ends the iteration they were issued in, two iterations before anybody reads the buffer.  No
// compiler-generated vector memory access is in flight while these are (the accumulator read-modify-write comes after
// the loop's last wait), so hipcc's own vmcnt bookkeeping is not disturbed.
__device__ __forceinline__ void m3_glds16(const void* gsrc, char* lds_wave_base) { spy_glds16(gsrc, lds_wave_base); }

// G: which 17 sub-tiles; WPG = 8 waves per workgroup (one workgroup per frequency, two waves per SIMD); wave WV of the
// workgroup stages rows (16 / WPG) WV ... of every chunk
// EXACT = false: the spectra carry a.C <= CH channels per frequency (any count, odd ones included); the LDS image keeps
// its CH-channel geometry - every lane fetches the 16 bytes of ITS two LDS elements from wherever they sit in the
// narrower rows (8-byte aligned sources for odd a.C), lanes of the padding channels a.C ... CH - 1 copy nothing (what
// the padding holds only reaches accumulator rows / columns >= a.C, which are not stored).  For odd a.C the lane of
// channel a.C - 1 reads 8 bytes past its frequency: the host keeps the very last row of a launch away from this kernel.
// Channel sub-ranges (a.ctot > 0, more than 512 channels): the image's channels are a.n0 channels from a.ch0 of rows
// that are a.ctot channels wide (Hermitian block of one range) or, RECT, the two halves of the 512-element image are
// ranges (a.ch0, a.n0) and (a.ch1, a.n1) and the wave's sub-tiles are those of the rectangle range 1 x range 0.
// M4: the 4-multiplication product in the same tiling (p3 accumulates Im = Xi Yr - Xr Yi directly instead of the third
// product of the 3M scheme, whose imaginary part is a difference of large terms): spyhip_csd_set_phase_exact on more
// than 512 channels, where the 32 x 32-tile kernels of csd_kernel.h do not fit a row into LDS.
template <int CH, int G, int WPG, bool EXACT, bool RECT, bool M4>
__device__ __forceinline__ void m3_wave(const CsdArgs& a, char* Xb, int f, int lane) {
    using TAB = M3Tab<CH, RECT>;
    constexpr int M3_NT = TAB::NT, M3_NB = TAB::NB;
    constexpr int ROWLEN = TAB::ROWLEN, KB = TAB::KB;  // LDS row length (elements), rows per chunk: 32 KiB per buffer
    static_assert(KB * ROWLEN * 8 == M3_CHUNK_BYTES, "chunk geometry");
    constexpr int FPR = ROWLEN / CH;                 // frequencies per LDS row; f = index of the packed row
    constexpr int WV = G % WPG;                      // this wave's place in its workgroup
    constexpr int PPR = ROWLEN * 8 / 1024;           // 1-KiB pieces per row (one wave-wide 16-byte copy each)
    constexpr int NPW = KB * PPR / WPG;              // pieces a wave stages per chunk
    constexpr int RG = 4 * ROWLEN * 8;               // bytes per group of four rows
    const int l15 = lane & 15, lq = lane >> 4;
    const int C = EXACT ? CH : (a.ctot ? a.ctot : a.C);             // channels per frequency in memory
    const int ch0 = (!EXACT && a.ctot) ? a.ch0 : 0, n0 = EXACT ? CH : (a.ctot ? a.n0 : a.C);
    const int ch1 = RECT ? a.ch1 : 0, n1 = RECT ? a.n1 : 0;
    // image channel c -> channel of the rows in memory, or -1 (padding)
    auto gchan = [&](int c) -> int {
        if (RECT) return c < 256 ? (c < n0 ? ch0 + c : -1) : (c - 256 < n1 ? ch1 + c - 256 : -1);
        return c < n0 ? ch0 + c : -1;
    };
    const size_t rowstride = (size_t)a.F * C;                       // float2 elements between rows
    const size_t rowbytes = rowstride * 8;
    // source of this lane's 16 bytes of (row 0, piece p): row-major spectra (r, f, c): 16 consecutive bytes of the
    // row; channel-quad-blocked spectra (r, c/4, f, 4) (spyhip_fft_plan_set_blocked, 256 channels only): lanes
    // (2q, 2q+1) take the two halves of quad q's 32 bytes - the copy gathers, the LDS image is the same
    const char* const gbase = reinterpret_cast<const char*>(a.spec) +
                              (a.blocked ? ((size_t)(lane >> 1) * a.F + f) * 32 + (lane & 1) * 16 : (size_t)f * FPR * CH * 8 + lane * 16);
    const size_t piecestep = a.blocked ? (size_t)32 * a.F * 32 : 1024;    // from piece p (128 channels) to piece p + 1
    // valid bytes of this packed row: its FPR frequencies (fewer in the last one), CH channels each
    const int vbytes = ((a.F - f * FPR) < FPR ? (a.F - f * FPR) : FPR) * CH * 8;
    // EXACT = false: byte offset of this lane's two elements of piece p inside a row of spectra, or -1 (padding / no
    // such frequency)
    long long poff[PPR];
#pragma unroll
    for (int p = 0; p < PPR; ++p) {
        const int e0 = p * 128 + 2 * lane, d = e0 / CH, c = e0 % CH;
        const int gc = EXACT ? -1 : gchan(c);
        poff[p] = (gc >= 0 && f * FPR + d < a.F) ? ((long long)(f * FPR + d) * C + gc) * 8 : -1;
    }
    const char* const sbase = reinterpret_cast<const char*>(a.spec);
    const long long nrows = a.nrows;
    const long long nchunk = (nrows + KB - 1) / KB;

    // ---- staging: 1-KiB pieces, NPW per wave and chunk
    auto stage = [&](long long c, int buf) {
        const long long r0 = c * KB;
        const long long left = nrows - r0;
        const int rleft = left < KB ? (int)left : KB;
        char* const dst = Xb + buf * M3_CHUNK_BYTES;
#pragma unroll
        for (int v = 0; v < NPW; ++v) {
            const int id = NPW * WV + v, row = id / PPR, piece = id % PPR;
            // (a row shorter than the LDS row - fewer channels, or a last packed row with fewer frequencies: the bytes
            // behind it belong to the next row of spectra, or to nobody, and are not copied: those lanes sit the copy
            // out and leave stale LDS behind, which only sub-tiles of missing frequencies read - never stored)
            if (row < rleft) {                                          // wave-uniform
                if (!EXACT) {
                    if (poff[piece] >= 0)
                        m3_glds16(sbase + (size_t)(r0 + row) * rowbytes + poff[piece], dst + row * (ROWLEN * 8) + piece * 1024);
                } else if ((FPR == 1 && CH == ROWLEN) || piece * 1024 + lane * 16 < vbytes)
                    m3_glds16(gbase + (size_t)(r0 + row) * rowbytes + piece * piecestep, dst + row * (ROWLEN * 8) + piece * 1024);
            } else {
                *reinterpret_cast<float4*>(dst + row * (ROWLEN * 8) + piece * 1024 + lane * 16) =
                    make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };

    f32x4 p1[M3_NT], p2[M3_NT], p3[M3_NT];
#pragma unroll
    for (int t = 0; t < M3_NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            p1[t][r] = 0.f;
            p2[t][r] = 0.f;
            p3[t][r] = 0.f;
        }

    stage(0, 0);
    if (nchunk > 1) stage(1, 1);
    if (nchunk > 2) stage(2, 2);
    spy_wait_vmem();
    __syncthreads();

    // LDS address of this lane's fragments for the current group of four rows: channel (lane & 15) of a block, row
    // (lane >> 4) of the group; the block is an immediate offset (16 channels = 128 bytes apart)
    const char* fp = Xb + (unsigned)(lq * ROWLEN + l15) * 8u;
    float2 x[M3_NB];
    float sm[M3_NB], df[M3_NB];
    auto load = [&]() {
        m3_for<0, M3_NB>([&](auto ic) {
            constexpr int i = decltype(ic)::value;
            x[i] = *reinterpret_cast<const float2*>(fp + 128 * TAB::blk(G, i));
        });
    };
    // An MFMA blocks the wave that issued it for its 32 cycles, so everything else of a group - 8 fragment reads, 13
    // sums / differences, the address bump - is time the matrix pipe only keeps working through if ANOTHER wave of
    // the SIMD has MFMAs to issue (WPG = 8).  Kept minimal either way.
    load();
    int b0 = 0;
    for (long long c = 0; c < nchunk; ++c) {
        if (c >= 1 && c + 2 < nchunk) stage(c + 2, b0 == 0 ? 2 : b0 - 1);     // buffer (c + 2) % 3: read last iteration
        // fragment addresses advance by 8 KiB per group of four rows; after the fourth group of a chunk on to the
        // next buffer (+ 32 KiB, or back by 64 KiB).  The next chunk has landed: its DMA was waited for before
        // the barrier that ended the previous iteration.
        const int wrap = ((b0 == 2) ? -2 * M3_CHUNK_BYTES : M3_CHUNK_BYTES) - (KB / 4 - 1) * RG;
#pragma unroll 1
        for (int st = 0; st < KB / 4; ++st) {
            m3_for<0, M3_NB>([&](auto ic) {
                constexpr int i = decltype(ic)::value;
                if constexpr (M4) {
                    df[i] = m3_is_col<TAB>(G, i) ? -x[i].y : 0.f;            // (sm is not used)
                } else {
                    sm[i] = m3_is_row<TAB>(G, i) ? x[i].x + x[i].y : 0.f;
                    df[i] = m3_is_col<TAB>(G, i) ? x[i].x - x[i].y : 0.f;
                }
            });
            float re[M3_NB], im[M3_NB];
            m3_for<0, M3_NB>([&](auto ic) { constexpr int i = decltype(ic)::value; re[i] = x[i].x; im[i] = x[i].y; });
            fp += st + 1 < KB / 4 ? RG : wrap;
            m3_sched_fence();
            // the P3 products first: they free the fragment registers' successors (the reads of the next group
            // overwrite x) only after the P1 / P2 products, which come last
            m3_for<0, M3_NT>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                if constexpr (t < TAB::cnt(G)) {
                    if constexpr (M4) {
                        p3[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(im[TAB::ta(G, t)], re[TAB::tb(G, t)], p3[t], 0, 0, 0);
                        p3[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(re[TAB::ta(G, t)], df[TAB::tb(G, t)], p3[t], 0, 0, 0);
                    } else {
                        p3[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(sm[TAB::ta(G, t)], df[TAB::tb(G, t)], p3[t], 0, 0, 0);
                    }
                }
            });
            m3_for<0, M3_NT>([&](auto tc) {
                constexpr int t = decltype(tc)::value;
                if constexpr (t < TAB::cnt(G)) {
                    p1[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(re[TAB::ta(G, t)], re[TAB::tb(G, t)], p1[t], 0, 0, 0);
                    p2[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(im[TAB::ta(G, t)], im[TAB::tb(G, t)], p2[t], 0, 0, 0);
                }
            });
            m3_sched_fence();
            if (st + 1 < KB / 4 || c + 1 < nchunk) load();
        }
        spy_wait_vmem();                                                // this wave's share of chunk c + 2 has landed
        __syncthreads();
        b0 = b0 == 2 ? 0 : b0 + 1;
    }

    // ---- acc += sub-tile (each has one owner: plain read-modify-write).  Lane l holds column (l & 15) and rows
    // 4 (l >> 4) + r of the 16 x 16 block.
    constexpr int BPF = CH / 16;                     // blocks per frequency
    m3_for<0, M3_NT>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        constexpr int bi = TAB::blk(G, TAB::ta(G, t)), bj = TAB::blk(G, TAB::tb(G, t));
        static_assert(bi / BPF == bj / BPF && bi >= bj, "a sub-tile lies inside one frequency, on or below the diagonal");
        const int fr = f * FPR + bi / BPF;           // this sub-tile's frequency
        if (t < TAB::cnt(G) && fr < a.F) {           // wave-uniform
            if constexpr (EXACT) {
                float2* const pb = a.acc + (size_t)fr * CH * CH + (size_t)((bi % BPF) * 16 + 4 * lq) * CH + (bj % BPF) * 16 + l15;
                float2 old[4];
#pragma unroll
                for (int r = 0; r < 4; ++r) old[r] = pb[(size_t)r * CH];
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    pb[(size_t)r * CH] =
                        make_float2(old[r].x + (p1[t][r] + p2[t][r]), old[r].y + (M4 ? p3[t][r] : (p3[t][r] - p1[t][r]) + p2[t][r]));
            } else {
                const int row0 = (bi % BPF) * 16 + 4 * lq, gcol = gchan((bj % BPF) * 16 + l15);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int grow = gchan(row0 + r);
                    if (grow >= 0 && gcol >= 0) {
                        float2* const pe = a.acc + (size_t)fr * C * C + (size_t)grow * C + gcol;
                        const float2 old = *pe;
                        *pe = make_float2(old.x + (p1[t][r] + p2[t][r]), old.y + (M4 ? p3[t][r] : (p3[t][r] - p1[t][r]) + p2[t][r]));
                    }
                }
            }
        }
    });
}

// one workgroup of WPG = 8 waves per frequency (block b -> frequency item_base / 36 + b); run-time wave index ->
// compile-time sub-tile set
template <int CH, int WPG, bool EXACT, bool RECT, bool M4, int G0, int G1>
__device__ __forceinline__ void m3_dispatch(int g, const CsdArgs& a, char* Xb, int f, int lane) {
    if constexpr (G0 + 1 == G1) {
        m3_wave<CH, G0, WPG, EXACT, RECT, M4>(a, Xb, f, lane);
    } else {
        constexpr int GM = (G0 + G1) / 2;
        if (g < GM) m3_dispatch<CH, WPG, EXACT, RECT, M4, G0, GM>(g, a, Xb, f, lane);
        else m3_dispatch<CH, WPG, EXACT, RECT, M4, GM, G1>(g, a, Xb, f, lane);
    }
}

// CH: channels.  Up to 256: one workgroup per packed row of floor(256 / CH) frequencies, block b -> packed row
// item_base / 36 + b.  272 ... 512: NP workgroups per frequency; block b -> XCD b % 8, slot b / 8, frequency
// (slot / NP) * 8 + XCD, part slot % NP - all parts of a frequency run on ONE XCD, one after the other in its
// dispatch order, so the rows they all stage are fetched from HBM once and found in that XCD's L2 afterwards.
template <int CH, int WPG, bool EXACT = true, bool RECT = false, bool M4 = false>
__global__ void __launch_bounds__(64 * WPG) SPY_M3_KATTR(WPG) csd3m_kernel(CsdArgs a) {
    static_assert(WPG == 8, "8 waves per workgroup, two per SIMD (one wave per SIMD measured worse and was removed)");
    static_assert(!RECT || !EXACT, "rectangles come with channel sub-ranges");
    constexpr int NP = M3Tab<CH, RECT>::NP;
    SPY_DYN_SMEM(char, Xb);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = spy_wave_index(tid);
    int f, g;
    if (NP == 1) {
        f = (int)blockIdx.x;
        // channel-quad-blocked spectra (256 channels): a 128-byte line holds one quad of FOUR consecutive frequencies,
        // so those four workgroups must share an L2: runs of four frequencies per XCD (block b -> XCD b % 8)
        if (CH == 256 && a.blocked) f = ((f >> 5) << 5) + ((f & 7) << 2) + ((f >> 3) & 3);
        g = wave;
    } else {
        const int xcd = (int)(blockIdx.x & 7u), slot = (int)(blockIdx.x >> 3);
        f = (slot / NP) * 8 + xcd;
        g = 8 * (slot % NP) + wave;
    }
    f += (int)(a.item_base / M3_TILES_PER_F);
    if ((long long)(f + 1) * M3_TILES_PER_F > a.item_end) return;
    if (CH == 256 && EXACT && a.only_flagged && !a.only_flagged[f]) return;      // (wave-uniform)
    m3_dispatch<CH, WPG, EXACT, RECT, M4, 0, M3Tab<CH, RECT>::NW>(g, a, Xb, f, lane);
}

}  // namespace spycsd
