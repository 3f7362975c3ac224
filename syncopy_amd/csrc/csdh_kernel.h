// K4h - the cross-spectral update for 256 channels on the HALF-PRECISION matrix cores with SPLIT float32 operands:
//
//     acc[f,i,j] += sum_r X[r,f,i] * conj(X[r,f,j])                                  (i >= j at 16-channel granularity)
//
// Reference semantics as csd_kernel.h (connectivity/csd.py:94-102 + the trial sum of
// shared/computational_routine.py:1022-1032).  The float32-input matrix instructions of csd3m_kernel.h run at 1/16 of
// the half-precision rate; this kernel feeds the fast ones without giving up float32-class products:
//
//   * every real number y = x * 2^k(channel) (k from the channel's largest |re|, |im| of the batch, `absmax`, so that
//     |y| < 2^15: a power of two, exact, taken out again in the epilogue) is split ONCE, on its way into LDS, into
//     hi = fp16(y) and lo = fp16(y - hi): hi + lo carries 22 significant bits (|y - hi - lo| <= 2^-22 |y|, or
//     2^-25 absolute where lo is subnormal);
//   * a real product is hi hi' + hi lo' + lo hi' (the dropped lo lo' is <= 2^-22 of it), products exact in the matrix
//     core, float32 accumulation: three v_mfma_f32_16x16x32_f16 (16 cycles, K = 32 rows) where the float32 path needs
//     eight v_mfma_f32_16x16x4_f32 (32 cycles each) - 5.3 x less matrix time per real product;
//   * the complex product is the plain 4-multiplication one (Re = Ar Br + Ai Bi, Im = Ai Br - Ar Bi, each a K-extended
//     sum of the above): 12 instructions per 16 x 16 sub-tile and 32 rows.  The imaginary part is summed directly
//     (no difference of large sums as in the 3-multiplication scheme), so this kernel also serves the phase-exact
//     outputs.  The minus sign costs no operand negation: the Im accumulator changes sign once per chunk between the
//     two product groups (Ai Br first, negate, Ar Bi; the next chunk finds the planes swapped by the loader and so runs
//     the same instructions as Ar Bi first, negate, Ai Br), and the epilogue takes the parity of the chunk count out.
//
// Work decomposition: one 512-thread workgroup per frequency, the 136 lower-triangle sub-tiles dealt to the 8 waves by
// M3Tab<256> (csd3m_kernel.h), 2 x 17 accumulators of 4 registers per wave.  A chunk = 32 rows of X[., f, :]
// (64 KiB of spectra): wave w fetches rows 8 (w / 2) ... + 7 of channels 128 (w % 2) + 64 phase + lane (8-byte loads,
// 512 contiguous bytes per wave instruction), splits, and writes one 16-byte unit = 8 rows of one channel of one
// plane per ds_write_b128 into the layout the fragments are read in:
//
//     LDS[buffer][plane: re hi, re lo, im hi, im lo][row group kg = 0..3][channel 0..255][8 rows] fp16
//
// A fragment read (lane l: channel 16 b + l % 16, row group l / 16) is one conflict-free ds_read_b128 whatever the
// block.  Two buffers: chunk c + 1 is converted into the other buffer while chunk c is multiplied; one barrier per
// chunk.
//
// Validity: a channel whose rms at this frequency sits more than 2^18 below the channel's batch-wide peak would see
// lo go subnormal for its TYPICAL value (absolute error 2^-25 against an rms below 2^-3).  The workgroup checks the
// diagonal of what it accumulated (sum |y|^2 >= rows 2^-6 and finite, or 0 for a channel whose range is 0) BEFORE committing; a frequency
// that fails is left untouched and flagged, and the float32 kernel (csd3m_kernel<256, 8> with CsdArgs::only_flagged)
// redoes exactly those.  Non-finite input takes the same route, so NaN / Inf propagate as before.
#pragma once
#include "csd3m_kernel.h"

#ifndef CSDH_FENCE
#define CSDH_FENCE 0x78f    // __builtin_amdgcn_sched_barrier mask behind the loader's fetches: everything but vector memory may cross
#endif
#ifndef CSDH_S1
#define CSDH_S1 7
#define CSDH_S2 15
#endif

#ifndef CSDH_ABL
#define CSDH_ABL 0      // development ablations (tools/csdh_probe.hip): 1 no global loads, 2 no conversion, 4 no LDS writes,
#endif                  // 8 no fragment reads, 16 no barrier, 32 no negation

#ifdef CSDH_STAMPS
#define CSDH_STAMP(k) do { if (blockIdx.x == 0 && c >= 20 && c < 24) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
        const unsigned long long t__ = __builtin_readcyclecounter(); \
        if (lane == 0) a.stamps[((c - 20) * 8 + G) * 8 + (k)] = t__; } } while (0)
#else
#define CSDH_STAMP(k) do { } while (0)
#endif

namespace spycsd {

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

struct CsdhArgs {
    const float2* spec;     // (nrows, F, 256) complex64
    long long nrows;
    int F;
    float2* acc;            // (F, 256, 256) complex64
    const float* absmax;    // [256]: max over the batch of |re|, |im| per channel (>= the true maximum)
    int* flags;             // [F]: 1 = this launch did NOT add frequency f (float32 kernel must), 0 = added
    int f0;                 // block b -> frequency f0 + b
    int nf;                 // frequencies of this launch
    unsigned long long* stamps;   // development timeline (tools/csdh_probe.hip, -DCSDH_STAMPS): [4 chunks][8 waves][8 points]
    long long rs, fs;       // strides (complex elements) between rows and between frequencies: (nrows, F, 256) -> F * 256, 256
};

constexpr int CSDH_PLANE = 16 * 1024;                 // 4 row groups x 256 channels x 16 bytes
constexpr int CSDH_BUF = 4 * CSDH_PLANE;              // 32 rows of 256 complex values as fp16 pairs
constexpr int CSDH_LDS_BYTES = 2 * CSDH_BUF + 1024 + 16;   // + the channels' scale exponents + the validity word
constexpr int CSDH_KROWS = 32;

// ---- per-wave schedule: the order of the wave's sub-tiles and which fragment slot every operand block sits in.
// Two A slots (ping-pong: the next row block lands while the current one is multiplied), two B slots; a diagonal
// sub-tile reads its column operand from the A slot.  Sub-tiles are walked column pair by column pair so that a 4 x 4
// square loads each column block once and each row block twice.
template <int G>
struct HPlan {
    using TAB = M3Tab<256>;
    static constexpr int NT = TAB::NT;
    struct P {
        int tile[NT];       // walk order -> index into TAB::ta / tb
        int ablk[NT], bblk[NT];      // block indices (0..15) of the operands
        int aslot[NT], bslot[NT];    // bslot 2 = the A slot (diagonal)
        bool aload[NT], bload[NT];   // the operand is fetched for this step (else it is still in its slot)
    };
    static constexpr P make() {
        P p{};
        int key[NT] = {};
        for (int t = 0; t < NT; ++t) {
            p.tile[t] = t;
            key[t] = (TAB::tb(G, t) >> 1) * 64 + TAB::ta(G, t) * 8 + TAB::tb(G, t);
        }
        for (int i = 1; i < NT; ++i)            // insertion sort by (column pair, row block, column block)
            for (int j = i; j > 0 && key[j] < key[j - 1]; --j) {
                int k = key[j]; key[j] = key[j - 1]; key[j - 1] = k;
                int u = p.tile[j]; p.tile[j] = p.tile[j - 1]; p.tile[j - 1] = u;
            }
        int ac[2] = {-1, -1}, bc[2] = {-1, -1};
        int alast = 1, blast = 1;               // slot used by the previous step (the other one is the victim)
        for (int s = 0; s < NT; ++s) {
            const int t = p.tile[s];
            const int x = TAB::blk(G, TAB::ta(G, t)), y = TAB::blk(G, TAB::tb(G, t));
            p.ablk[s] = x;
            p.bblk[s] = y;
            if (ac[0] == x) { p.aslot[s] = 0; p.aload[s] = false; }
            else if (ac[1] == x) { p.aslot[s] = 1; p.aload[s] = false; }
            else { p.aslot[s] = 1 - alast; p.aload[s] = true; ac[1 - alast] = x; }
            alast = p.aslot[s];
            if (x == y) { p.bslot[s] = 2; p.bload[s] = false; }
            else {
                if (bc[0] == y) { p.bslot[s] = 0; p.bload[s] = false; }
                else if (bc[1] == y) { p.bslot[s] = 1; p.bload[s] = false; }
                else { p.bslot[s] = 1 - blast; p.bload[s] = true; bc[1 - blast] = y; }
                blast = p.bslot[s];
            }
        }
        return p;
    }
    static constexpr P T = make();
};

// scale exponent of a channel: y = x * 2^k with |y| < 2^15 for every |x| <= absmax
__device__ __forceinline__ int csdh_exponent(float absmax) {
    const int e = (int)((__float_as_uint(absmax) >> 23) & 0xffu);    // biased exponent; 0 for zero / subnormal
    int k = 141 - e;                                                  // 14 - (e - 127)
    return k > 127 ? 127 : k;                                         // (e = 255, Inf / NaN: k = -114, the data decide)
}

// One sub-tile, 32 rows: X / Y = [plane 0 hi, plane 0 lo, plane 1 hi, plane 1 lo] fragments of the row / column block.
// Re += P0 P0' + P1 P1'; Im: first group P1 P0', then the accumulator changes sign, then P0 P1' (see the header: with the
// planes (re, im) in even and (im, re) in odd chunks this is Ai Br - Ar Bi up to the parity of the chunk count).
__device__ __forceinline__ void csdh_tile(const f16x8 (&X)[4], const f16x8 (&Y)[4], f32x4& re, f32x4& im) {
    __builtin_amdgcn_s_setprio(1);            // the matrix instructions of a sub-tile ahead of the partner wave's vector work
    im = __builtin_amdgcn_mfma_f32_16x16x32_f16(X[2], Y[0], im, 0, 0, 0);
    im = __builtin_amdgcn_mfma_f32_16x16x32_f16(X[2], Y[1], im, 0, 0, 0);
    im = __builtin_amdgcn_mfma_f32_16x16x32_f16(X[3], Y[0], im, 0, 0, 0);
    re = __builtin_amdgcn_mfma_f32_16x16x32_f16(X[0], Y[0], re, 0, 0, 0);
    re = __builtin_amdgcn_mfma_f32_16x16x32_f16(X[0], Y[1], re, 0, 0, 0);
    re = __builtin_amdgcn_mfma_f32_16x16x32_f16(X[1], Y[0], re, 0, 0, 0);
    re = __builtin_amdgcn_mfma_f32_16x16x32_f16(X[2], Y[2], re, 0, 0, 0);
    re = __builtin_amdgcn_mfma_f32_16x16x32_f16(X[2], Y[3], re, 0, 0, 0);
    re = __builtin_amdgcn_mfma_f32_16x16x32_f16(X[3], Y[2], re, 0, 0, 0);
    if (!(CSDH_ABL & 32)) im = -im;
    im = __builtin_amdgcn_mfma_f32_16x16x32_f16(X[0], Y[2], im, 0, 0, 0);
    im = __builtin_amdgcn_mfma_f32_16x16x32_f16(X[0], Y[3], im, 0, 0, 0);
    im = __builtin_amdgcn_mfma_f32_16x16x32_f16(X[1], Y[2], im, 0, 0, 0);
    __builtin_amdgcn_s_setprio(0);
}

// After the last chunk: validity of the frequency, then acc += sub-tile / (2^k_i 2^k_j).  `kexp(channel)` = the scale
// exponent the operands were split with.
template <int G, class KEXP>
__device__ __forceinline__ void csdh_finish(const CsdhArgs& a, f32x4 (&re)[M3Tab<256>::NT], f32x4 (&im)[M3Tab<256>::NT], int nchunk,
                                            int f, int lane, int* vword, KEXP kexp) {
    using TAB = M3Tab<256>;
    constexpr int NT = TAB::NT;
    const int l15 = lane & 15, lq = lane >> 4;
    const long long nrows = a.nrows;
    // ---- the Im accumulators carry the sign (-1)^nchunk relative to (even chunks) Im = Ai Br - Ar Bi ... :
    // even chunk: planes (re, im): first group = Ai Br (+), then negated, + Ar Bi  ->  -(I + Ai Br - Ar Bi)
    // odd chunk:  planes (im, re): first group = Ar Bi added to -(I...), negated -> I... - Ar Bi, + Ai Br
    const float isign = (nchunk & 1) ? -1.f : 1.f;

    // ---- validity of the frequency: the diagonal of the scaled accumulation (waves 6 and 7 own the diagonal sub-tiles)
    {
        const float floor2 = (float)nrows * 0.015625f;            // rms^2 >= 2^-6 in scaled units
        bool bad = false;
        m3_for<0, NT>([&](auto tc) {
            constexpr int t = decltype(tc)::value;
            if constexpr (TAB::blk(G, TAB::ta(G, t)) == TAB::blk(G, TAB::tb(G, t))) {
                // an all-zero diagonal entry is only innocent for a channel that IS zero (a scale of 2^-114 for an
                // Inf "maximum" would flush a whole channel to zero otherwise)
                const bool dead = a.absmax[TAB::blk(G, TAB::ta(G, t)) * 16 + l15] == 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r)
                    if (4 * lq + r == l15) {
                        const float v = re[t][r];
                        bad = bad || !((v >= floor2 && v < __builtin_inff()) || (v == 0.f && dead));
                    }
            }
        });
        if (__any(bad) && lane == 0) *vword = 0;
    }
    __syncthreads();
    const bool valid = *vword != 0;
    if (a.flags && G == 0 && lane == 0) a.flags[f] = valid ? 0 : 1;
    if (!valid) return;

    // ---- acc += sub-tile / (2^k_i 2^k_j).  Lane l holds column (l & 15) and rows 4 (l >> 4) + r of the 16 x 16 block.
    m3_for<0, NT>([&](auto tc) {
        constexpr int t = decltype(tc)::value;
        constexpr int bi = TAB::blk(G, TAB::ta(G, t)), bj = TAB::blk(G, TAB::tb(G, t));
        static_assert(bi >= bj, "a sub-tile lies on or below the diagonal");
        float2* const pb = a.acc + (size_t)f * 65536 + (size_t)(bi * 16 + 4 * lq) * 256 + bj * 16 + l15;
        const int kj = kexp(bj * 16 + l15);
        float2 old[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) old[r] = pb[(size_t)r * 256];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int kk = -(kexp(bi * 16 + 4 * lq + r) + kj);
            pb[(size_t)r * 256] = make_float2(old[r].x + ldexpf(re[t][r], kk), old[r].y + ldexpf(isign * im[t][r], kk));
        }
    });
}

template <int G>
__device__ __forceinline__ void csdh_wave(const CsdhArgs& a, char* lds, int f, int lane) {
    using TAB = M3Tab<256>;
    using PL = HPlan<G>;
    constexpr int NT = TAB::NT;
    constexpr int KG = G >> 1, CHH = G & 1;          // this wave converts rows 8 KG ... of channels 128 CHH ...
    const int l15 = lane & 15, lq = lane >> 4;
    int* const kexp = reinterpret_cast<int*>(lds + 2 * CSDH_BUF);          // [256] scale exponents
    int* const vword = reinterpret_cast<int*>(lds + 2 * CSDH_BUF + 1024);  // validity of the frequency

    // ---- scales of this lane's two loader channels
    const int lc0 = 128 * CHH + lane, lc1 = lc0 + 64;
    const int k0 = csdh_exponent(a.absmax[lc0]), k1 = csdh_exponent(a.absmax[lc1]);
    const float s0 = __uint_as_float((unsigned)(k0 + 127) << 23), s1 = __uint_as_float((unsigned)(k1 + 127) << 23);
    if (KG == 0) { kexp[lc0] = k0; kexp[lc1] = k1; }
    if (G == 0 && lane == 0) *vword = 1;

    const long long nrows = a.nrows;
    const int nchunk = (int)((nrows + CSDH_KROWS - 1) / CSDH_KROWS);
    const size_t rowstride = (size_t)a.rs;                            // float2 elements between rows

    // ---- loader: phase ph = channel lc0 + 64 ph, 8 rows.  Everything that is not a matrix instruction competes with the
    // matrix instructions of BOTH waves of the SIMD for its one issue port (the timeline of tools/csdh_probe.hip
    // -DCSDH_STAMPS: the wave that loses the arbitration finishes a chunk 2500 cycles after its partner), so the
    // common case - all 32 rows of the chunk exist - is kept lean: row addresses = one scalar pointer per chunk + i row
    // strides, two v_fma_mix per value for the split (product with the scale, rounding to fp16 and packing in one
    // instruction; the residual the same way), no per-row selects.  Only the last one or two chunks of a launch take
    // the general path (rows past the end are fetched from the last row and given the scale 0).
    f32x2 st[8];
    const int nrows_i = (int)nrows;                  // (the host keeps launches below 2^31 rows)
    const int nfull = nrows_i / CSDH_KROWS;          // chunks whose 32 rows all exist
    const char* const gb = reinterpret_cast<const char*>(a.spec) + ((size_t)f * a.fs + 128 * CHH) * 8;   // wave-uniform
    const size_t rowbytes = rowstride * 8;
    const unsigned voff = (unsigned)lane * 8u;
    const char* nextp = gb + (size_t)(CSDH_KROWS + 8 * KG) * rowbytes;       // rows 8 KG ... of the chunk fetched next
    auto fetch = [&](int c, int ph) {
        if (c < nfull) {                                                      // (uniform) nextp points at chunk c
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (CSDH_ABL & 1) st[i] = f32x2{(float)(i + lane), (float)(c - lane)};
                else if ((CSDH_ABL & 128) && c > 2) asm volatile("" : "+v"(st[i]));      // keep the (real) values of chunk 2
                else st[i] = *reinterpret_cast<const f32x2*>(nextp + i * rowbytes + 512 * ph + voff);
            }
        } else {
            const int r0 = c * CSDH_KROWS + 8 * KG;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                const int r = min(r0 + i, nrows_i - 1);
                st[i] = *reinterpret_cast<const f32x2*>(gb + (size_t)r * rowbytes + 512 * ph + voff);
            }
        }
    };
    // (hi, lo) fp16 pairs of x0 s and x1 s, packed (x0 in the low halves): hi = fp16(x s), lo = fp16(x s - hi)
    auto split2 = [](float x0, float x1, float s, unsigned& hi, unsigned& lo) {
        asm("v_fma_mixlo_f16 %0, %2, %4, 0\n\t"
            "v_fma_mixhi_f16 %0, %3, %4, 0\n\t"
            "v_fma_mixlo_f16 %1, %2, %4, -%0 op_sel_hi:[0,0,1]\n\t"
            "v_fma_mixhi_f16 %1, %3, %4, -%0 op_sel:[0,0,1] op_sel_hi:[0,0,1]"
            : "=&v"(hi), "=&v"(lo)
            : "v"(x0), "v"(x1), "v"(s));
    };
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    // chunk c lives in buffer c & 1 with the planes (re, im) in the order (c & 1) ? (im, re) : (re, im)
    auto convert = [&](int c, int ph) {
        const float sc = ph ? s1 : s0;
        u32x4 rh, rl, ih, il;
        if (CSDH_ABL & 2) {
#pragma unroll
            for (int i = 0; i < 4; ++i) { rh[i] = rl[i] = __float_as_uint(st[2 * i][0]); ih[i] = il[i] = __float_as_uint(st[2 * i + 1][1]); }
        } else if (c < nfull) {
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                unsigned h, l;
                split2(st[2 * i][0], st[2 * i + 1][0], sc, h, l);
                rh[i] = h; rl[i] = l;
                split2(st[2 * i][1], st[2 * i + 1][1], sc, h, l);
                ih[i] = h; il[i] = l;
            }
        } else {
            // (a row past the end was fetched from the last row: scale 0 - if that row holds Inf / NaN the product is NaN,
            // in channels that the row itself has already made non-finite)
            const int r0 = c * CSDH_KROWS + 8 * KG;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const float sa = (r0 + 2 * i < nrows_i) ? sc : 0.f, sb = (r0 + 2 * i + 1 < nrows_i) ? sc : 0.f;
                unsigned h0, l0, h1, l1;
                split2(st[2 * i][0], st[2 * i][1], sa, h0, l0);            // (re, im) of row 2 i
                split2(st[2 * i + 1][0], st[2 * i + 1][1], sb, h1, l1);    // (re, im) of row 2 i + 1
                rh[i] = (h0 & 0xffffu) | (h1 << 16);
                rl[i] = (l0 & 0xffffu) | (l1 << 16);
                ih[i] = (h0 >> 16) | (h1 & 0xffff0000u);
                il[i] = (l0 >> 16) | (l1 & 0xffff0000u);
            }
        }
        const int odd = c & 1;
        char* const w = lds + odd * CSDH_BUF + KG * 4096 + (lc0 + 64 * ph) * 16;
        char* const wr = w + (odd ? 2 * CSDH_PLANE : 0);
        char* const wi = w + (odd ? 0 : 2 * CSDH_PLANE);
        if (CSDH_ABL & 4) {
            if (rh[0] + rl[1] + ih[2] + il[3] == 123456u) *reinterpret_cast<u32x4*>(wr) = rh;
            return;
        }
        *reinterpret_cast<u32x4*>(wr) = rh;
        *reinterpret_cast<u32x4*>(wr + CSDH_PLANE) = rl;
        *reinterpret_cast<u32x4*>(wi) = ih;
        *reinterpret_cast<u32x4*>(wi + CSDH_PLANE) = il;
    };

    f32x4 re[NT], im[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) { re[t][r] = 0.f; im[t][r] = 0.f; }

    nextp -= (size_t)CSDH_KROWS * rowbytes;      // chunk 0
    fetch(0, 0);
    convert(0, 0);
    fetch(0, 1);
    convert(0, 1);
    nextp += (size_t)CSDH_KROWS * rowbytes;
    __syncthreads();

    // fragment slots: [plane 0 hi, plane 0 lo, plane 1 hi, plane 1 lo]
    f16x8 A[2][4], B[2][4];
    unsigned rb = (unsigned)(lq * 4096 + l15 * 16);          // this lane's fragment base in the current buffer
    bool nofrag = false;
    auto ldfrag = [&](f16x8 (&dst)[4], int blk) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            if (CSDH_ABL & 8) {
                asm volatile("" : "+v"(dst[p]));            // keep the registers alive, read nothing
                continue;
            }
            if ((CSDH_ABL & 64) && nofrag) continue;        // real operand bits, fetched in the first chunk only
            dst[p] = *reinterpret_cast<const f16x8*>(lds + rb + p * CSDH_PLANE + blk * 256);
        }
    };
    constexpr int S1 = CSDH_S1, S2 = CSDH_S2;                 // the conversions go behind these steps

    for (int c = 0; c < nchunk; ++c) {
        CSDH_STAMP(0);
        ldfrag(A[PL::T.aslot[0]], PL::T.ablk[0]);
        if constexpr (PL::T.bload[0]) ldfrag(B[PL::T.bslot[0] & 1], PL::T.bblk[0]);
        fetch(c + 1, 0);          // (the last iteration converts rows that do not exist into the idle buffer: zeros)
        __builtin_amdgcn_sched_barrier(CSDH_FENCE);      // the loads are issued HERE, ahead of the chunk's matrix work
        m3_for<0, NT>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            constexpr int t = PL::T.tile[s];
            // operands of the next step land while this one multiplies
            if constexpr (s + 1 < NT) {
                if constexpr (PL::T.aload[s + 1]) ldfrag(A[PL::T.aslot[s + 1]], PL::T.ablk[s + 1]);
                if constexpr (PL::T.bload[s + 1]) ldfrag(B[PL::T.bslot[s + 1]], PL::T.bblk[s + 1]);
            }
            const f16x8(&X)[4] = A[PL::T.aslot[s]];
            const f16x8(&Y)[4] = PL::T.bslot[s] == 2 ? A[PL::T.aslot[s]] : B[PL::T.bslot[s] & 1];
            csdh_tile(X, Y, re[t], im[t]);
            if constexpr (s == S1) {
                CSDH_STAMP(2);
#ifdef CSDH_STAMPS
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
                CSDH_STAMP(3);
                convert(c + 1, 0);
                fetch(c + 1, 1);
                __builtin_amdgcn_sched_barrier(CSDH_FENCE);
                CSDH_STAMP(4);
            }
            if constexpr (s == 0) CSDH_STAMP(1);
            if constexpr (s == S2) {
                CSDH_STAMP(5);
#ifdef CSDH_STAMPS
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#endif
                CSDH_STAMP(6);
                convert(c + 1, 1);
            }
        });
        CSDH_STAMP(7);
        if (!(CSDH_ABL & 16)) __syncthreads();
        if (CSDH_ABL & 64) nofrag = true;
        rb ^= (unsigned)CSDH_BUF;
        nextp += (size_t)CSDH_KROWS * rowbytes;
    }

    csdh_finish<G>(a, re, im, nchunk, f, lane, vword, [&](int ch) { return kexp[ch]; });
}

template <int G0, int G1>
__device__ __forceinline__ void csdh_dispatch(int g, const CsdhArgs& a, char* lds, int f, int lane) {
    if constexpr (G0 + 1 == G1) {
        csdh_wave<G0>(a, lds, f, lane);
    } else {
        constexpr int GM = (G0 + G1) / 2;
        if (g < GM) csdh_dispatch<G0, GM>(g, a, lds, f, lane);
        else csdh_dispatch<GM, G1>(g, a, lds, f, lane);
    }
}

__global__ void __launch_bounds__(512) SPY_M3_KATTR(8) csdh_kernel(CsdhArgs a) {
    SPY_DYN_SMEM(char, lds);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int f = a.f0 + (int)blockIdx.x;
    csdh_dispatch<0, 8>(wave, a, lds, f, lane);
}

// max over a batch of spectra of |re|, |im| per channel (the kernels of mtmfft.hip deliver it for free; this pass is for
// spectra that come from somewhere else): out[c] = max(out[c], ...), bit pattern compare (non-negative floats)
__global__ void __launch_bounds__(256) csdh_absmax_kernel(const float4* __restrict__ spec, long long nquads, int C,
                                                          unsigned* __restrict__ out) {
    // a thread walks elements of constant channel pair: stride = multiple of C / 2 float4s
    __shared__ unsigned sm[512];
    for (int i = threadIdx.x; i < 512; i += 256) sm[i] = 0u;
    __syncthreads();
    const long long per = C / 2;                                  // float4 (two channels) per (row, frequency)
    const long long stride = (long long)gridDim.x * 256;
    float m0 = 0.f, m1 = 0.f;
    long long q = (long long)blockIdx.x * 256 + threadIdx.x;
    const int cp = (int)(q % per);                                // stride is a multiple of `per` (host: grid * 256 % per == 0)
    for (; q < nquads; q += stride) {
        const float4 v = spec[q];
        m0 = fmaxf(m0, fmaxf(fabsf(v.x), fabsf(v.y)));
        m1 = fmaxf(m1, fmaxf(fabsf(v.z), fabsf(v.w)));
        // NaN: fmaxf drops it - a NaN / Inf spectrum fails the validity check of csdh_kernel instead
    }
    atomicMax(&sm[2 * cp], __float_as_uint(m0));
    atomicMax(&sm[2 * cp + 1], __float_as_uint(m1));
    __syncthreads();
    for (int i = threadIdx.x; i < C; i += 256)
        if (sm[i]) atomicMax(&out[i], sm[i]);
}

}  // namespace spycsd
