// MEASURED ALTERNATIVE, NOT PART OF THE LIBRARY (tools/csdh_probe.hip builds it; result in profiles/r5_k4h_presplit_probe.txt):
// 2.90 vs 3.17 ms at 7000 rows x 512 frequencies - 10 % fewer cycles, but the chip answers the higher matrix-pipe
// occupancy (61 -> 69 %) with a lower clock (1.68 -> 1.59 GHz): the kernel is power-limited near 1.0-1.1 PFLOP/s of FP16
// matrix work on real operand bits.  With the transform kernel paying ~150 extra vector instructions per taper for the
// split, the step as a whole would gain ~2 %; the hand-over stays complex64.
//
// K4h fed with spectra that are ALREADY split (as the transform kernel would write them): the same
// matrix work, accumulators, validity rule and epilogue as csdh_kernel.h, but nothing else in the inner loop - no staging
// registers, no conversion, no LDS writes.  What csdh_kernel spends per chunk and wave on turning complex64 rows into
// operand planes (~100 vector instructions, 8 ds_write_b128, 16 loads through registers) is where its two waves per SIMD
// fail to cover each other.
//
// Hand-over layout ("planes"): per (row, frequency) 2 KiB, channel quad q = 4 q ... 4 q + 3 at byte 32 q:
//     [ re hi (4 x fp16) | re lo | im hi | im lo ]          value = (hi + lo) 2^-k(channel), k = csdh_exponent(absmax[channel])
// i.e. the 32 bytes one thread of the transform kernel owns per bin - it stores them exactly as it stored its two 16-byte
// halves of complex64 values.
//
// A chunk = 32 rows x 2 KiB travels global -> LDS by LDS-DMA (global_load_lds_dwordx4, 1 KiB per wave instruction, no
// registers): wave w copies rows 4 w ... 4 w + 3 of the NEXT chunk at the start of a chunk and waits for them
// (s_waitcnt vmcnt(0)) before the barrier that ends it.  In LDS a row keeps its row-major form; MFMA fragments (lane =
// channel, 8 consecutive rows) come out of it through ds_read_b64_tr_b16: each 16-lane group reads a [4 rows][16 channels]
// tile, every lane giving the address of the 8 bytes (one plane of one quad) it contributes, and gets it back transposed.
// Row r sits at r * 2304 + 8 (r % 4) + 128 ((r / 8) % 2) bytes: the 8 rows two lane groups touch in one LDS cycle then
// cover the 64 banks exactly once.  Two buffers of 72 KiB.
#pragma once
#include "../syncopy_amd/csrc/csdh_kernel.h"

namespace spycsd {

typedef _Float16 f16x4 __attribute__((ext_vector_type(4)));

constexpr int CSDHP_ROW = 2304;                         // LDS bytes reserved per row (2048 of data + room for the bank offsets)
constexpr int CSDHP_BUF = CSDH_KROWS * CSDHP_ROW;        // 73 728
constexpr int CSDHP_LDS_BYTES = 2 * CSDHP_BUF + 16;      // + the validity word

__host__ __device__ constexpr int csdhp_row_base(int r) { return r * CSDHP_ROW + 8 * (r % 4) + 128 * ((r / 8) % 2); }

template <int G>
__device__ __forceinline__ void csdhp_wave(const CsdhArgs& a, char* lds, int f, int lane) {
    using TAB = M3Tab<256>;
    using PL = HPlan<G>;
    constexpr int NT = TAB::NT;
    int* const vword = reinterpret_cast<int*>(lds + 2 * CSDHP_BUF);
    if (G == 0 && lane == 0) *vword = 1;

    const int nrows_i = (int)a.nrows;
    const int nchunk = (nrows_i + CSDH_KROWS - 1) / CSDH_KROWS;
    const size_t rowbytes = (size_t)a.rs * 8;                                   // bytes between rows of the hand-over
    const char* const gb = reinterpret_cast<const char*>(a.spec) + (size_t)f * a.fs * 8 + lane * 16;

    // ---- loader: rows 4 G ... 4 G + 3 of chunk c into buffer c & 1 (rows past the end of the spectra: zeros)
    auto stage = [&](int c) {
        char* const dst = lds + (c & 1) * CSDHP_BUF;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = 4 * G + i, row = c * CSDH_KROWS + r;
            if (row < nrows_i) {                                                // (wave-uniform)
                const char* const src = gb + (size_t)row * rowbytes;
                spy_glds16(src, dst + csdhp_row_base(r));
                spy_glds16(src + 1024, dst + csdhp_row_base(r) + 1024);
            } else {
                *reinterpret_cast<float4*>(dst + csdhp_row_base(r) + lane * 16) = make_float4(0.f, 0.f, 0.f, 0.f);
                *reinterpret_cast<float4*>(dst + csdhp_row_base(r) + 1024 + lane * 16) = make_float4(0.f, 0.f, 0.f, 0.f);
            }
        }
    };

    f32x4 re[NT], im[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) { re[t][r] = 0.f; im[t][r] = 0.f; }

    stage(0);
    spy_wait_vmem();
    __syncthreads();

    // ---- fragments.  Lane l of a 16-lane group g = l / 16 contributes, to the tile of rows 8 g + 4 h ... + 3 (h = 0, 1) and
    // the 16 channels of block b, the 8 bytes of row 8 g + 4 h + (l % 16) / 4, quad 4 b + l % 4, plane p - and receives
    // channel 16 b + l % 16 of those four rows.  Two reads (h = 0, 1) make the 8 rows of a 16x16x32 operand.
    // Planes: even chunks (re, im), odd chunks (im, re) (csdh_tile's sign rule): logical plane j sits at byte 16 (j ^ odd).
    const int li = lane & 15, lg = lane >> 4;
    const unsigned lanebase = (unsigned)(csdhp_row_base(8 * lg + (li >> 2)) + 32 * (li & 3));
    unsigned a0 = lanebase, a1 = lanebase + 16;              // logical plane 0 / 1 in the current buffer
    f16x8 A[2][4], B[2][4];
    typedef __attribute__((address_space(3))) char lds_char;
    typedef __fp16 fp16v4 __attribute__((__vector_size__(4 * sizeof(__fp16))));       // the builtin's own vector type
    typedef __attribute__((address_space(3))) fp16v4 lds_f16x4;
    lds_char* const lds3 = (lds_char*)lds;                   // the workgroup's LDS as an LDS-space pointer
    auto ldfrag = [&](f16x8 (&dst)[4], int blk) {
#pragma unroll
        for (int p = 0; p < 4; ++p) {
            const unsigned base = (p < 2 ? a0 : a1) + (unsigned)((p & 1) * 8 + blk * 128);
            const fp16v4 lo4 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_f16x4*)(lds3 + base));
            const fp16v4 hi4 = __builtin_amdgcn_ds_read_tr16_b64_v4f16((lds_f16x4*)(lds3 + base + 4 * CSDHP_ROW));
            typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            const u32x2 lo = __builtin_bit_cast(u32x2, lo4), hi = __builtin_bit_cast(u32x2, hi4);
            dst[p] = __builtin_bit_cast(f16x8, u32x4{lo[0], lo[1], hi[0], hi[1]});
        }
    };

    for (int c = 0; c < nchunk; ++c) {
        ldfrag(A[PL::T.aslot[0]], PL::T.ablk[0]);
        if constexpr (PL::T.bload[0]) ldfrag(B[PL::T.bslot[0] & 1], PL::T.bblk[0]);
        if (c + 1 < nchunk) stage(c + 1);
        __builtin_amdgcn_sched_barrier(0);                   // the copies are issued HERE, ahead of the chunk's matrix work
        m3_for<0, NT>([&](auto sc) {
            constexpr int s = decltype(sc)::value;
            constexpr int t = PL::T.tile[s];
            if constexpr (s + 1 < NT) {
                if constexpr (PL::T.aload[s + 1]) ldfrag(A[PL::T.aslot[s + 1]], PL::T.ablk[s + 1]);
                if constexpr (PL::T.bload[s + 1]) ldfrag(B[PL::T.bslot[s + 1]], PL::T.bblk[s + 1]);
            }
            const f16x8(&X)[4] = A[PL::T.aslot[s]];
            const f16x8(&Y)[4] = PL::T.bslot[s] == 2 ? A[PL::T.aslot[s]] : B[PL::T.bslot[s] & 1];
            csdh_tile(X, Y, re[t], im[t]);
        });
        spy_wait_vmem();                                     // this wave's rows of chunk c + 1 have landed
        __syncthreads();
        // the other buffer, the planes the other way round
        const unsigned odd = (unsigned)((c + 1) & 1);
        a0 = lanebase + odd * (CSDHP_BUF + 16);
        a1 = lanebase + odd * CSDHP_BUF + 16 * (1 - odd);
    }

    csdh_finish<G>(a, re, im, nchunk, f, lane, vword, [&](int ch) { return csdh_exponent(a.absmax[ch]); });
}

template <int G0, int G1>
__device__ __forceinline__ void csdhp_dispatch(int g, const CsdhArgs& a, char* lds, int f, int lane) {
    if constexpr (G0 + 1 == G1) {
        csdhp_wave<G0>(a, lds, f, lane);
    } else {
        constexpr int GM = (G0 + G1) / 2;
        if (g < GM) csdhp_dispatch<G0, GM>(g, a, lds, f, lane);
        else csdhp_dispatch<GM, G1>(g, a, lds, f, lane);
    }
}

// a.spec = the planes hand-over (nrows, F, 64 quads, 4 planes, 4) fp16; a.rs / a.fs in units of 8 bytes as for complex64
__global__ void __launch_bounds__(512) SPY_M3_KATTR(8) csdhp_kernel(CsdhArgs a) {
    SPY_DYN_SMEM(char, lds);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = spy_wave_index(tid);
    const int f = a.f0 + (int)blockIdx.x;
    csdhp_dispatch<0, 8>(wave, a, lds, f, lane);
}

// complex64 spectra -> planes hand-over with the scales of `absmax` (tests, and spectra that were not written split):
// one thread per (row, frequency, quad)
__global__ void __launch_bounds__(256) csdhp_split_kernel(const float4* __restrict__ spec, long long nquads, const float* __restrict__ absmax,
                                                          uint4* __restrict__ planes) {
    for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < nquads; q += (long long)gridDim.x * 256) {
        const int c0 = (int)(q & 63) * 4;
        const float4 u = spec[2 * q], v = spec[2 * q + 1];           // (re, im) of channels c0, c0 + 1 | c0 + 2, c0 + 3
        const float x[4][2] = {{u.x, u.y}, {u.z, u.w}, {v.x, v.y}, {v.z, v.w}};
        _Float16 h[2][4], l[2][4];
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const float s = __uint_as_float((unsigned)(csdh_exponent(absmax[c0 + i]) + 127) << 23);
#pragma unroll
            for (int p = 0; p < 2; ++p) {
                const float y = x[i][p] * s;
                h[p][i] = (_Float16)y;
                l[p][i] = (_Float16)(y - (float)h[p][i]);
            }
        }
        auto pack = [](const _Float16 (&e)[4], unsigned& w0, unsigned& w1) {
            unsigned short b[4];
            __builtin_memcpy(b, e, 8);
            w0 = b[0] | ((unsigned)b[1] << 16);
            w1 = b[2] | ((unsigned)b[3] << 16);
        };
        uint4 o0, o1;
        pack(h[0], o0.x, o0.y);
        pack(l[0], o0.z, o0.w);
        pack(h[1], o1.x, o1.y);
        pack(l[1], o1.z, o1.w);
        planes[2 * q] = o0;
        planes[2 * q + 1] = o1;
    }
}

}  // namespace spycsd
