// K6: Wilson spectral matrix factorisation + Granger-Geweke causality in complex128.
//
// Reference semantics: regularize_csd / wilson_sf / _plusOperator / _psi0_initial / max_rel_err
// (connectivity/wilson_sf.py:16-254) and granger (connectivity/granger.py:10-79), driven by
// granger_cF (connectivity/AV_compRoutines.py:293-412).
//
// The reference mirrors the CSD to negative frequencies (wilson_sf.py:63) and works on 2(F-1)
// matrices; every matrix at -f is the complex conjugate of the one at +f and stays so through
// inverse, products and the plus operator, so all batched linear algebra here runs on the F
// non-negative frequencies only and the lag-domain step uses the conjugate-symmetric extension
// (real(ifft(full)) == irfft(half) exactly).
#pragma once
#include "cd_math.h"
#include "f64_stockham.h"

namespace spywil {

// ---- complex64 (F,C,C) -> complex128, + eps on the diagonal (regularize_csd: CSD + eps*I)
__global__ void __launch_bounds__(256) widen_kernel(const float2* in, cd* out, int C, long long n, double eps) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) {
        const int j = (int)(e % C), i = (int)((e / C) % C);
        const float2 v = in[e];
        out[e] = make_double2((double)v.x + (i == j ? eps : 0.0), (double)v.y);
    }
}

// ---- batched C[b] = A[b] * op(B[b]) (+ I), n x n, opB: 0 = B, 1 = B^H; strideB = 0 broadcasts B
constexpr int GT = 32;   // output tile
constexpr int GK = 8;    // k step
__global__ void __launch_bounds__(256) zgemm_kernel(const cd* A, const cd* B, cd* Cm, int n, long long sA, long long sB,
                                                    long long sC, int opB, int addI) {
    __shared__ cd As[GT][GK + 1];
    __shared__ cd Bs[GK][GT + 1];
    const int b = blockIdx.z, ti = blockIdx.y * GT, tj = blockIdx.x * GT;
    const cd* Ab = A + (size_t)b * sA;
    const cd* Bb = B + (size_t)b * sB;
    const int tid = threadIdx.x, tx = tid & 15, ty = tid >> 4;   // 16 x 16 threads, 2 x 2 outputs each
    cd acc[2][2];
    for (int u = 0; u < 2; ++u)
        for (int w = 0; w < 2; ++w) acc[u][w] = make_double2(0.0, 0.0);
    for (int k0 = 0; k0 < n; k0 += GK) {
        {   // 256 threads stage 32x8 of A and 8x32 of op(B)
            const int r = tid >> 3, kk = tid & 7;
            const int gi = ti + r, gk = k0 + kk;
            As[r][kk] = (gi < n && gk < n) ? Ab[(size_t)gi * n + gk] : make_double2(0.0, 0.0);
            const int kk2 = tid >> 5, cc = tid & 31;
            const int gk2 = k0 + kk2, gj = tj + cc;
            cd v = make_double2(0.0, 0.0);
            if (gk2 < n && gj < n) {
                if (opB == 0) v = Bb[(size_t)gk2 * n + gj];
                else { v = Bb[(size_t)gj * n + gk2]; v.y = -v.y; }
            }
            Bs[kk2][cc] = v;
        }
        __syncthreads();
#pragma unroll
        for (int kk = 0; kk < GK; ++kk) {
            const cd a0 = As[ty * 2][kk], a1 = As[ty * 2 + 1][kk];
            const cd b0 = Bs[kk][tx * 2], b1 = Bs[kk][tx * 2 + 1];
            acc[0][0] = cadd(acc[0][0], cmul(a0, b0));
            acc[0][1] = cadd(acc[0][1], cmul(a0, b1));
            acc[1][0] = cadd(acc[1][0], cmul(a1, b0));
            acc[1][1] = cadd(acc[1][1], cmul(a1, b1));
        }
        __syncthreads();
    }
    cd* Cb = Cm + (size_t)b * sC;
    for (int u = 0; u < 2; ++u)
        for (int w = 0; w < 2; ++w) {
            const int gi = ti + ty * 2 + u, gj = tj + tx * 2 + w;
            if (gi < n && gj < n) {
                cd v = acc[u][w];
                if ((addI & 1) && gi == gj) v.x += 1.0;
                Cb[(size_t)gi * n + gj] = v;
            }
        }
}

// ---- the same product on the fp64 matrix cores: 64 x 64 output tile per workgroup, wave w owns the 32 x 32
// quadrant (w>>1, w&1) as 2 x 2 sub-tiles of v_mfma_f64_16x16x4_f64 (A: one double per lane at
// [row = lane&15][k = lane>>4], B: [k = lane>>4][col = lane&15], D: 4 doubles per lane at
// row = (lane>>4) + 4*reg, col = lane&15).  Complex product = 4 real MFMAs per k-step of 4:
//   Cr += Ar Br ; Cr += (-Ai) Bi ; Ci += Ar Bi ; Ci += Ai Br.
// The 32 x 32 VALU tile above moves 16 bytes per 16 flop through L2 and stalls at ~28 TFLOP/s; this tile
// halves the traffic per flop and leaves the multiply-adds to the matrix pipe.
constexpr int MT = 64;   // output tile
constexpr int MK = 8;    // k per stage
// Badd (n x n, or nullptr): op(B) + Badd is multiplied - the skew matrix S of wilson_sf.py:97-98 joins g+ here instead
// of in a pass of its own.  Ref / part (or nullptr): instead of storing C, the workgroup writes
// max |Ref - C| / |Ref| over its tile to part[linear block id] (max_rel_err, wilson_sf.py:190-194: psi psi^H is only
// ever compared with the CSD, never kept).
// MODE 3: Hermitian product (op(B) = A^H given as B = A, opB = 1): only the tiles on and below the diagonal are
// computed, tiles below it are also stored mirrored (conjugated) above it - 10 tile products instead of 16 at
// n = 256.  MODE 2 likewise skips the tiles above the diagonal (|Ref - C| / |Ref| is symmetric for Hermitian Ref, C).
// MODE 0: plain, 1: with Badd, 2: error check (separate instances: the plain product keeps its 90 registers and
// three waves per SIMD - with the extra operands in one kernel it dropped to two and ran at half speed)
#define SPY_ZGEMM_KATTR SPY_WAVES_PER_EU(3, 3)
// 64 x 64 output tiles per matrix: all of them, or the lower triangle for the Hermitian modes (2: error check, 3: X X^H)
__host__ __device__ inline int zgemm_tiles(int n, bool hermitian) {
    const int ntx = (n + 63) / 64;
    return hermitian ? ntx * (ntx + 1) / 2 : ntx * ntx;
}

// one 64 x 64 output tile (by, bx) of matrix b; `pidx`: where MODE 2 leaves its maximum
template <int MODE>
__device__ __forceinline__ void zgemm_tile(const cd* A, const cd* B, cd* Cm, int n, long long sA, long long sB, long long sC,
                                           int opB, int addI, const cd* Badd, const cd* Ref, double* part, int b, int by,
                                           int bx, size_t pidx) {
    __shared__ cd As[MT][MK + 1];
    // op(B) = B: Bs[k][j] (row stride MT + 1); op(B) = B^H: stored as it is read from memory, Bs[j][k] (row stride
    // MK + 1, like As)
    __shared__ cd Bsm[MT * (MK + 1) > MK * (MT + 1) ? MT * (MK + 1) : MK * (MT + 1)];
    const int ti = by * MT, tj = bx * MT;
    const cd* Ab = A + (size_t)b * sA;
    const cd* Bb = B + (size_t)b * sB;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = (wave >> 1) * 32, wc = (wave & 1) * 32;     // quadrant of this wave
    const int l15 = lane & 15, l4 = lane >> 4;
    f64x4 cr[2][2], ci[2][2];
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int w = 0; w < 2; ++w)
#pragma unroll
            for (int r = 0; r < 4; ++r) { cr[u][w][r] = 0.0; ci[u][w][r] = 0.0; }
    // 256 threads stage 64 x 8 of A and 8 x 64 of op(B): two elements each.  The elements of the NEXT k-step are
    // requested before the MFMAs of the current one (register prefetch), and B^H is read along its rows (k contiguous:
    // whole 128-byte lines) and transposed on the way into LDS - read by columns every lane touched its own line.
    cd pa[2], pb[2];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int e = tid + 256 * q;
            const int r = e >> 3, kk = e & 7;
            const int gi = ti + r, gk = k0 + kk;
            pa[q] = (gi < n && gk < n) ? Ab[(size_t)gi * n + gk] : make_double2(0.0, 0.0);
            cd v = make_double2(0.0, 0.0);
            if (opB == 0) {
                const int kk2 = e >> 6, cc = e & 63;
                const int gk2 = k0 + kk2, gj = tj + cc;
                if (gk2 < n && gj < n) {
                    v = Bb[(size_t)gk2 * n + gj];
                    if (MODE == 1) v = cadd(v, Badd[(size_t)gk2 * n + gj]);
                }
            } else {
                const int gj = tj + r;                           // row r of B = column r of B^H, k = kk
                if (gj < n && gk < n) {
                    v = Bb[(size_t)gj * n + gk];
                    v.y = -v.y;
                    if (MODE == 1) v = cadd(v, Badd[(size_t)gk * n + gj]);
                }
            }
            pb[q] = v;
        }
    };
    // addI & 2: B is LOWER triangular (the Cholesky factor U in psi^-1 U): rows k < tj of this column tile are zero
    const int kbeg = (addI & 2) ? (tj / MK) * MK : 0;
    fetch(kbeg);
    for (int k0 = kbeg; k0 < n; k0 += MK) {
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int e = tid + 256 * q;
            As[e >> 3][e & 7] = pa[q];
            if (opB == 0) Bsm[(e >> 6) * (MT + 1) + (e & 63)] = pb[q];
            else Bsm[(e >> 3) * (MK + 1) + (e & 7)] = pb[q];
        }
        if (k0 + MK < n) fetch(k0 + MK);
        __syncthreads();
#pragma unroll
        for (int ks = 0; ks < MK; ks += 4) {
            cd a[2], bb[2];
#pragma unroll
            for (int u = 0; u < 2; ++u) a[u] = As[wr + 16 * u + l15][ks + l4];
#pragma unroll
            for (int w = 0; w < 2; ++w)
                bb[w] = opB == 0 ? Bsm[(ks + l4) * (MT + 1) + wc + 16 * w + l15] : Bsm[(wc + 16 * w + l15) * (MK + 1) + ks + l4];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int w = 0; w < 2; ++w) {
                    cr[u][w] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u].x, bb[w].x, cr[u][w], 0, 0, 0);
                    ci[u][w] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u].x, bb[w].y, ci[u][w], 0, 0, 0);
                    cr[u][w] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a[u].y, bb[w].y, cr[u][w], 0, 0, 0);
                    ci[u][w] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[u].y, bb[w].x, ci[u][w], 0, 0, 0);
                }
        }
        __syncthreads();
    }
    if constexpr (MODE == 2) {
        const cd* Rb = Ref + (size_t)b * sC;
        double m = 0.0;
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int w = 0; w < 2; ++w)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int gi = ti + wr + 16 * u + l4 + 4 * r, gj = tj + wc + 16 * w + l15;
                    if (gi < n && gj < n) {
                        const cd a = Rb[(size_t)gi * n + gj];
                        const cd d = make_double2(a.x - cr[u][w][r], a.y - ci[u][w][r]);
                        const double e = sqrt(cabs2(d)) / sqrt(cabs2(a));
                        if (e > m || e != e) m = e;
                    }
                }
        __shared__ double red[256];
        red[tid] = m;
        __syncthreads();
        for (int st = 128; st > 0; st >>= 1) {
            if (tid < st) {
                const double o = red[tid + st];
                if (o > red[tid] || o != o) red[tid] = o;
            }
            __syncthreads();
        }
        if (tid == 0) part[pidx] = red[0];
        return;
    }
    cd* Cb = Cm + (size_t)b * sC;
#pragma unroll
    for (int u = 0; u < 2; ++u)
#pragma unroll
        for (int w = 0; w < 2; ++w)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int gi = ti + wr + 16 * u + l4 + 4 * r, gj = tj + wc + 16 * w + l15;
                if (gi < n && gj < n) {
                    cd v = make_double2(cr[u][w][r], ci[u][w][r]);
                    if ((addI & 1) && gi == gj) v.x += 1.0;
                    Cb[(size_t)gi * n + gj] = v;
                    if (MODE == 3 && bx < by) Cb[(size_t)gj * n + gi] = make_double2(v.x, -v.y);
                }
            }
}

// XCD-aware 1-D grid: workgroup ids are dealt round-robin to the 8 XCDs, so id % 8 picks the XCD and all tiles of
// one matrix get ids that are congruent mod 8 and consecutive in that XCD's order: the A / B panels the tiles share
// meet in ONE L2 (with a 3-D grid the 16 tiles of a 256 x 256 matrix landed on all 8 XCDs and every panel was
// fetched from HBM / Infinity Cache up to 8 times: the kernel ran at memory speed, 16.8 GB per product).
// The dispatcher hands out about one workgroup per microsecond and XCD, and a 64 x 64 x 256 tile is 28 us of matrix
// work on a CU that holds three of them: with one tile per workgroup the DISPATCHER set the pace (the Hermitian modes
// took as long as the full product while 6 of their 16 workgroups exited at once).  So the Hermitian modes launch
// the lower-triangle tiles only (zgemm_tiles), and every workgroup works through zgemm_tpw(MODE) tiles.
// (measured per mode at 2049 x 256 x 256: the plain product - whose four tiles of a row share the A panel - likes 4
// tiles per workgroup, 3.66 -> 2.84 ms with the triangular factor; the others are best with 2)
__host__ __device__ constexpr int zgemm_tpw(int mode) { return mode == 0 ? 4 : 2; }
__host__ __device__ inline int zgemm_groups(int n, int mode) {
    const int tpw = zgemm_tpw(mode);
    return (zgemm_tiles(n, mode == 2 || mode == 3) + tpw - 1) / tpw;
}

template <int MODE>
__global__ void __launch_bounds__(256) SPY_ZGEMM_KATTR zgemm_mfma_kernel(const cd* A, const cd* B, cd* Cm, int n, long long sA, long long sB,
                                                         long long sC, int opB, int addI, const cd* Badd, const cd* Ref,
                                                         double* part, int nbatch) {
    constexpr bool HERM = MODE == 2 || MODE == 3;
    constexpr int ZGEMM_TPW = zgemm_tpw(MODE);
    const int ntx = (n + MT - 1) / MT, ntile = zgemm_tiles(n, HERM), ngrp = zgemm_groups(n, MODE);
    const int lin = blockIdx.x, slot = lin >> 3;
    const int b = (slot / ngrp) * 8 + (lin & 7), grp = slot % ngrp;
    if (b >= nbatch) return;
    for (int rep = 0; rep < ZGEMM_TPW; ++rep) {
        const int tt = grp * ZGEMM_TPW + rep;
        if (tt >= ntile) break;
        int by, bx;
        if (HERM) {
            by = 0;
            while ((by + 1) * (by + 2) / 2 <= tt) ++by;
            bx = tt - by * (by + 1) / 2;
        } else {
            by = tt / ntx;
            bx = tt - by * ntx;
        }
        zgemm_tile<MODE>(A, B, Cm, n, sA, sB, sC, opB, addI, Badd, Ref, part, b, by, bx, (size_t)b * ntile + tt);
        __syncthreads();
    }
}

// S = triu(g0) - triu(g0)^H (wilson_sf.py:97-98) and g0 + S, both n x n
__global__ void __launch_bounds__(256) skew_kernel(const cd* g0, cd* S, cd* g0S, int n) {
    const int e = blockIdx.x * 256 + threadIdx.x;
    if (e >= n * n) return;
    const int i = e / n, j = e - i * n;
    cd v = make_double2(0.0, 0.0);
    if (j > i) v = g0[(size_t)i * n + j];
    else if (j < i) { const cd t = g0[(size_t)j * n + i]; v = make_double2(-t.x, t.y); }
    else { const cd t = g0[(size_t)i * n + i]; v = make_double2(0.0, 2.0 * t.y); }
    S[e] = v;
    g0S[e] = cadd(g0[e], v);
}

// max of a vector of partial maxima (NaN wins), one workgroup
__global__ void __launch_bounds__(256) maxred_kernel(const double* part, int n, double* out) {
    __shared__ double red[256];
    double m = 0.0;
    for (int e = threadIdx.x; e < n; e += 256) {
        const double v = part[e];
        if (v > m || v != v) m = v;
    }
    red[threadIdx.x] = m;
    __syncthreads();
    for (int st = 128; st > 0; st >>= 1) {
        if (threadIdx.x < st) {
            const double o = red[threadIdx.x + st];
            if (o > red[threadIdx.x] || o != o) red[threadIdx.x] = o;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) out[0] = red[0];
}

// ---- batched in-place inverse, blocked: Gauss-Jordan on ZB x ZB blocks.  The unblocked kernel below
// sweeps the whole matrix once per pivot (n sweeps of n*n*16 bytes: HBM/L2-bound, 230 ms for 2049
// matrices of 256 x 256); here one sweep serves ZB pivots:
//   for every block row k:  D = A_kk^-1;  R = D A_k* (with R_kk = D);  A_i* <- [A_i* off block k] - A_ik R  (i != k);  A_k* <- R
// The row block R (ZB x n) and D live in LDS.  Pivots are taken inside the diagonal block only, in
// order: `info[b] = 2` flags a (relatively) tiny pivot - the caller then repeats with the pivoted
// kernel.  Hermitian positive definite inputs (the CSD itself) never trip it.
constexpr int ZB = 16;
__global__ void __launch_bounds__(256) zinv_blocked_kernel(cd* M, int n, int* info) {
    SPY_DYN_SMEM(char, raw);
    const int npad = ((n + ZB - 1) / ZB) * ZB;
    cd* Rk = reinterpret_cast<cd*>(raw);             // ZB x npad
    cd* D = Rk + (size_t)ZB * npad;                  // ZB x ZB
    __shared__ double s_scale;
    cd* A = M + (size_t)blockIdx.x * n * n;
    const int tid = threadIdx.x;
    const int r = tid >> 4, c = tid & 15;            // element of the diagonal block
    const int cg = tid & 7, rl = tid >> 3;           // 32 rows x 8 column groups of the trailing update
    int bad = 0;
    for (int k0 = 0; k0 < npad; k0 += ZB) {
        // (a) D = A_kk (identity padding beyond n), inverted in LDS
        {
            const int i = k0 + r, j = k0 + c;
            D[r * ZB + c] = (i < n && j < n) ? A[(size_t)i * n + j] : make_double2(i == j ? 1.0 : 0.0, 0.0);
        }
        __syncthreads();
        if (tid == 0) {
            double m = 0.0;
            for (int e = 0; e < ZB * ZB; ++e) m = fmax(m, cabs2(D[e]));
            s_scale = m;
        }
        __syncthreads();
        for (int p = 0; p < ZB; ++p) {
            const cd piv = D[p * ZB + p], prow = D[p * ZB + c], pcol = D[r * ZB + p], own = D[r * ZB + c];
            __syncthreads();
            const double d = cabs2(piv);
            if (!(d > 1e-26 * s_scale)) bad = 1;
            const cd pinv = d > 0.0 ? make_double2(piv.x / d, -piv.y / d) : make_double2(0.0, 0.0);
            cd v;
            if (r == p) {
                v = (c == p) ? pinv : cmul(prow, pinv);
            } else {
                const cd f = cmul(pcol, pinv);
                v = (c == p) ? make_double2(-f.x, -f.y) : csub(own, cmul(f, prow));
            }
            D[r * ZB + c] = v;
            __syncthreads();
        }
        // (b) R = D A_k*, block k of R = D
        for (int j = tid; j < npad; j += 256) {
            if (j >= k0 && j < k0 + ZB) {
#pragma unroll
                for (int m = 0; m < ZB; ++m) Rk[(size_t)m * npad + j] = D[m * ZB + (j - k0)];
            } else {
                cd a[ZB];
#pragma unroll
                for (int q = 0; q < ZB; ++q)
                    a[q] = (k0 + q < n && j < n) ? A[(size_t)(k0 + q) * n + j] : make_double2(0.0, 0.0);
#pragma unroll
                for (int m = 0; m < ZB; ++m) {
                    cd acc = make_double2(0.0, 0.0);
#pragma unroll
                    for (int q = 0; q < ZB; ++q) acc = cadd(acc, cmul(D[m * ZB + q], a[q]));
                    Rk[(size_t)m * npad + j] = acc;
                }
            }
        }
        __syncthreads();
        // (c) rows outside block k: A_ij <- (j in block k ? 0 : A_ij) - sum_m A_i,k0+m R_mj
        //     (the 8 threads of a row all hold T = A_i,block k before any of them overwrites it: barrier)
        for (int i0 = 0; i0 < n; i0 += 32) {
            const int i = i0 + rl;
            const bool valid = i < n && !(i >= k0 && i < k0 + ZB);
            cd T[ZB];
#pragma unroll
            for (int m = 0; m < ZB; ++m)
                T[m] = (valid && k0 + m < n) ? A[(size_t)i * n + k0 + m] : make_double2(0.0, 0.0);
            __syncthreads();
            if (valid) {
                for (int j = cg; j < n; j += 8) {
                    cd acc = (j >= k0 && j < k0 + ZB) ? make_double2(0.0, 0.0) : A[(size_t)i * n + j];
#pragma unroll
                    for (int m = 0; m < ZB; ++m) acc = csub(acc, cmul(T[m], Rk[(size_t)m * npad + j]));
                    A[(size_t)i * n + j] = acc;
                }
            }
        }
        // (d) rows of block k
        for (int e = tid; e < ZB * n; e += 256) {
            const int m = e / n, j = e - m * n;
            if (k0 + m < n) A[(size_t)(k0 + m) * n + j] = Rk[(size_t)m * npad + j];
        }
        __syncthreads();
    }
    if (tid == 0) info[blockIdx.x] = 0;
    __syncthreads();
    if (bad) info[blockIdx.x] = 2;
}

// ---- the blocked inverse on the fp64 matrix cores.  Same algorithm as zinv_blocked_kernel with blocks of ZM = 32:
//   for every block row k:  D = A_kk^-1;  R = D A_k* (R_kk = D);  A_i* <- [A_i* off block k] - A_ik R  (i != k);  A_k* <- R
// but the two products run as v_mfma_f64_16x16x4_f64 tiles (layouts as zgemm_mfma_kernel): the trailing update - all
// of the 8 n^3 flops - is 16 x 16 output tiles C = A_ij + (-A_ik) R_kj whose accumulators start as the tile itself,
// a wave owns whole rows of tiles (so the in-place update has no hazards between waves) and reads R from LDS.
// Twice the block size halves the sweeps over the matrix (8 x 2 MB instead of 16 x 2 MB at n = 256: the VALU
// kernel sat at HBM / Infinity-Cache speed, 23 ms for 2049 matrices).  Pivots inside the diagonal block only, in
// order; info = 2 flags a (relatively) tiny pivot exactly as zinv_blocked_kernel does.
constexpr int ZM = 32;
constexpr int ZT = 512;      // threads: 8 waves = two per SIMD (an MFMA blocks its wave; the partner keeps the pipe busy)
constexpr int ZJ = 4;        // column tiles per pass of the trailing update: 8 independent accumulators per wave
// `src` != nullptr: out of place - the first sweep reads src, everything lands in M (saves the caller a copy)
__global__ void __launch_bounds__(ZT) zinv_mfma_kernel(cd* M, const cd* src, int n, int* info) {
    SPY_DYN_SMEM(char, raw);
    const int npad = ((n + ZM - 1) / ZM) * ZM;
    const int ldr = npad + 1;                        // row stride of R in LDS (cd units; odd: spreads the banks)
    cd* Rk = reinterpret_cast<cd*>(raw);             // ZM x ldr
    cd* D = Rk + (size_t)ZM * ldr;                   // ZM x (ZM + 1)
    constexpr int LDD = ZM + 1;
    __shared__ double s_scale;
    cd* A = M + (size_t)blockIdx.x * n * n;
    const cd* S0 = src ? src + (size_t)blockIdx.x * n * n : A;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int dr = tid >> 4, dc0 = tid & 15;         // diagonal block: row dr, columns dc0 and dc0 + 16
    const int ntile = npad / 16;
    int bad = 0;
    for (int k0 = 0; k0 < npad; k0 += ZM) {
        const cd* S = k0 == 0 ? S0 : A;              // where this sweep reads the matrix
        // (a) D = A_kk (identity padding beyond n), inverted in LDS
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            const int i = k0 + dr, j = k0 + dc0 + 16 * q;
            D[dr * LDD + dc0 + 16 * q] = (i < n && j < n) ? S[(size_t)i * n + j] : make_double2(i == j ? 1.0 : 0.0, 0.0);
        }
        __syncthreads();
        if (tid < 64) {                              // largest |entry|^2 of the block (pivot threshold)
            double m = 0.0;
            for (int e = tid; e < ZM * ZM; e += 64) m = fmax(m, cabs2(D[(e / ZM) * LDD + (e % ZM)]));
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) m = fmax(m, __shfl_xor(m, off));
            if (tid == 0) s_scale = m;
        }
        __syncthreads();
        for (int p = 0; p < ZM; ++p) {
            const cd piv = D[p * LDD + p], pcol = D[dr * LDD + p];
            cd prow[2], own[2];
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                prow[q] = D[p * LDD + dc0 + 16 * q];
                own[q] = D[dr * LDD + dc0 + 16 * q];
            }
            __syncthreads();
            const double d = cabs2(piv);
            if (!(d > 1e-26 * s_scale)) bad = 1;
            const cd pinv = d > 0.0 ? make_double2(piv.x / d, -piv.y / d) : make_double2(0.0, 0.0);
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                const int c = dc0 + 16 * q;
                cd v;
                if (dr == p) {
                    v = (c == p) ? pinv : cmul(prow[q], pinv);
                } else {
                    const cd f = cmul(pcol, pinv);
                    v = (c == p) ? make_double2(-f.x, -f.y) : csub(own[q], cmul(f, prow[q]));
                }
                D[dr * LDD + c] = v;
            }
            __syncthreads();
        }
        // (b) R = D A_k* on the matrix cores (column tiles dealt to the waves); the two tiles of block k get D itself
        for (int jt = wave; jt < ntile; jt += ZT / 64) {
            const int j = 16 * jt + l15;
            const bool inblk = 16 * jt >= k0 && 16 * jt < k0 + ZM;
            f64x4 cr[2], ci[2];
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) { cr[u][r] = 0.0; ci[u][r] = 0.0; }
            if (!inblk) {
#pragma unroll
                for (int ks = 0; ks < ZM; ks += 4) {
                    const int gk = k0 + ks + l4;
                    const cd bb = (gk < n && j < n) ? S[(size_t)gk * n + j] : make_double2(0.0, 0.0);
#pragma unroll
                    for (int u = 0; u < 2; ++u) {
                        const cd a = D[(16 * u + l15) * LDD + ks + l4];
                        cr[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, bb.x, cr[u], 0, 0, 0);
                        ci[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, bb.y, ci[u], 0, 0, 0);
                        cr[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a.y, bb.y, cr[u], 0, 0, 0);
                        ci[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.y, bb.x, ci[u], 0, 0, 0);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int m = 16 * u + l4 + 4 * r;
                    Rk[(size_t)m * ldr + j] = inblk ? D[m * LDD + (j - k0)] : make_double2(cr[u][r], ci[u][r]);
                }
        }
        __syncthreads();
        // (c) rows outside block k, one row of 16 x 16 tiles per wave, ZJ column tiles at a time:
        //     A_ij <- (j in block k ? 0 : A_ij) + (-A_i,blockk) R_kj
        for (int it = wave; it < ntile; it += ZT / 64) {
            if (16 * it >= k0 && 16 * it < k0 + ZM) continue;           // wave-uniform
            const int gi = 16 * it + l15;
            cd ta[ZM / 4];                                               // -A[gi][k0 + 4 q + l4]
#pragma unroll
            for (int q = 0; q < ZM / 4; ++q) {
                const int gk = k0 + 4 * q + l4;
                const cd t = (gi < n && gk < n) ? S[(size_t)gi * n + gk] : make_double2(0.0, 0.0);
                ta[q] = make_double2(-t.x, -t.y);
            }
            // the tiles of the NEXT pass are requested before the MFMAs of the current one (an MFMA blocks its wave:
            // loads issued after them would only start when they are over)
            auto fetch = [&](int jt0, cd (&c)[ZJ][4]) {
#pragma unroll
                for (int v = 0; v < ZJ; ++v) {
                    const int jt = jt0 + v, j = 16 * jt + l15;
                    const bool inblk = 16 * jt >= k0 && 16 * jt < k0 + ZM;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = 16 * it + l4 + 4 * r;
                        c[v][r] = (jt < ntile && !inblk && i < n && j < n) ? S[(size_t)i * n + j] : make_double2(0.0, 0.0);
                    }
                }
            };
            cd cn[ZJ][4];
            fetch(0, cn);
            for (int jt0 = 0; jt0 < ntile; jt0 += ZJ) {
                f64x4 cr[ZJ], ci[ZJ];
#pragma unroll
                for (int v = 0; v < ZJ; ++v)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        cr[v][r] = cn[v][r].x;
                        ci[v][r] = cn[v][r].y;
                    }
                if (jt0 + ZJ < ntile) fetch(jt0 + ZJ, cn);
#pragma unroll
                for (int q = 0; q < ZM / 4; ++q) {
                    cd bb[ZJ];
#pragma unroll
                    for (int v = 0; v < ZJ; ++v)
                        bb[v] = jt0 + v < ntile ? Rk[(size_t)(4 * q + l4) * ldr + 16 * (jt0 + v) + l15] : make_double2(0.0, 0.0);
#pragma unroll
                    for (int v = 0; v < ZJ; ++v) {
                        cr[v] = __builtin_amdgcn_mfma_f64_16x16x4f64(ta[q].x, bb[v].x, cr[v], 0, 0, 0);
                        ci[v] = __builtin_amdgcn_mfma_f64_16x16x4f64(ta[q].x, bb[v].y, ci[v], 0, 0, 0);
                    }
#pragma unroll
                    for (int v = 0; v < ZJ; ++v) {
                        cr[v] = __builtin_amdgcn_mfma_f64_16x16x4f64(-ta[q].y, bb[v].y, cr[v], 0, 0, 0);
                        ci[v] = __builtin_amdgcn_mfma_f64_16x16x4f64(ta[q].y, bb[v].x, ci[v], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int v = 0; v < ZJ; ++v) {
                    const int jt = jt0 + v, j = 16 * jt + l15;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = 16 * it + l4 + 4 * r;
                        if (jt < ntile && i < n && j < n) A[(size_t)i * n + j] = make_double2(cr[v][r], ci[v][r]);
                    }
                }
            }
        }
        // (d) rows of block k
        for (int e = tid; e < ZM * n; e += ZT) {
            const int m = e / n, j = e - m * n;
            if (k0 + m < n) A[(size_t)(k0 + m) * n + j] = Rk[(size_t)m * ldr + j];
        }
        __syncthreads();
    }
    if (tid == 0) info[blockIdx.x] = 0;
    __syncthreads();
    if (bad) info[blockIdx.x] = 2;
}

// ---- the same block Gauss-Jordan with blocks of ZW = 64 rows: FOUR sweeps over a 256 x 256 matrix instead of eight
// (the 32-row kernel moves 8 x 2 MB per matrix at ~2.8 TB/s: it is bound by those sweeps, not by its 8 n^3 flops).
// A 64-row R panel does not fit LDS next to D - it does not have to: a wave owns a COLUMN tile (16 columns), computes
// its 64 x 16 piece of R = D A_k* into registers (64 VGPRs) and keeps it there for the whole block step; what
// passes through LDS is the column block k of the other rows (the A operand of the trailing update), 64 rows at a time.
// LDS: D and that panel, 64 x 65 complex128 each (133 KB) whatever n is.  Pivots inside the diagonal block only, in
// order; info = 2 flags a (relatively) tiny pivot exactly as the other blocked kernels do.
// Measured at 2049 x 256 x 256 (profiles/r3_wilson_inverse_variants.txt): 32-row blocks 12.2 ms; 64-row blocks with the
// matrix walked in column quarters through an R panel in LDS 10.2; + register-resident diagonal-block inversion 8.9;
// this version 7.5 (its parts ADD UP - memory skeleton 3.4 + diagonal blocks 1.6 + MFMA phases 3.0 - one workgroup per
// CU runs them one after the other); two 4-wave workgroups per CU (R pieces parked in the rows of block k, D's LDS
// reused for the panel) 9.2: 512 matrices in flight no longer fit the 256 MB Infinity Cache between sweeps.
constexpr int ZW = 64;
__global__ void __launch_bounds__(ZT) zinv64_mfma_kernel(cd* M, const cd* src, int n, int* info) {
    SPY_DYN_SMEM(char, raw);
    constexpr int LDD = ZW + 1;
    cd* const D = reinterpret_cast<cd*>(raw);        // ZW x LDD
    cd* const Rq = D + (size_t)ZW * LDD;             // ZW x LDD
    __shared__ double s_scale;
    const int npad = ((n + ZW - 1) / ZW) * ZW, nq = npad / ZW, ntile = npad / 16;
    cd* A = M + (size_t)blockIdx.x * n * n;
    const cd* S0 = src ? src + (size_t)blockIdx.x * n * n : A;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int l15 = lane & 15, l4 = lane >> 4;
    const int dr = tid >> 3, dc0 = tid & 7;          // diagonal block: row dr, columns dc0 + 8 q
    int bad = 0;
    for (int kb = 0; kb < nq; ++kb) {
        const int k0 = kb * ZW;
        const cd* S = kb == 0 ? S0 : A;              // where this sweep reads the matrix
        // (a) D = A_kk (identity padding beyond n), inverted by in-block Gauss-Jordan.  A thread OWNS row dr, columns
        // dc0 + 8 q in registers for all 64 pivots; a pivot step only publishes pivot row and pivot column in LDS
        // (two buffers in turn: one barrier per pivot) - 10 LDS reads per thread and pivot instead of 18 + 8 writes
        cd own[8];
        double m2 = 0.0;
#pragma unroll
        for (int q = 0; q < 8; ++q) {
            const int i = k0 + dr, j = k0 + dc0 + 8 * q;
            own[q] = (i < n && j < n) ? S[(size_t)i * n + j] : make_double2(i == j ? 1.0 : 0.0, 0.0);
            m2 = fmax(m2, cabs2(own[q]));
        }
        {                                            // largest |entry|^2 of the block (pivot threshold)
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) m2 = fmax(m2, __shfl_xor(m2, off));
            double* const red = reinterpret_cast<double*>(Rq);
            if (lane == 0) red[wave] = m2;
            __syncthreads();
            if (tid == 0) {
                double m = red[0];
                for (int w = 1; w < ZT / 64; ++w) m = fmax(m, red[w]);
                s_scale = m;
            }
            __syncthreads();
        }
        const double thresh = 1e-26 * s_scale;
        __syncthreads();                             // (everybody has read s_scale / red before Rq is reused)
        // publication buffers in Rq (free until the first quarter): [buf][0..63] pivot row, [buf][64..127] pivot column
        cd* const pub = Rq;
#pragma unroll
        for (int pq = 0; pq < 8; ++pq) {             // pivot p = pr + 8 pq: held in own[pq] by the threads with dc0 = pr
#pragma unroll 1
            for (int pr = 0; pr < 8; ++pr) {
                const int p = pr + 8 * pq;
                cd* const pb = pub + (p & 1) * 2 * ZW;
                if (dr == p) {
#pragma unroll
                    for (int q = 0; q < 8; ++q) pb[dc0 + 8 * q] = own[q];
                }
                if (dc0 == pr) pb[ZW + dr] = own[pq];
                __syncthreads();
                const cd piv = pb[p], pcol = pb[ZW + dr];
                const double d = cabs2(piv);
                if (!(d > thresh)) bad = 1;
                const double rd = d > 0.0 ? 1.0 / d : 0.0;         // (one division on the pivot chain instead of two)
                const cd pinv = make_double2(piv.x * rd, -piv.y * rd);
                const cd f = cmul(pcol, pinv);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int c = dc0 + 8 * q;
                    const cd prow = pb[c];
                    cd v;
                    if (dr == p) v = (c == p) ? pinv : cmul(prow, pinv);
                    else v = (c == p) ? make_double2(-f.x, -f.y) : csub(own[q], cmul(f, prow));
                    own[q] = v;
                }
            }
        }
        __syncthreads();                             // the last publication has been read: Rq is free
#pragma unroll
        for (int q = 0; q < 8; ++q) D[dr * LDD + dc0 + 8 * q] = own[q];
        __syncthreads();
        // (b) - (d): a wave owns ONE column tile per pass of 8 tiles (the tiles of block k last: nobody reads column block
        // k from memory after it has been overwritten).  Its R fragment - R_k,jt = D A_k,jt, 64 x 16 - is computed
        // into registers and never leaves them: the accumulator layout of the four 16 x 16 products IS the B-operand
        // layout of the trailing update (lane (l15, l4) holds rows 4 ks + l4, ks = 4 u + r, of column l15).  The
        // (negated) column block k of the other rows passes through LDS, 64 rows at a time.
        const int nout = ntile - 4;
        for (int e0 = 0; e0 < ntile; e0 += ZT / 64) {
            const int e = e0 + wave;
            const bool have = e < ntile;                                  // wave-uniform
            const int jt = !have ? 0 : (e < nout ? (e < 4 * kb ? e : e + 4) : 4 * kb + (e - nout));
            const bool own_t = have && e >= nout;                         // a tile of block k itself: R = D
            const int j = 16 * jt + l15;
            cd Rf[ZW / 4];
            if (own_t) {
#pragma unroll
                for (int ks = 0; ks < ZW / 4; ++ks) Rf[ks] = D[(4 * ks + l4) * LDD + 16 * (jt - 4 * kb) + l15];
            } else if (have) {
                f64x4 cr[4], ci[4];
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int r = 0; r < 4; ++r) { cr[u][r] = 0.0; ci[u][r] = 0.0; }
                cd bb[ZW / 4];
#pragma unroll
                for (int ks = 0; ks < ZW / 4; ++ks) {
                    const int gk = k0 + 4 * ks + l4;
                    bb[ks] = (gk < n && j < n) ? S[(size_t)gk * n + j] : make_double2(0.0, 0.0);
                }
#pragma unroll
                for (int ks = 0; ks < ZW / 4; ++ks) {
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        const cd a = D[(16 * u + l15) * LDD + 4 * ks + l4];
                        cr[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, bb[ks].x, cr[u], 0, 0, 0);
                        ci[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.x, bb[ks].y, ci[u], 0, 0, 0);
                        cr[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(-a.y, bb[ks].y, cr[u], 0, 0, 0);
                        ci[u] = __builtin_amdgcn_mfma_f64_16x16x4f64(a.y, bb[ks].x, ci[u], 0, 0, 0);
                    }
                }
#pragma unroll
                for (int u = 0; u < 4; ++u)
#pragma unroll
                    for (int r = 0; r < 4; ++r) Rf[4 * u + r] = make_double2(cr[u][r], ci[u][r]);
            }
            // (d) rows of block k of this column tile
            if (have) {
#pragma unroll
                for (int ks = 0; ks < ZW / 4; ++ks) {
                    const int i = k0 + 4 * ks + l4;
                    if (i < n && j < n) A[(size_t)i * n + j] = Rf[ks];
                }
            }
            // (c) the other block rows: A_i,jt <- (tile of block k ? 0 : A_i,jt) + (-A_i,k) R_k,jt
            auto fetch = [&](int it0, cd (&c)[2][4]) {                   // accumulator start of row tiles it0, it0 + 1
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = 16 * (it0 + t) + l4 + 4 * r;
                        c[t][r] = (have && !own_t && i < n && j < n) ? S[(size_t)i * n + j] : make_double2(0.0, 0.0);
                    }
            };
            cd cn[2][4];
            if (nq > 1) fetch(4 * (kb == 0 ? 1 : 0), cn);
            for (int cb = 0; cb < nq; ++cb) {
                if (cb == kb) continue;
                __syncthreads();                                         // the previous panel has been read
                // (prefetching the next panel into registers across the MFMAs was measured: no gain, 34 more spills)
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int i = ZW * cb + dr, jj = k0 + dc0 + 8 * q;
                    const cd t = (i < n && jj < n) ? S[(size_t)i * n + jj] : make_double2(0.0, 0.0);
                    Rq[dr * LDD + dc0 + 8 * q] = make_double2(-t.x, -t.y);
                }
                __syncthreads();
#pragma unroll 1
                for (int pr = 0; pr < 2; ++pr) {
                    const int it0 = 4 * cb + 2 * pr;
                    f64x4 cr[2], ci[2];
#pragma unroll
                    for (int t = 0; t < 2; ++t)
#pragma unroll
                        for (int r = 0; r < 4; ++r) { cr[t][r] = cn[t][r].x; ci[t][r] = cn[t][r].y; }
                    // the next pair's tiles are requested before the MFMAs of this one
                    int nxt = pr == 0 ? it0 + 2 : -1;
                    if (pr == 1) {
                        int cbn = cb + 1;
                        if (cbn == kb) ++cbn;
                        if (cbn < nq) nxt = 4 * cbn;
                    }
                    if (nxt >= 0) fetch(nxt, cn);
                    if (have) {
#pragma unroll
                        for (int ks = 0; ks < ZW / 4; ++ks) {
                            cd aa[2];
#pragma unroll
                            for (int t = 0; t < 2; ++t) aa[t] = Rq[(16 * (2 * pr + t) + l15) * LDD + 4 * ks + l4];
#pragma unroll
                            for (int t = 0; t < 2; ++t) {
                                cr[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(aa[t].x, Rf[ks].x, cr[t], 0, 0, 0);
                                ci[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(aa[t].x, Rf[ks].y, ci[t], 0, 0, 0);
                            }
#pragma unroll
                            for (int t = 0; t < 2; ++t) {
                                cr[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(-aa[t].y, Rf[ks].y, cr[t], 0, 0, 0);
                                ci[t] = __builtin_amdgcn_mfma_f64_16x16x4f64(aa[t].y, Rf[ks].x, ci[t], 0, 0, 0);
                            }
                        }
#pragma unroll
                        for (int t = 0; t < 2; ++t)
#pragma unroll
                            for (int r = 0; r < 4; ++r) {
                                const int i = 16 * (it0 + t) + l4 + 4 * r;
                                if (i < n && j < n) A[(size_t)i * n + j] = make_double2(cr[t][r], ci[t][r]);
                            }
                    }
                }
            }
            __syncthreads();                         // panel and D are free (next pass of tiles / next block)
        }
    }
    if (tid == 0) info[blockIdx.x] = 0;
    __syncthreads();
    if (bad) info[blockIdx.x] = 2;
}

// ---- batched in-place inverse: Gauss-Jordan with partial pivoting, one workgroup per matrix.
// `info[b]` = 1 if a zero pivot was met.
__global__ void __launch_bounds__(256) zinv_kernel(cd* M, int n, int* info) {
    SPY_DYN_SMEM(char, raw);
    cd* rowk = reinterpret_cast<cd*>(raw);            // n
    cd* colk = rowk + n;                              // n
    int* perm = reinterpret_cast<int*>(colk + n);     // n
    __shared__ double redv[256];
    __shared__ int redi[256];
    cd* A = M + (size_t)blockIdx.x * n * n;
    const int tid = threadIdx.x;
    int bad = 0;
    for (int k = 0; k < n; ++k) {
        // pivot search in column k, rows k..n-1
        double best = -1.0;
        int bi = k;
        for (int i = k + tid; i < n; i += 256) {
            const double v = cabs2(A[(size_t)i * n + k]);
            if (v > best) { best = v; bi = i; }
        }
        redv[tid] = best;
        redi[tid] = bi;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (tid < s) {
                if (redv[tid + s] > redv[tid] || (redv[tid + s] == redv[tid] && redi[tid + s] < redi[tid])) {
                    redv[tid] = redv[tid + s];
                    redi[tid] = redi[tid + s];
                }
            }
            __syncthreads();
        }
        const int p = redi[0];
        if (redv[0] <= 0.0) bad = 1;
        if (tid == 0) perm[k] = p;
        // swap rows k and p, cache the (new) pivot row
        for (int j = tid; j < n; j += 256) {
            const cd a = A[(size_t)k * n + j], c = A[(size_t)p * n + j];
            if (p != k) { A[(size_t)p * n + j] = a; }
            rowk[j] = c;
        }
        __syncthreads();
        const cd piv = rowk[k];
        const double d = cabs2(piv);
        const cd pinv = d > 0.0 ? make_double2(piv.x / d, -piv.y / d) : make_double2(0.0, 0.0);
        for (int j = tid; j < n; j += 256) {
            cd v = (j == k) ? pinv : cmul(rowk[j], pinv);
            A[(size_t)k * n + j] = v;
            // column k of the other rows (read before it is overwritten)
            colk[j] = (j == k) ? make_double2(0.0, 0.0) : A[(size_t)j * n + k];
        }
        __syncthreads();
        for (int j = tid; j < n; j += 256) rowk[j] = A[(size_t)k * n + j];
        __syncthreads();
        // eliminate: rows i != k
        for (int e = tid; e < n * n; e += 256) {
            const int i = e / n, j = e - i * n;
            if (i == k) continue;
            const cd f = colk[i];
            if (j == k) A[e] = make_double2(-(f.x * rowk[k].x - f.y * rowk[k].y), -(f.x * rowk[k].y + f.y * rowk[k].x));
            else A[e] = csub(A[e], cmul(f, rowk[j]));
        }
        __syncthreads();
    }
    // undo the row permutation: swap columns in reverse order
    for (int k = n - 1; k >= 0; --k) {
        const int p = perm[k];
        if (p != k) {
            for (int i = tid; i < n; i += 256) {
                const cd a = A[(size_t)i * n + k];
                A[(size_t)i * n + k] = A[(size_t)i * n + p];
                A[(size_t)i * n + p] = a;
            }
        }
        __syncthreads();
    }
    if (tid == 0 && info) info[blockIdx.x] = bad;
}

// ---- batched Cholesky A = L L^H (lower, in place; the strict upper triangle is zeroed), one workgroup per matrix
__global__ void __launch_bounds__(256) zchol_kernel(cd* M, int n, int* info) {
    SPY_DYN_SMEM(char, raw);
    cd* colk = reinterpret_cast<cd*>(raw);   // n
    cd* A = M + (size_t)blockIdx.x * n * n;
    const int tid = threadIdx.x;
    int bad = 0;
    for (int k = 0; k < n; ++k) {
        const double dk = A[(size_t)k * n + k].x;
        __syncthreads();                       // every thread has read the pivot before it is overwritten
        if (!(dk > 0.0)) bad = 1;
        const double l = sqrt(dk > 0.0 ? dk : 1.0);
        for (int i = k + tid; i < n; i += 256) {
            cd v = A[(size_t)i * n + k];
            v = (i == k) ? make_double2(l, 0.0) : make_double2(v.x / l, v.y / l);
            A[(size_t)i * n + k] = v;
            colk[i] = v;
        }
        __syncthreads();
        const int m = n - k - 1;
        for (int e = tid; e < m * m; e += 256) {
            const int i = k + 1 + e / m, j = k + 1 + e % m;
            if (j <= i) A[(size_t)i * n + j] = csub(A[(size_t)i * n + j], cmulc(colk[i], colk[j]));
        }
        __syncthreads();
    }
    for (int e = tid; e < n * n; e += 256) {
        const int i = e / n, j = e - i * n;
        if (j > i) A[e] = make_double2(0.0, 0.0);
    }
    if (tid == 0 && info) info[blockIdx.x] = bad;
}

// ---- the same factorisation, left-looking in panels of 32 columns (n <= 256): the right-looking kernel above sweeps
// the whole trailing matrix once per COLUMN (48 GB of HBM traffic for 2049 matrices of 256 x 256, 46 ms); here a
// panel is read once, receives the contributions of all earlier columns from registers (one thread per row, 32
// accumulators; the 32 rows of L the panel's columns need are staged through LDS in chunks of 32 columns), is
// factorised inside LDS and written once: ~3 MB per matrix.  Same arithmetic per element, other summation order.
constexpr int CHP = 32;
__global__ void __launch_bounds__(256) zchol_panel_kernel(cd* M, int n, int* info) {
    SPY_DYN_SMEM(char, raw);
    cd* P = reinterpret_cast<cd*>(raw);                  // n x (CHP + 1): the panel, row i at P[i * (CHP + 1)]
    cd* Lr = P + (size_t)n * (CHP + 1);                  // CHP x (CHP + 1): rows J0 .. J0+31 of L, columns of the current chunk
    constexpr int LDP = CHP + 1;
    cd* A = M + (size_t)blockIdx.x * n * n;
    const int i = threadIdx.x;                           // this thread's row
    int bad = 0;
    for (int J0 = 0; J0 < n; J0 += CHP) {
        const int pw = n - J0 < CHP ? n - J0 : CHP;      // panel width
        cd acc[CHP];
#pragma unroll
        for (int c = 0; c < CHP; ++c)
            acc[c] = (i >= J0 && i < n && c < pw) ? A[(size_t)i * n + J0 + c] : make_double2(0.0, 0.0);
        // contributions of the columns before the panel: acc[c] -= sum_k L[i, k] conj(L[J0 + c, k])
        for (int k0 = 0; k0 < J0; k0 += CHP) {
            __syncthreads();
            for (int e = threadIdx.x; e < CHP * CHP; e += 256) {
                const int r = e / CHP, kk = e % CHP;
                Lr[r * LDP + kk] = (J0 + r < n) ? A[(size_t)(J0 + r) * n + k0 + kk] : make_double2(0.0, 0.0);
            }
            __syncthreads();
            if (i >= J0 && i < n) {
                for (int kk = 0; kk < CHP; ++kk) {
                    const cd a = A[(size_t)i * n + k0 + kk];
#pragma unroll
                    for (int c = 0; c < CHP; ++c) acc[c] = csub(acc[c], cmulc(a, Lr[c * LDP + kk]));
                }
            }
        }
        __syncthreads();
        if (i >= J0 && i < n) {
#pragma unroll
            for (int c = 0; c < CHP; ++c) P[(size_t)i * LDP + c] = acc[c];
        }
        __syncthreads();
        // factorise the panel inside LDS, column by column
        for (int c = 0; c < pw; ++c) {
            const int pr = J0 + c;                       // pivot row
            const double dk = P[(size_t)pr * LDP + c].x;
            if (!(dk > 0.0)) bad = 1;
            const double l = sqrt(dk > 0.0 ? dk : 1.0);
            __syncthreads();                             // everybody has the pivot
            if (i >= pr && i < n) {
                const cd v = P[(size_t)i * LDP + c];
                P[(size_t)i * LDP + c] = (i == pr) ? make_double2(l, 0.0) : make_double2(v.x / l, v.y / l);
            }
            __syncthreads();
            if (i > pr && i < n) {
                const cd a = P[(size_t)i * LDP + c];
                for (int c2 = c + 1; c2 < pw; ++c2)
                    if (i >= J0 + c2) P[(size_t)i * LDP + c2] = csub(P[(size_t)i * LDP + c2], cmulc(a, P[(size_t)(J0 + c2) * LDP + c]));
            }
            __syncthreads();
        }
        // the panel's columns: L on and below the diagonal, zeros above it
        if (i < n) {
            for (int c = 0; c < pw; ++c)
                A[(size_t)i * n + J0 + c] = (i >= J0 + c) ? P[(size_t)i * LDP + c] : make_double2(0.0, 0.0);
        }
        __syncthreads();
    }
    if (threadIdx.x == 0 && info) info[blockIdx.x] = bad;
}

// ---- gamma0 = sum over the full (mirrored) frequency axis = A[0] + A[F-1] + sum_{0<f<F-1} 2 Re-part...
// (fft(CSD_full)[0], wilson_sf.py:135-140), then symmetrised real part: out[i,j] = Re((g[i,j] + conj(g[j,i]))/2)
// (A holds the F bins [f_lo, f_lo + F) of Ftot: a frequency shard contributes its part of the sum)
__global__ void __launch_bounds__(256) gamma0_kernel(const cd* A, int F, int n, cd* out, int f_lo, int Ftot) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n * n) return;
    const int i = e / n, j = e - i * n;
    double s = 0.0;
    for (int f = 0; f < F; ++f) {
        const double w = (f + f_lo == 0 || f + f_lo == Ftot - 1) ? 1.0 : 2.0;   // +f and -f: a + conj(a) = 2 Re(a) ... per entry pair
        // full-spectrum sum of entry (i,j): A[f][i][j] + conj(A[f][i][j]) for mirrored f
        const cd a = A[((size_t)f * n + i) * n + j], b = A[((size_t)f * n + j) * n + i];
        // Re( (g_ij + conj(g_ji)) / 2 ) with g = sum over full spectrum
        s += 0.5 * w * (a.x + b.x);
        (void)a;
    }
    out[e] = make_double2(s, 0.0);
}

// ---- plus operator along the frequency axis for every matrix entry (wilson_sf.py:154-184)
// one workgroup per entry (i,j): half spectrum g[0..F-1] -> conjugate-symmetric sequence of length
// L = 2(F-1) -> inverse DFT (real) -> halve lag 0 and lag L/2, zero negative lags -> forward DFT.
// Generic length: Stockham passes with radices 2..16 and O(R^2) butterflies for other primes.
// (PlusPlan and the Stockham passes po_pass / po_pass_r24 / po_pass_any: f64_stockham.h, shared with the
// reference-precision transform of any length)

// body of the plus operator for entry e with the two length-L working arrays a, b (LDS or global scratch)
__device__ __forceinline__ void plus_entry(const cd* g, int F, size_t fs, long long e, const PlusPlan& pl, const cd* tw, cd* gp,
                                           cd* g0, cd* a, cd* b, int tid) {
    const int L = pl.L;
    for (int f = tid; f < F; f += 256) {
        const cd v = g[(size_t)f * fs + e];
        a[f] = v;
        if (f > 0 && f < F - 1) a[L - f] = make_double2(v.x, -v.y);
    }
    __syncthreads();
    int Ns = 1;
    for (int p = 0; p < pl.nfac; ++p) {          // inverse DFT (unnormalised)
        po_pass_any(a, b, L, pl.radix[p], Ns, tw, +1, tid);
        __syncthreads();
        Ns *= pl.radix[p];
        cd* t = a; a = b; b = t;
    }
    const double invL = 1.0 / (double)L;
    const int half = L / 2;
    for (int t = tid; t < L; t += 256) {
        double beta = a[t].x * invL;             // np.real(ifft(g))
        if (t == 0 || t == half) beta *= 0.5;
        if (t > half) beta = 0.0;
        a[t] = make_double2(beta, 0.0);
    }
    __syncthreads();
    if (tid == 0) g0[e] = a[0];
    __syncthreads();
    Ns = 1;
    for (int p = 0; p < pl.nfac; ++p) {          // forward DFT
        po_pass_any(a, b, L, pl.radix[p], Ns, tw, -1, tid);
        __syncthreads();
        Ns *= pl.radix[p];
        cd* t = a; a = b; b = t;
    }
    for (int f = tid; f < F; f += 256) gp[(size_t)f * fs + e] = a[f];
}

// nent = entries per frequency row (n^2 for the whole matrix, fewer for an entry shard of the sharded factorisation)
__global__ void __launch_bounds__(256) plus_kernel(const cd* g, int F, long long nent, PlusPlan pl, const cd* tw, cd* gp, cd* g0) {
    SPY_DYN_SMEM(cd, buf);          // 2 x L
    plus_entry(g, F, (size_t)nent, blockIdx.x, pl, tw, gp, g0, buf, buf + pl.L, threadIdx.x);
}

// The same operator for lag-domain lengths whose working arrays (2 x L complex128) do not fit LDS - trials longer than
// 5120 samples (wilson_sf.py:154-184 has no length limit): the arrays live in global scratch, 2 L entries per workgroup
// (512 KiB at L = 16384: L2-resident while the workgroup works on them; the passes are the same Stockham passes, the
// workgroup barrier orders its own global stores and loads).  One workgroup per entry e0 + blockIdx.x.
__global__ void __launch_bounds__(256) plus_long_kernel(const cd* g, int F, long long nent, PlusPlan pl, const cd* tw, cd* gp, cd* g0,
                                                        cd* scr, long long e0) {
    cd* const a = scr + (size_t)blockIdx.x * 2 * (size_t)pl.L;
    plus_entry(g, F, (size_t)nent, e0 + blockIdx.x, pl, tw, gp, g0, a, a + pl.L, threadIdx.x);
}

// S = triu(g0) - triu(g0)^H ; out_b[f] = gp[f] + S (all f) ; out0 = g0 + S
__global__ void __launch_bounds__(256) add_S_kernel(cd* gp, const cd* g0, cd* out0, int F, int n) {
    const long long tot = (long long)F * n * n;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += stride) {
        const int j = (int)(e % n), i = (int)((e / n) % n);
        cd S = make_double2(0.0, 0.0);
        if (j > i) S = g0[(size_t)i * n + j];
        else if (j < i) { const cd t = g0[(size_t)j * n + i]; S = make_double2(-t.x, t.y); }
        else { const cd t = g0[(size_t)i * n + i]; S = make_double2(0.0, 2.0 * t.y); }   // t - conj(t)
        gp[e] = cadd(gp[e], S);
        if (e < (long long)n * n) out0[e] = cadd(g0[e], S);
    }
}

// max over all entries of |A - B| / |A|  (max_rel_err, wilson_sf.py:190-194): per-block maxima
__global__ void __launch_bounds__(256) relerr_kernel(const cd* A, const cd* B, long long n, double* partial) {
    __shared__ double red[256];
    double m = 0.0;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) {
        const cd d = csub(A[e], B[e]);
        const double r = sqrt(cabs2(d)) / sqrt(cabs2(A[e]));
        if (r > m || r != r) m = r;
    }
    red[threadIdx.x] = m;
    __syncthreads();
    for (int s = 128; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            const double o = red[threadIdx.x + s];
            if (o > red[threadIdx.x] || o != o) red[threadIdx.x] = o;
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) partial[blockIdx.x] = red[0];
}

// ---- 2-norm condition number of Hermitian matrices by power iteration: lam[b] = largest |eigenvalue|
// of M[b] (run on A and on A^-1).  One workgroup per matrix.  The matrices are Hermitian, so row i of
// A x is read as the conjugated COLUMN i (lanes = adjacent addresses: coalesced); iteration stops once
// the estimate has moved by less than 1e-7 (relative) three times in a row.
__global__ void __launch_bounds__(256) power_kernel(const cd* M, int n, int iters, double* lam) {
    SPY_DYN_SMEM(cd, vec);   // x[n], y[n]
    __shared__ double red[256];
    cd* x = vec;
    cd* y = vec + n;
    const cd* A = M + (size_t)blockIdx.x * n * n;
    const int tid = threadIdx.x;
    for (int i = tid; i < n; i += 256) x[i] = make_double2(1.0 + 0.37 * ((i * 7919) % 13), 0.11 * ((i * 104729) % 7));
    __syncthreads();
    double nrm = 0.0, prev = -1.0;
    int calm = 0;
    for (int it = 0; it < iters; ++it) {
        for (int i = tid; i < n; i += 256) {
            cd s0 = make_double2(0.0, 0.0), s1 = make_double2(0.0, 0.0);
            int j = 0;
            for (; j + 1 < n; j += 2) {          // y_i = sum_j conj(A_ji) x_j
                s0 = cadd(s0, cmulc(x[j], A[(size_t)j * n + i]));
                s1 = cadd(s1, cmulc(x[j + 1], A[(size_t)(j + 1) * n + i]));
            }
            if (j < n) s0 = cadd(s0, cmulc(x[j], A[(size_t)j * n + i]));
            y[i] = cadd(s0, s1);
        }
        __syncthreads();
        double p = 0.0;
        for (int i = tid; i < n; i += 256) p += cabs2(y[i]);
        red[tid] = p;
        __syncthreads();
        for (int s = 128; s > 0; s >>= 1) {
            if (tid < s) red[tid] += red[tid + s];
            __syncthreads();
        }
        nrm = sqrt(red[0]);
        __syncthreads();
        const double inv = nrm > 0.0 ? 1.0 / nrm : 0.0;
        for (int i = tid; i < n; i += 256) x[i] = make_double2(y[i].x * inv, y[i].y * inv);
        __syncthreads();
        calm = (fabs(nrm - prev) <= 1e-7 * nrm) ? calm + 1 : 0;      // workgroup-uniform
        prev = nrm;
        if (calm >= 3) break;
    }
    if (tid == 0) lam[blockIdx.x] = nrm;     // |A x| with |x| = 1 -> largest singular value
}

// ---- Granger-Geweke causality (granger.py:53-77); H: (F,n,n), Sigma: (n,n) complex128
__global__ void __launch_bounds__(256) granger_kernel(const cd* CSD, const cd* H, const cd* Sigma, int F, int n, float* out) {
    const long long tot = (long long)F * n * n;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += stride) {
        const int c = (int)(e % n), r = (int)((e / n) % n);
        const long long f = e / ((long long)n * n);
        const double Smat = sqrt(cabs2(CSD[((size_t)f * n + c) * n + c]));          // |S_cc|
        const double Hmat = cabs2(H[((size_t)f * n + c) * n + r]);                   // |H[f,c,r]|^2
        const double SigJI = sqrt(cabs2(Sigma[(size_t)c * n + r]));                  // |Sigma^T|[r,c]
        const double SigII_rc = sqrt(cabs2(Sigma[(size_t)c * n + c]));               // SigmaII[r,c] = |Sigma_cc|
        const double SigII_T = sqrt(cabs2(Sigma[(size_t)r * n + r]));                // SigmaII^T[r,c] = |Sigma_rr|
        const double denom = Smat - (SigII_T - SigJI * SigJI / SigII_rc) * Hmat;
        out[e] = (float)log(Smat / denom);
    }
}

// out[f] = src (n x n) for every f  (np.tile(psi0, (nFreq,1,1)), wilson_sf.py:69)
__global__ void __launch_bounds__(256) tile_kernel(const cd* src, cd* out, int F, int n) {
    const long long tot = (long long)F * n * n, nn = (long long)n * n;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < tot; e += stride) out[e] = src[e % nn];
}

// out = in^T (plain transpose, n x n)
__global__ void __launch_bounds__(256) transpose_kernel(const cd* in, cd* out, int n) {
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e < n * n) out[(size_t)(e % n) * n + e / n] = in[e];
}

}  // namespace spywil
