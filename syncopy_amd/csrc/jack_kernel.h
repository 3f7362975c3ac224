// K9: streaming leave-one-out (jackknife) replicates of the coherence.
//
// Reference (connectivity_analysis.py:601-606,736-757; statistics/jackknifing.py:14-184): keep all T single-trial
// cross spectra, form T leave-one-out averages (T*S - S_t)/(T-1), run normalize_csd on each, then bias and variance
// from the replicates.  Here one pass over the tapered spectra of a batch of trials does all of it per (frequency,
// channel pair): S_t from the trial's K tapers, the replicate, its coherence, d_t = replicate - direct, and the
// float64 sums of d_t and |d_t|^2 stay in registers; one read-modify-write of the two sum arrays per launch.
#pragma once
#include "spy_intrinsics.h"
#include "../../include/spyhip.h"
#include "csd_kernel.h"

namespace spycsd {

struct JackArgs {
    const float2* spec;   // (ntrials * K, F, C) complex64 tapered spectra, the K rows of a trial adjacent
    const float2* S;      // (F, C, C) complex64: trial- and taper-averaged cross spectra, full Hermitian array
    const void* direct;   // (F, C, C) float32 (real kinds) or complex64: coherence of S in the requested output kind
    int ntrials, K, F, C, kind;
    float T;              // total number of trials of the estimate
    double* sum_d;        // (F, C, C) float64, or (F, C, C, 2) for the complex kind: += sum_t d_t
    double* sum_d2;       // (F, C, C) float64: += sum_t |d_t|^2
};

__device__ __forceinline__ float fast_rsqrt(float x) { return spy_rsqrt(x); }
__device__ __forceinline__ float fast_sqrt(float x) { return spy_sqrt(x); }

// Workgroup = (frequency, 32 x 32 tile of the lower triangle of the channel square), 256 threads: thread (ti, tq) owns the pairs
// (i = ti, j = 4 tq .. 4 tq + 3).  Staging as in K7 (ntaper x 64 spectra per trial through LDS, double-buffered).
template <bool CPLX>
__global__ void __launch_bounds__(256) jack_coh_kernel(JackArgs a) {
    SPY_DYN_SMEM(float2, jk_lds);
    const int tid = threadIdx.x, ti = tid & 31, tq = tid >> 5;
    const int nt = (a.C + 31) / 32, ntl = nt * (nt + 1) / 2;
    // all tiles of a frequency on ONE XCD, one after the other (as K7)
    const int fchunk = (a.F + 7) >> 3;
    const unsigned yid = blockIdx.x >> 3;
    const int f = (int)(blockIdx.x & 7u) * fchunk + (int)(yid / ntl);
    if ((int)(yid / ntl) >= fchunk || f >= a.F) return;
    int bi, bj;
    tile_of((int)(yid % ntl), bi, bj);                  // bi >= bj: the upper triangle is mirrored on the way out
    const int K = a.K, per = 2 * K * 32;
    // (buffers addressed as jk_lds + n * per: a pointer array would decay to flat addressing)

    // staging offsets inside a trial are the same for every trial: computed once (as in K7)
    constexpr int NST = 4;
    unsigned soff[NST];
    bool sok[NST];
#pragma unroll
    for (int n = 0; n < NST; ++n) {
        const int e = tid + 256 * n;
        const int side = e / (K * 32), k = (e - side * K * 32) >> 5, c = e & 31;
        const int ch = (side ? bj : bi) * 32 + c;
        sok[n] = e < per && ch < a.C;
        soff[n] = sok[n] ? (unsigned)(((size_t)k * a.F + f) * a.C + ch) : 0u;
    }
    const size_t tstride = (size_t)K * a.F * a.C;
    const bool small = per <= 256 * NST && tstride < (1ull << 31);
    auto stage = [&](int t, float2* dst) {
        if (small) {
            const float2* base = a.spec + (size_t)t * tstride;
#pragma unroll
            for (int n = 0; n < NST; ++n) {
                const int e = tid + 256 * n;
                if (256 * n < per && e < per) dst[e] = sok[n] ? base[soff[n]] : make_float2(0.f, 0.f);
            }
            return;
        }
        for (int e = tid; e < per; e += 256) {
            const int side = e / (K * 32), k = (e - side * K * 32) >> 5, c = e & 31;
            const int ch = (side ? bj : bi) * 32 + c;
            float2 v = make_float2(0.f, 0.f);
            if (ch < a.C) v = a.spec[((size_t)((size_t)t * K + k) * a.F + f) * a.C + ch];
            dst[e] = v;
        }
    };

    const int i = bi * 32 + ti, ic = i < a.C ? i : a.C - 1;
    const size_t fb = (size_t)f * a.C * a.C;
    const float Sii = a.S[fb + (size_t)ic * a.C + ic].x;
    float2 Sij[4], dir[4];
    float Sjj[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int j = bj * 32 + tq * 4 + q, jc = j < a.C ? j : a.C - 1;
        Sij[q] = a.S[fb + (size_t)ic * a.C + jc];
        Sjj[q] = a.S[fb + (size_t)jc * a.C + jc].x;
        if (CPLX) dir[q] = reinterpret_cast<const float2*>(a.direct)[fb + (size_t)ic * a.C + jc];
        else dir[q] = make_float2(reinterpret_cast<const float*>(a.direct)[fb + (size_t)ic * a.C + jc], 0.f);
    }
    const float invK = 1.0f / (float)K, T = a.T, invT1 = 1.0f / (a.T - 1.0f);
    double sd[4], sdi[4], sd2[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) sd[q] = sdi[q] = sd2[q] = 0.0;

    stage(0, jk_lds);
    __syncthreads();
    for (int t = 0; t < a.ntrials; ++t) {
        const float2* b = jk_lds + (t & 1) * per;
        if (t + 1 < a.ntrials) stage(t + 1, jk_lds + ((t + 1) & 1) * per);
        // xi * conj(xj) = xi * xj.re + (xi.im, -xi.re) * xj.im: two packed FMAs per pair and taper; the powers of the
        // four column channels as two packed FMA chains over (re^2 + im^2)
        typedef float pk2 __attribute__((ext_vector_type(2)));
        pk2 s[4], pj01 = pk2{0.f, 0.f}, pj23 = pk2{0.f, 0.f};
        float pi = 0.f;
#pragma unroll
        for (int q = 0; q < 4; ++q) s[q] = pk2{0.f, 0.f};
        for (int k = 0; k < K; ++k) {
            const float2 xi2 = b[k * 32 + ti];
            const float4* p4 = reinterpret_cast<const float4*>(b + (K + k) * 32 + tq * 4);
            const float4 j01 = p4[0], j23 = p4[1];
            const pk2 xi = pk2{xi2.x, xi2.y}, xs = pk2{xi2.y, -xi2.x};
            pi += xi2.x * xi2.x + xi2.y * xi2.y;
            s[0] = xi * pk2{j01.x, j01.x} + s[0];
            s[1] = xi * pk2{j01.z, j01.z} + s[1];
            s[2] = xi * pk2{j23.x, j23.x} + s[2];
            s[3] = xi * pk2{j23.z, j23.z} + s[3];
            s[0] = xs * pk2{j01.y, j01.y} + s[0];
            s[1] = xs * pk2{j01.w, j01.w} + s[1];
            s[2] = xs * pk2{j23.y, j23.y} + s[2];
            s[3] = xs * pk2{j23.w, j23.w} + s[3];
            pj01 = pk2{j01.x, j01.z} * pk2{j01.x, j01.z} + pj01;
            pj01 = pk2{j01.y, j01.w} * pk2{j01.y, j01.w} + pj01;
            pj23 = pk2{j23.x, j23.z} * pk2{j23.x, j23.z} + pj23;
            pj23 = pk2{j23.y, j23.w} * pk2{j23.y, j23.w} + pj23;
        }
        const float pj[4] = {pj01.x, pj01.y, pj23.x, pj23.y};
        // leave-one-out average in complex64 as the reference forms it: (T * S - S_t) / (T - 1)
        const float lii = (T * Sii - pi * invK) * invT1;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float lx = (T * Sij[q].x - s[q].x * invK) * invT1;
            const float ly = (T * Sij[q].y - s[q].y * invK) * invT1;
            const float ljj = (T * Sjj[q] - pj[q] * invK) * invT1;
            const float rden = fast_rsqrt(lii * ljj);                // 1 ulp: d_t itself is good to ~1e-7 |c| only
            const float2 c = make_float2(lx * rden, ly * rden);
            if (CPLX) {
                const float dx = c.x - dir[q].x, dy = c.y - dir[q].y;
                sd[q] += (double)dx;
                sdi[q] += (double)dy;
                sd2[q] += (double)dx * dx + (double)dy * dy;
            } else {
                const float d = (a.kind == SPYHIP_OUT_ABS ? fast_sqrt(c.x * c.x + c.y * c.y) : coh_convert(c, a.kind)) - dir[q].x;
                sd[q] += (double)d;
                sd2[q] += (double)d * d;
            }
        }
        __syncthreads();
    }
    if (i >= a.C) return;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int j = bj * 32 + tq * 4 + q;
        if (j >= a.C) continue;
        const size_t o = fb + (size_t)i * a.C + j;
        if (CPLX) {
            a.sum_d[2 * o] += sd[q];
            a.sum_d[2 * o + 1] += sdi[q];
        } else {
            a.sum_d[o] += sd[q];
        }
        a.sum_d2[o] += sd2[q];
        if (bi == bj) continue;                          // diagonal tiles hold both triangles themselves
        // mirror (j, i): the coherency is Hermitian - imaginary part and phase change sign, the rest is symmetric
        const size_t m = fb + (size_t)j * a.C + i;
        if (CPLX) {
            a.sum_d[2 * m] += sd[q];
            a.sum_d[2 * m + 1] -= sdi[q];
        } else {
            a.sum_d[m] += (a.kind == SPYHIP_OUT_IMAG || a.kind == SPYHIP_OUT_ANGLE) ? -sd[q] : sd[q];
        }
        a.sum_d2[m] += sd2[q];
    }
}

}  // namespace spycsd
