// K7: pairwise phase consistency (Vinck 2010) of single-trial cross spectra.
//
// The reference evaluates all T(T-1)/2 trial pairs (connectivity/ST_compRoutines.py:159-233,
// connectivity_analysis.py:624-663):  ppc = 2/(T(T-1)) * sum_{j<k} cos(arg(S_j conj(S_k))).
// With the unit phasors u_t = S_t/|S_t| the pair sum collapses,
//     sum_{j<k} Re(u_j conj(u_k)) = (|sum_t u_t|^2 - T) / 2,
// so ONE pass over the trials suffices: accumulate U = sum_t u_t per (frequency, channel pair), then
//     ppc = (|U|^2 - T) / (T (T-1)).
// A cross spectrum that is exactly zero has arg 0 in the reference (np.angle(0) = 0): u_t = 1.
#pragma once
#include "spy_intrinsics.h"
#include "../../include/spyhip.h"

namespace spyppc {

struct PpcArgs {
    const float2* spec;   // (ntrials * ntaper, F, C) complex64: tapered spectra, rows of a trial adjacent
    int ntrials, ntaper, F, C;
    float2* acc;          // (F, C, C) complex64: U on the lower triangle (32 x 32 tile granularity)
};

// unit phasor of s (scaled first: |s|^2 of small spectra underflows in fp32)
__device__ __forceinline__ float2 unit_phasor(float2 s) {
    // branch-free: m = 0 -> r = inf, x = y = NaN, selected away at the end
    const float m = fmaxf(fabsf(s.x), fabsf(s.y));
    const float r = __builtin_amdgcn_rcpf(m);
    const float x = s.x * r, y = s.y * r;
    const float inv = __builtin_amdgcn_rsqf(x * x + y * y);
    const bool ok = m > 0.f;
    return make_float2(ok ? x * inv : 1.f, ok ? y * inv : 0.f);
}

// Workgroup = (frequency, 32 x 32 tile of the lower triangle), 256 threads: thread (ti, tq) owns the pairs
// (i = ti, j = 4 tq .. 4 tq + 3).  Per trial the ntaper x 64 spectra of the tile's channels go through LDS
// (double-buffered: the next trial's rows are in flight while this one is evaluated); the taper sum, the
// normalisation and the trial sum stay in registers; one read-modify-write of the accumulator per launch.
__global__ void __launch_bounds__(256) ppc_accum_kernel(PpcArgs a) {
    SPY_DYN_SMEM(float2, ppc_lds);
    const int tid = threadIdx.x, ti = tid & 31, tq = tid >> 5;
    const int nt = (a.C + 31) / 32, ntl = nt * (nt + 1) / 2;
    // all tiles of a frequency run on ONE XCD (ids congruent mod 8), one after the other: the frequency's rows are
    // fetched from HBM into that L2 once instead of into all eight
    const int fchunk = (a.F + 7) >> 3;
    const unsigned yid = blockIdx.x >> 3;
    const int f = (int)(blockIdx.x & 7u) * fchunk + (int)(yid / ntl);
    if ((int)(yid / ntl) >= fchunk || f >= a.F) return;
    int rem = (int)(yid % ntl), bi = 0;
    while (rem >= bi + 1) { rem -= bi + 1; ++bi; }
    const int bj = rem;
    const int K = a.ntaper, per = 2 * K * 32;
    // (buffers addressed as ppc_lds + n * per: a pointer array would decay to flat addressing)

    // staging offsets inside a trial are the same for every trial: computed once for up to 4 elements per thread
    // (16 tapers); the generic walk below serves anything larger
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

    float2 u[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) u[q] = make_float2(0.f, 0.f);
    stage(0, ppc_lds);
    __syncthreads();
    for (int t = 0; t < a.ntrials; ++t) {
        const float2* b = ppc_lds + (t & 1) * per;
        if (t + 1 < a.ntrials) stage(t + 1, ppc_lds + ((t + 1) & 1) * per);
        // xi * conj(xj) = xi * xj.re + (xi.im, -xi.re) * xj.im: two packed FMAs per pair and taper
        typedef float pk2 __attribute__((ext_vector_type(2)));
        pk2 s[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) s[q] = pk2{0.f, 0.f};
        for (int k = 0; k < K; ++k) {
            const float2 xi2 = b[k * 32 + ti];
            const float4* pj = reinterpret_cast<const float4*>(b + (K + k) * 32 + tq * 4);
            const float4 j01 = pj[0], j23 = pj[1];
            const pk2 xi = pk2{xi2.x, xi2.y}, xs = pk2{xi2.y, -xi2.x};
            s[0] = xi * pk2{j01.x, j01.x} + s[0];
            s[1] = xi * pk2{j01.z, j01.z} + s[1];
            s[2] = xi * pk2{j23.x, j23.x} + s[2];
            s[3] = xi * pk2{j23.z, j23.z} + s[3];
            s[0] = xs * pk2{j01.y, j01.y} + s[0];
            s[1] = xs * pk2{j01.w, j01.w} + s[1];
            s[2] = xs * pk2{j23.y, j23.y} + s[2];
            s[3] = xs * pk2{j23.w, j23.w} + s[3];
        }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const float2 p = unit_phasor(make_float2(s[q].x, s[q].y));
            u[q].x += p.x;
            u[q].y += p.y;
        }
        __syncthreads();
    }
    const int i = bi * 32 + ti;
    if (i >= a.C) return;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int j = bj * 32 + tq * 4 + q;
        if (j >= a.C) continue;
        float2* o = a.acc + ((size_t)f * a.C + i) * a.C + j;
        float2 v = *o;
        v.x += u[q].x;
        v.y += u[q].y;
        *o = v;
    }
}

// the same accumulation from single-trial cross spectra that exist already: csd (ntrials, n) -> acc (n) += unit phasors
__global__ void __launch_bounds__(256) ppc_accum_csd_kernel(const float2* csd, long long n, int ntrials, float2* acc) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    if (idx >= n) return;
    float2 u = acc[idx];
    for (int t = 0; t < ntrials; ++t) {
        const float2 p = unit_phasor(csd[(size_t)t * n + idx]);
        u.x += p.x;
        u.y += p.y;
    }
    acc[idx] = u;
}

// ppc[f,i,j] = (|U|^2 - T) / (T (T-1)); lower_only: U[f,i,j] for i < j is read from its mirror U[f,j,i]
// (|conj U| = |U|: the result is symmetric)
__global__ void __launch_bounds__(256) ppc_finalize_kernel(const float2* acc, int F, int ni, int nj, int lower_only,
                                                           double T, float* out) {
    const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
    const long long n = (long long)F * ni * nj;
    if (idx >= n) return;
    const int j = (int)(idx % nj), i = (int)((idx / nj) % ni);
    const long long f = idx / ((long long)ni * nj);
    const bool mirror = lower_only && j > i;                   // exactly symmetric output
    const float2 u = mirror ? acc[(f * ni + j) * nj + i] : acc[idx];
    const double m2 = (double)u.x * u.x + (double)u.y * u.y;
    out[idx] = (float)((m2 - T) / (T * (T - 1.0)));
}

}  // namespace spyppc
