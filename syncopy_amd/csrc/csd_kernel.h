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

#ifndef SPY_HOST_EMU
typedef float f32x16 __attribute__((ext_vector_type(16)));
#endif

namespace spycsd {

constexpr int CSD_THREADS = 256;

struct CsdArgs {
    const float2* spec;   // (nrows, F, C) complex64
    long long nrows;
    int F, C;
    float2* acc;          // (F, C, C) complex64
    int nt;               // channel tiles = ceil(C/32)
    int ntiles;           // nt*(nt+1)/2
    long long nitems;     // F*ntiles
    int cpad;             // nt*32
    int kb;               // rows per LDS chunk (even)
};

__device__ __forceinline__ void tile_of(int tt, int& ti, int& tj) {
    // inverse of tt = ti*(ti+1)/2 + tj, tj <= ti
    int i = (int)((sqrtf(8.0f * (float)tt + 1.0f) - 1.0f) * 0.5f);
    while ((i + 1) * (i + 2) / 2 <= tt) ++i;
    while (i * (i + 1) / 2 > tt) --i;
    ti = i;
    tj = tt - i * (i + 1) / 2;
}

template <int TPW>
__global__ void __launch_bounds__(CSD_THREADS) csd_accum_kernel(CsdArgs a) {
    SPY_DYN_SMEM(float2, X);   // [kb][rowlen]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    const int l31 = lane & 31, lhi = lane >> 5;

    const long long item0 = (long long)blockIdx.x * (4 * TPW);
    long long last = item0 + 4 * TPW;
    if (last > a.nitems) last = a.nitems;
    if (item0 >= a.nitems) return;
    const int f_lo = (int)(item0 / a.ntiles);
    const int nfb = (int)((last - 1) / a.ntiles) - f_lo + 1;
    const int rowlen = nfb * a.cpad;

    int aoff[TPW], boff[TPW], tf[TPW], tti[TPW], ttj[TPW];
    bool valid[TPW];
    f32x16 accr[TPW], acci[TPW];
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        const long long item = item0 + (long long)wave * TPW + t;
        valid[t] = item < a.nitems;
        int f = f_lo, ti = 0, tj = 0;
        if (valid[t]) {
            f = (int)(item / a.ntiles);
            tile_of((int)(item % a.ntiles), ti, tj);
        }
        tf[t] = f;
        tti[t] = ti;
        ttj[t] = tj;
        aoff[t] = (f - f_lo) * a.cpad + ti * 32 + l31;
        boff[t] = (f - f_lo) * a.cpad + tj * 32 + l31;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            accr[t][r] = 0.f;
            acci[t][r] = 0.f;
        }
    }

    const int total = a.kb * rowlen;
    for (long long r0 = 0; r0 < a.nrows; r0 += a.kb) {
        // ---- stage kb rows of the nfb frequencies (zero fill: padding channels, rows past the end)
        for (int u = tid; u < total; u += CSD_THREADS) {
            const int kr = u / rowlen, cc = u - kr * rowlen;
            const int fb = cc / a.cpad, c = cc - fb * a.cpad;
            const long long row = r0 + kr;
            float2 val = make_float2(0.f, 0.f);
            if (row < a.nrows && c < a.C) val = a.spec[((size_t)row * a.F + (f_lo + fb)) * a.C + c];
            X[u] = val;
        }
        __syncthreads();
        // ---- rank-2 updates
        for (int ks = 0; ks < a.kb; ks += 2) {
            const float2* xr = X + (ks + lhi) * rowlen;
#pragma unroll
            for (int t = 0; t < TPW; ++t) {
                const float2 av = xr[aoff[t]];
                const float2 bv = xr[boff[t]];
                accr[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.x, bv.x, accr[t], 0, 0, 0);
                accr[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.y, accr[t], 0, 0, 0);
                acci[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(av.y, bv.x, acci[t], 0, 0, 0);
                acci[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(-av.x, bv.y, acci[t], 0, 0, 0);
            }
        }
        __syncthreads();
    }

    // ---- acc += tile (each (f, tile) is owned by exactly one wave: plain read-modify-write)
#pragma unroll
    for (int t = 0; t < TPW; ++t) {
        if (!valid[t]) continue;
        const int j = ttj[t] * 32 + l31;
        if (j >= a.C) continue;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int i = tti[t] * 32 + (r & 3) + 8 * (r >> 2) + 4 * lhi;
            if (i < a.C) {
                float2* p = a.acc + ((size_t)tf[t] * a.C + i) * a.C + j;
                float2 v = *p;
                v.x += accr[t][r];
                v.y += acci[t][r];
                *p = v;
            }
        }
    }
}

// acc[f,i,j] *= scale (i >= j), zero the diagonal's imaginary part, mirror to the upper triangle
__global__ void __launch_bounds__(256) csd_finalize_kernel(float2* acc, int F, int C, float scale) {
    const long long n = (long long)F * C * C;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long e = (long long)blockIdx.x * blockDim.x + threadIdx.x; e < n; e += stride) {
        const int j = (int)(e % C);
        const long long fi = e / C;
        const int i = (int)(fi % C);
        if (i < j) continue;
        float2 v = acc[e];
        v.x *= scale;
        v.y = (i == j) ? 0.f : v.y * scale;
        acc[e] = v;
        if (i != j) acc[(fi - i + j) * C + i] = make_float2(v.x, -v.y);
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

}  // namespace spycsd
