// Argument block and packed load / store helpers of the transforms that go through HBM (mtmfft_long.h, mtmfft_declong.h)
#pragma once
#include "fft2_device.h"
#include "mtmfft_kernel.h"

namespace spyfft {

struct LongArgs {
    MtmArgs m;                  // trial matrix, segments, tapers, output description (tw/chirp/bhat unused here)
    int M1, M2;                 // M = M1*M2
    const float2* tw1;          // exp(-2 pi i m / M1)
    const float2* tw2;          // exp(-2 pi i m / M2)
    const float2* twM;          // exp(-2 pi i m / M), M entries
    const float2* chirp;        // nfft entries
    const float2* bhat;         // M entries in [k1][k2] order, 1/M folded in
    float4* scratch;            // [item][M] packed elements, item = (segment of the chunk, quad, taper)
    const double* stats;        // [seg][chan][2 + ntaper]: sum x, sum (n-mid) x, sum w_k x   (valid samples)
    const double* wsum;         // [ntaper][2]: sum w_k, sum w_k (n - mid)
    int seg0, nsegc;            // segments [seg0, seg0 + nsegc) are in flight
    int nquad;
    int direct;                 // nfft == M (power of two): no chirp, one forward four-step transform, the
                                // spectrum stays in [k1][k2] order (bin f at (f % M1) * M2 + f / M1)
};

__device__ __forceinline__ C2 ld_c2(const float4* p) {
    const float4 t = *p;
    return C2{v2f{t.x, t.y}, v2f{t.z, t.w}};
}
__device__ __forceinline__ void st_c2(float4* p, C2 v) { *p = make_float4(v.r[0], v.r[1], v.i[0], v.i[1]); }
__device__ __forceinline__ C2 conj2(C2 a) { return C2{a.r, -a.i}; }

}  // namespace spyfft
