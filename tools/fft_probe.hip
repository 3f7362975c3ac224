// Timing / ablation probe of mtmfft_quad_kernel (development aid).  Build variants with
//   -DPG=<pairs per workgroup> -DPOUTK=<0|2> -DPMEAN=<0|1> -DSPYFFT_ABL=<bits> -DSPYFFT_KATTR='...'
// ABL bits: 1 no taper loads, 2 no twiddle loads, 4 no stores.  Results of ablated builds are wrong by design.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#define SPY_DYN_SMEM(type, name) extern __shared__ __attribute__((aligned(16))) char name##_raw[]; \
    type* name = reinterpret_cast<type*>(name##_raw)
#include "../include/spyhip.h"
#include "../syncopy_amd/csrc/mtmfft2_kernel.h"
#ifndef PG
#define PG 1
#endif
#ifndef POUTK
#define POUTK 2
#endif
#ifndef PMEAN
#define PMEAN 0
#endif
#ifndef PLOG2N
#define PLOG2N 12
#endif
using namespace spyfft;
int main(int argc, char** argv) {
    const int B = argc > 1 ? atoi(argv[1]) : 125, C = 256, K = 7;
    constexpr int N = 1 << PLOG2N, F = N / 2 + 1, G = PG;
    using Cf = Cfg2<PLOG2N, G>;
    float *data, *tap; float2* tw; void* out; long long *st, *lo, *hi;
    hipMalloc(&data, (size_t)B * N * C * 4); hipMemset(data, 0x3c, (size_t)B * N * C * 4);
    hipMalloc(&tap, (size_t)K * N * 4); hipMemset(tap, 0x3c, (size_t)K * N * 4);
    std::vector<float2> htw(N);
    for (int m = 0; m < N; ++m) htw[m] = make_float2((float)cos(-2 * M_PI * m / N), (float)sin(-2 * M_PI * m / N));
    hipMalloc(&tw, N * 8); hipMemcpy(tw, htw.data(), N * 8, hipMemcpyHostToDevice);
    const size_t osz = (size_t)B * (PMEAN ? 1 : K) * F * C * (POUTK == 2 ? 8 : 4);
    hipMalloc(&out, osz);
    std::vector<long long> hs(B), hl(B), hh(B);
    for (int b = 0; b < B; ++b) { hs[b] = hl[b] = (long long)b * N; hh[b] = hs[b] + N; }
    hipMalloc(&st, B * 8); hipMalloc(&lo, B * 8); hipMalloc(&hi, B * 8);
    hipMemcpy(st, hs.data(), B * 8, hipMemcpyHostToDevice); hipMemcpy(lo, hl.data(), B * 8, hipMemcpyHostToDevice);
    hipMemcpy(hi, hh.data(), B * 8, hipMemcpyHostToDevice);
    MtmArgs a{};
    a.data = data; a.ld = C; a.chan_idx = nullptr; a.seg_start = st; a.seg_lo = lo; a.seg_hi = hi; a.nseg = B; a.nsig = N;
    a.nchan = C; a.ntaper = K; a.tapers = tap; a.tw = tw; a.scale = 0.001f; a.detrend = 0; a.demean_taper = 0; a.fpos = nullptr;
    a.nfsel = F; a.out_kind = 0; a.out = out;
    const int nquad = C / 4;
    a.npg = (nquad + G - 1) / G;
    int S = 8 / G; if (S < 1) S = 1; if (S > a.npg) S = a.npg;
    a.S = S; a.ncl = (a.npg + S - 1) / S;
    const long long nclusters = (long long)B * a.ncl;
    const unsigned grid = (unsigned)(((nclusters + 7) / 8) * S * 8);
    auto kern = mtmfft_quad_kernel<PLOG2N, G, POUTK, (bool)PMEAN>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cf::LDS_BYTES);
    hipFuncAttributes fa; hipFuncGetAttributes(&fa, (const void*)kern);
    int occ = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, Cf::NTHREADS, Cf::LDS_BYTES);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kern<<<grid, Cf::NTHREADS, Cf::LDS_BYTES>>>(a);
    hipEventRecord(e0);
    for (int i = 0; i < 3; ++i) kern<<<grid, Cf::NTHREADS, Cf::LDS_BYTES>>>(a);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    hipError_t err = hipGetLastError();
    printf("N=%d G=%d outk=%d mean=%d abl=%d regs=%d blocks/CU=%d : %.3f ms / %d trials = %.2f us/trial (%s)\n", N, G, POUTK, PMEAN,
           SPYFFT_ABL, fa.numRegs, occ, ms, B, 1e3 * ms / B, hipGetErrorString(err));
    return 0;
}
