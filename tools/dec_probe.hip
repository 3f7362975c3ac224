// Timing probe of mtmfft_dec_kernel (development aid): -DDV -DDR1 -DDR2 -DDR3 -DDG choose the schedule, -DPOUTK / -DPMEAN
// the output mode; prints registers, occupancy and us per trial of 256 channels x 7 tapers.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#include "../syncopy_amd/csrc/spy_intrinsics.h"
#include "../include/spyhip.h"
#include "../syncopy_amd/csrc/mtmfft_dec_kernel.h"
#ifndef DV
#define DV 10
#endif
#ifndef DR1
#define DR1 10
#endif
#ifndef DR2
#define DR2 10
#endif
#ifndef DR3
#define DR3 2
#endif
#ifndef DG
#define DG 1
#endif
#ifndef POUTK
#define POUTK 0
#endif
#ifndef PMEAN
#define PMEAN 1
#endif
using namespace spyfft;
__global__ void fillr(float* p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u; h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = (float)(int)(h & 0xffff) * (1.f / 32768.f) - 1.f;
    }
}
int main(int argc, char** argv) {
    using Cf = CfgD<DV, DR1, DR2, DR3, DG>;
    const int B = argc > 1 ? atoi(argv[1]) : 500, C = 256, K = 7;
    constexpr int N = Cf::N, F = N / 2 + 1, G = DG;
    float *data, *tap; float2* tw; void* out; long long *st, *hi;
    hipMalloc(&data, (size_t)B * N * C * 4); fillr<<<2048, 256>>>(data, (size_t)B * N * C);
    hipMalloc(&tap, (size_t)K * N * 4); fillr<<<64, 256>>>(tap, (size_t)K * N);
    std::vector<float2> htw(N);
    for (int m = 0; m < N; ++m) htw[m] = make_float2((float)cos(-2 * M_PI * m / N), (float)sin(-2 * M_PI * m / N));
    hipMalloc(&tw, N * 8); hipMemcpy(tw, htw.data(), N * 8, hipMemcpyHostToDevice);
    const size_t osz = (size_t)B * (PMEAN ? 1 : K) * F * C * (POUTK == 2 ? 8 : 4);
    hipMalloc(&out, osz);
    std::vector<long long> hs(B), hh(B);
    for (int b = 0; b < B; ++b) { hs[b] = (long long)b * N; hh[b] = hs[b] + N; }
    hipMalloc(&st, B * 8); hipMalloc(&hi, B * 8);
    hipMemcpy(st, hs.data(), B * 8, hipMemcpyHostToDevice); hipMemcpy(hi, hh.data(), B * 8, hipMemcpyHostToDevice);
    MtmArgs a{};
    a.data = data; a.ld = C; a.seg_start = st; a.seg_lo = st; a.seg_hi = hi; a.nseg = B; a.nsig = N;
    a.nchan = C; a.ntaper = K; a.tapers = tap; a.tw = tw; a.scale = 0.001f; a.detrend = -1; a.nfsel = F; a.out_kind = 0; a.out = out;
    const int nitem = C / 4;
    a.npg = (nitem + G - 1) / G; int S = 8 / G; if (S < 1) S = 1; if (S > a.npg) S = a.npg;
    auto kern = mtmfft_dec_kernel<Cf, POUTK, (bool)PMEAN>;
    a.S = S; a.ncl = (a.npg + S - 1) / S;
    const long long nclusters = (long long)B * a.ncl;
    const unsigned grid = (unsigned)(((nclusters + 7) / 8) * S * 8);
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cf::LDS_BYTES);
    hipFuncAttributes fa; hipFuncGetAttributes(&fa, (const void*)kern);
    int occ = 0; hipOccupancyMaxActiveBlocksPerMultiprocessor(&occ, kern, Cf::NTHREADS, Cf::LDS_BYTES);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kern<<<grid, Cf::NTHREADS, Cf::LDS_BYTES>>>(a);
    hipEventRecord(e0);
    for (int i = 0; i < 3; ++i) kern<<<grid, Cf::NTHREADS, Cf::LDS_BYTES>>>(a);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    printf("N=%d = %d x %d x %d x %d, G=%d threads=%d outk=%d mean=%d regs=%d lds=%zu blocks/CU=%d : %.3f ms / %d trials = %.2f us/trial, %.4f ns per channel-sample (%s)\n",
           N, DV, DR1, DR2, DR3, G, Cf::NTHREADS, POUTK, PMEAN, fa.numRegs, (size_t)Cf::LDS_BYTES, occ, ms, B, 1e3 * ms / B,
           1e6 * ms / B / (N * 256.0), hipGetErrorString(hipGetLastError()));
    return 0;
}
