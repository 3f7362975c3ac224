// Phase timeline of mtmfft_quad_kernel (development aid): s_memtime stamps from lane 0 of every wave of a few
// workgroups in the middle of the grid; prints the mean cycles between consecutive stamps.
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 [-DPG=1 -DPOUTK=0 -DPMEAN=1] tools/fft_stamp_probe.hip -o stamp
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <cmath>
#define SPY_DYN_SMEM(type, name) extern __shared__ __attribute__((aligned(16))) char name##_raw[]; \
    type* name = reinterpret_cast<type*>(name##_raw)
#define SPYFFT_STAMPS 1
#define SPYFFT_STAMP_B0 16000
#define SPYFFT_STAMP_NB 8
#include "../include/spyhip.h"
#include "../syncopy_amd/csrc/mtmfft2_kernel.h"
#ifndef PG
#define PG 1
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
    const int B = argc > 1 ? atoi(argv[1]) : 500, C = 256, K = 7;
    constexpr int LOG2N = 12, N = 1 << LOG2N, F = N / 2 + 1, G = PG;
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
    using Cf = Cfg2<LOG2N, G>;
    const int nitem = C / 4;
    a.npg = (nitem + G - 1) / G; int S = 8 / G; if (S < 1) S = 1; if (S > a.npg) S = a.npg;
    auto kern = mtmfft_quad_kernel<LOG2N, G, POUTK, (bool)PMEAN>;
    a.S = S; a.ncl = (a.npg + S - 1) / S;
    const long long nclusters = (long long)B * a.ncl;
    const unsigned grid = (unsigned)(((nclusters + 7) / 8) * S * 8);
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)Cf::LDS_BYTES);
    const size_t nst = (size_t)SPYFFT_STAMP_NB * 16 * 1024;
    unsigned long long* sb; hipMalloc(&sb, nst * 8); hipMemset(sb, 0, nst * 8);
    hipMemcpyToSymbol(HIP_SYMBOL(spy_stamp_buf), &sb, sizeof(sb));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kern<<<grid, Cf::NTHREADS, Cf::LDS_BYTES>>>(a);
    hipEventRecord(e0);
    kern<<<grid, Cf::NTHREADS, Cf::LDS_BYTES>>>(a);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("G=%d outk=%d mean=%d grid=%u : %.3f ms / %d trials = %.2f us/trial (%s) [stamped build]\n", G, POUTK, PMEAN, grid, ms, B,
           1e3 * ms / B, hipGetErrorString(hipGetLastError()));
    std::vector<unsigned long long> h(nst);
    hipMemcpy(h.data(), sb, nst * 8, hipMemcpyDeviceToHost);
    const int nw = Cf::NTHREADS / 64;
    int ns = 0;
    while (ns < 1023 && h[ns + 1] != 0) ++ns;
    printf("stamps per wave: %d; mean cycles between stamps over %d workgroups x %d waves (min..max)\n", ns + 1, SPYFFT_STAMP_NB, nw);
    double tot = 0;
    for (int s = 1; s <= ns; ++s) {
        double sum = 0, mn = 1e30, mx = 0; int n = 0;
        for (int b = 0; b < SPYFFT_STAMP_NB; ++b)
            for (int w = 0; w < nw; ++w) {
                const unsigned long long* p = h.data() + ((size_t)(b * 16 + w) << 10);
                if (!p[s] || !p[s - 1]) continue;
                const double d = (double)(p[s] - p[s - 1]);
                sum += d; ++n; if (d < mn) mn = d; if (d > mx) mx = d;
            }
        if (n) { printf("  %3d: %8.0f  (%6.0f .. %6.0f)\n", s, sum / n, mn, mx); tot += sum / n; }
    }
    printf("total %.0f cycles per workgroup\n", tot);
    return 0;
}
