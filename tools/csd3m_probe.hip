// Timing probe of csd3m_kernel (development aid): build variants with -DNW=<4|8> (waves per workgroup).
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -DNW=8 -o tools/bin/csd3m_probe8 tools/csd3m_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define SPY_DYN_SMEM(type, name) extern __shared__ __attribute__((aligned(16))) char name##_raw[]; \
    type* name = reinterpret_cast<type*>(name##_raw)
#include "../include/spyhip.h"
#include "../syncopy_amd/csrc/csd_kernel.h"
#ifndef NW
#define NW 8
#endif
#include "../syncopy_amd/csrc/csd3m_kernel.h"
__global__ void fill(float* p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        unsigned h = (unsigned)i * 2654435761u;
        h ^= h >> 15; h *= 2246822519u; h ^= h >> 13;
        p[i] = (float)(int)(h & 0xffff) * (1.f / 32768.f) - 1.f;
    }
}
int main(int argc, char** argv) {
    const int rows = argc > 1 ? atoi(argv[1]) : 3500, F = argc > 2 ? atoi(argv[2]) : 1024, C = 256;
    spycsd::CsdArgs a{};
    void *spec, *acc;
    hipMalloc(&spec, (size_t)rows * F * C * 8);
    hipMalloc(&acc, (size_t)F * C * C * 8);
    fill<<<4096, 256>>>((float*)spec, (size_t)rows * F * C * 2);
    hipMemset(acc, 0, (size_t)F * C * C * 8);
    a.spec = (const float2*)spec; a.nrows = rows; a.F = F; a.C = C; a.acc = (float2*)acc;
    a.nt = 8; a.ntiles = 36; a.nitems = (long long)F * 36; a.cpad = 256; a.item_base = 0; a.item_end = a.nitems;
    auto kern = spycsd::csd3m_kernel<256, NW>;
    const int grid = NW == 8 ? F : 2 * F, threads = 64 * NW;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, spycsd::M3_LDS_BYTES);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kern<<<grid, threads, spycsd::M3_LDS_BYTES>>>(a);
    hipEventRecord(e0);
    for (int i = 0; i < 3; ++i) kern<<<grid, threads, spycsd::M3_LDS_BYTES>>>(a);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    double fl = 8.0 * rows * F * C * (C + 1) / 2;
    const double groups = (double)((rows + 15) / 16) * 4, rounds = (grid + 255) / 256;
    printf("NW=%d rows=%d F=%d: %.3f ms, %.1f TF algorithmic, %.0f cycles per group of 4 rows and wave pair at 2.2 GHz (%d = matrix pipe full) %s\n",
           NW, rows, F, ms, fl / ms / 1e9, ms * 1e-3 * 2.2e9 / (groups * rounds), NW == 8 ? 3264 : 1632, hipGetErrorString(hipGetLastError()));
    return 0;
}
