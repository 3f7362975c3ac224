// Ablation probe of csd_accum_kernel<5,4> (development aid): build with -DCSD_DBG_* variants.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define SPY_DYN_SMEM(type, name) extern __shared__ __attribute__((aligned(16))) char name##_raw[]; \
    type* name = reinterpret_cast<type*>(name##_raw)
#include "../include/spyhip.h"
#include "../syncopy_amd/csrc/csd_kernel.h"
int main(int argc, char** argv) {
    const int rows = argc > 1 ? atoi(argv[1]) : 875, F = argc > 2 ? atoi(argv[2]) : 2049, C = 256;
    spycsd::CsdArgs a{};
    void *spec, *acc;
    hipMalloc(&spec, (size_t)rows * F * C * 8);
    hipMalloc(&acc, (size_t)F * C * C * 8);
    hipMemset(spec, 0x3c, (size_t)rows * F * C * 8);
    hipMemset(acc, 0, (size_t)F * C * C * 8);
    a.spec = (const float2*)spec; a.nrows = rows; a.F = F; a.C = C; a.acc = (float2*)acc;
    a.nt = 8; a.ntiles = 36; a.nitems = (long long)F * 36; a.cpad = 256; a.kb = 16; a.item_base = 0; a.item_end = a.nitems;
    const size_t lds = 3 * 16 * 256 * 8;
    auto kern = spycsd::csd_accum_kernel<5, 4>;
    hipFuncSetAttribute((const void*)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    kern<<<F, 512, lds>>>(a);
    hipEventRecord(e0);
    for (int i = 0; i < 3; ++i) kern<<<F, 512, lds>>>(a);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 3;
    double fl = 8.0 * rows * F * C * (C + 1) / 2;
    printf("rows=%d F=%d: %.3f ms, %.1f TF algorithmic, %.1f TF issued (36 tiles)\n", rows, F, ms, fl / ms / 1e9,
           8.0 * rows * F * 36.0 * 1024 / ms / 1e9);
    return 0;
}
