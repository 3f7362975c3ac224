// declong_sub_kernel instances (mtmfft_declong.h): sub-transform lengths 2000, 4096, 5000
#include "spy_common.h"
#include "mtmfft_declong.h"

namespace spyfft {

template <class C>
static int declong_sub(hipStream_t stream, const LongArgs& a, int P, long long nblocks) {
    if (nblocks > 0x7fffffffLL) { spy::set_error("fft_exec: grid too large (%lld blocks)", nblocks); return -1; }
    auto kern = declong_sub_kernel<C>;
    SPY_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)C::LDS_BYTES));
    hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(C::NTHREADS), C::LDS_BYTES, stream, a, P);
    SPY_HIP_CHECK(hipGetLastError());
    return 0;
}

// quads per workgroup of the schedule that serves sub-transform length M (0: none)
int declong_group(int M) {
    switch (M) {
        case 2000: case 4096: case 5000: case 4000: case 8000: case 10000: return 1;
        default: return 0;
    }
}

int declong_launch_sub_a(hipStream_t stream, const LongArgs& a, int M, int P, long long nblocks) {
    switch (M) {
        case 2000: return declong_sub<CfgD<10, 10, 10, 2, 1>>(stream, a, P, nblocks);
        case 4096: return declong_sub<CfgD<16, 16, 16, 1, 1>>(stream, a, P, nblocks);
        case 5000: return declong_sub<CfgD<10, 10, 10, 5, 1>>(stream, a, P, nblocks);
        default: return -100;
    }
}

}  // namespace spyfft
