// declong64_sub_kernel instances (mtmfft_declong64.h): sub-transform lengths 4000, 8000, 10000
#include "spy_common.h"
#include "mtmfft_dec64_cfg.h"
#include "mtmfft_declong64.h"

namespace spyfft {

template <class C>
static int declong64_sub(hipStream_t stream, const Long64Args& a, int P, long long nblocks) {
    if (nblocks > 0x7fffffffLL) { spy::set_error("fft_exec: grid too large (%lld blocks)", nblocks); return -1; }
    auto kern = declong64_sub_kernel<C>;
    SPY_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)C::LDS_BYTES));
    hipLaunchKernelGGL(kern, dim3((unsigned)nblocks), dim3(C::NTHREADS), C::LDS_BYTES, stream, a, P);
    SPY_HIP_CHECK(hipGetLastError());
    return 0;
}

int declong64_launch_sub_b(hipStream_t stream, const Long64Args& a, int M, int P, long long nblocks) {
    switch (M) {
        case 4000: return declong64_sub<D64_4000>(stream, a, P, nblocks);
        case 8000: return declong64_sub<D64_8000>(stream, a, P, nblocks);
        case 10000: return declong64_sub<D64_10000>(stream, a, P, nblocks);
        default: return -100;
    }
}

}  // namespace spyfft
