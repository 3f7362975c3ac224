// declong64_sub_kernel instances (mtmfft_declong64.h): sub-transform lengths 2000, 4096, 5000
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

// channel pairs per workgroup of the schedule that serves sub-transform length M (0: none)
int declong64_group(int M) {
    switch (M) {
        case 2000: return D64_2000::G;
        case 4096: return D64_4096::G;
        case 5000: return D64_5000::G;
        case 4000: return D64_4000::G;
        case 8000: return D64_8000::G;
        case 10000: return D64_10000::G;
        default: return 0;
    }
}

int declong64_launch_sub_a(hipStream_t stream, const Long64Args& a, int M, int P, long long nblocks) {
    switch (M) {
        case 2000: return declong64_sub<D64_2000>(stream, a, P, nblocks);
        case 4096: return declong64_sub<D64_4096>(stream, a, P, nblocks);
        case 5000: return declong64_sub<D64_5000>(stream, a, P, nblocks);
        default: return -100;
    }
}

}  // namespace spyfft
