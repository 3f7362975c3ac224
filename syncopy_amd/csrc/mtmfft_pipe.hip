// Launchers of the pipelined tapered-FFT kernel (mtmfft_pipe_kernel.h); its own translation unit so that the
// instances compile next to mtmfft.hip.
#include "spy_common.h"
#include "mtmfft_pipe_kernel.h"

namespace spyfft {

template <int LOG2N, int OUTK, bool MEAN>
static int pipe_launch_one(hipStream_t stream, const MtmArgs& a, unsigned grid) {
    using P = CfgP<LOG2N>;
    auto kern = mtmfft_pipe_kernel<LOG2N, OUTK, MEAN>;
    SPY_HIP_CHECK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize,
                                      (int)P::LDS_BYTES));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(P::NTHREADS), P::LDS_BYTES, stream, a);
    SPY_HIP_CHECK(hipGetLastError());
    return 0;
}

template <int LOG2N>
static int pipe_launch_mode(hipStream_t stream, const MtmArgs& a, unsigned grid, int outk, bool mean) {
    switch (outk * 2 + (mean ? 1 : 0)) {
        case 0: return pipe_launch_one<LOG2N, 0, false>(stream, a, grid);
        case 1: return pipe_launch_one<LOG2N, 0, true>(stream, a, grid);
        case 2: return pipe_launch_one<LOG2N, 1, false>(stream, a, grid);
        case 3: return pipe_launch_one<LOG2N, 1, true>(stream, a, grid);
        case 4: return pipe_launch_one<LOG2N, 2, false>(stream, a, grid);
        default: return pipe_launch_one<LOG2N, 2, true>(stream, a, grid);
    }
}

int pipe_max_tapers_demean() { return CfgP<12>::KMAX; }

int pipe_launch(hipStream_t stream, const MtmArgs& a, int log2n, unsigned grid, int outk, bool mean) {
    switch (log2n) {
        case 10: return pipe_launch_mode<10>(stream, a, grid, outk, mean);
        case 11: return pipe_launch_mode<11>(stream, a, grid, outk, mean);
        case 12: return pipe_launch_mode<12>(stream, a, grid, outk, mean);
        default: spy::set_error("no pipelined FFT kernel for N = 2^%d", log2n); return -1;
    }
}

}  // namespace spyfft
