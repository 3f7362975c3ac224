// 3-multiplication cross-spectral kernels for up to 16, 32, 48, 64, 80, 96, 112, 128 channels per instance, any channel count below an instance's
// (rows narrower than the LDS image: csd3m_kernel<CH, 8, false>; see csd3m_launch.h)
#include "csd3m_launch_impl.h"

namespace spycsd {
int m3_launch_a(int chp, hipStream_t stream, CsdArgs a, long long nprow) {
    switch (chp) {
        case 16: return m3_launch_one<16, false>(stream, a, nprow);
        case 32: return m3_launch_one<32, false>(stream, a, nprow);
        case 48: return m3_launch_one<48, false>(stream, a, nprow);
        case 64: return m3_launch_one<64, false>(stream, a, nprow);
        case 80: return m3_launch_one<80, false>(stream, a, nprow);
        case 96: return m3_launch_one<96, false>(stream, a, nprow);
        case 112: return m3_launch_one<112, false>(stream, a, nprow);
        case 128: return m3_launch_one<128, false>(stream, a, nprow);
        default: return -100;
    }
}
}  // namespace spycsd
