// 3-multiplication cross-spectral kernels for up to 144, 160, 176, 192, 208, 224, 240, 256 channels per instance, any channel count below an instance's
// (rows narrower than the LDS image: csd3m_kernel<CH, 8, false>; see csd3m_launch.h)
#include "csd3m_launch_impl.h"

namespace spycsd {
int m3_launch_b(int chp, hipStream_t stream, CsdArgs a, long long nprow) {
    switch (chp) {
        case 144: return m3_launch_one<144, false>(stream, a, nprow);
        case 160: return m3_launch_one<160, false>(stream, a, nprow);
        case 176: return m3_launch_one<176, false>(stream, a, nprow);
        case 192: return m3_launch_one<192, false>(stream, a, nprow);
        case 208: return m3_launch_one<208, false>(stream, a, nprow);
        case 224: return m3_launch_one<224, false>(stream, a, nprow);
        case 240: return m3_launch_one<240, false>(stream, a, nprow);
        case 256: return m3_launch_one<256, false>(stream, a, nprow);
        default: return -100;
    }
}
}  // namespace spycsd
