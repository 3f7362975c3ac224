// 3-multiplication cross-spectral kernels for up to 448, 464, 480 channels per instance, any channel count below an instance's
// (rows narrower than the LDS image: csd3m_kernel<CH, 8, false>; see csd3m_launch.h)
#include "csd3m_launch_impl.h"

namespace spycsd {
int m3_launch_g(int chp, hipStream_t stream, CsdArgs a, long long nprow) {
    switch (chp) {
        case 448: return m3_launch_one<448, false>(stream, a, nprow);
        case 464: return m3_launch_one<464, false>(stream, a, nprow);
        case 480: return m3_launch_one<480, false>(stream, a, nprow);
        default: return -100;
    }
}
}  // namespace spycsd
