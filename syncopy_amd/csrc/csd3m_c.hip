// 3-multiplication cross-spectral kernels for up to 272, 288, 304 channels per instance, any channel count below an instance's
// (rows narrower than the LDS image: csd3m_kernel<CH, 8, false>; see csd3m_launch.h)
#include "csd3m_launch_impl.h"

namespace spycsd {
int m3_launch_c(int chp, hipStream_t stream, CsdArgs a, long long nprow) {
    switch (chp) {
        case 272: return m3_launch_one<272, false>(stream, a, nprow);
        case 288: return m3_launch_one<288, false>(stream, a, nprow);
        case 304: return m3_launch_one<304, false>(stream, a, nprow);
        default: return -100;
    }
}
}  // namespace spycsd
