// 3-multiplication cross-spectral kernels for 256, 272, 288 channels (see csd3m_launch.h)
#include "csd3m_launch_impl.h"

namespace spycsd {
int m3_launch_c(int nchan, hipStream_t stream, CsdArgs a, long long nprow) {
    switch (nchan) {
        case 256: return m3_launch_one<256>(stream, a, nprow);
        case 272: return m3_launch_one<272>(stream, a, nprow);
        case 288: return m3_launch_one<288>(stream, a, nprow);
        default: return -100;
    }
}
}  // namespace spycsd
