// 3-multiplication cross-spectral kernels for 496, 512 channels (see csd3m_launch.h)
#include "csd3m_launch_impl.h"

namespace spycsd {
int m3_launch_h(int nchan, hipStream_t stream, CsdArgs a, long long nprow) {
    switch (nchan) {
        case 496: return m3_launch_one<496>(stream, a, nprow);
        case 512: return m3_launch_one<512>(stream, a, nprow);
        default: return -100;
    }
}
}  // namespace spycsd
