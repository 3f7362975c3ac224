// 3-multiplication cross-spectral kernels for 448, 464, 480 channels (see csd3m_launch.h)
#include "csd3m_launch_impl.h"

namespace spycsd {
int m3_launch_g(int nchan, hipStream_t stream, CsdArgs a, long long nprow) {
    switch (nchan) {
        case 448: return m3_launch_one<448>(stream, a, nprow);
        case 464: return m3_launch_one<464>(stream, a, nprow);
        case 480: return m3_launch_one<480>(stream, a, nprow);
        default: return -100;
    }
}
}  // namespace spycsd
