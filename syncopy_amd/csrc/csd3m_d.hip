// 3-multiplication cross-spectral kernels for 304, 320, 336 channels (see csd3m_launch.h)
#include "csd3m_launch_impl.h"

namespace spycsd {
int m3_launch_d(int nchan, hipStream_t stream, CsdArgs a, long long nprow) {
    switch (nchan) {
        case 304: return m3_launch_one<304>(stream, a, nprow);
        case 320: return m3_launch_one<320>(stream, a, nprow);
        case 336: return m3_launch_one<336>(stream, a, nprow);
        default: return -100;
    }
}
}  // namespace spycsd
