// 3-multiplication cross-spectral kernels for 400, 416, 432 channels (see csd3m_launch.h)
#include "csd3m_launch_impl.h"

namespace spycsd {
int m3_launch_f(int nchan, hipStream_t stream, CsdArgs a, long long nprow) {
    switch (nchan) {
        case 400: return m3_launch_one<400>(stream, a, nprow);
        case 416: return m3_launch_one<416>(stream, a, nprow);
        case 432: return m3_launch_one<432>(stream, a, nprow);
        default: return -100;
    }
}
}  // namespace spycsd
