// 3-multiplication cross-spectral kernels for 160, 192, 224 channels (see csd3m_launch.h)
#include "csd3m_launch_impl.h"

namespace spycsd {
int m3_launch_b(int nchan, hipStream_t stream, CsdArgs a, long long nprow) {
    switch (nchan) {
        case 160: return m3_launch_one<160>(stream, a, nprow);
        case 192: return m3_launch_one<192>(stream, a, nprow);
        case 224: return m3_launch_one<224>(stream, a, nprow);
        default: return -100;
    }
}
}  // namespace spycsd
