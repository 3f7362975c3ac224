// 3-multiplication cross-spectral kernels for 512 channels (see csd3m_launch.h)
#include "csd3m_launch_impl.h"

namespace spycsd {
int m3_launch_e(int nchan, hipStream_t stream, CsdArgs a, long long nprow) {
    switch (nchan) {
        case 512: return m3_launch_one<512>(stream, a, nprow);
        default: return -100;
    }
}
}  // namespace spycsd
