// 3-multiplication cross-spectral kernels for 352, 368, 384 channels (see csd3m_launch.h)
#include "csd3m_launch_impl.h"

namespace spycsd {
int m3_launch_e(int nchan, hipStream_t stream, CsdArgs a, long long nprow) {
    switch (nchan) {
        case 352: return m3_launch_one<352>(stream, a, nprow);
        case 368: return m3_launch_one<368>(stream, a, nprow);
        case 384: return m3_launch_one<384>(stream, a, nprow);
        default: return -100;
    }
}
}  // namespace spycsd
