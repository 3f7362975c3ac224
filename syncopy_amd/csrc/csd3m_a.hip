// 3-multiplication cross-spectral kernels for 32, 64, 96, 128 channels (see csd3m_launch.h)
#include "csd3m_launch_impl.h"

namespace spycsd {
int m3_launch_a(int nchan, hipStream_t stream, CsdArgs a, long long nprow) {
    switch (nchan) {
        case 32: return m3_launch_one<32>(stream, a, nprow);
        case 64: return m3_launch_one<64>(stream, a, nprow);
        case 96: return m3_launch_one<96>(stream, a, nprow);
        case 128: return m3_launch_one<128>(stream, a, nprow);
        default: return -100;
    }
}
}  // namespace spycsd
