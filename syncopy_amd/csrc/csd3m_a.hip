// 3-multiplication cross-spectral kernels for 16, 32, 48, 64, 80, 96, 112, 128 channels (see csd3m_launch.h)
#include "csd3m_launch_impl.h"

namespace spycsd {
int m3_launch_a(int nchan, hipStream_t stream, CsdArgs a, long long nprow) {
    switch (nchan) {
        case 16: return m3_launch_one<16>(stream, a, nprow);
        case 32: return m3_launch_one<32>(stream, a, nprow);
        case 48: return m3_launch_one<48>(stream, a, nprow);
        case 64: return m3_launch_one<64>(stream, a, nprow);
        case 80: return m3_launch_one<80>(stream, a, nprow);
        case 96: return m3_launch_one<96>(stream, a, nprow);
        case 112: return m3_launch_one<112>(stream, a, nprow);
        case 128: return m3_launch_one<128>(stream, a, nprow);
        default: return -100;
    }
}
}  // namespace spycsd
