// 3-multiplication cross-spectral kernels for 144, 160, 176, 192, 208, 224, 240 channels (see csd3m_launch.h)
#include "csd3m_launch_impl.h"

namespace spycsd {
int m3_launch_b(int nchan, hipStream_t stream, CsdArgs a, long long nprow) {
    switch (nchan) {
        case 144: return m3_launch_one<144>(stream, a, nprow);
        case 160: return m3_launch_one<160>(stream, a, nprow);
        case 176: return m3_launch_one<176>(stream, a, nprow);
        case 192: return m3_launch_one<192>(stream, a, nprow);
        case 208: return m3_launch_one<208>(stream, a, nprow);
        case 224: return m3_launch_one<224>(stream, a, nprow);
        case 240: return m3_launch_one<240>(stream, a, nprow);
        default: return -100;
    }
}
}  // namespace spycsd
