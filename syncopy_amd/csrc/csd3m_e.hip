// 3-multiplication cross-spectral kernels for up to 352, 368, 384 channels per instance, any channel count below an instance's
// (rows narrower than the LDS image: csd3m_kernel<CH, 8, false>; see csd3m_launch.h)
#include "csd3m_launch_impl.h"

namespace spycsd {
int m3_launch_e(int chp, hipStream_t stream, CsdArgs a, long long nprow) {
    switch (chp) {
        case 352: return m3_launch_one<352, false>(stream, a, nprow);
        case 368: return m3_launch_one<368, false>(stream, a, nprow);
        case 384: return m3_launch_one<384, false>(stream, a, nprow);
        default: return -100;
    }
}
}  // namespace spycsd
