// 3-multiplication cross-spectral kernel for exactly 256 channels (both hand-over layouts; see csd3m_launch.h)
#include "csd3m_launch_impl.h"

namespace spycsd {
int m3_launch_exact256(hipStream_t stream, CsdArgs a, long long nprow) { return m3_launch_one<256, true>(stream, a, nprow); }
}  // namespace spycsd
