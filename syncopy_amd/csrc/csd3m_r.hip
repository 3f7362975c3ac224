// 3-multiplication cross-spectral kernel for the rectangle between two 256-channel blocks of a recording with more than
// 512 channels (csd3m_kernel<512, 8, false, true>; see csd3m_launch.h)
#include "csd3m_launch_impl.h"

namespace spycsd {
int m3_launch_rect(hipStream_t stream, CsdArgs a, long long nfreq) { return m3_launch_one<512, false, true>(stream, a, nfreq); }
// the 4-multiplication product in the same tiling (phase-exact accumulation of more than 512 channels)
int m4_launch_rect(hipStream_t stream, CsdArgs a, long long nfreq) { return m3_launch_one<512, false, true, true>(stream, a, nfreq); }
int m4_launch_block(hipStream_t stream, CsdArgs a, long long nfreq) { return m3_launch_one<256, false, false, true>(stream, a, nfreq); }
}  // namespace spycsd
