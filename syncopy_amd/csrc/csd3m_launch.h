// Launchers of the 3-multiplication cross-spectral kernels (csd3m_kernel.h).  The kernels are instantiated per channel
// count in several translation units (csd3m_*.hip) that compile in parallel; csd.hip only sees this interface.
#pragma once
#include <hip/hip_runtime.h>

#include "csd_args.h"

namespace spycsd {

// workgroups per packed row (csd3m_kernel.h: M3Tab<CH>::NP): 1 up to 256 channels, ceil(sub-tiles / 112) above
inline int m3_parts(int nchan) {
    if (nchan <= 256) return 1;
    const int nb = nchan / 16;
    return (nb * (nb + 1) / 2 + 111) / 112;
}

// launch csd3m_kernel<nchan, 8> over the packed rows [0, nprow); 0, a negative spyhip error code, or -100 if this
// build has no instance for `nchan`
int m3_launch(int nchan, hipStream_t stream, CsdArgs a, long long nprow);

// channel counts this build instantiates: every multiple of 16 up to 512 (csd3m_{a..h}.hip)
bool m3_available(int nchan);

}  // namespace spycsd
