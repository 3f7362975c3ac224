// Launchers of the 3-multiplication cross-spectral kernels (csd3m_kernel.h).  The kernels are instantiated per channel
// count in several translation units (csd3m_*.hip) that compile in parallel; csd.hip only sees this interface.
#pragma once
#include <hip/hip_runtime.h>

#include "csd_args.h"

namespace spycsd {

// channel count of the kernel instance that serves `nchan` channels: the next multiple of 16 (the LDS image of a
// narrower row is padded, csd3m_kernel.h)
inline int m3_padded(int nchan) { return (nchan + 15) & ~15; }

// workgroups per packed row (csd3m_kernel.h: M3Tab<CH>::NP): 1 up to 256 channels, ceil(sub-tiles / 112) above
inline int m3_parts(int nchan) {
    const int chp = m3_padded(nchan);
    if (chp <= 256) return 1;
    const int nb = chp / 16;
    return (nb * (nb + 1) / 2 + 111) / 112;
}

// launch the kernel instance for `nchan` channels (a.C = nchan; csd3m_kernel<256, 8, true> for exactly 256,
// csd3m_kernel<m3_padded(nchan), 8, false> otherwise) over the packed rows [0, nprow); 0, a negative spyhip error code,
// or -100 if this build has no such instance.  Odd channel counts: the caller must keep the last row of spectra out of
// a.nrows (the lane of the last channel reads 8 bytes beyond its frequency).
int m3_launch(int nchan, hipStream_t stream, CsdArgs a, long long nprow);

// the instance csd3m_kernel<chp, 8, false> (chp a multiple of 16 up to 512) whatever the channel count: channel
// sub-ranges of wider rows (a.ctot, a.ch0, a.n0)
int m3_launch_padded(int chp, hipStream_t stream, CsdArgs a, long long nprow);

// the rectangle of the lower triangle between two channel blocks (a.ch1, a.n1 <= 256: rows) x (a.ch0, a.n0 <= 256:
// columns) of rows that are a.ctot channels wide, all `nfreq` frequencies
int m3_launch_rect(hipStream_t stream, CsdArgs a, long long nfreq);
// the same tiling with the 4-multiplication product (csd3m_kernel<..., M4 = true>): rectangle, and one Hermitian block of
// up to 256 channels (padded inside the 256-channel image)
int m4_launch_rect(hipStream_t stream, CsdArgs a, long long nfreq);
int m4_launch_block(hipStream_t stream, CsdArgs a, long long nfreq);

// channel counts served: 1 ... 512 (instances for every multiple of 16, csd3m_{a..h}.hip + csd3m_x.hip)
bool m3_available(int nchan);

}  // namespace spycsd
