// Dispatcher over the per-channel-count translation units of the 3-multiplication kernels (csd3m_launch.h)
#include "csd3m_launch.h"

namespace spycsd {
int m3_launch_a(int nchan, hipStream_t stream, CsdArgs a, long long nprow);
int m3_launch_b(int nchan, hipStream_t stream, CsdArgs a, long long nprow);
int m3_launch_c(int nchan, hipStream_t stream, CsdArgs a, long long nprow);
int m3_launch_d(int nchan, hipStream_t stream, CsdArgs a, long long nprow);
int m3_launch_e(int nchan, hipStream_t stream, CsdArgs a, long long nprow);
int m3_launch_f(int nchan, hipStream_t stream, CsdArgs a, long long nprow);
int m3_launch_g(int nchan, hipStream_t stream, CsdArgs a, long long nprow);
int m3_launch_h(int nchan, hipStream_t stream, CsdArgs a, long long nprow);

int m3_launch(int nchan, hipStream_t stream, CsdArgs a, long long nprow) {
    int rc;
    if ((rc = m3_launch_a(nchan, stream, a, nprow)) != -100) return rc;
    if ((rc = m3_launch_b(nchan, stream, a, nprow)) != -100) return rc;
    if ((rc = m3_launch_c(nchan, stream, a, nprow)) != -100) return rc;
    if ((rc = m3_launch_d(nchan, stream, a, nprow)) != -100) return rc;
    if ((rc = m3_launch_e(nchan, stream, a, nprow)) != -100) return rc;
    if ((rc = m3_launch_f(nchan, stream, a, nprow)) != -100) return rc;
    if ((rc = m3_launch_g(nchan, stream, a, nprow)) != -100) return rc;
    if ((rc = m3_launch_h(nchan, stream, a, nprow)) != -100) return rc;
    return -100;
}

// every multiple of 16 up to 512 channels is built (M3Tab generates the sub-tile tables at compile time)
bool m3_available(int nchan) { return nchan >= 16 && nchan <= 512 && nchan % 16 == 0; }
}  // namespace spycsd
