// Dispatcher over the translation units of the 3-multiplication kernels (csd3m_launch.h)
#include "csd3m_launch.h"

namespace spycsd {
int m3_launch_exact256(hipStream_t stream, CsdArgs a, long long nprow);
int m3_launch_a(int chp, hipStream_t stream, CsdArgs a, long long nprow);
int m3_launch_b(int chp, hipStream_t stream, CsdArgs a, long long nprow);
int m3_launch_c(int chp, hipStream_t stream, CsdArgs a, long long nprow);
int m3_launch_d(int chp, hipStream_t stream, CsdArgs a, long long nprow);
int m3_launch_e(int chp, hipStream_t stream, CsdArgs a, long long nprow);
int m3_launch_f(int chp, hipStream_t stream, CsdArgs a, long long nprow);
int m3_launch_g(int chp, hipStream_t stream, CsdArgs a, long long nprow);
int m3_launch_h(int chp, hipStream_t stream, CsdArgs a, long long nprow);

int m3_launch(int nchan, hipStream_t stream, CsdArgs a, long long nprow) {
    if (nchan == 256) return m3_launch_exact256(stream, a, nprow);
    return m3_launch_padded(m3_padded(nchan), stream, a, nprow);
}

int m3_launch_padded(int chp, hipStream_t stream, CsdArgs a, long long nprow) {
    int rc;
    if ((rc = m3_launch_a(chp, stream, a, nprow)) != -100) return rc;
    if ((rc = m3_launch_b(chp, stream, a, nprow)) != -100) return rc;
    if ((rc = m3_launch_c(chp, stream, a, nprow)) != -100) return rc;
    if ((rc = m3_launch_d(chp, stream, a, nprow)) != -100) return rc;
    if ((rc = m3_launch_e(chp, stream, a, nprow)) != -100) return rc;
    if ((rc = m3_launch_f(chp, stream, a, nprow)) != -100) return rc;
    if ((rc = m3_launch_g(chp, stream, a, nprow)) != -100) return rc;
    if ((rc = m3_launch_h(chp, stream, a, nprow)) != -100) return rc;
    return -100;
}

bool m3_available(int nchan) { return nchan >= 1 && nchan <= 512; }
}  // namespace spycsd
