// Dispatcher over the per-channel-count translation units of the 3-multiplication kernels (csd3m_launch.h)
#include "csd3m_launch.h"

namespace spycsd {
int m3_launch_a(int nchan, hipStream_t stream, CsdArgs a, long long nprow);
int m3_launch_b(int nchan, hipStream_t stream, CsdArgs a, long long nprow);
int m3_launch_c(int nchan, hipStream_t stream, CsdArgs a, long long nprow);
int m3_launch_d(int nchan, hipStream_t stream, CsdArgs a, long long nprow);
int m3_launch_e(int nchan, hipStream_t stream, CsdArgs a, long long nprow);

int m3_launch(int nchan, hipStream_t stream, CsdArgs a, long long nprow) {
    int rc;
    if ((rc = m3_launch_a(nchan, stream, a, nprow)) != -100) return rc;
    if ((rc = m3_launch_b(nchan, stream, a, nprow)) != -100) return rc;
    if ((rc = m3_launch_c(nchan, stream, a, nprow)) != -100) return rc;
    if ((rc = m3_launch_d(nchan, stream, a, nprow)) != -100) return rc;
    if ((rc = m3_launch_e(nchan, stream, a, nprow)) != -100) return rc;
    return -100;
}

bool m3_available(int nchan) {
    switch (nchan) {
        case 32:
        case 64:
        case 96:
        case 128:
        case 160:
        case 192:
        case 224:
        case 256:
        case 320:
        case 384:
        case 512:
            return true;
        default: return false;
    }
}
}  // namespace spycsd
