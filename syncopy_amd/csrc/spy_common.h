// Shared host-side plumbing of libspyhip: context, error reporting, launch checks.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/spyhip.h"

#include "spy_intrinsics.h"

struct spyhip_ctx {
    int device = 0;
    hipStream_t stream = nullptr;
    void* scratch = nullptr;        // library-owned device scratch (partial sums of split launches), grown on demand
    size_t scratch_bytes = 0;
    void* comm = nullptr;           // ncclComm_t of spyhip_comm_init (comm.hip), or nullptr
    int comm_rank = -1, comm_nranks = 0;
    void* comm_buf = nullptr;       // packed lower triangle travelling through spyhip_allreduce_csd
    size_t comm_buf_bytes = 0;
    void* arena = nullptr;          // work arrays of spyhip_granger (5 x F n^2 complex128 ...): kept between calls - a
    size_t arena_bytes = 0;         // hipMalloc + hipFree of 11 GB per call cost 0.05 ... 1 s; spyhip_ctx_trim frees it
    void* k4h_buf = nullptr;        // spyhip_csd_accumulate_split: 256 floats (the library's own range pass) + one flag per frequency
    size_t k4h_bytes = 0;
    int k4h_nf = 0;                 // frequencies the half-precision kernel was launched on in the last call
    hipEvent_t k4h_done = nullptr;  // recorded behind the last reader of k4h_buf: the next call's stream waits on it, so two
                                    // calls issued on different streams cannot trade flags (the buffer is per context)
    int csd_phase_exact = 0;        // spyhip_csd_set_phase_exact: 4-multiplication K4 kernels only (csd.hip)
    int granger_iters = 0;          // Wilson iterations of the last spyhip_granger call on this context
    int num_cu = 256;
    size_t lds_per_block = 160 * 1024;
};

namespace spy {

void set_error(const char* fmt, ...);

#define SPY_HIP_CHECK(expr)                                                                  \
    do {                                                                                     \
        hipError_t e__ = (expr);                                                             \
        if (e__ != hipSuccess) {                                                             \
            spy::set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(e__), __FILE__, \
                           __LINE__);                                                        \
            return -2;                                                                       \
        }                                                                                    \
    } while (0)

// device buffer owned by a plan
template <typename T>
struct DevBuf {
    T* p = nullptr;
    size_t n = 0;
    int alloc(size_t count) {
        n = count;
        if (count == 0) return 0;
        SPY_HIP_CHECK(hipMalloc(reinterpret_cast<void**>(&p), count * sizeof(T)));
        return 0;
    }
    int upload(const std::vector<T>& h, hipStream_t s) {
        if (alloc(h.size())) return -2;
        if (h.empty()) return 0;
        SPY_HIP_CHECK(hipMemcpyAsync(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice, s));
        SPY_HIP_CHECK(hipStreamSynchronize(s));
        return 0;
    }
    ~DevBuf() {
        if (p) (void)hipFree(p);
    }
};

static inline int ilog2(unsigned v) {
    int l = 0;
    while ((1u << l) < v) ++l;
    return l;
}
static inline bool is_pow2(unsigned v) { return v && !(v & (v - 1)); }

}  // namespace spy
