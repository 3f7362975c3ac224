// C1: sums over ranks with RCCL, inside the library (spyhip_comm_* / spyhip_allreduce* of include/spyhip.h).
//
// Replaces the mutex-guarded `+=` of the reference's parallel trial map (shared/kwarg_decorators.py:723-735,
// shared/computational_routine.py:939-942).  librccl is resolved at run time: symbols already in the process
// (a host that imported PyTorch has RCCL loaded) win, otherwise librccl.so.1 is dlopen'ed - single-GPU users never
// touch it and the library has no link-time dependency on it.
#include <dlfcn.h>

#include <mutex>
#include <string>

#include "spy_common.h"

namespace {

// the handful of RCCL declarations used here (rccl/rccl.h: ncclUniqueId :43, ncclDataType_t :466-467, ncclRedOp_t :448)
struct UniqueId { char internal[SPYHIP_UNIQUE_ID_BYTES]; };
using Comm = void*;
constexpr int kSum = 0, kFloat32 = 7, kFloat64 = 8;

struct Rccl {
    int (*GetUniqueId)(UniqueId*) = nullptr;
    int (*CommInitRank)(Comm*, int, UniqueId, int) = nullptr;
    int (*CommDestroy)(Comm) = nullptr;
    int (*AllReduce)(const void*, void*, size_t, int, int, Comm, hipStream_t) = nullptr;
    const char* (*GetErrorString)(int) = nullptr;
    bool ok = false;
};

Rccl g_rccl;
std::once_flag g_once;
std::string g_dlerr;            // dlerror() of the failed dlopen, captured where it happened

void load_rccl() {
    void* h = nullptr;
    // (1) a copy that is in the process already, (2) the system library
    if (dlsym(RTLD_DEFAULT, "ncclCommInitRank")) h = RTLD_DEFAULT;
    if (!h) h = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
    if (!h) h = dlopen("/opt/rocm/lib/librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
    if (!h) {
        const char* e = dlerror();          // one call: it clears the state it reports
        g_dlerr = e ? e : "dlopen failed";
        return;
    }
    g_rccl.GetUniqueId = reinterpret_cast<decltype(g_rccl.GetUniqueId)>(dlsym(h, "ncclGetUniqueId"));
    g_rccl.CommInitRank = reinterpret_cast<decltype(g_rccl.CommInitRank)>(dlsym(h, "ncclCommInitRank"));
    g_rccl.CommDestroy = reinterpret_cast<decltype(g_rccl.CommDestroy)>(dlsym(h, "ncclCommDestroy"));
    g_rccl.AllReduce = reinterpret_cast<decltype(g_rccl.AllReduce)>(dlsym(h, "ncclAllReduce"));
    g_rccl.GetErrorString = reinterpret_cast<decltype(g_rccl.GetErrorString)>(dlsym(h, "ncclGetErrorString"));
    g_rccl.ok = g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.CommDestroy && g_rccl.AllReduce;
}

int need_rccl() {
    std::call_once(g_once, load_rccl);
    if (!g_rccl.ok) {
        spy::set_error("RCCL (librccl.so.1) could not be loaded: %s", g_dlerr.empty() ? "symbols missing" : g_dlerr.c_str());
        return -5;
    }
    return 0;
}

#define SPY_RCCL_CHECK(expr)                                                                              \
    do {                                                                                                  \
        int r__ = (expr);                                                                                 \
        if (r__ != 0) {                                                                                   \
            spy::set_error("%s failed: %s", #expr, g_rccl.GetErrorString ? g_rccl.GetErrorString(r__) : "RCCL error"); \
            return -5;                                                                                    \
        }                                                                                                 \
    } while (0)

}  // namespace

extern "C" int spyhip_comm_unique_id(void* id_out) {
    if (!id_out) { spy::set_error("comm_unique_id: null argument"); return -1; }
    if (int rc = need_rccl()) return rc;
    UniqueId id;
    SPY_RCCL_CHECK(g_rccl.GetUniqueId(&id));
    std::memcpy(id_out, id.internal, SPYHIP_UNIQUE_ID_BYTES);
    return 0;
}

extern "C" int spyhip_comm_init(spyhip_ctx* ctx, const void* id, int rank, int nranks) {
    if (!ctx || !id) { spy::set_error("comm_init: null argument"); return -1; }
    if (nranks < 1 || rank < 0 || rank >= nranks) { spy::set_error("comm_init: rank %d of %d", rank, nranks); return -1; }
    if (ctx->comm) { spy::set_error("comm_init: this context has a communicator already"); return -1; }
    if (int rc = need_rccl()) return rc;
    SPY_HIP_CHECK(hipSetDevice(ctx->device));
    UniqueId uid;
    std::memcpy(uid.internal, id, SPYHIP_UNIQUE_ID_BYTES);
    Comm c = nullptr;
    SPY_RCCL_CHECK(g_rccl.CommInitRank(&c, nranks, uid, rank));
    ctx->comm = c;
    ctx->comm_rank = rank;
    ctx->comm_nranks = nranks;
    return 0;
}

extern "C" int spyhip_comm_destroy(spyhip_ctx* ctx) {
    if (!ctx) { spy::set_error("comm_destroy: null ctx"); return -1; }
    if (!ctx->comm) return 0;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    const int r = g_rccl.CommDestroy(ctx->comm);
    ctx->comm = nullptr;
    ctx->comm_rank = -1;
    ctx->comm_nranks = 0;
    if (r != 0) { spy::set_error("ncclCommDestroy failed"); return -5; }
    return 0;
}

extern "C" int spyhip_comm_info(const spyhip_ctx* ctx, int* rank, int* nranks) {
    if (!ctx || !ctx->comm) return -1;
    if (rank) *rank = ctx->comm_rank;
    if (nranks) *nranks = ctx->comm_nranks;
    return 0;
}

extern "C" int spyhip_allreduce(spyhip_ctx* ctx, void* buf_d, int64_t n, int dtype) {
    if (!ctx || !buf_d) { spy::set_error("allreduce: null argument"); return -1; }
    if (!ctx->comm) { spy::set_error("allreduce: no communicator (call spyhip_comm_init first)"); return -1; }
    if (dtype != 0 && dtype != 1) { spy::set_error("allreduce: dtype %d (0 = float32, 1 = float64)", dtype); return -1; }
    if (n <= 0) return 0;
    SPY_HIP_CHECK(hipSetDevice(ctx->device));
    SPY_RCCL_CHECK(g_rccl.AllReduce(buf_d, buf_d, (size_t)n, dtype ? kFloat64 : kFloat32, kSum, ctx->comm, ctx->stream));
    return 0;
}

extern "C" int spyhip_allreduce_csd(spyhip_ctx* ctx, void* acc_d, int nfreq, int nchan) {
    if (!ctx || !acc_d) { spy::set_error("allreduce_csd: null argument"); return -1; }
    if (!ctx->comm) { spy::set_error("allreduce_csd: no communicator (call spyhip_comm_init first)"); return -1; }
    if (nfreq < 1 || nchan < 1) { spy::set_error("allreduce_csd: bad shape"); return -1; }
    SPY_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t npack = (size_t)nfreq * ((size_t)nchan * (nchan + 1) / 2);
    const size_t need = npack * sizeof(float2);
    if (need > ctx->comm_buf_bytes) {
        if (ctx->comm_buf) { SPY_HIP_CHECK(hipStreamSynchronize(ctx->stream)); (void)hipFree(ctx->comm_buf); ctx->comm_buf = nullptr; ctx->comm_buf_bytes = 0; }
        SPY_HIP_CHECK(hipMalloc(&ctx->comm_buf, need));
        ctx->comm_buf_bytes = need;
    }
    int rc = spyhip_csd_tril_pack(ctx, acc_d, nfreq, nchan, ctx->comm_buf);
    if (rc) return rc;
    SPY_RCCL_CHECK(g_rccl.AllReduce(ctx->comm_buf, ctx->comm_buf, 2 * npack, kFloat32, kSum, ctx->comm, ctx->stream));
    return spyhip_csd_tril_unpack(ctx, ctx->comm_buf, nfreq, nchan, acc_d);
}
