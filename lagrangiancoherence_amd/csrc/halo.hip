// Halo exchange of the departure points between row-sharded GPUs, on RCCL.
//
// The reference has no parallelism; this is the one exchange step of the sharded path: after the
// advection and before the sigma kernel every rank needs the 2 boundary rows of (x_dep, y_dep) of the
// previous and the next rank (4th-order stencil, LCS/tools.py:202-207).  Non-periodic in rank.  The rows
// are contiguous in memory (longitude is never split), so the sends and receives work in place on the
// halo-extended buffers lc_advect writes into: per neighbour 2 sends + 2 receives of 2*nx elements, all
// inside one ncclGroupStart/End on the context's stream (64 KiB per message at nx = 4096 float32:
// latency-bound on xGMI; no collective anywhere).
//
// RCCL is resolved at run time (dlopen) so that the library loads -- and every single-GPU entry point
// works -- on a machine without it, and so that a host process that already carries an RCCL (PyTorch
// does) shares that copy instead of loading a second one.
#include <dlfcn.h>

#include "lcs_common.h"

// The handful of RCCL declarations this file uses, stated here so that the library builds without the RCCL
// headers (the functions themselves are resolved with dlopen at run time).  Values as in <rccl/rccl.h>
// (NCCL ABI: ncclFloat32 = 7, ncclFloat64 = 8, 128-byte unique id); a static_assert-style check against the
// real header runs in tests/test_capi_symbols.py where that header is installed.
typedef struct ncclComm *ncclComm_t;
typedef struct {
    char internal[128];
} ncclUniqueId;
typedef enum { ncclSuccess = 0 } ncclResult_t;
typedef enum { ncclUint32 = 3, ncclFloat32 = 7, ncclFloat64 = 8 } ncclDataType_t;
typedef enum { ncclMax = 2 } ncclRedOp_t;

struct lc_comm {
    ncclComm_t comm;
    int nranks, rank;
    lc_ctx *ctx;  // the context lc_comm_create was given (its stream carries lc_comm_flag_allreduce)
};

namespace {

struct Rccl {
    void *handle = nullptr;
    ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
    ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
    ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
    ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
    ncclResult_t (*GroupStart)() = nullptr;
    ncclResult_t (*GroupEnd)() = nullptr;
    ncclResult_t (*Send)(const void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*Recv)(void *, size_t, ncclDataType_t, int, ncclComm_t, hipStream_t) = nullptr;
    ncclResult_t (*AllReduce)(const void *, void *, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
    const char *(*GetErrorString)(ncclResult_t) = nullptr;
};

Rccl g_rccl;

bool load_rccl() {
    if (g_rccl.handle) return true;
    const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    void *h = nullptr;
    for (const char *n : names)  // a copy the process already holds first
        if ((h = dlopen(n, RTLD_NOW | RTLD_NOLOAD))) break;
    if (!h)
        for (const char *n : names)
            if ((h = dlopen(n, RTLD_NOW | RTLD_GLOBAL))) break;
    if (!h) {
        lc_set_error("RCCL not found (librccl.so.1): %s", dlerror());
        return false;
    }
    Rccl r;
    r.handle = h;
#define LC_SYM(field, name)                                        \
    *(void **)(&r.field) = dlsym(h, name);                         \
    if (!r.field) {                                                \
        lc_set_error("RCCL symbol %s missing", name);              \
        return false;                                              \
    }
    LC_SYM(GetUniqueId, "ncclGetUniqueId")
    LC_SYM(CommInitRank, "ncclCommInitRank")
    LC_SYM(CommDestroy, "ncclCommDestroy")
    LC_SYM(CommCount, "ncclCommCount")
    LC_SYM(CommUserRank, "ncclCommUserRank")
    LC_SYM(GroupStart, "ncclGroupStart")
    LC_SYM(GroupEnd, "ncclGroupEnd")
    LC_SYM(Send, "ncclSend")
    LC_SYM(Recv, "ncclRecv")
    LC_SYM(AllReduce, "ncclAllReduce")
    LC_SYM(GetErrorString, "ncclGetErrorString")
#undef LC_SYM
    g_rccl = r;
    return true;
}

#define LC_RCCL_CHECK(expr)                                                                       \
    do {                                                                                          \
        ncclResult_t _r = (expr);                                                                 \
        if (_r != ncclSuccess) {                                                                  \
            lc_set_error("%s failed: %s (%s:%d)", #expr, g_rccl.GetErrorString(_r), __FILE__, __LINE__); \
            return LC_ERCCL;                                                                      \
        }                                                                                         \
    } while (0)

}  // namespace

extern "C" int lc_comm_unique_id(void *id_out, size_t id_bytes) {
    LC_REQUIRE(id_out && id_bytes >= sizeof(ncclUniqueId), "lc_comm_unique_id: need a %zu-byte buffer", sizeof(ncclUniqueId));
    if (!load_rccl()) return LC_ERCCL;
    ncclUniqueId id;
    LC_RCCL_CHECK(g_rccl.GetUniqueId(&id));
    __builtin_memcpy(id_out, &id, sizeof(id));
    return LC_OK;
}

extern "C" int lc_comm_create(lc_ctx *ctx, int nranks, int rank, const void *id, size_t id_bytes, lc_comm **out) {
    LC_REQUIRE(ctx && out, "lc_comm_create: null pointer");
    LC_REQUIRE(nranks >= 1 && rank >= 0 && rank < nranks, "lc_comm_create: rank %d not in [0,%d)", rank, nranks);
    LC_REQUIRE(id && id_bytes >= sizeof(ncclUniqueId), "lc_comm_create: need the %zu-byte id of lc_comm_unique_id",
               sizeof(ncclUniqueId));
    if (!load_rccl()) return LC_ERCCL;
    LC_HIP_CHECK(hipSetDevice(ctx->device));
    ncclUniqueId uid;
    __builtin_memcpy(&uid, id, sizeof(uid));
    lc_comm *c = new lc_comm{nullptr, nranks, rank, ctx};
    ncclResult_t r = g_rccl.CommInitRank(&c->comm, nranks, uid, rank);
    if (r != ncclSuccess) {
        lc_set_error("ncclCommInitRank failed: %s", g_rccl.GetErrorString(r));
        delete c;
        return LC_ERCCL;
    }
    *out = c;
    return LC_OK;
}

extern "C" int lc_comm_destroy(lc_comm *comm) {
    if (!comm) return LC_OK;
    if (g_rccl.handle && comm->comm) g_rccl.CommDestroy(comm->comm);
    delete comm;
    return LC_OK;
}

extern "C" int lc_comm_count(const lc_comm *comm, int *nranks_out, int *rank_out) {
    LC_REQUIRE(comm && comm->comm && g_rccl.handle, "lc_comm_count: null communicator");
    int n = 0, r = 0;
    LC_RCCL_CHECK(g_rccl.CommCount(comm->comm, &n));  // what RCCL itself says, not what lc_comm_create was told
    LC_RCCL_CHECK(g_rccl.CommUserRank(comm->comm, &r));
    if (nranks_out) *nranks_out = n;
    if (rank_out) *rank_out = r;
    return LC_OK;
}

extern "C" int lc_comm_flag_allreduce(void *comm_, void *flags_dev, size_t count) {
    lc_comm *comm = (lc_comm *)comm_;
    LC_REQUIRE(comm && comm->comm && comm->ctx && g_rccl.handle, "lc_comm_flag_allreduce: null communicator");
    LC_REQUIRE(flags_dev || count == 0, "lc_comm_flag_allreduce: null buffer");
    if (comm->nranks == 1 || count == 0) return LC_OK;
    LC_RCCL_CHECK(g_rccl.AllReduce(flags_dev, flags_dev, count, ncclUint32, ncclMax, comm->comm, comm->ctx->stream));
    return LC_OK;
}

extern "C" int lc_halo_exchange(lc_ctx *ctx, lc_comm *comm, void *x_ext, void *y_ext, int dtype, int n_rows, int nx,
                                int n_lo, int n_hi) {
    LC_REQUIRE(ctx && comm && x_ext && y_ext, "lc_halo_exchange: null pointer");
    LC_REQUIRE(dtype == LC_F32 || dtype == LC_F64, "lc_halo_exchange: dtype must be LC_F32 or LC_F64");
    constexpr int HALO = 2;  // LCS/tools.py:202-207
    const int rank = comm->rank, nranks = comm->nranks;
    LC_REQUIRE(n_lo == (rank > 0 ? HALO : 0) && n_hi == (rank < nranks - 1 ? HALO : 0),
               "lc_halo_exchange: rank %d of %d needs halos (%d,%d), got (%d,%d)", rank, nranks, rank > 0 ? HALO : 0,
               rank < nranks - 1 ? HALO : 0, n_lo, n_hi);
    const int n = n_rows - n_lo - n_hi;
    LC_REQUIRE(nx >= 1 && n >= HALO, "lc_halo_exchange: block of %d rows is thinner than the halo", n);
    if (nranks == 1) return LC_OK;
    LC_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t es = dtype == LC_F32 ? 4 : 8, row = (size_t)nx * es, count = (size_t)HALO * nx;
    const ncclDataType_t dt = dtype == LC_F32 ? ncclFloat32 : ncclFloat64;
    char *bufs[2] = {(char *)x_ext, (char *)y_ext};
    LC_RCCL_CHECK(g_rccl.GroupStart());
    ncclResult_t r = ncclSuccess;
    auto both = [&](char *b, size_t send_row, size_t recv_row, int peer) {
        if (r == ncclSuccess) r = g_rccl.Send(b + send_row * row, count, dt, peer, comm->comm, ctx->stream);
        if (r == ncclSuccess) r = g_rccl.Recv(b + recv_row * row, count, dt, peer, comm->comm, ctx->stream);
    };
    for (char *b : bufs) {
        if (rank > 0) both(b, (size_t)n_lo, 0, rank - 1);                                   // rows just below ours
        if (rank < nranks - 1) both(b, (size_t)(n_lo + n - HALO), (size_t)(n_lo + n), rank + 1);  // rows above ours
    }
    const ncclResult_t rend = g_rccl.GroupEnd();  // always close the group
    if (r == ncclSuccess) r = rend;
    if (r != ncclSuccess) {
        lc_set_error("lc_halo_exchange: RCCL send/recv failed: %s", g_rccl.GetErrorString(r));
        return LC_ERCCL;
    }
    return LC_OK;
}
