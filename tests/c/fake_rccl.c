/* A recording LOOPBACK stand-in for librccl.so.1, for tests only (tests/test_fake_rccl.py).
 *
 * liblcs_hip.so resolves RCCL with dlopen at run time and prefers a copy the process already holds
 * (csrc/halo.hip load_rccl).  This library carries the SONAME librccl.so.1 and exports the 11 symbols halo.hip uses, so a
 * process that loads it first exercises lc_comm_* / lc_halo_exchange / lc_comm_flag_allreduce without a second GPU:
 *   - every call is RECORDED (a text log the test reads back: peer, pointer, count, dtype, stream of each send / recv,
 *     group brackets, all-reduces);
 *   - all "ranks" live in ONE process on one device.  A send is matched with the receive posted by its peer (same
 *     unique id, src/dst ranks crossed) whichever comes first, and the bytes are copied device-to-device on the
 *     receiver's stream; an all-reduce completes (element-wise max through the host) when every rank of the communicator
 *     has contributed.  So a test can run rank 0, 1, 2 of 3 one after the other and check real results;
 *   - fake_rccl_fail_at(n) makes the n-th send / recv from now on return an error (error-path tests).
 * Not a product component; nothing under lagrangiancoherence_amd/ knows it exists. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define __HIP_PLATFORM_AMD__ 1
#include <hip/hip_runtime_api.h>

typedef struct fake_comm {
    int nranks, rank;
    unsigned id;
} fake_comm;
typedef fake_comm *ncclComm_t;
typedef struct {
    char internal[128];
} ncclUniqueId;
typedef int ncclResult_t;

#define MAXLOG 4096
#define MAXOPS 256
static char *g_log[MAXLOG];
static int g_nlog = 0;
static int g_group_depth = 0;
static int g_fail_at = 0;    /* countdown over send / recv calls; 1 = the next one fails */
static unsigned g_next_id = 1;

typedef struct {
    int live, is_send, src, dst;
    unsigned id;
    void *buf;
    size_t bytes;
    hipStream_t stream;
} op_t;
static op_t g_ops[MAXOPS];

typedef struct {
    int live, rank;
    unsigned id;
    void *buf;
    size_t count;
    hipStream_t stream;
} red_t;
static red_t g_red[MAXOPS];

static void logf_(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
#include <stdarg.h>
static void logf_(const char *fmt, ...) {
    char line[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(line, sizeof line, fmt, ap);
    va_end(ap);
    if (g_nlog < MAXLOG) g_log[g_nlog++] = strdup(line);
}

/* ---- test controls ---- */
int fake_rccl_nlog(void) { return g_nlog; }
const char *fake_rccl_log_line(int i) { return (i >= 0 && i < g_nlog) ? g_log[i] : ""; }
void fake_rccl_reset(void) {
    for (int i = 0; i < g_nlog; ++i) free(g_log[i]);
    g_nlog = 0;
    g_fail_at = 0;
    memset(g_ops, 0, sizeof g_ops);
    memset(g_red, 0, sizeof g_red);
}
void fake_rccl_fail_at(int n) { g_fail_at = n; }
int fake_rccl_pending(void) {
    int n = 0;
    for (int i = 0; i < MAXOPS; ++i) n += g_ops[i].live + g_red[i].live;
    return n;
}
int fake_rccl_group_depth(void) { return g_group_depth; }

/* ---- the RCCL entry points halo.hip resolves ---- */
ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
    memset(id, 0, sizeof *id);
    memcpy(id->internal, "FAKE", 4);
    memcpy(id->internal + 4, &g_next_id, sizeof g_next_id);
    logf_("GetUniqueId id=%u", g_next_id);
    ++g_next_id;
    return 0;
}

ncclResult_t ncclCommInitRank(ncclComm_t *comm, int nranks, ncclUniqueId id, int rank) {
    if (memcmp(id.internal, "FAKE", 4) != 0) return 4; /* ncclInvalidArgument */
    fake_comm *c = (fake_comm *)calloc(1, sizeof *c);
    c->nranks = nranks;
    c->rank = rank;
    memcpy(&c->id, id.internal + 4, sizeof c->id);
    *comm = c;
    logf_("CommInitRank id=%u nranks=%d rank=%d", c->id, nranks, rank);
    return 0;
}
ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    logf_("CommDestroy rank=%d", comm->rank);
    free(comm);
    return 0;
}
ncclResult_t ncclCommCount(const ncclComm_t comm, int *n) {
    *n = comm->nranks;
    return 0;
}
ncclResult_t ncclCommUserRank(const ncclComm_t comm, int *r) {
    *r = comm->rank;
    return 0;
}
ncclResult_t ncclGroupStart(void) {
    ++g_group_depth;
    logf_("GroupStart");
    return 0;
}
ncclResult_t ncclGroupEnd(void) {
    --g_group_depth;
    logf_("GroupEnd");
    return 0;
}
const char *ncclGetErrorString(ncclResult_t r) { return r == 0 ? "no error" : (r == 1 ? "fake: injected failure" : "fake: error"); }

static size_t esize(int dtype) { return dtype == 8 ? 8 : (dtype == 7 || dtype == 3 ? 4 : 0); }

static ncclResult_t post(int is_send, void *buf, size_t count, int dtype, int peer, ncclComm_t comm, hipStream_t stream) {
    logf_("%s rank=%d peer=%d ptr=%llu count=%zu dtype=%d stream=%llu in_group=%d", is_send ? "Send" : "Recv", comm->rank, peer,
          (unsigned long long)(uintptr_t)buf, count, dtype, (unsigned long long)(uintptr_t)stream, g_group_depth);
    if (g_fail_at > 0 && --g_fail_at == 0) {
        logf_("INJECTED FAILURE");
        return 1;
    }
    if (peer < 0 || peer >= comm->nranks || peer == comm->rank || esize(dtype) == 0) return 4;
    const size_t bytes = count * esize(dtype);
    const int src = is_send ? comm->rank : peer, dst = is_send ? peer : comm->rank;
    for (int i = 0; i < MAXOPS; ++i) { /* the peer's matching op, oldest first */
        op_t *o = &g_ops[i];
        if (o->live && o->id == comm->id && o->is_send != is_send && o->src == src && o->dst == dst) {
            if (o->bytes != bytes) {
                logf_("SIZE MISMATCH %zu vs %zu", o->bytes, bytes);
                return 5;
            }
            void *from = is_send ? buf : o->buf, *to = is_send ? o->buf : buf;
            hipStream_t st = is_send ? o->stream : stream;
            if (hipMemcpyAsync(to, from, bytes, hipMemcpyDeviceToDevice, st) != hipSuccess) return 1;
            o->live = 0;
            return 0;
        }
    }
    for (int i = 0; i < MAXOPS; ++i)
        if (!g_ops[i].live) {
            op_t o = {1, is_send, src, dst, comm->id, buf, bytes, stream};
            g_ops[i] = o;
            return 0;
        }
    return 3;
}
ncclResult_t ncclSend(const void *buf, size_t count, int dtype, int peer, ncclComm_t comm, hipStream_t stream) {
    return post(1, (void *)buf, count, dtype, peer, comm, stream);
}
ncclResult_t ncclRecv(void *buf, size_t count, int dtype, int peer, ncclComm_t comm, hipStream_t stream) {
    return post(0, buf, count, dtype, peer, comm, stream);
}

ncclResult_t ncclAllReduce(const void *send, void *recv, size_t count, int dtype, int op, ncclComm_t comm, hipStream_t stream) {
    logf_("AllReduce rank=%d send=%llu recv=%llu count=%zu dtype=%d op=%d stream=%llu", comm->rank, (unsigned long long)(uintptr_t)send,
          (unsigned long long)(uintptr_t)recv, count, dtype, op, (unsigned long long)(uintptr_t)stream);
    if (dtype != 3 || op != 2 || send != recv) return 4; /* the one form lc_comm_flag_allreduce uses: uint32, max, in place */
    int have = 1;
    for (int i = 0; i < MAXOPS; ++i)
        if (g_red[i].live && g_red[i].id == comm->id) {
            if (g_red[i].count != count) return 5;
            ++have;
        }
    if (have < comm->nranks) {
        for (int i = 0; i < MAXOPS; ++i)
            if (!g_red[i].live) {
                red_t r = {1, comm->rank, comm->id, recv, count, stream};
                g_red[i] = r;
                return 0;
            }
        return 3;
    }
    /* last contributor: element-wise max over every rank's buffer, written back to all of them */
    uint32_t *acc = (uint32_t *)calloc(count, 4), *tmp = (uint32_t *)malloc(count * 4);
    if (hipStreamSynchronize(stream) != hipSuccess || hipMemcpy(acc, recv, count * 4, hipMemcpyDeviceToHost) != hipSuccess) return 1;
    for (int i = 0; i < MAXOPS; ++i)
        if (g_red[i].live && g_red[i].id == comm->id) {
            if (hipStreamSynchronize(g_red[i].stream) != hipSuccess || hipMemcpy(tmp, g_red[i].buf, count * 4, hipMemcpyDeviceToHost) != hipSuccess) return 1;
            for (size_t k = 0; k < count; ++k)
                if (tmp[k] > acc[k]) acc[k] = tmp[k];
        }
    if (hipMemcpy(recv, acc, count * 4, hipMemcpyHostToDevice) != hipSuccess) return 1;
    for (int i = 0; i < MAXOPS; ++i)
        if (g_red[i].live && g_red[i].id == comm->id) {
            if (hipMemcpy(g_red[i].buf, acc, count * 4, hipMemcpyHostToDevice) != hipSuccess) return 1;
            g_red[i].live = 0;
        }
    free(acc);
    free(tmp);
    return 0;
}
