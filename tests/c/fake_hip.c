/* A recording stand-in for the HIP runtime, for running the HOST side of liblcs_hip under AddressSanitizer / UBSan on a
 * machine without a GPU (tests/test_host_orchestration_asan.py).
 *
 * "Device" memory is host memory from malloc (so a buffer overrun of a copy, a use after free, a double free or a leak is
 * the sanitizer's to report), copies are memcpy / memset, kernel launches are counted and do nothing (outputs are
 * whatever the buffers held: the tests look at status codes and at what was allocated and freed, not at numbers),
 * streams are opaque tokens.  hipMalloc / hipMallocAsync fail with hipErrorOutOfMemory at the call number
 * fake_hip_fail_malloc_at() names, and any copy at fake_hip_fail_memcpy_at(): every early-return path of the host routes
 * is then walked once.  Callable from several threads at once, as the real runtime is (the books behind one lock, the last error
 * and the pushed launch configuration per thread).  Only what csrc/ *.hip call is here; a new call shows up as an undefined symbol at link time. */
#include <hip/hip_runtime_api.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define MAX_LIVE 4096
static void *g_live[MAX_LIVE];
static size_t g_live_bytes[MAX_LIVE];
static int g_nlive, g_mallocs, g_frees, g_launches, g_copies, g_fail_malloc_at, g_fail_memcpy_at, g_bad_free;
static _Thread_local hipError_t g_last = hipSuccess;
static pthread_mutex_t g_books = PTHREAD_MUTEX_INITIALIZER;
#define LOCKED(stmt) do { pthread_mutex_lock(&g_books); stmt; pthread_mutex_unlock(&g_books); } while (0)

int fake_hip_live(void) { int n; LOCKED(n = g_nlive); return n; }
int fake_hip_mallocs(void) { return g_mallocs; }
int fake_hip_frees(void) { return g_frees; }
int fake_hip_launches(void) { return g_launches; }
int fake_hip_copies(void) { return g_copies; }
int fake_hip_bad_frees(void) { return g_bad_free; }
void fake_hip_fail_malloc_at(int n) { g_fail_malloc_at = n; }   /* the n-th allocation from now (1 = the next) fails; 0 = never */
void fake_hip_fail_memcpy_at(int n) { g_fail_memcpy_at = n; }
void fake_hip_reset_counts(void) { g_mallocs = g_frees = g_launches = g_copies = 0; }

static hipError_t do_malloc(void **p, size_t bytes) {
    int refuse;
    LOCKED(++g_mallocs; refuse = (g_fail_malloc_at > 0 && --g_fail_malloc_at == 0) || g_nlive == MAX_LIVE);
    *p = refuse ? NULL : malloc(bytes ? bytes : 1);
    if (!*p) return g_last = hipErrorOutOfMemory;
    LOCKED(g_live[g_nlive] = *p; g_live_bytes[g_nlive++] = bytes);
    return hipSuccess;
}
static hipError_t do_free(void *p) {
    if (!p) return hipSuccess;
    int found = 0;
    pthread_mutex_lock(&g_books);
    for (int i = 0; i < g_nlive && !found; ++i)
        if (g_live[i] == p) {
            g_live[i] = g_live[--g_nlive];
            g_live_bytes[i] = g_live_bytes[g_nlive];
            ++g_frees;
            found = 1;
        }
    if (!found) ++g_bad_free; /* not ours, or freed twice */
    pthread_mutex_unlock(&g_books);
    if (!found) return g_last = hipErrorInvalidValue;
    free(p);
    return hipSuccess;
}
static hipError_t do_copy(void *dst, const void *src, size_t n) {
    int refuse;
    LOCKED(++g_copies; refuse = g_fail_memcpy_at > 0 && --g_fail_memcpy_at == 0);
    if (refuse) return g_last = hipErrorInvalidValue;
    if (n) memcpy(dst, src, n); /* an overrun of either side is ASan's to catch */
    return hipSuccess;
}

hipError_t hipGetDeviceCount(int *n) { *n = 1; return hipSuccess; }
hipError_t hipDeviceGetAttribute(int *v, hipDeviceAttribute_t a, int d) { (void)a; (void)d; *v = 256; return hipSuccess; }
hipError_t hipSetDevice(int d) { return d == 0 ? hipSuccess : (g_last = hipErrorInvalidDevice); }
hipError_t hipStreamCreateWithFlags(hipStream_t *s, unsigned flags) { (void)flags; *s = (hipStream_t)malloc(1); return *s ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipStreamDestroy(hipStream_t s) { free(s); return hipSuccess; }
hipError_t hipStreamSynchronize(hipStream_t s) { (void)s; return hipSuccess; }
/* pinned host memory, events, cross-stream waits (the host routes' staging ring, csrc/hostxfer.h): host memory is host memory here,
 * every "asynchronous" operation has completed when its call returns, so events are tokens and waits are no-ops.  Pinned
 * allocations are counted like device ones (a leak of either shows in fake_hip_live()). */
static int g_fail_host_malloc = 0;
void fake_hip_fail_host_malloc(int on) { g_fail_host_malloc = on; }    /* 1: hipHostMalloc fails (the routes must fall back to plain copies) */
hipError_t hipHostMalloc(void **p, size_t bytes, unsigned flags) {
    (void)flags;
    if (g_fail_host_malloc) { *p = NULL; return g_last = hipErrorOutOfMemory; }
    return do_malloc(p, bytes);
}
hipError_t hipHostFree(void *p) { return do_free(p); }
hipError_t hipEventCreateWithFlags(hipEvent_t *e, unsigned flags) { (void)flags; *e = (hipEvent_t)malloc(1); return *e ? hipSuccess : hipErrorOutOfMemory; }
hipError_t hipEventDestroy(hipEvent_t e) { free(e); return hipSuccess; }
hipError_t hipEventRecord(hipEvent_t e, hipStream_t s) { (void)e; (void)s; return hipSuccess; }
hipError_t hipEventSynchronize(hipEvent_t e) { (void)e; return hipSuccess; }
hipError_t hipStreamWaitEvent(hipStream_t s, hipEvent_t e, unsigned flags) { (void)s; (void)e; (void)flags; return hipSuccess; }
hipError_t hipMalloc(void **p, size_t bytes) { return do_malloc(p, bytes); }
hipError_t hipMallocAsync(void **p, size_t bytes, hipStream_t s) { (void)s; return do_malloc(p, bytes); }
hipError_t hipFree(void *p) { return do_free(p); }
hipError_t hipFreeAsync(void *p, hipStream_t s) { (void)s; return do_free(p); }
hipError_t hipMemcpy(void *d, const void *s, size_t n, hipMemcpyKind k) { (void)k; return do_copy(d, s, n); }
hipError_t hipMemcpyAsync(void *d, const void *s, size_t n, hipMemcpyKind k, hipStream_t st) { (void)k; (void)st; return do_copy(d, s, n); }
hipError_t hipMemsetAsync(void *d, int v, size_t n, hipStream_t st) { (void)st; memset(d, v, n); return hipSuccess; }
hipError_t hipGetLastError(void) { hipError_t e = g_last; g_last = hipSuccess; return e; }
const char *hipGetErrorString(hipError_t e) { return e == hipSuccess ? "no error" : (e == hipErrorOutOfMemory ? "out of memory (injected)" : "fake HIP error"); }
hipError_t hipLaunchKernel(const void *f, dim3 grid, dim3 block, void **args, size_t shmem, hipStream_t st) {
    (void)f; (void)args; (void)shmem; (void)st;
    LOCKED(++g_launches);
    if (grid.x == 0 || grid.y == 0 || grid.z == 0 || block.x * block.y * block.z == 0 || block.x * block.y * block.z > 1024 || grid.y > 65535 || grid.z > 65535)
        return g_last = hipErrorInvalidConfiguration; /* what the real runtime refuses */
    return hipSuccess;
}
/* what hipcc's host stubs call */
static _Thread_local struct { dim3 grid, block; size_t shmem; hipStream_t st; } g_cfg;
void **__hipRegisterFatBinary(const void *data) { (void)data; static void *h; return &h; }
void __hipUnregisterFatBinary(void **h) { (void)h; }
void __hipRegisterFunction(void **h, const void *host, char *dev, const char *name, int tl, void *a, void *b, void *c, void *d, int *e) {
    (void)h; (void)host; (void)dev; (void)name; (void)tl; (void)a; (void)b; (void)c; (void)d; (void)e;
}
void __hipRegisterVar(void **h, void *var, char *a, char *b, int ext, size_t size, int constant, int global) {
    (void)h; (void)var; (void)a; (void)b; (void)ext; (void)size; (void)constant; (void)global;
}
hipError_t __hipPushCallConfiguration(dim3 grid, dim3 block, size_t shmem, hipStream_t st) {
    g_cfg.grid = grid; g_cfg.block = block; g_cfg.shmem = shmem; g_cfg.st = st;
    return hipSuccess;
}
hipError_t __hipPopCallConfiguration(dim3 *grid, dim3 *block, size_t *shmem, hipStream_t *st) {
    *grid = g_cfg.grid; *block = g_cfg.block; *shmem = g_cfg.shmem; *st = g_cfg.st;
    return hipSuccess;
}
