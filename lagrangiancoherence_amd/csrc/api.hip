// C ABI glue: contexts, device memory, error strings, field packing entry point,
// Gaussian smoothing of the departure fields, and the one-call host entry point.
#include <chrono>
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "hostxfer.h"
#include "lcs_common.h"

static thread_local char g_err[1024] = "";

void lc_set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" const char *lc_last_error(void) { return g_err; }
extern "C" int lc_version(void) { return LC_VERSION; }
#ifndef LCS_BUILD_ID
#define LCS_BUILD_ID "unstamped"
#endif
extern "C" const char *lc_build_id(void) { return LCS_BUILD_ID; }

extern "C" int lc_ctx_create(int device, lc_ctx **out) {
    LC_REQUIRE(out, "lc_ctx_create: null out pointer");
    *out = nullptr;
    int ndev = 0;
    LC_HIP_CHECK(hipGetDeviceCount(&ndev));
    LC_REQUIRE(device >= 0 && device < ndev, "lc_ctx_create: device %d not in [0,%d)", device, ndev);
    LC_HIP_CHECK(hipSetDevice(device));
    lc_ctx *c = new (std::nothrow) lc_ctx;
    if (!c) {
        lc_set_error("lc_ctx_create: out of host memory");
        return LC_ENOMEM;
    }
    c->device = device;
    c->own_stream = nullptr;
    hipError_t e = hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        lc_set_error("hipStreamCreate failed: %s", hipGetErrorString(e));
        delete c;
        return LC_EHIP;
    }
    c->stream = c->own_stream;
    c->n_cus = 256;
    {
        int n = 0;
        if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, c->device) == hipSuccess && n > 0) c->n_cus = n;
        else (void)hipGetLastError();
    }
    c->lds_tiles = 3;  // by size
    if (const char *ev = getenv("LCS_LDS_TILES")) c->lds_tiles = ev[0] == '0' ? 0 : (ev[0] == '2' ? 2 : (ev[0] == '1' ? 1 : 3));  // read once, here
    c->xcd_chunk_rows = 1;  // tile rows dealt to the XCDs cyclically (measured: C3 -1.7 %, C4 -2.8 %, C5 -5 % against contiguous bands)
    if (const char *ev = getenv("LCS_XCD_CHUNK_ROWS")) c->xcd_chunk_rows = atoi(ev) > 0 ? atoi(ev) : 0;  // read once, here
    c->xcd_split = -1;  // by the launch's shape (lcplan::xcd_chunk_tiles)
    if (const char *ev = getenv("LCS_XCD_SPLIT")) c->xcd_split = atoi(ev) >= 0 ? atoi(ev) : -1;  // read once, here (0: whole tile rows always)
    c->tile_order = -1;
    if (const char *ev = getenv("LCS_TILE_ORDER")) c->tile_order = (ev[0] >= '0' && ev[0] <= '3') ? ev[0] - '0' : -1;  // read once, here
    c->pole_blocks = 1;
    if (const char *ev = getenv("LCS_POLE_BLOCKS")) c->pole_blocks = ev[0] != '0';  // read once, here
    c->fir_prefilter = 1;
    if (const char *ev = getenv("LCS_FIR_PREFILTER")) c->fir_prefilter = ev[0] == '0' ? 0 : (ev[0] == '2' ? 2 : 1);  // read once, here
    c->fused_prefilter = 1;
    if (const char *ev = getenv("LCS_FUSED_PREFILTER")) c->fused_prefilter = ev[0] != '0';
    c->sigma_march = 2;  // by size
    if (const char *ev = getenv("LCS_SIGMA_MARCH")) c->sigma_march = ev[0] == '0' ? 0 : (ev[0] == '1' ? 1 : 2);  // read once, here
    c->level_chunk = -1;  // by size (advect.hip: 32 levels per launch from 2^22 seeds per call)
    if (const char *ev = getenv("LCS_LEVEL_CHUNK")) c->level_chunk = atoi(ev) >= 0 ? atoi(ev) : -1;  // read once, here
    c->f64_fidelity = LC_F64_AUTO;
    if (const char *ev = getenv("LCS_F64_FIDELITY")) c->f64_fidelity = ev[0] == 'e' ? LC_F64_EXACT_ORDER : (ev[0] == 'f' ? LC_F64_FAST : LC_F64_AUTO);  // read once, here
    c->patch_mode = -1;
    if (const char *ev = getenv("LCS_PATCH_MODE")) c->patch_mode = (ev[0] >= '0' && ev[0] <= '3') ? ev[0] - '0' : -1;  // read once, here
    c->flag_reduce = nullptr;
    c->flag_reduce_user = nullptr;
    c->lds_tiles_init = c->lds_tiles;
    c->sigma_march_init = c->sigma_march;
    c->last_advect_kernel = "";
    c->last_advect_launches = 0;
    c->last_sigma_kernel = "";
    c->last_pack_kernel = "";
    c->verify_dev = nullptr;
    c->trunc = nullptr;
    c->xfer = nullptr;
    c->host_ws = nullptr;
    c->host_cache = 1;
    if (const char *ev = getenv("LCS_HOST_CACHE")) c->host_cache = ev[0] != '0';  // read once, here
    c->host_timing = getenv("LCS_HOST_TIMING") != nullptr;
    for (double &m : c->host_marks) m = 0.0;
    c->host_pipeline = 1;
    if (const char *ev = getenv("LCS_HOST_PIPELINE")) c->host_pipeline = ev[0] != '0';  // read once, here
    c->f64_wg_tile = 0;
    if (const char *ev = getenv("LCS_F64_WG_TILE")) c->f64_wg_tile = ev[0] == '1';  // read once, here
    c->host_threads = -1;
    c->host_piece_mb = 0;
    if (const char *ev = getenv("LCS_HOST_THREADS")) c->host_threads = atoi(ev);
    if (const char *ev = getenv("LCS_HOST_PIECE_MB")) c->host_piece_mb = std::min(std::max(atoi(ev), 1), 256);
    *out = c;
    return LC_OK;
}

extern "C" int lc_ctx_set_lds_tiles(lc_ctx *ctx, int mode) {
    LC_REQUIRE(ctx, "lc_ctx_set_lds_tiles: null context");
    LC_REQUIRE(mode >= -1 && mode <= 2, "lc_ctx_set_lds_tiles: mode must be -1, 0, 1 or 2");
    ctx->lds_tiles = mode < 0 ? ctx->lds_tiles_init : mode;
    return LC_OK;
}

extern "C" int lc_ctx_set_sigma_march(lc_ctx *ctx, int on) {
    LC_REQUIRE(ctx, "lc_ctx_set_sigma_march: null context");
    LC_REQUIRE(on >= -1 && on <= 1, "lc_ctx_set_sigma_march: on must be -1, 0 or 1");
    ctx->sigma_march = on < 0 ? ctx->sigma_march_init : on;
    return LC_OK;
}

extern "C" int lc_ctx_set_flag_allreduce(lc_ctx *ctx, lc_flag_allreduce_fn fn, void *user) {
    LC_REQUIRE(ctx, "lc_ctx_set_flag_allreduce: null context");
    ctx->flag_reduce = fn;
    ctx->flag_reduce_user = user;
    return LC_OK;
}

extern "C" int lc_ctx_set_f64_fidelity(lc_ctx *ctx, int mode) {
    LC_REQUIRE(ctx, "lc_ctx_set_f64_fidelity: null context");
    LC_REQUIRE(mode >= LC_F64_AUTO && mode <= LC_F64_FAST, "lc_ctx_set_f64_fidelity: mode %d (LC_F64_AUTO / LC_F64_EXACT_ORDER / LC_F64_FAST)", mode);
    ctx->f64_fidelity = mode;
    return LC_OK;
}

extern "C" int lc_ctx_set_host_pipeline(lc_ctx *ctx, int on) {
    LC_REQUIRE(ctx, "lc_ctx_set_host_pipeline: null context");
    LC_REQUIRE(on == 0 || on == 1, "lc_ctx_set_host_pipeline: 0 or 1");
    ctx->host_pipeline = on;
    return LC_OK;
}

extern "C" int lc_ctx_last_host_marks(const lc_ctx *ctx, double *ms4_out) {
    LC_REQUIRE(ctx && ms4_out, "lc_ctx_last_host_marks: null pointer");
    for (int i = 0; i < 4; ++i) ms4_out[i] = ctx->host_marks[i];
    return LC_OK;
}

extern "C" int lc_ctx_set_xcd_split(lc_ctx *ctx, int split) {
    LC_REQUIRE(ctx, "lc_ctx_set_xcd_split: null context");
    LC_REQUIRE(split >= -1 && split <= 64, "lc_ctx_set_xcd_split: -1 (by shape), 0 (whole rows) or 1 .. 64");
    ctx->xcd_split = split;
    return LC_OK;
}

extern "C" int lc_ctx_get_f64_fidelity(const lc_ctx *ctx, int *mode_out) {
    LC_REQUIRE(ctx && mode_out, "lc_ctx_get_f64_fidelity: null pointer");
    *mode_out = ctx->f64_fidelity;
    return LC_OK;
}

// lc_advect with the raw planes as the order-1 source where a kernel reads them (lc_advect_ex): the host routes then
// neither allocate nor pack the order-1 image
static bool host_route_needs_lin(int dtype, int interp_order) { return dtype == LC_F32 && interp_order == 1; }
static int advect_with_raw(lc_ctx *ctx, const void *lin, const void *cub, const void *ext, const void *u, const void *v, int dtype,
                           int nt, int ny_f, int nx_f, double lat_min, double lat_max, double lon_min, double lon_max,
                           const void *slat, int ny, const void *slon, int nx, double timestep, int K, int order, int cyclic_x,
                           int t0, int nsteps, void *x, void *y, void *tx, void *ty) {
    lc_advect_args a = {};
    a.struct_size = sizeof(a);
    a.packed_lin = lin;
    a.packed_cub = cub;
    a.packed_ext = ext;
    a.u_raw = u;
    a.v_raw = v;
    a.dtype = dtype;
    a.nt = nt;
    a.ny_f = ny_f;
    a.nx_f = nx_f;
    a.lat_min = lat_min;
    a.lat_max = lat_max;
    a.lon_min = lon_min;
    a.lon_max = lon_max;
    a.seed_lat_dev = slat;
    a.ny = ny;
    a.seed_lon_dev = slon;
    a.nx = nx;
    a.row0 = 0;
    a.ny_global = ny;
    a.timestep = timestep;
    a.settls_order = K;
    a.interp_order = order;
    a.cyclic_x = cyclic_x;
    a.t0 = t0;
    a.nsteps = nsteps;
    a.n_members = 1;
    a.t0_stride = 0;
    a.x_out = x;
    a.y_out = y;
    a.traj_x = tx;
    a.traj_y = ty;
    return lc_advect_ex(ctx, &a);
}

// float64 on the one-call host routes: the reference's operation order (no fused-level image) or the fast form
static bool f64_exact_order(const lc_ctx *ctx, int dtype, int ny, int nx) {
    if (dtype != LC_F64) return false;
    return ctx->f64_fidelity == LC_F64_EXACT_ORDER ||
           (ctx->f64_fidelity == LC_F64_AUTO && (long long)ny * nx <= LC_EXACT_ORDER_MAX_SEEDS);
}

extern "C" int lc_ctx_get_level_chunk(const lc_ctx *ctx, int *levels_out) {
    LC_REQUIRE(ctx && levels_out, "lc_ctx_get_level_chunk: null pointer");
    *levels_out = ctx->level_chunk;
    return LC_OK;
}

extern "C" int lc_ctx_set_level_chunk(lc_ctx *ctx, int levels) {
    LC_REQUIRE(ctx, "lc_ctx_set_level_chunk: null context");
    LC_REQUIRE(levels >= -1, "lc_ctx_set_level_chunk: levels must be >= -1 (0 = one launch for the whole series, -1 = by size)");
    ctx->level_chunk = levels;
    return LC_OK;
}

// Wave-state audit of the one-seed order-1 LDS kernel (advect.hip, VERIFY instances): 16 counters in device memory.
extern "C" int lc_ctx_set_verify(lc_ctx *ctx, int mode) {
    LC_REQUIRE(ctx, "lc_ctx_set_verify: null context");
    LC_REQUIRE(mode >= 0 && mode <= 2, "lc_ctx_set_verify: mode must be 0 (off), 1 (audit) or 2 (audit + one injected corruption)");
    LC_HIP_CHECK(hipSetDevice(ctx->device));
    LC_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    if (mode == 0) {
        if (ctx->verify_dev) LC_HIP_CHECK(hipFree(ctx->verify_dev));
        ctx->verify_dev = nullptr;
        return LC_OK;
    }
    if (!ctx->verify_dev) {
        hipError_t e = hipMalloc((void **)&ctx->verify_dev, LC_VERIFY_WORDS * sizeof(unsigned));
        if (e != hipSuccess) {
            ctx->verify_dev = nullptr;
            (void)hipGetLastError();
            lc_set_error("lc_ctx_set_verify: hipMalloc failed: %s", hipGetErrorString(e));
            return e == hipErrorOutOfMemory ? LC_ENOMEM : LC_EHIP;
        }
    }
    unsigned init[LC_VERIFY_WORDS] = {};
    init[15] = mode == 2 ? 0xBADu : 0u;
    LC_HIP_CHECK(hipMemcpy(ctx->verify_dev, init, sizeof(init), hipMemcpyHostToDevice));
    return LC_OK;
}

extern "C" int lc_ctx_read_verify(lc_ctx *ctx, unsigned *out16, int reset) {
    LC_REQUIRE(ctx && out16, "lc_ctx_read_verify: null pointer");
    LC_REQUIRE(ctx->verify_dev, "lc_ctx_read_verify: the audit is off (lc_ctx_set_verify)");
    LC_HIP_CHECK(hipSetDevice(ctx->device));
    LC_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    LC_HIP_CHECK(hipMemcpy(out16, ctx->verify_dev, LC_VERIFY_WORDS * sizeof(unsigned), hipMemcpyDeviceToHost));
    if (reset) {
        unsigned init[LC_VERIFY_WORDS] = {};
        init[15] = out16[15];
        LC_HIP_CHECK(hipMemcpy(ctx->verify_dev, init, sizeof(init), hipMemcpyHostToDevice));
    }
    return LC_OK;
}

extern "C" const char *lc_ctx_last_advect_kernel(const lc_ctx *ctx) { return ctx ? ctx->last_advect_kernel : ""; }
extern "C" int lc_ctx_last_advect_launches(const lc_ctx *ctx) { return ctx ? ctx->last_advect_launches : 0; }
extern "C" const char *lc_ctx_last_sigma_kernel(const lc_ctx *ctx) { return ctx ? ctx->last_sigma_kernel : ""; }
extern "C" const char *lc_ctx_last_pack_kernel(const lc_ctx *ctx) { return ctx ? ctx->last_pack_kernel : ""; }

static void host_ws_destroy(lc_ctx *ctx);  // (below, with the one-call host route)

extern "C" int lc_ctx_destroy(lc_ctx *ctx) {
    if (!ctx) return LC_OK;
    (void)hipSetDevice(ctx->device);
    (void)hipStreamSynchronize(ctx->stream);
    lc_trunc_cache_free(ctx->trunc);
    lc_host_xfer::release(ctx->xfer);
    host_ws_destroy(ctx);
    if (ctx->verify_dev) (void)hipFree(ctx->verify_dev);
    (void)hipStreamDestroy(ctx->own_stream);
    delete ctx;
    return LC_OK;
}

extern "C" int lc_ctx_set_stream(lc_ctx *ctx, void *hip_stream) {
    LC_REQUIRE(ctx, "lc_ctx_set_stream: null context");
    ctx->stream = (hipStream_t)hip_stream;  // NULL is the legacy default stream, a valid stream
    return LC_OK;
}

extern "C" int lc_ctx_use_own_stream(lc_ctx *ctx) {
    LC_REQUIRE(ctx, "lc_ctx_use_own_stream: null context");
    ctx->stream = ctx->own_stream;
    return LC_OK;
}

extern "C" int lc_sync(lc_ctx *ctx) {
    LC_REQUIRE(ctx, "lc_sync: null context");
    LC_HIP_CHECK(hipSetDevice(ctx->device));
    LC_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return LC_OK;
}

extern "C" int lc_malloc(lc_ctx *ctx, size_t bytes, void **dev_out) {
    LC_REQUIRE(ctx && dev_out, "lc_malloc: null argument");
    *dev_out = nullptr;
    LC_HIP_CHECK(hipSetDevice(ctx->device));
    hipError_t e = hipMalloc(dev_out, bytes ? bytes : 1);
    if (e != hipSuccess) {
        (void)hipGetLastError();  // (consumed with the failure it belongs to: see LC_HIP_CHECK)
        lc_set_error("hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e));
        return e == hipErrorOutOfMemory ? LC_ENOMEM : LC_EHIP;
    }
    return LC_OK;
}

extern "C" int lc_free(lc_ctx *ctx, void *dev) {
    LC_REQUIRE(ctx, "lc_free: null context");
    if (!dev) return LC_OK;
    LC_HIP_CHECK(hipSetDevice(ctx->device));
    LC_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    LC_HIP_CHECK(hipFree(dev));
    return LC_OK;
}

extern "C" int lc_memcpy_h2d(lc_ctx *ctx, void *dev_dst, const void *host_src, size_t bytes) {
    LC_REQUIRE(ctx && (bytes == 0 || (dev_dst && host_src)), "lc_memcpy_h2d: null argument");
    LC_HIP_CHECK(hipSetDevice(ctx->device));
    LC_HIP_CHECK(hipMemcpyAsync(dev_dst, host_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    LC_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return LC_OK;
}

extern "C" int lc_memcpy_d2h(lc_ctx *ctx, void *host_dst, const void *dev_src, size_t bytes) {
    LC_REQUIRE(ctx && (bytes == 0 || (host_dst && dev_src)), "lc_memcpy_d2h: null argument");
    LC_HIP_CHECK(hipSetDevice(ctx->device));
    LC_HIP_CHECK(hipMemcpyAsync(host_dst, dev_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    LC_HIP_CHECK(hipStreamSynchronize(ctx->stream));
    return LC_OK;
}

extern "C" size_t lc_packed_elems(int nt, int ny_f, int nx_f) {
    if (nt < 1 || ny_f < 1 || nx_f < 1) return 0;
    return (size_t)nt * lc_level_elems(ny_f, nx_f);
}

extern "C" int lc_field_pack(lc_ctx *ctx, const void *u_dev, const void *v_dev, int dtype, int nt, int ny_f, int nx_f,
                             int interp_order, void *packed_dev, void *ext_dev) {
    LC_REQUIRE(ctx, "lc_field_pack: null context");
    LC_REQUIRE(dtype == LC_F32 || dtype == LC_F64 || dtype == LC_F64_WIND_F32, "lc_field_pack: bad dtype %d", dtype);
    LC_REQUIRE(dtype != LC_F64_WIND_F32 || (interp_order >= 2 && packed_dev && !ext_dev),
               "lc_field_pack: LC_F64_WIND_F32 (float32 planes in, float64 spline coefficients out) is for interp_order 2..5 without ext_dev; "
               "at order 1 pack the float32 wind as LC_F32");
    LC_REQUIRE(u_dev && v_dev, "lc_field_pack: null pointer");
    LC_REQUIRE(packed_dev || (interp_order == 1 && ext_dev && nt >= 2),
               "lc_field_pack: packed_dev may only be NULL at interp_order 1 with ext_dev set (fused-level image alone)");
    LC_REQUIRE(nt >= 1 && ny_f >= 4 && nx_f >= 4, "lc_field_pack: field too small (nt=%d ny_f=%d nx_f=%d)", nt, ny_f,
               nx_f);
    if (interp_order < 1 || interp_order > 5) {
        lc_set_error("lc_field_pack: interp_order %d unsupported (scipy's spline orders 1..5)", interp_order);
        return LC_EUNSUPPORTED;
    }
    LC_HIP_CHECK(hipSetDevice(ctx->device));
    LC_REQUIRE(ext_dev != packed_dev, "lc_field_pack: ext_dev must not alias packed_dev");
    return lc_launch_pack(ctx, u_dev, v_dev, dtype, nt, ny_f, nx_f, interp_order, packed_dev, ext_dev);
}

extern "C" int lc_field_extrapolate(lc_ctx *ctx, const void *packed_dev, int dtype, int nt, int ny_f, int nx_f,
                                    void *ext_dev) {
    LC_REQUIRE(ctx, "lc_field_extrapolate: null context");
    LC_REQUIRE(dtype == LC_F32 || dtype == LC_F64, "lc_field_extrapolate: bad dtype %d", dtype);
    LC_REQUIRE(packed_dev && ext_dev && packed_dev != ext_dev, "lc_field_extrapolate: bad pointers");
    LC_REQUIRE(nt >= 2 && ny_f >= 4 && nx_f >= 4, "lc_field_extrapolate: field too small");
    LC_HIP_CHECK(hipSetDevice(ctx->device));
    return lc_launch_extrapolate(ctx, packed_dev, dtype, nt, ny_f, nx_f, ext_dev);
}

// ---------------------------------------------------------------------------
// Gaussian smoothing (LCS/LCS.py:187-190 -> scipy.ndimage.gaussian_filter defaults:
// truncate=4.0, mode='reflect', axis 0 then axis 1, output dtype = input dtype,
// accumulation in double, symmetric-kernel summation order of ni_filters.c).
// ---------------------------------------------------------------------------
namespace {

constexpr int GAUSS_MAX_RADIUS = 256;

struct GaussW {
    double w[GAUSS_MAX_RADIUS + 1];  // w[0] centre ... w[radius]
    int radius;
};

__device__ __forceinline__ int reflect_index(int i, int n) {
    // scipy 'reflect' (d c b a | a b c d | d c b a), any distance
    if (n == 1) return 0;
    const int period = 2 * n;
    i %= period;
    if (i < 0) i += period;
    return i < n ? i : period - 1 - i;
}

template <typename T, int AXIS>
__global__ void gauss_kernel(const T *__restrict__ in, T *__restrict__ out, int ny, int nx, const GaussW G) {
    const size_t total = (size_t)ny * nx;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int y = (int)(i / nx), x = (int)(i - (size_t)y * nx);
        double acc = (double)in[i] * G.w[0];
        for (int j = G.radius; j >= 1; --j) {  // outermost pair first, as correlate1d does
            double lo, hi;
            if (AXIS == 0) {
                lo = (double)in[(size_t)reflect_index(y - j, ny) * nx + x];
                hi = (double)in[(size_t)reflect_index(y + j, ny) * nx + x];
            } else {
                lo = (double)in[(size_t)y * nx + reflect_index(x - j, nx)];
                hi = (double)in[(size_t)y * nx + reflect_index(x + j, nx)];
            }
            acc += (lo + hi) * G.w[j];
        }
        out[i] = (T)acc;
    }
}

template <typename T>
int gauss_impl(lc_ctx *ctx, const T *in, int ny, int nx, const GaussW &G, T *tmp, T *out) {
    const size_t total = (size_t)ny * nx;
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL((gauss_kernel<T, 0>), dim3(blocks), dim3(256), 0, ctx->stream, in, tmp, ny, nx, G);
    hipLaunchKernelGGL((gauss_kernel<T, 1>), dim3(blocks), dim3(256), 0, ctx->stream, (const T *)tmp, out, ny, nx, G);
    LC_HIP_CHECK(hipGetLastError());
    return LC_OK;
}

}  // namespace

extern "C" int lc_gaussian_filter(lc_ctx *ctx, const void *in_dev, int dtype, int ny, int nx, double sigma,
                                  void *tmp_dev, void *out_dev) {
    LC_REQUIRE(ctx, "lc_gaussian_filter: null context");
    LC_REQUIRE(dtype == LC_F32 || dtype == LC_F64, "lc_gaussian_filter: bad dtype %d", dtype);
    LC_REQUIRE(in_dev && tmp_dev && out_dev && in_dev != out_dev && in_dev != tmp_dev && tmp_dev != out_dev,
               "lc_gaussian_filter: in, tmp and out must be three distinct buffers");
    LC_REQUIRE(ny >= 1 && nx >= 1 && sigma > 0, "lc_gaussian_filter: bad size or sigma");
    GaussW G;
    G.radius = (int)(4.0 * sigma + 0.5);  // scipy: int(truncate * sd + 0.5)
    if (G.radius > GAUSS_MAX_RADIUS) {
        lc_set_error("lc_gaussian_filter: sigma %g needs radius %d > %d", sigma, G.radius, GAUSS_MAX_RADIUS);
        return LC_EUNSUPPORTED;
    }
    // scipy _gaussian_kernel1d: phi = exp(-0.5/sigma^2 * x^2); phi /= phi.sum()
    double sum = 0.0;
    std::vector<double> phi(2 * G.radius + 1);
    for (int k = -G.radius; k <= G.radius; ++k) phi[k + G.radius] = std::exp(-0.5 / (sigma * sigma) * (double)k * k);
    for (double p : phi) sum += p;
    for (int k = 0; k <= G.radius; ++k) G.w[k] = phi[k + G.radius] / sum;
    LC_HIP_CHECK(hipSetDevice(ctx->device));
    if (dtype == LC_F32) return gauss_impl<float>(ctx, (const float *)in_dev, ny, nx, G, (float *)tmp_dev, (float *)out_dev);
    return gauss_impl<double>(ctx, (const double *)in_dev, ny, nx, G, (double *)tmp_dev, (double *)out_dev);
}

// ---------------------------------------------------------------------------
// One-call host entry point (LCS/LCS.py:129-157 on host arrays).
// ---------------------------------------------------------------------------
namespace {
// The one-call host routes' device buffers.  hipMalloc + hipFree of configs[2]'s 2.6 GB cost 1.6 ms of a 21 ms call, every call:
// a context keeps the buffers of its last call and hands them to the next one that fits them (same shapes: all of them);
// lc_ctx_trim / lc_ctx_destroy free them.  No buffer is ever shared between two live DevBufs.
struct HostWorkspace {
    struct Slot {
        void *p;
        size_t bytes;
    };
    std::vector<Slot> free_list;
    static constexpr size_t MAX_KEPT = 24;
    void *take(size_t bytes) {  // the smallest kept buffer that holds `bytes` without being more than half as large again
        size_t best = free_list.size();
        for (size_t i = 0; i < free_list.size(); ++i)
            if (free_list[i].bytes >= bytes && free_list[i].bytes <= bytes + bytes / 2 + (1u << 20) &&
                (best == free_list.size() || free_list[i].bytes < free_list[best].bytes))
                best = i;
        if (best == free_list.size()) return nullptr;
        void *p = free_list[best].p;
        free_list[best] = free_list.back();
        free_list.pop_back();
        return p;
    }
    bool give(void *p, size_t bytes) {
        if (free_list.size() >= MAX_KEPT) return false;
        try {
            free_list.push_back(Slot{p, bytes});
        } catch (...) {
            return false;
        }
        return true;
    }
    void trim() {
        for (auto &s : free_list) (void)hipFree(s.p);
        free_list.clear();
    }
};

struct DevBuf {
    void *p = nullptr;
    size_t bytes = 0;
    HostWorkspace *ws = nullptr;  // NULL: plain hipMalloc / hipFree (lc_lcs_global_host, a context with the cache off)
    ~DevBuf() {
        if (!p) return;
        if (!(ws && ws->give(p, bytes))) (void)hipFree(p);
    }
    int alloc(size_t n) {
        bytes = n ? n : 1;
        if (ws && (p = ws->take(bytes)) != nullptr) return LC_OK;
        hipError_t e = hipMalloc(&p, bytes);
        if (e != hipSuccess && ws && !ws->free_list.empty()) {  // out of memory with buffers kept for later: let them go, once
            (void)hipGetLastError();
            ws->trim();
            e = hipMalloc(&p, bytes);
        }
        if (e != hipSuccess) {
            lc_set_error("hipMalloc(%zu) failed: %s", n, hipGetErrorString(e));
            (void)hipGetLastError();
            p = nullptr;
            return e == hipErrorOutOfMemory ? LC_ENOMEM : LC_EHIP;
        }
        return LC_OK;
    }
};

}  // namespace

static void host_ws_destroy(lc_ctx *ctx) {
    if (!ctx->host_ws) return;
    ((HostWorkspace *)ctx->host_ws)->trim();
    delete (HostWorkspace *)ctx->host_ws;
    ctx->host_ws = nullptr;
}

extern "C" int lc_ctx_trim(lc_ctx *ctx) {
    LC_REQUIRE(ctx, "lc_ctx_trim: null context");
    LC_HIP_CHECK(hipSetDevice(ctx->device));
    if (ctx->host_ws) ((HostWorkspace *)ctx->host_ws)->trim();
    return LC_OK;
}

// The context's staging ring (the device's one ring, shared: hostxfer.h), taken on first use; NULL (and nothing left in HIP's sticky error) when it cannot be had.
static lc_host_xfer *host_xfer_of(lc_ctx *ctx) {
    if (!ctx->host_pipeline) return nullptr;
    if (!ctx->xfer) {
        hipError_t e = hipSuccess;
        ctx->xfer = lc_host_xfer::acquire(ctx->device, &e, ctx->host_threads, (size_t)ctx->host_piece_mb << 20);
        if (!ctx->xfer) (void)hipGetLastError();
    }
    return ctx->xfer;
}

// Staged copies between a caller's pageable host array and device memory, ordered with the context's stream: what the Python
// host (Engine.to_device, the drop-in's results) moves large arrays with -- hipMemcpy from pages the runtime has not pinned
// before runs at a quarter of the bus rate (hostxfer.h).  to_device: work enqueued on the context's stream after the call sees
// the data.  to_host: sees everything enqueued on the context's stream before the call; returns when the bytes are in `host`.
extern "C" int lc_copy_to_device(lc_ctx *ctx, void *dev, const void *host, size_t bytes) {
    LC_REQUIRE(ctx && (bytes == 0 || (dev && host)), "lc_copy_to_device: null pointer");
    if (!bytes) return LC_OK;
    LC_HIP_CHECK(hipSetDevice(ctx->device));
    lc_host_xfer *hx = host_xfer_of(ctx);
    if (!hx) {
        LC_HIP_CHECK(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, ctx->stream));
        LC_HIP_CHECK(hipStreamSynchronize(ctx->stream));   // (a pageable source may be reused by the caller at once)
        return LC_OK;
    }
    std::lock_guard<std::mutex> ring(hx->use);
    hipEvent_t ev = nullptr;
    LC_HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    struct Ev {
        hipEvent_t e;
        ~Ev() { (void)hipEventDestroy(e); }
    } guard{ev};
    LC_HIP_CHECK(hipEventRecord(ev, ctx->stream));            // the destination may still be in use by earlier work of the stream
    LC_HIP_CHECK(hipStreamWaitEvent(hx->copy, ev, 0));
    hipError_t e = hx->upload(dev, host, bytes);
    if (e != hipSuccess) {
        hx->drain();
        lc_set_error("lc_copy_to_device: staged upload failed: %s", hipGetErrorString(e));
        (void)hipGetLastError();
        return LC_EHIP;
    }
    LC_HIP_CHECK(hipEventRecord(ev, hx->copy));
    LC_HIP_CHECK(hipStreamWaitEvent(ctx->stream, ev, 0));
    return LC_OK;
}

extern "C" int lc_copy_to_host(lc_ctx *ctx, void *host, const void *dev, size_t bytes) {
    LC_REQUIRE(ctx && (bytes == 0 || (dev && host)), "lc_copy_to_host: null pointer");
    if (!bytes) return LC_OK;
    LC_HIP_CHECK(hipSetDevice(ctx->device));
    lc_host_xfer *hx = host_xfer_of(ctx);
    if (!hx) {
        LC_HIP_CHECK(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, ctx->stream));
        LC_HIP_CHECK(hipStreamSynchronize(ctx->stream));
        return LC_OK;
    }
    std::lock_guard<std::mutex> ring(hx->use);
    hipEvent_t ev = nullptr;
    LC_HIP_CHECK(hipEventCreateWithFlags(&ev, hipEventDisableTiming));
    struct Ev {
        hipEvent_t e;
        ~Ev() { (void)hipEventDestroy(e); }
    } guard{ev};
    LC_HIP_CHECK(hipEventRecord(ev, ctx->stream));
    LC_HIP_CHECK(hipStreamWaitEvent(hx->copy, ev, 0));
    hipError_t e = hx->download(host, dev, bytes);
    if (e != hipSuccess) {
        hx->drain();
        lc_set_error("lc_copy_to_host: staged download failed: %s", hipGetErrorString(e));
        (void)hipGetLastError();
        return LC_EHIP;
    }
    return LC_OK;
}

extern "C" int lc_ctx_set_host_cache(lc_ctx *ctx, int on) {
    LC_REQUIRE(ctx, "lc_ctx_set_host_cache: null context");
    LC_REQUIRE(on == 0 || on == 1, "lc_ctx_set_host_cache: 0 or 1");
    ctx->host_cache = on;
    if (!on) return lc_ctx_trim(ctx);
    return LC_OK;
}

namespace {
template <typename T>
void coord_extremes(const void *lat, int n, double *lo, double *hi) {
    const T *p = (const T *)lat;
    *lo = (double)p[0];
    *hi = (double)p[n - 1];
}
}  // namespace

#define LC_TRY(expr)              \
    do {                          \
        int _s = (expr);          \
        if (_s != LC_OK) return _s; \
    } while (0)

extern "C" int lc_lcs_host(lc_ctx *ctx, const void *u_host, const void *v_host, int dtype, int nt, int ny_f, int nx_f,
                           const void *lat_f_host, const void *lon_f_host, const void *seed_lat_host, int ny,
                           const void *seed_lon_host, int nx, double timestep, int settls_order, int interp_order,
                           int cyclic_x, int t0, int nsteps, double gauss_sigma, int fd_fp32_cast, int tensor_layout,
                           void *sigma_out, void *x_out, void *y_out, void *traj_x, void *traj_y) {
    LC_REQUIRE(ctx, "lc_lcs_host: null context");
    LC_REQUIRE(dtype == LC_F32 || dtype == LC_F64, "lc_lcs_host: bad dtype %d", dtype);
    LC_REQUIRE(u_host && v_host && lat_f_host && lon_f_host && seed_lat_host && seed_lon_host,
               "lc_lcs_host: null input pointer");
    LC_REQUIRE(nt >= 2 && ny_f >= 4 && nx_f >= 4 && ny >= 1 && nx >= 1, "lc_lcs_host: bad sizes");
    LC_REQUIRE((traj_x == nullptr) == (traj_y == nullptr), "lc_lcs_host: traj_x/traj_y must be set together");
    LC_REQUIRE(!sigma_out || (ny >= 5 && nx >= 5), "lc_lcs_host: sigma needs at least a 5x5 seed grid");
    // validated before any allocation (a negative nsteps must not turn into a huge hipMalloc)
    LC_REQUIRE(settls_order >= 0, "lc_lcs_host: SETTLS_order must be >= 0");
    LC_REQUIRE(t0 >= 0 && nsteps >= 0 && t0 + nsteps <= nt - 1, "lc_lcs_host: steps [%d,%d) need levels up to %d, have %d",
               t0, t0 + nsteps, t0 + nsteps, nt);
    LC_REQUIRE(gauss_sigma >= 0.0 || gauss_sigma != gauss_sigma, "lc_lcs_host: gauss_sigma must be >= 0 (0 = no smoothing)");
    if (interp_order < 1 || interp_order > 5) {
        lc_set_error("lc_lcs_host: interp_order %d unsupported (scipy's spline orders 1..5)", interp_order);
        return LC_EUNSUPPORTED;
    }
    LC_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t es = dtype == LC_F32 ? 4 : 8;
    const size_t fbytes = (size_t)nt * ny_f * nx_f * es;
    const size_t pbytes = lc_packed_elems(nt, ny_f, nx_f) * es;
    const size_t sbytes = (size_t)ny * nx * es;
    double lat_min, lat_max, lon_min, lon_max, s_lat0, s_lat1, s_lon0, s_lon1;
    if (dtype == LC_F32) {
        coord_extremes<float>(lat_f_host, ny_f, &lat_min, &lat_max);
        coord_extremes<float>(lon_f_host, nx_f, &lon_min, &lon_max);
        s_lat0 = ((const float *)seed_lat_host)[0];
        s_lat1 = ((const float *)seed_lat_host)[ny > 1 ? 1 : 0];
        s_lon0 = ((const float *)seed_lon_host)[0];
        s_lon1 = ((const float *)seed_lon_host)[nx > 1 ? 1 : 0];
    } else {
        coord_extremes<double>(lat_f_host, ny_f, &lat_min, &lat_max);
        coord_extremes<double>(lon_f_host, nx_f, &lon_min, &lon_max);
        s_lat0 = ((const double *)seed_lat_host)[0];
        s_lat1 = ((const double *)seed_lat_host)[ny > 1 ? 1 : 0];
        s_lon0 = ((const double *)seed_lon_host)[0];
        s_lon1 = ((const double *)seed_lon_host)[nx > 1 ? 1 : 0];
    }
    // spacing in the coordinate dtype, as lat[1]-lat[0] evaluates in numpy (tools.py:255-256)
    const double dlat = dtype == LC_F32 ? (double)((float)s_lat1 - (float)s_lat0) : s_lat1 - s_lat0;
    const double dlon = dtype == LC_F32 ? (double)((float)s_lon1 - (float)s_lon0) : s_lon1 - s_lon0;

    // LCS_HOST_TIMING: wall-clock marks of this call (ms since entry) -- allocations done, last upload piece handed to the DMA
    // engine, kernels finished, results in the caller's buffers
    const auto t_enter = std::chrono::steady_clock::now();
    auto ms_since = [&] { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_enter).count(); };
    double t_alloc = 0, t_up = 0, t_kernels = 0;
    struct ExitMark {  // (declared before the buffers: runs after they have been freed)
        const std::chrono::steady_clock::time_point t0;
        bool on;
        ~ExitMark() {
            if (on) std::fprintf(stderr, "lc_lcs_host: device buffers freed, returning at %.2f ms\n",
                                 std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count());
        }
    } exit_mark{t_enter, ctx->host_timing != 0};
    DevBuf u, v, lin, cub, ext, slat, slon, x, y, tx, ty, sig, gx, gy, gtmp;
    if (ctx->host_cache) {
        if (!ctx->host_ws) ctx->host_ws = new (std::nothrow) HostWorkspace;
        for (DevBuf *b : {&u, &v, &lin, &cub, &ext, &slat, &slon, &x, &y, &tx, &ty, &sig, &gx, &gy, &gtmp}) b->ws = (HostWorkspace *)ctx->host_ws;
    }
    LC_TRY(u.alloc(fbytes));
    LC_TRY(v.alloc(fbytes));
    const bool need_lin = host_route_needs_lin(dtype, interp_order);
    if (need_lin) LC_TRY(lin.alloc(pbytes));
    if (interp_order != 1) LC_TRY(cub.alloc(pbytes));
    LC_TRY(slat.alloc(ny * es));
    LC_TRY(slon.alloc(nx * es));
    LC_TRY(x.alloc(sbytes));
    LC_TRY(y.alloc(sbytes));
    if (traj_x) {
        LC_TRY(tx.alloc(sbytes * (size_t)(nsteps + 1)));
        LC_TRY(ty.alloc(sbytes * (size_t)(nsteps + 1)));
    }
    hipStream_t st = ctx->stream;
    // one combined sample per SETTLS iteration (ext image of the matching order) in float32, and in float64 beyond the
    // size / setting where the reference's own operation order is kept (lc_ctx_set_f64_fidelity)
    const bool fusable = (interp_order == 1 || interp_order == 3) && !f64_exact_order(ctx, dtype, ny, nx);
    if (settls_order > 0 && fusable) LC_TRY(ext.alloc(lc_packed_elems(nt - 1, ny_f, nx_f) * es));
    if (sigma_out) LC_TRY(sig.alloc(sbytes));

    // ---- transfers: the staging ring (hostxfer.h), or plain hipMemcpyAsync when it cannot be had / is switched off --------
    lc_host_xfer *hx = host_xfer_of(ctx);  // (NULL: switched off, or no pinned memory / stream to be had -- the plain copies below)
    // the ring is the device's, shared by every context: this call's until it returns (declared before `drain`: released last)
    std::unique_lock<std::mutex> ring;
    if (hx) ring = std::unique_lock<std::mutex>(hx->use);
    // whatever happens below, no DMA of this call is left running into / out of buffers that are about to be freed
    struct Drain {
        lc_host_xfer *hx;
        hipStream_t st;
        ~Drain() {
            if (hx) hx->drain();
            (void)hipStreamSynchronize(st);
        }
    } drain{hx, st};
    lc_prefault touch;  // (declared after `drain`: joined first)
    hipEvent_t ev_ready = nullptr, ev_up = nullptr, ev_done = nullptr;
    struct Events {
        hipEvent_t *e[3];
        ~Events() {
            for (auto p : e)
                if (*p) (void)hipEventDestroy(*p);
        }
    } events{{&ev_ready, &ev_up, &ev_done}};
    auto up = [&](void *dev, const void *host, size_t bytes) -> int {
        if (!bytes) return LC_OK;
        if (hx) {
            LC_HIP_CHECK(hx->upload(dev, host, bytes));
        } else {
            LC_HIP_CHECK(hipMemcpyAsync(dev, host, bytes, hipMemcpyHostToDevice, st));
        }
        return LC_OK;
    };
    auto down = [&](void *host, const void *dev, size_t bytes) -> int {
        if (hx) {
            LC_HIP_CHECK(hx->download(host, dev, bytes));
        } else {
            LC_HIP_CHECK(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, st));
        }
        return LC_OK;
    };
    if (hx) {
        LC_HIP_CHECK(hipEventCreateWithFlags(&ev_ready, hipEventDisableTiming));
        LC_HIP_CHECK(hipEventCreateWithFlags(&ev_up, hipEventDisableTiming));
        LC_HIP_CHECK(hipEventCreateWithFlags(&ev_done, hipEventDisableTiming));
        // the copy stream starts after whatever the caller's stream was doing with this memory (nothing, normally)
        LC_HIP_CHECK(hipEventRecord(ev_ready, st));
        LC_HIP_CHECK(hipStreamWaitEvent(hx->copy, ev_ready, 0));
        std::vector<std::pair<void *, size_t>> outs;
        if (sigma_out) outs.emplace_back(sigma_out, sbytes);
        if (x_out) outs.emplace_back(x_out, sbytes);
        if (y_out) outs.emplace_back(y_out, sbytes);
        if (traj_x) {
            outs.emplace_back(traj_x, sbytes * (size_t)(nsteps + 1));
            outs.emplace_back(traj_y, sbytes * (size_t)(nsteps + 1));
        }
        touch.start(outs);
    }
    t_alloc = ms_since();
    LC_TRY(up(slat.p, seed_lat_host, ny * es));
    LC_TRY(up(slon.p, seed_lon_host, nx * es));

    // ---- upload || pack + advect, level chunk by level chunk ------------------------------------------------------------------
    // Chunk c covers the steps [s0, s1): it needs the wind levels t0 + s0 ... t0 + s1 (the first of them arrived with chunk
    // c - 1), packs the images of exactly those levels (lc_field_pack on the sub-range: the level two chunks share is packed by
    // both, same values) and continues the advection in place (lc_advect_from: LCS/trajectory.py:80-126 carries only the
    // positions from level to level) -- bit-identical to the serial form.  While its kernels run on the context's stream, the
    // host threads and the DMA engine are already moving chunk c + 1.  Only the levels the call uses travel.
    // Serial form (one upload of the used levels, one pack, one advect): the reference's non-cyclic outer-product clamp (it
    // restarts the series), trajectories, the exact-order float64 form, orders other than 1 / 3, short series.
    const size_t lvl_bytes = (size_t)ny_f * nx_f * es, le = lc_packed_elems(1, ny_f, nx_f);
    // Chunk lengths HALVE towards the end of the series (96 steps: 32, 24, 16, 12, 8, 4): what is left to compute when the last
    // upload lands is the last chunk alone, so it is the shortest; a chunk's kernels (0.07 ms per level on configs[2]) are done
    // well before the next, at least half as long, has travelled (0.14 ms per level).
    const int PIPE_LEVELS = 16, PIPE_LAST = 4;
    const bool piped = hx && cyclic_x == LC_X_CYCLIC && ext.p && !traj_x && nsteps >= 2 * PIPE_LEVELS;
    auto chunk_len = [&](int left) { return left <= PIPE_LAST ? left : std::max(PIPE_LAST, (left / 3 + 3) / 4 * 4); };
    auto pack_levels = [&](int l0, int nlev) -> int {  // images of wind levels [l0, l0 + nlev), ext of [l0, l0 + nlev - 1)
        const char *ul = (const char *)u.p + (size_t)l0 * lvl_bytes, *vl = (const char *)v.p + (size_t)l0 * lvl_bytes;
        void *el = ext.p ? (char *)ext.p + (size_t)l0 * le * es : nullptr;
        if (need_lin || (interp_order == 1 && ext.p))
            LC_TRY(lc_field_pack(ctx, ul, vl, dtype, nlev, ny_f, nx_f, 1, lin.p ? (char *)lin.p + (size_t)l0 * le * es : nullptr,
                                 interp_order == 1 ? el : nullptr));
        if (interp_order != 1)
            LC_TRY(lc_field_pack(ctx, ul, vl, dtype, nlev, ny_f, nx_f, interp_order, (char *)cub.p + (size_t)l0 * le * es, el));
        return LC_OK;
    };
    if (piped) {
        for (int s0 = 0, s1; s0 < nsteps; s0 = s1) {
            s1 = s0 + chunk_len(nsteps - s0);
            const int first = t0 + s0 + (s0 ? 1 : 0), last = t0 + s1;
            LC_TRY(up((char *)u.p + (size_t)first * lvl_bytes, (const char *)u_host + (size_t)first * lvl_bytes, (size_t)(last - first + 1) * lvl_bytes));
            LC_TRY(up((char *)v.p + (size_t)first * lvl_bytes, (const char *)v_host + (size_t)first * lvl_bytes, (size_t)(last - first + 1) * lvl_bytes));
            LC_HIP_CHECK(hipEventRecord(ev_up, hx->copy));
            LC_HIP_CHECK(hipStreamWaitEvent(st, ev_up, 0));
            LC_TRY(pack_levels(t0 + s0, s1 - s0 + 1));
            lc_advect_args a = {};
            a.struct_size = sizeof(a);
            a.packed_lin = lin.p;
            a.packed_cub = cub.p;
            a.packed_ext = ext.p;
            a.u_raw = u.p;
            a.v_raw = v.p;
            a.dtype = dtype;
            a.nt = nt;
            a.ny_f = ny_f;
            a.nx_f = nx_f;
            a.lat_min = lat_min;
            a.lat_max = lat_max;
            a.lon_min = lon_min;
            a.lon_max = lon_max;
            a.seed_lat_dev = slat.p;
            a.ny = ny;
            a.seed_lon_dev = slon.p;
            a.nx = nx;
            a.row0 = 0;
            a.ny_global = ny;
            a.x_start = s0 ? x.p : nullptr;
            a.y_start = s0 ? y.p : nullptr;
            a.timestep = timestep;
            a.settls_order = settls_order;
            a.interp_order = interp_order;
            a.cyclic_x = cyclic_x;
            a.t0 = t0 + s0;
            a.nsteps = s1 - s0;
            a.n_members = 1;
            a.x_out = x.p;
            a.y_out = y.p;
            LC_TRY(lc_advect_ex(ctx, &a));
        }
    } else {
        // only the levels [t0, t0 + nsteps] are read (the pack of the whole series below touches the others' device memory:
        // they travel too unless the call uses a sub-range, in which case the images of the used levels alone are packed)
        const bool sub = hx && nsteps >= 1 && (t0 > 0 || t0 + nsteps < nt - 1) && cyclic_x != LC_X_CLAMP_REFERENCE_OUTER;
        const int l0 = sub ? t0 : 0, nlev = sub ? nsteps + 1 : nt;
        LC_TRY(up((char *)u.p + (size_t)l0 * lvl_bytes, (const char *)u_host + (size_t)l0 * lvl_bytes, (size_t)nlev * lvl_bytes));
        LC_TRY(up((char *)v.p + (size_t)l0 * lvl_bytes, (const char *)v_host + (size_t)l0 * lvl_bytes, (size_t)nlev * lvl_bytes));
        if (hx) {
            LC_HIP_CHECK(hipEventRecord(ev_up, hx->copy));
            LC_HIP_CHECK(hipStreamWaitEvent(st, ev_up, 0));
        }
        LC_TRY(pack_levels(l0, nlev));
        LC_TRY(advect_with_raw(ctx, lin.p, cub.p, ext.p, u.p, v.p, dtype, nt, ny_f, nx_f, lat_min, lat_max, lon_min, lon_max, slat.p, ny,
                               slon.p, nx, timestep, settls_order, interp_order, cyclic_x, t0, nsteps, x.p, y.p, tx.p, ty.p));
    }
    if (sigma_out) {
        const void *xs = x.p, *ys = y.p;
        if (gauss_sigma > 1e-15) {  // sigma = 0: scipy returns an unsmoothed copy
            LC_TRY(gx.alloc(sbytes));
            LC_TRY(gy.alloc(sbytes));
            LC_TRY(gtmp.alloc(sbytes));
            LC_TRY(lc_gaussian_filter(ctx, x.p, dtype, ny, nx, gauss_sigma, gtmp.p, gx.p));
            LC_TRY(lc_gaussian_filter(ctx, y.p, dtype, ny, nx, gauss_sigma, gtmp.p, gy.p));
            xs = gx.p;
            ys = gy.p;
        }
        LC_TRY(lc_sigma(ctx, xs, ys, dtype, 0, ny, nx, ny, slat.p, dlat, dlon, fd_fp32_cast, tensor_layout, 0, ny,
                        sig.p));
    }
    t_up = ms_since();
    // ---- results down: the departure points travel while nothing else does; sigma (0.07 ms of kernel) last ---------------
    if (hx) {
        LC_HIP_CHECK(hipEventRecord(ev_done, st));
        LC_HIP_CHECK(hipStreamWaitEvent(hx->copy, ev_done, 0));
        touch.join();
        LC_HIP_CHECK(hipEventSynchronize(ev_done));   // (the first download DMA waits for this event anyway: the host has nothing else to do)
        t_kernels = ms_since();
    }
    {
        std::vector<lc_host_xfer::Range> res;
        if (x_out) res.push_back({x_out, x.p, sbytes});
        if (y_out) res.push_back({y_out, y.p, sbytes});
        if (sigma_out) res.push_back({sigma_out, sig.p, sbytes});
        if (traj_x) {
            res.push_back({traj_x, tx.p, sbytes * (size_t)(nsteps + 1)});
            res.push_back({traj_y, ty.p, sbytes * (size_t)(nsteps + 1)});
        }
        if (hx) {
            LC_HIP_CHECK(hx->download(res));   // one pipelined sequence of pieces over all of them
        } else {
            for (auto &r : res) LC_TRY(down(r.host, r.dev, r.bytes));
        }
    }
    LC_HIP_CHECK(hipStreamSynchronize(st));
    ctx->host_marks[0] = t_alloc;
    ctx->host_marks[1] = t_up;
    ctx->host_marks[2] = hx ? t_kernels : 0.0;
    ctx->host_marks[3] = ms_since();
    if (ctx->host_timing)
        std::fprintf(stderr, "lc_lcs_host: %s, buffers allocated %.2f ms, uploads and launches issued %.2f, kernels done %.2f, results down %.2f"
                             " (download: %.2f waiting for DMAs, %.2f copying out of the ring)\n",
                     piped ? "pipelined" : (hx ? "staged" : "plain copies"), t_alloc, t_up, t_kernels, ms_since(),
                     hx ? hx->last_down_wait_ms : 0.0, hx ? hx->last_down_copy_ms : 0.0);
    return LC_OK;  // `drain` waits for both streams, then the DevBuf destructors free
}

// ---------------------------------------------------------------------------
// One-call host entry point for the reference's DEFAULT global call form,
// LCS(...)(ds, isglobal=True) (LCS/LCS.py:105-157; examples/ideal_vortex.py:280-287):
// regrid to the common 0.5 degree grid -> T-truncation -> advect (cyclic) -> sigma.
// ---------------------------------------------------------------------------
extern "C" int lc_common_grid(int *ny_out, int *nx_out, double *lats_out, double *lons_out) {
    // LCS.py:107-108: np.linspace(-89.75, 89.75, 360), np.linspace(-180, 179.5, 721)  (numpy's formula: start + i*step)
    const int ny = 360, nx = 721;
    if (ny_out) *ny_out = ny;
    if (nx_out) *nx_out = nx;
    if (lats_out) {
        const double step = (89.75 - -89.75) / (ny - 1);
        for (int i = 0; i < ny; ++i) lats_out[i] = -89.75 + i * step;
        lats_out[ny - 1] = 89.75;
    }
    if (lons_out) {
        const double step = (179.5 - -180.0) / (nx - 1);
        for (int i = 0; i < nx; ++i) lons_out[i] = -180.0 + i * step;
        lons_out[nx - 1] = 179.5;
    }
    return LC_OK;
}

extern "C" int lc_lcs_global_host(lc_ctx *ctx, const void *u_host, const void *v_host, int dtype, int nt, int ny_f, int nx_f,
                                  const double *lat_f_host, const double *lon_f_host, int interp_to_common_grid,
                                  int truncation, double timestep, int settls_order, int interp_order,
                                  double gauss_sigma, int fd_fp32_cast, int tensor_layout, void *sigma_out, void *x_out,
                                  void *y_out) {
    LC_REQUIRE(ctx, "lc_lcs_global_host: null context");
    LC_REQUIRE(dtype == LC_F32 || dtype == LC_F64, "lc_lcs_global_host: bad dtype %d", dtype);
    LC_REQUIRE(u_host && v_host && lat_f_host && lon_f_host, "lc_lcs_global_host: null input pointer");
    LC_REQUIRE(nt >= 2 && ny_f >= 4 && nx_f >= 4, "lc_lcs_global_host: bad sizes");
    LC_REQUIRE(settls_order >= 0, "lc_lcs_global_host: SETTLS_order must be >= 0");
    if (interp_order < 1 || interp_order > 5) {
        lc_set_error("lc_lcs_global_host: interp_order %d unsupported (scipy's spline orders 1..5)", interp_order);
        return LC_EUNSUPPORTED;
    }
    LC_HIP_CHECK(hipSetDevice(ctx->device));
    hipStream_t st = ctx->stream;
    const size_t es_in = dtype == LC_F32 ? 4 : 8;
    const size_t nin = (size_t)nt * ny_f * nx_f;
    // grid and dtype the wind ends up on: regridding gives float64 (xarray's interp result)
    int ny = ny_f, nx = nx_f, wdtype = dtype;
    std::vector<double> lat(lat_f_host, lat_f_host + ny_f), lon(lon_f_host, lon_f_host + nx_f);
    if (interp_to_common_grid) {
        lc_common_grid(&ny, &nx, nullptr, nullptr);
        lat.resize(ny);
        lon.resize(nx);
        lc_common_grid(nullptr, nullptr, lat.data(), lon.data());
        wdtype = LC_F64;
    }
    int gridtype = LC_GRID_REGULAR;
    if (truncation >= 0) LC_TRY(lc_inspect_gridtype(lat.data(), ny, &gridtype));  // windspharm's grid inspection, LCS.py:116 via VectorWind
    const size_t es = wdtype == LC_F32 ? 4 : 8;
    const size_t nw = (size_t)nt * ny * nx, sbytes = (size_t)ny * nx * es;
    DevBuf uin, vin, ur, vr, ut, vt, lin, cub, ext, slat, slon, x, y, sig, gx, gy, gtmp;
    LC_TRY(uin.alloc(nin * es_in));
    LC_TRY(vin.alloc(nin * es_in));
    LC_HIP_CHECK(hipMemcpyAsync(uin.p, u_host, nin * es_in, hipMemcpyHostToDevice, st));
    LC_HIP_CHECK(hipMemcpyAsync(vin.p, v_host, nin * es_in, hipMemcpyHostToDevice, st));
    void *uw = uin.p, *vw = vin.p;
    if (interp_to_common_grid) {
        LC_TRY(ur.alloc(nw * 8));
        LC_TRY(vr.alloc(nw * 8));
        LC_TRY(lc_regrid_common_grid(ctx, uin.p, dtype, nt, ny_f, nx_f, lat_f_host, lon_f_host, lat.data(), ny, lon.data(), nx,
                                     (double *)ur.p));
        LC_TRY(lc_regrid_common_grid(ctx, vin.p, dtype, nt, ny_f, nx_f, lat_f_host, lon_f_host, lat.data(), ny, lon.data(), nx,
                                     (double *)vr.p));
        uw = ur.p;
        vw = vr.p;
    }
    if (truncation >= 0) {
        LC_TRY(ut.alloc(nw * es));
        LC_TRY(vt.alloc(nw * es));
        LC_TRY(lc_spectral_truncate(ctx, uw, wdtype, nt, ny, nx, truncation, gridtype, ut.p));
        LC_TRY(lc_spectral_truncate(ctx, vw, wdtype, nt, ny, nx, truncation, gridtype, vt.p));
        uw = ut.p;
        vw = vt.p;
    }
    // seeds = the (new) grid nodes, in the wind's dtype (trajectory.py:68-70)
    std::vector<char> hl(ny * es), ho(nx * es);
    for (int i = 0; i < ny; ++i)
        if (wdtype == LC_F32) ((float *)hl.data())[i] = (float)lat[i]; else ((double *)hl.data())[i] = lat[i];
    for (int i = 0; i < nx; ++i)
        if (wdtype == LC_F32) ((float *)ho.data())[i] = (float)lon[i]; else ((double *)ho.data())[i] = lon[i];
    LC_TRY(slat.alloc(ny * es));
    LC_TRY(slon.alloc(nx * es));
    LC_HIP_CHECK(hipMemcpyAsync(slat.p, hl.data(), ny * es, hipMemcpyHostToDevice, st));
    LC_HIP_CHECK(hipMemcpyAsync(slon.p, ho.data(), nx * es, hipMemcpyHostToDevice, st));
    LC_HIP_CHECK(hipStreamSynchronize(st));  // hl / ho are pageable locals
    const size_t pbytes = lc_packed_elems(nt, ny, nx) * es;
    const bool need_lin = host_route_needs_lin(wdtype, interp_order);
    if (need_lin) LC_TRY(lin.alloc(pbytes));
    if (interp_order != 1) LC_TRY(cub.alloc(pbytes));
    const bool fusable = (interp_order == 1 || interp_order == 3) && !f64_exact_order(ctx, wdtype, ny, nx);
    if (settls_order > 0 && fusable) LC_TRY(ext.alloc(lc_packed_elems(nt - 1, ny, nx) * es));
    if (need_lin || (interp_order == 1 && ext.p))
        LC_TRY(lc_field_pack(ctx, uw, vw, wdtype, nt, ny, nx, 1, lin.p, interp_order == 1 ? ext.p : nullptr));
    if (interp_order != 1) LC_TRY(lc_field_pack(ctx, uw, vw, wdtype, nt, ny, nx, interp_order, cub.p, ext.p));
    LC_TRY(x.alloc(sbytes));
    LC_TRY(y.alloc(sbytes));
    const double lat_min = wdtype == LC_F32 ? (double)(float)lat[0] : lat[0], lat_max = wdtype == LC_F32 ? (double)(float)lat[ny - 1] : lat[ny - 1];
    const double lon_min = wdtype == LC_F32 ? (double)(float)lon[0] : lon[0], lon_max = wdtype == LC_F32 ? (double)(float)lon[nx - 1] : lon[nx - 1];
    LC_TRY(advect_with_raw(ctx, lin.p, cub.p, ext.p, uw, vw, wdtype, nt, ny, nx, lat_min, lat_max, lon_min, lon_max, slat.p, ny,
                           slon.p, nx, timestep, settls_order, interp_order, LC_X_CYCLIC /* LCS.py:119 */, 0, nt - 1, x.p, y.p,
                           nullptr, nullptr));
    if (sigma_out) {
        LC_REQUIRE(ny >= 5 && nx >= 5, "lc_lcs_global_host: sigma needs at least a 5x5 grid");
        LC_TRY(sig.alloc(sbytes));
        const void *xs = x.p, *ys = y.p;
        if (gauss_sigma > 1e-15) {
            LC_TRY(gx.alloc(sbytes));
            LC_TRY(gy.alloc(sbytes));
            LC_TRY(gtmp.alloc(sbytes));
            LC_TRY(lc_gaussian_filter(ctx, x.p, wdtype, ny, nx, gauss_sigma, gtmp.p, gx.p));
            LC_TRY(lc_gaussian_filter(ctx, y.p, wdtype, ny, nx, gauss_sigma, gtmp.p, gy.p));
            xs = gx.p;
            ys = gy.p;
        }
        const double dlat = wdtype == LC_F32 ? (double)((float)lat[1] - (float)lat[0]) : lat[1] - lat[0];
        const double dlon = wdtype == LC_F32 ? (double)((float)lon[1] - (float)lon[0]) : lon[1] - lon[0];
        LC_TRY(lc_sigma(ctx, xs, ys, wdtype, 0, ny, nx, ny, slat.p, dlat, dlon, fd_fp32_cast, tensor_layout, 0, ny, sig.p));
        LC_HIP_CHECK(hipMemcpyAsync(sigma_out, sig.p, sbytes, hipMemcpyDeviceToHost, st));
    }
    if (x_out) LC_HIP_CHECK(hipMemcpyAsync(x_out, x.p, sbytes, hipMemcpyDeviceToHost, st));
    if (y_out) LC_HIP_CHECK(hipMemcpyAsync(y_out, y.p, sbytes, hipMemcpyDeviceToHost, st));
    LC_HIP_CHECK(hipStreamSynchronize(st));
    return LC_OK;
}
