/*
 * lcs_hip.h -- C ABI of the MI355X-native FTLE engine (liblcs_hip.so).
 *
 * Drop-in boundary for the parcel-advection -> flow-map-gradient -> sigma_max
 * hot path of gabrielmpp/LagrangianCoherence.  The reference is pure Python
 * (no FFI of its own), so every entry point below names the reference Python
 * function it replaces (file:line relative to the reference checkout).  The
 * reference-side binding a maintainer would add is a ctypes stub; it is shown
 * in INTEGRATION.md.
 *
 * Conventions
 *   - plain C: pointers, sizes, doubles.  No torch / C++ types.
 *   - every function returns an lc_status (0 = ok, <0 = error) and records a
 *     message retrievable with lc_last_error() (thread-local).
 *   - "dev" pointers are HIP device pointers valid on the context's device;
 *     "host" pointers are ordinary process memory.  The library never keeps a
 *     caller pointer after the call returns and never hands out memory it
 *     owns, except through lc_malloc (freed with lc_free).
 *   - arrays are C-contiguous; fields are (time, latitude, longitude) with
 *     latitude and longitude ASCENDING (what the reference reaches after its
 *     sortby calls, LCS/trajectory.py:49-52, LCS/LCS.py:101-104).
 *   - dtype selects the arithmetic type of fields AND positions:
 *     LC_F32 (all float) or LC_F64 (all double).
 *   - device work is enqueued on the context's stream and is asynchronous
 *     unless stated; lc_sync() waits for it.
 */
#ifndef LCS_HIP_H
#define LCS_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* 101: lc_spectral_truncate gained `gridtype`, lc_fourth_order_derivative gained `isglobal` (arguments inserted: a
 * client built against 100 must be rebuilt), lc_ctx_get_level_chunk, lc_ctx_set/get_f64_fidelity, lc_advect_ex and
 * lc_sample_raw added, lc_field_pack accepts packed_dev == NULL at order 1 (fused-level image only).  lc_version() returns the value the LIBRARY
 * was built with: compare it with this macro before any other call (tests/c/abi_smoke.c, _capi.load do). */
#define LC_VERSION 104 /* 0.1.4: + lc_ctx_set_host_pipeline, lc_copy_to_device, lc_copy_to_host, lc_ctx_set_host_cache, lc_ctx_trim, lc_ctx_last_host_marks, lc_ctx_set_xcd_split (0.1.3: + lc_ctx_last_pack_kernel; 0.1.2: + lc_ctx_set_verify, lc_ctx_read_verify, LC_F64_WIND_F32_LIN32) */

typedef struct lc_ctx lc_ctx;

enum lc_dtype {
    LC_F32 = 0,
    LC_F64 = 1,
    /* lc_advect only: float64 positions and images whose wind values are float32-valued; the
     * arithmetic follows numpy's promotion for that mix in the reference (samples rounded to float,
     * latitude increments formed in float; LCS/trajectory.py:86-87,110-112 with SURVEY Q10). */
    LC_F64_WIND_F32 = 2,
    /* lc_advect_ex only, interp_order 1 or 3, cyclic or per-point boundaries: the same arithmetic with the wind KEPT float32
     * wherever scipy keeps it; positions, seeds and outputs are float64; results equal LC_F64_WIND_F32's bit for bit.
     *   interp_order 1: packed_lin is the order-1 image lc_field_pack builds for LC_F32 (half the bytes, no float64 copy of
     *                   the wind: a node is widened as it is read), every other image pointer NULL; per-wave LDS tiles of
     *                   levels t and t + 1.
     *   interp_order 3: packed_cub is the FLOAT64 coefficient image of the float32 planes -- lc_field_pack(LC_F64_WIND_F32,
     *                   order 3): scipy's spline_filter(output=float64) inside map_coordinates -- and u_raw / v_raw are the
     *                   float32 planes themselves (the order-1 source of the pole rows); packed_lin NULL.
     * (lc_field_pack accepts LC_F64_WIND_F32 for interp_order 2..5 without ext_dev: float32 planes in, float64 image out.) */
    LC_F64_WIND_F32_LIN32 = 3
};

enum lc_status {
    LC_OK = 0,
    LC_EINVAL = -1,       /* bad argument (null pointer, bad size, bad enum) */
    LC_EUNSUPPORTED = -2, /* e.g. interp_order outside 1..5                  */
    LC_EHIP = -3,         /* a HIP runtime call failed                       */
    LC_ENOMEM = -4,
    LC_ERCCL = -5         /* RCCL missing or an RCCL call failed             */
};

/* lc_advect's cyclic_x argument: what happens to a longitude that leaves [lon_min, lon_max]
 * (LCS/trajectory.py:92-97, 118-123). */
enum lc_xboundary {
    LC_X_CLAMP_POINT = 0, /* cyclic_xboundary=False, each parcel clamped on its own (NOT the reference when a parcel
                             leaves the box: see LC_X_CLAMP_REFERENCE_OUTER) */
    LC_X_CYCLIC = 1,      /* cyclic_xboundary=True: wrap hard-coded to +-180 with Python's floor-mod (Q7)          */
    /* cyclic_xboundary=False exactly as the reference computes it: `positions_x[np.where(x < x_min)] = x_min` on a
     * DataArray is orthogonal indexing, so every (row, col) in the cross product of offending rows and offending
     * columns is set (trajectory.py:96-97, 122-123).  lc_advect runs the fused kernel in chunks of 16 time
     * levels, reads a "clamp fired" flag back after each (the call is SYNCHRONOUS in this mode: one stream
     * synchronisation per chunk) and, from the chunk in which a parcel first left the box, continues sub-step by
     * sub-step with that rule (2 (K+1) launches per time level, positions in global memory) from the positions saved
     * before that chunk; if no parcel ever leaves, the fused result stands.
     * The rule couples every seed row through the offending COLUMNS: a call on a row block (row0 != 0 or
     * ny != ny_global) needs lc_ctx_set_flag_allreduce, through which the ranks of a row-sharded grid OR their
     * column flags after every sub-step; without it such a call is refused. */
    LC_X_CLAMP_REFERENCE_OUTER = 2
};

enum lc_tensor_layout {
    LC_LAYOUT_REFERENCE = 0, /* 9 comps reshaped row-major to 3x3 (LCS/LCS.py:152-153) */
    LC_LAYOUT_PHYSICAL = 1   /* Jacobian d(X,Y,Z)/d(x,y); NOT reference behaviour      */
};

/* ---- library / context ------------------------------------------------- */
int lc_version(void);
const char *lc_last_error(void);
/* What this binary was built from: the hash of the kernel sources + this header (comments and white space stripped;
 * lagrangiancoherence_amd/build.py csrc_hash), followed by "+<flags>" when it is an experiment build with extra -D
 * flags; "unstamped" for a build that bypassed build.py.  A benchmark replays committed profiler counters only for
 * the binary they were measured on. */
const char *lc_build_id(void);

/* One context per device; distinct contexts may be used from distinct host
 * threads.  The context owns one HIP stream unless lc_ctx_set_stream lends it
 * an external one (e.g. the stream a host framework is already using). */
int lc_ctx_create(int device, lc_ctx **out);
int lc_ctx_destroy(lc_ctx *ctx);
int lc_ctx_set_stream(lc_ctx *ctx, void *hip_stream /* hipStream_t; NULL = the device's default stream */);
int lc_ctx_use_own_stream(lc_ctx *ctx); /* back to the context's private stream */
int lc_sync(lc_ctx *ctx);

/* Kernel choice of lc_advect: -1 = what lc_ctx_create set (LCS_LDS_TILES, else the default): per-wave LDS tiles -- float32
 * orders 1 and 3 with two seeds per lane from 2^23 seeds per call upwards and one seed per lane below (smaller launches
 * want the waves), float64 order 1 with packed_ext one seed per lane; 1 = LDS tiles, two seeds per lane whatever the
 * size; 2 = LDS tiles, one seed per lane; 0 = direct gathers (float32 and float64).  SETTLS_order = 0 stages no tile at
 * order 1: direct gathers, two seeds per lane from 2^23 seeds per call upwards (the two-seed kernel compiled without its tile
 * and iteration blocks), one per lane below; the LDS kernels at order 3.  The environment variable LCS_LDS_TILES (0/1/2) sets the initial
 * value, read ONCE in lc_ctx_create (profiling A/B; results are bit-identical either way).  No reference counterpart. */
int lc_ctx_set_lds_tiles(lc_ctx *ctx, int mode);
/* Kernel choice of lc_sigma for float32 sigma-only calls on grids of even width: 1 = marching kernel (a wave walks
 * down its rows with five rows of X, Y, Z in registers and takes the x-neighbours by wavefront shuffle), 0 = the
 * LDS-tile kernel that also serves odd widths, -1 = what lc_ctx_create set (LCS_SIGMA_MARCH, else the default): the marching kernel from 2^23 cells per call upwards
 * (its waves walk 24 rows one after the other: smaller grids finish sooner as many short-lived tiles).
 * LCS_SIGMA_MARCH (0/1) sets the initial value, read ONCE in lc_ctx_create.  Results are bit-identical either way.
 * No reference counterpart. */
int lc_ctx_set_sigma_march(lc_ctx *ctx, int on);
/* Row-sharded grids with LC_X_CLAMP_REFERENCE_OUTER: `fn(user, flags_dev, count)` must replace the `count` uint32
 * flags at `flags_dev` (device memory of this context) by their element-wise MAXIMUM over all ranks that hold row
 * blocks of the same seed grid, ordered after the work already enqueued on the context's stream and before
 * anything enqueued on it afterwards (an all-reduce on that stream, or one that synchronises with it); it returns
 * 0 on success.  Every rank's lc_advect calls it the same number of times: once for "did any parcel leave the box
 * anywhere", then -- only if one did -- twice per sub-step for the offending-column flags (nx of them) of the
 * reference's two assignments (LCS/trajectory.py:96-97, 122-123).  NULL (the default) = single process.
 * lc_comm_flag_allreduce (below) is such a function for an lc_comm. */
typedef int (*lc_flag_allreduce_fn)(void *user, void *flags_dev, size_t count);
int lc_ctx_set_flag_allreduce(lc_ctx *ctx, lc_flag_allreduce_fn fn, void *user);
/* lc_advect / lc_advect_from run a series of nsteps time levels as consecutive launches of at most `levels` levels,
 * each continuing from the positions the previous one stored (0 = one launch; -1 = by size, the default: from 2^18
 * seeds per call upwards 32 levels per launch for SETTLS_order >= 3, 64 for 2, one launch below).  Results are bit-identical whatever the value; it shapes the launches
 * only (workgroups of one launch stay within `levels` levels of each other, so their tiles of the wind images meet in
 * L2 / the Infinity Cache, and the launch's tail is one chunk long: long series and sparse seed grids gain 7-17 %).
 * The environment variable LCS_LEVEL_CHUNK sets the initial value, read ONCE in lc_ctx_create.  No reference
 * counterpart (the reference's loop over time levels is LCS/trajectory.py:80-126). */
int lc_ctx_set_level_chunk(lc_ctx *ctx, int levels);
/* The value in force (what lc_ctx_set_level_chunk was last given, or what LCS_LEVEL_CHUNK set at creation): a caller
 * that changes it for one call restores THIS, not a guess. */
int lc_ctx_get_level_chunk(const lc_ctx *ctx, int *levels_out);
/* float64 on the ONE-CALL host routes (lc_lcs_host, lc_lcs_global_host -- what a reference-side binding calls, where a
 * user expects the reference's numbers): which form of the SETTLS iteration they take.
 *   LC_F64_EXACT_ORDER  numpy / scipy's operation order (two samples per iteration, true divisions, scipy's tap sums;
 *                       LCS/trajectory.py:86-87,110-112): ~1e-13 degrees from the reference's float64 result;
 *   LC_F64_FAST         one sample of the fused-level image 2F[t]-F[t+1] per iteration, multiplied index map, fused
 *                       lerps: the same mathematics with different ROUNDING, 1.4-2x faster at BASELINE config 2's size;
 *                       distance from the reference <= 1e-9 degrees, or the flow's own amplification of a 1e-12 degree
 *                       seed shift if that is larger (ill-conditioned flows; INTEGRATION.md "Behavioural notes");
 *   LC_F64_AUTO         (default) exact order up to LC_EXACT_ORDER_MAX_SEEDS seeds per call -- there the time is launch
 *                       latency, not arithmetic: the reference's example (89 x 180) and the 360 x 721 common grid of
 *                       isglobal=True are on this side -- and the fast form above.
 * float32 always takes the fused form (it cannot be bit-comparable with scipy's float64 interpolation anyway).
 * lc_advect itself is explicit: packed_ext == NULL is the exact order.  LCS_F64_FIDELITY (auto / exact / fast) sets the
 * initial value, read ONCE in lc_ctx_create. */
enum lc_f64_fidelity { LC_F64_AUTO = 0, LC_F64_EXACT_ORDER = 1, LC_F64_FAST = 2 };
#define LC_EXACT_ORDER_MAX_SEEDS (1 << 18)
int lc_ctx_set_f64_fidelity(lc_ctx *ctx, int mode);
int lc_ctx_get_f64_fidelity(const lc_ctx *ctx, int *mode_out);
/* How the one-call host routes move their host buffers (no reference counterpart: LCS/LCS.py:129-157 runs on host arrays).
 *   1 (default)  through a ring of four 32 MB pinned staging buffers filled / emptied by a few host threads (ONE ring per device
 *                and process, shared by its contexts -- a process's second copy stream runs 25 % slower --, taken by a context's
 *                first staged transfer, released by lc_ctx_destroy, freed with the last context that holds it; transfers of
 *                different contexts take turns): pageable caller memory
 *                then travels at the bus rate whatever state its pages are in (hipMemcpy from pages the runtime has not pinned
 *                before runs at a quarter of it), the upload is cut at time-level boundaries and the pack + advect kernels of level
 *                chunk c run while chunk c + 1 is on the bus (cyclic_x = LC_X_CYCLIC, fused levels, no trajectories, 32 steps or
 *                more; lc_advect_from continuation: bit-identical to the serial form), only the levels [t0, t0 + nsteps] travel,
 *                and the caller's output pages are populated by background threads meanwhile.  A host with no pinned memory to
 *                give falls back to 0 by itself.
 *   0            plain hipMemcpyAsync of the whole series on the context's stream, then pack, advect, sigma, copies back.
 * LCS_HOST_PIPELINE (0 / 1) sets the initial value, read ONCE in lc_ctx_create. */
int lc_ctx_set_host_pipeline(lc_ctx *ctx, int on);
/* Copies between a caller's ordinary (pageable) host array and device memory through the context's staging ring (see
 * lc_ctx_set_host_pipeline; plain hipMemcpy with it switched off), ordered with the context's stream: work enqueued on the
 * stream after lc_copy_to_device sees the data (the source may be reused as soon as the call returns); lc_copy_to_host sees
 * everything enqueued before it and returns when the bytes are in `host`.  What the Python host moves large arrays with
 * (numpy -> device tensor, results -> numpy): a first-touch hipMemcpy of pageable memory runs at 12-14 GB/s, these at the bus's 55. */
int lc_copy_to_device(lc_ctx *ctx, void *dst_dev, const void *src_host, size_t bytes);
int lc_copy_to_host(lc_ctx *ctx, void *dst_host, const void *src_dev, size_t bytes);
/* lc_lcs_host keeps the device buffers of its last call on the context (1, the default) and hands them to the next call that
 * fits them: hipMalloc + hipFree of BASELINE configs[2]'s 2.6 GB are 1.6 ms of a 21 ms call.  lc_ctx_trim frees what is kept
 * (device memory returns to the system; the next call allocates again), lc_ctx_destroy does too; 0 = allocate and free per call.
 * LCS_HOST_CACHE (0 / 1) sets the initial value, read ONCE in lc_ctx_create. */
int lc_ctx_set_host_cache(lc_ctx *ctx, int on);
int lc_ctx_trim(lc_ctx *ctx);
/* Wall-clock marks of the context's last lc_lcs_host call, milliseconds since its entry: [0] device buffers allocated, [1] the
 * last upload piece handed to the DMA engine and the last kernel launched, [2] kernels finished (0 with plain copies), [3]
 * results in the caller's buffers.  (A benchmark reports the upload / compute tail / download split with them.) */
int lc_ctx_last_host_marks(const lc_ctx *ctx, double *ms4_out);
/* How lc_advect deals the workgroups of a launch to the 8 XCDs: -1 (default) by the launch's shape -- whole workgroup rows
 * cyclically, eighths of a row when the row count would leave the XCDs more than 15 % apart; 0 whole rows always; n > 0 a
 * 1 / n part of a row per chunk.  (A seed block whose rows do not cost alike -- the polar rows of a global grid take several
 * times longer -- ends on the XCD that drew them; 8 spreads every row over all XCDs at ~4-15 % on launches that were even.)
 * LCS_XCD_SPLIT sets the initial value, read ONCE in lc_ctx_create. */
int lc_ctx_set_xcd_split(lc_ctx *ctx, int split);
/* Name of the kernel the context's last lc_advect call launched (static string, "" before the first call);
 * what a profiler shows, so a benchmark labels its numbers with the kernel that actually ran. */
const char *lc_ctx_last_advect_kernel(const lc_ctx *ctx);
/* Kernel launches that call made (level chunks: lc_ctx_set_level_chunk); a benchmark divides its event time by it to
 * quote the duration of ONE launch, the figure a profiler's per-kernel average shows. */
int lc_ctx_last_advect_launches(const lc_ctx *ctx);
/* The same for the context's last lc_sigma / lc_flowmap_gradient call. */
const char *lc_ctx_last_sigma_kernel(const lc_ctx *ctx);
/* The kernel the last lc_field_pack launched for its first stage (interleave / spline prefilter; "" before any): e.g.
 * "pack_fused_kernel", "prefilter_fir_kernel", "prefilter_fused_stream_kernel<double>" (float64 order 3, both axes of 64
 * nodes or more: both prefilter sweeps in one pass), "prefilter_cols_stream_kernel + prefilter_rows_stream_kernel".  The pads
 * / fused-level pass that follows is not named.  Same lifetime as lc_ctx_last_advect_kernel's string. */
const char *lc_ctx_last_pack_kernel(const lc_ctx *ctx);
/* Wave-state audit (diagnostic; no reference counterpart).  mode 1: float32 lc_advect calls that dispatch to the one-seed
 * LDS-tile kernels (orders 1 and 3 below 2^23 seeds per call, SETTLS_order > 0, no whole-line trajectory stores) run their "verify" instances:
 * after the iterations of every time level each wave reads back the tile of the wind image it staged in LDS for that
 * level and compares it, 16 bytes per lane, with the registers it staged it from, and compares HW_REG_HW_ID /
 * HW_REG_XCC_ID with the values it read when it started.  A wave's tile and registers are written by that wave only,
 * so a non-zero count means the wave's state was changed from OUTSIDE the kernel: its context saved and restored by
 * the driver (several processes time-sharing one GPU) with something lost on the way, or a hardware fault.  Results
 * are bit-identical to the plain instances (a few percent slower).  mode 2 additionally overwrites one tile entry once
 * (tile 5, wave 1, second level) so that a test can see the audit fire; mode 0 frees the counters.
 * lc_ctx_read_verify synchronises the context's stream and copies the LC_VERIFY_WORDS counters:
 *   [0] wave-levels whose tile differed, [1] 16-byte entries that differed, [2] wave-levels at which the wave sat in
 *   another hardware slot than one level earlier (a context switch; harmless by itself), [3] wave-levels audited,
 *   [4..11] the first event: workgroup, tile, wave, time level, HW_ID before / after, lane mask low / high word.
 * `reset` != 0 zeroes them afterwards. */
#define LC_VERIFY_WORDS 16
int lc_ctx_set_verify(lc_ctx *ctx, int mode);
int lc_ctx_read_verify(lc_ctx *ctx, unsigned *out16, int reset);

/* ---- device memory (so a ctypes-only host needs nothing else) ---------- */
int lc_malloc(lc_ctx *ctx, size_t bytes, void **dev_out);
int lc_free(lc_ctx *ctx, void *dev);
int lc_memcpy_h2d(lc_ctx *ctx, void *dev_dst, const void *host_src, size_t bytes); /* synchronous */
int lc_memcpy_d2h(lc_ctx *ctx, void *host_dst, const void *dev_src, size_t bytes); /* synchronous */

/* ---- field preparation --------------------------------------------------
 * Replaces the per-call work of tools.xr_map_coordinates that does not depend
 * on the seeds (LCS/tools.py:12-14 copies; for order 3 the whole-field spline
 * prefilter scipy redoes inside every map_coordinates call, LCS/tools.py:26).
 *
 * Builds the gather-ready image of a wind time series: per time level a
 * (ny_f+3) x (nx_f+3) array of interleaved (u,v) nodes -- 1 mirrored node in
 * front and 2 behind on each axis, so the 2x2 / 4x4 tap windows never need
 * index logic.  order 1: raw values.  orders 2..5: B-spline coefficients of that order
 * (mirror-boundary prefilter with scipy's pole lists, exact-sum initialisation ==
 * scipy.ndimage.spline_filter(order, mode='mirror'); orders 4, 5 agree with scipy to 1e-12: the pole
 * constants differ in the last bit).
 *
 * lc_packed_elems() = number of dtype elements the image needs.
 * interp_order 1 with packed_dev == NULL and ext_dev != NULL builds ONLY the fused-level image: what a float64
 * caller needs who hands lc_advect_ex the raw planes as the order-1 source (half the bytes written).          */
size_t lc_packed_elems(int nt, int ny_f, int nx_f);
int lc_field_pack(lc_ctx *ctx, const void *u_dev, const void *v_dev, int dtype,
                  int nt, int ny_f, int nx_f, int interp_order, void *packed_dev,
                  void *ext_dev /* NULL, or lc_packed_elems(nt-1,..): also build the
                                   lc_field_extrapolate image of this order in the same call */);

/* Optional third image for the SETTLS iterations: ext[t] = 2*packed[t] - packed[t+1],
 * t = 0..nt-2 (same layout, nt-1 levels).  Interpolation is linear in the field, so
 * one sample of ext[t] equals 2*interp(F[t]) - interp(F[t+1]) of trajectory.py:110-112
 * up to rounding, and halves the gathers of every SETTLS iteration.  Build it from
 * the image that matches interp_order (raw for 1, coefficients for 3). */
int lc_field_extrapolate(lc_ctx *ctx, const void *packed_dev, int dtype,
                         int nt, int ny_f, int nx_f, void *ext_dev);

/* ---- global pre-processing of LCS.__call__(isglobal=True) -----------------
 * lc_regrid_common_grid replaces LCS/LCS.py:107-114: `u.interp(latitude=lats, longitude=lons, method='linear')`
 * with the holes (targets outside the source range, NaN results) filled from `u.reindex(method='nearest')`.
 * One fused kernel: latitude lerp, longitude lerp (scipy.interpolate.interp1d's operation order), nearest
 * fill.  src [nt][ny_s][nx_s] in `dtype` on the device, coordinates ascending, as host doubles; out
 * [nt][ny_d][nx_d] float64 on the device (xarray's interp result is float64).  The reference's target grid is
 * lats = linspace(-89.75, 89.75, 360), lons = linspace(-180, 179.5, 721). */
int lc_regrid_common_grid(lc_ctx *ctx, const void *src_dev, int dtype, int nt, int ny_s, int nx_s,
                          const double *src_lat_host, const double *src_lon_host,
                          const double *dst_lat_host, int ny_d, const double *dst_lon_host, int nx_d,
                          double *out_dev);

/* lc_spectral_truncate replaces LCS/LCS.py:115-118: windspharm `VectorWind(u, v).truncate(f, truncation=T)`
 * = spherical-harmonic analysis, triangular truncation n <= T, synthesis.  gridtype is what windspharm's grid
 * inspection finds: LC_GRID_REGULAR = SPHEREPACK's equally spaced grid (theta_i = i pi / (nlat-1); shaes/shses),
 * LC_GRID_GAUSSIAN = Gauss-Legendre latitudes (shags/shsgs: Gaussian quadrature).  f, out: [nbatch][nlat][nlon] in
 * `dtype`, latitude ASCENDING; computed in float64.  Operators are built on the host once per (nlat, nlon, T,
 * gridtype) and cached on the context.  Any T <= min(nlat-1, (nlon-1)/2).  Restates the published algorithm
 * (Swarztrauber 1979); not pinned against pyspharm. */
enum lc_gridtype { LC_GRID_REGULAR = 0, LC_GRID_GAUSSIAN = 1 };
/* windspharm's latitude inspection (what VectorWind does with the grid it is handed, LCS/LCS.py:116): equally spaced
 * global latitudes -> LC_GRID_REGULAR, Gaussian latitudes -> LC_GRID_GAUSSIAN (both to 5e-4 degrees), anything else
 * LC_EINVAL with windspharm's message.  lat_ascending: host doubles.  No device work. */
int lc_inspect_gridtype(const double *lat_ascending, int nlat, int *gridtype_out);
int lc_spectral_truncate(lc_ctx *ctx, const void *f_dev, int dtype, int nbatch, int nlat, int nlon,
                         int truncation, int gridtype, void *out_dev);

/* ---- K1: parcel advection ------------------------------------------------
 * Replaces trajectory.parcel_propagation (LCS/trajectory.py:8-144) together
 * with every tools.xr_map_coordinates call it makes (LCS/tools.py:11-41):
 * Euler + settls_order accumulate-"SETTLS" sub-steps per time level, latitude
 * clamp, cyclic (+-180) or clamped longitude, all nsteps fused in one launch.
 *
 *   packed_lin   image from lc_field_pack(order=1)   (always required: the
 *                first/last interp_order seed rows use order 1 + 'constant')
 *   packed_cub   image from lc_field_pack(order=interp_order), or NULL when interp_order==1.  interp_order may be
 *                any of scipy's spline orders 1..5 (LCS/trajectory.py:16, LCS/tools.py:26-30 hand it to
 *                map_coordinates); 1 and 3 have the fused / LDS-tile kernels, 2, 4 and 5 a generic direct one
 *                (no packed_ext).  0 fails in the reference itself (empty interior slice).
 *   packed_ext   image from lc_field_extrapolate, or NULL.  NULL = two samples per
 *                SETTLS iteration in the reference's operation order (the float64
 *                default, results identical to numpy/scipy's); non-NULL = one sample
 *                of the combined field 2F[t]-F[t+1] (the float32 default; opt-in for
 *                LC_F64, where it moves results by rounding only, ~1e-13 degrees)
 *   lat_min..lon_max   extremes of the FIELD coordinates (index scale, tools.py:21-22,
 *                and the clamp bounds, trajectory.py:63-66)
 *   seed_lat[ny], seed_lon[nx]   seed coordinates (dtype elements, device).  The
 *                reference seeds at the field nodes (trajectory.py:68-70): pass the
 *                field coordinates for that.
 *   row0, ny_global   this call advects seed rows [row0, row0+ny) of a grid of
 *                ny_global rows (row sharding across GPUs); the pole-row rule
 *                applies to the global row index.  Single GPU: 0, ny.
 *   timestep     seconds, sign = direction; fields are always consumed in stored
 *                order t0, t0+1, ... (trajectory.py:58-60,80)
 *   cyclic_x     enum lc_xboundary
 *   t0, nsteps   first time level and number of steps (t0+nsteps <= nt-1);
 *                the reference is t0=0, nsteps=nt-1
 *   x_out,y_out  [ny*nx] departure longitude / latitude, degrees
 *   traj_x,traj_y  NULL, or [(nsteps+1)*ny*nx]: positions after every step,
 *                entry 0 = seed grid (return_traj=True, trajectory.py:125-139)
 */
int lc_advect(lc_ctx *ctx, const void *packed_lin, const void *packed_cub,
              const void *packed_ext, int dtype,
              int nt, int ny_f, int nx_f,
              double lat_min, double lat_max, double lon_min, double lon_max,
              const void *seed_lat_dev, int ny, const void *seed_lon_dev, int nx,
              int row0, int ny_global,
              double timestep, int settls_order, int interp_order, int cyclic_x,
              int t0, int nsteps,
              void *x_out, void *y_out, void *traj_x, void *traj_y);

/* lc_advect continuing from given positions instead of the seed grid: x_start, y_start [ny*nx] (dtype elements,
 * device; both NULL = lc_advect).  The seed coordinates are still needed -- conversion_x is a function of the SEED
 * latitude for the whole integration (LCS/trajectory.py:56-57, Q5) and the pole-row rule of the seed row index
 * (LCS/tools.py:24-39, Q3).  Running levels [t0, t0+a) with lc_advect and [t0+a, t0+a+b) with lc_advect_from on the
 * first call's outputs gives, bit for bit, what one call over a+b levels gives: the loop body of
 * LCS/trajectory.py:80-126 carries no state from one time level to the next except the positions.  x_start / y_start
 * may alias x_out / y_out (in place).  traj entry 0 = the start positions.  Not with LC_X_CLAMP_REFERENCE_OUTER. */
int lc_advect_from(lc_ctx *ctx, const void *packed_lin, const void *packed_cub,
                   const void *packed_ext, int dtype,
                   int nt, int ny_f, int nx_f,
                   double lat_min, double lat_max, double lon_min, double lon_max,
                   const void *seed_lat_dev, int ny, const void *seed_lon_dev, int nx,
                   int row0, int ny_global,
                   const void *x_start, const void *y_start,
                   double timestep, int settls_order, int interp_order, int cyclic_x,
                   int t0, int nsteps,
                   void *x_out, void *y_out, void *traj_x, void *traj_y);

/* lc_advect_from for an ENSEMBLE of n_members start times over one seed grid and one wind series, in one launch per
 * level chunk: member m integrates nsteps steps from time level t0 + m * t0_stride (BASELINE config 5: stride 1) and
 * keeps its positions in the m-th [ny*nx] plane of x_start / x_out ([n_members][ny*nx]; x_start NULL = every member
 * starts on the seed grid; in place allowed).  With lc_ctx_set_level_chunk(c) the members advance together c levels
 * at a time (level-major order), so the launches of a chunk share all but a few of their time levels in the caches
 * and a launch is n_members times deeper than a member's own (no tail of idle compute units between members).  In
 * float32 at order 1 with SETTLS_order > 0, from 2^23 seeds per call, consecutive members share a LANE (members 2p and
 * 2p + 1 at the same grid point: at field level l one takes its step l, the other its step l - t0_stride, from one staged
 * tile of the wind), and the launches walk the pair's level window instead of a step range.  Each
 * member's result is, bit for bit, what lc_advect(t0 + m * t0_stride) gives.  traj_x / traj_y must be NULL and
 * cyclic_x must not be LC_X_CLAMP_REFERENCE_OUTER when n_members > 1.  No reference counterpart: there the caller
 * loops over start times (LCS/trajectory.py:80 consumes one series). */
int lc_advect_batch(lc_ctx *ctx, const void *packed_lin, const void *packed_cub,
                    const void *packed_ext, int dtype,
                    int nt, int ny_f, int nx_f,
                    double lat_min, double lat_max, double lon_min, double lon_max,
                    const void *seed_lat_dev, int ny, const void *seed_lon_dev, int nx,
                    int row0, int ny_global,
                    const void *x_start, const void *y_start,
                    double timestep, int settls_order, int interp_order, int cyclic_x,
                    int t0, int nsteps, int n_members, int t0_stride,
                    void *x_out, void *y_out, void *traj_x, void *traj_y);

/* lc_advect_batch with its arguments in one structure (same names, same meaning), plus the RAW wind planes as the
 * order-1 source in place of the packed_lin image:
 *   u_raw, v_raw   NULL, or the [nt][ny_f][nx_f] arrays lc_field_pack was given (dtype elements, device, still alive and
 *                  unchanged).  With them packed_lin may be NULL when interp_order != 1 (the first / last interp_order
 *                  seed rows gather their order-1 / 'constant' samples, LCS/tools.py:31-39, straight from the planes) and
 *                  in LC_F64 / LC_F64_WIND_F32 at interp_order 1 (the Euler sample, LCS/trajectory.py:82-84, too): the
 *                  order-1 image is then never built, written or read -- one third of the pack's traffic at order 1 and a
 *                  fifth at order 3 -- and the results are bit-identical to the packed_lin form (same node values, same
 *                  arithmetic).  LC_F32 at interp_order 1 keeps packed_lin (its kernels read 16-byte {u, v} node pairs)
 *                  and ignores the planes.
 *   struct_size    sizeof(lc_advect_args) as the caller compiled it: a library built for another layout refuses.
 * lc_advect / lc_advect_from / lc_advect_batch are this call with u_raw = v_raw = NULL. */
typedef struct lc_advect_args {
    size_t struct_size;
    const void *packed_lin, *packed_cub, *packed_ext;
    const void *u_raw, *v_raw;
    int dtype, nt, ny_f, nx_f;
    double lat_min, lat_max, lon_min, lon_max;
    const void *seed_lat_dev;
    int ny;
    const void *seed_lon_dev;
    int nx;
    int row0, ny_global;
    const void *x_start, *y_start;
    double timestep;
    int settls_order, interp_order, cyclic_x;
    int t0, nsteps, n_members, t0_stride;
    void *x_out, *y_out, *traj_x, *traj_y;
    /* LC_F64 at interp_order 1 with u_raw / v_raw and packed_ext == NULL: 1 = take the fused-level form all the same -- the
     * value packed_ext would hold, 2 F[t] - F[t+1], is formed from the raw planes node by node inside the kernels (the
     * same expression, one rounding: results equal those with packed_ext bit for bit).  No packed image exists then:
     * lc_field_pack is not called at all for such a field.
     * LC_F64 at interp_order 3 with packed_cub and packed_ext == NULL: 1 = the same for the spline coefficients -- the
     * kernels form 2 c[t] - c[t+1] from packed_cub node by node (lc_field_pack's own expression: bit-identical to a call
     * with packed_ext), so lc_field_pack(order 3, ext_dev = NULL) is the whole pack: it neither reads the coefficients back
     * nor writes a second image, and the advect kernel streams one image series from memory instead of two.
     * 0 (and every other dtype / order): packed_ext == NULL means the reference's two-sample operation order, as in
     * lc_advect. */
    int fuse_levels_raw;
} lc_advect_args;
int lc_advect_ex(lc_ctx *ctx, const lc_advect_args *args);

/* One interpolation pass on its own: tools.xr_map_coordinates (LCS/tools.py:11-41) for the
 * u and v fields of time level `level` at the given positions (degrees), same index
 * scale, row classes and boundary modes as inside lc_advect.  pos_x/pos_y/out_u/out_v
 * are [ny*nx] dtype elements on the device. */
int lc_sample(lc_ctx *ctx, const void *packed_lin, const void *packed_cub, int dtype,
              int nt, int ny_f, int nx_f,
              double lat_min, double lat_max, double lon_min, double lon_max, int level,
              const void *pos_x_dev, const void *pos_y_dev, int ny, int nx,
              int row0, int ny_global, int interp_order, void *out_u, void *out_v);

/* lc_sample with the raw planes as the order-1 source (see lc_advect_ex): packed_lin may then be NULL (interp_order != 1:
 * only the pole rows sample order 1; LC_F64 at interp_order 1: every row does). */
int lc_sample_raw(lc_ctx *ctx, const void *packed_lin, const void *packed_cub, const void *u_raw, const void *v_raw, int dtype,
                  int nt, int ny_f, int nx_f,
                  double lat_min, double lat_max, double lon_min, double lon_max, int level,
                  const void *pos_x_dev, const void *pos_y_dev, int ny, int nx,
                  int row0, int ny_global, int interp_order, void *out_u, void *out_v);

/* ---- K3: flow-map gradient + largest singular value ------------------------
 * Replaces LCS.flowmap_gradient (LCS/LCS.py:171-225), tools.derivative_spherical_coords
 * + tools.fourth_order_derivative (LCS/tools.py:248-267, 190-228) and the eigen
 * step of LCS.__call__ (LCS/LCS.py:145-157) in one fused kernel.
 *
 *   x_dep,y_dep  [n_in_rows*nx] departure points for global seed rows
 *                [in_row0, in_row0+n_in_rows)
 *   out_row0, n_out_rows   rows to produce; each needs rows r-2..r+2 inside the
 *                input window unless it is one of the 2 first / 2 last GLOBAL rows
 *                (one-sided rule, tools.py:210-217)
 *   seed_lat     [n_in_rows] latitude of the input rows (dtype elements, device)
 *   dlat, dlon   seed spacing in degrees (lat[1]-lat[0], lon[1]-lon[0]; tools.py:255-256)
 *   fd_fp32_cast 1 = round X,Y,Z to float before differencing (reference, tools.py:258)
 *   sigma_out    [n_out_rows*nx]
 */
int lc_sigma(lc_ctx *ctx, const void *x_dep, const void *y_dep, int dtype,
             int in_row0, int n_in_rows, int nx, int ny_global,
             const void *seed_lat_dev, double dlat, double dlon,
             int fd_fp32_cast, int tensor_layout,
             int out_row0, int n_out_rows, void *sigma_out);

/* The 9-component "def_tensor" itself, for callers of LCS.flowmap_gradient
 * (LCS/LCS.py:171-225): planes dXdx,dXdy,dYdx,dYdy,dZdx,dZdy,dXdr,dYdr,dZdr (the last
 * three all zero, LCS.py:206-208), each [ny*nx], whole grid on one device. */
int lc_flowmap_gradient(lc_ctx *ctx, const void *x_dep, const void *y_dep, int dtype,
                        int ny, int nx, const void *seed_lat_dev, double dlat, double dlon,
                        int fd_fp32_cast, void *def_tensor_out);

/* tools.fourth_order_derivative (LCS/tools.py:190-245): 5-point index-space difference of a [ny*nx] array along
 * dim 0 (latitude; one-sided/2 on the 2 first/last rows, :210-217) or dim 1 (longitude: cyclic when isglobal != 0,
 * :220-228, else one-sided/2 on the 2 first/last columns, :229-244; the reference ignores isglobal for dim 0 and so
 * does this); result in the input dtype. */
int lc_fourth_order_derivative(lc_ctx *ctx, const void *in_dev, int dtype, int ny, int nx,
                               int dim, int isglobal, void *out_dev);

/* ---- optional smoothing of the departure fields ----------------------------
 * Replaces scipy.ndimage.gaussian_filter(x, sigma) at LCS/LCS.py:187-190
 * (truncate=4.0, mode='reflect', separable).  In place is not allowed. */
int lc_gaussian_filter(lc_ctx *ctx, const void *in_dev, int dtype, int ny, int nx,
                       double sigma, void *tmp_dev, void *out_dev);

/* ---- ridge classification (consumer of the sigma / FTLE field) ---------------------
 * Replaces the per-point Python loop of tools.find_ridges_spherical_hessian
 * (LCS/tools.py:99-138): numpy.linalg.eig of the symmetric 2x2 Hessian
 * [[hxx,hxy],[hxy,hyy]] (inf/NaN entries zeroed), the reference's row-indexed
 * eigenvector dotted with the gradient, the eigenvalue of largest magnitude, and the
 * mask (|dot| <= tolerance and that eigenvalue negative).  All arrays [n] doubles on the
 * device; dt_out (the raw dot product, tools.py:115) and eigvec_out ([2][n]: the two
 * components of that row-indexed eigenvector, tools.py:107, unmasked -- the
 * return_eigvectors=True outputs of tools.py:123-150 are built from it) may be NULL. */
int lc_ridge_classify(lc_ctx *ctx, const void *hxx, const void *hxy, const void *hyy,
                      const void *gx, const void *gy, size_t n, double tolerance,
                      void *mask_out, void *eigmin_out, void *dt_out, void *eigvec_out);

/* ---- multi-GPU: halo exchange on RCCL ----------------------------------------
 * New (the reference is single-process).  One process per GPU,
 * seed rows block-partitioned, wind replicated; lc_advect needs no communication
 * (row0 / ny_global).  Between lc_advect and lc_sigma every rank needs the 2 boundary
 * rows of (x_dep, y_dep) of the previous and the next rank (4th-order stencil,
 * LCS/tools.py:202-207): lc_halo_exchange does that in place, point-to-point over
 * RCCL (xGMI), on the context's stream, non-periodic in rank, no collective.
 *
 *   lc_comm_unique_id   rank 0 makes the 128-byte id; the caller hands it to every rank
 *                       (MPI, a file, torch.distributed ...)
 *   lc_comm_create      collective over the nranks processes; ctx fixes the device
 *   lc_halo_exchange    x_ext, y_ext: [n_rows][nx] device buffers whose middle
 *                       n_rows - n_lo - n_hi rows hold this rank's departure points
 *                       (write them there with lc_advect on a sub-view); the first n_lo /
 *                       last n_hi rows receive the neighbours' boundary rows.  n_lo must be
 *                       2 except on rank 0 (0), n_hi 2 except on the last rank (0).
 * RCCL is loaded at run time (dlopen), shared with the host process if it already
 * carries one; without it these calls return LC_ERCCL and everything else works. */
typedef struct lc_comm lc_comm;
int lc_comm_unique_id(void *id_out, size_t id_bytes /* >= 128 */);
int lc_comm_create(lc_ctx *ctx, int nranks, int rank, const void *id, size_t id_bytes, lc_comm **out);
int lc_comm_destroy(lc_comm *comm);
/* ncclCommCount / ncclCommUserRank of the communicator: the ranks RCCL itself spans (a benchmark reports it). */
int lc_comm_count(const lc_comm *comm, int *nranks_out, int *rank_out);
int lc_halo_exchange(lc_ctx *ctx, lc_comm *comm, void *x_ext, void *y_ext, int dtype,
                     int n_rows, int nx, int n_lo, int n_hi);
/* An lc_flag_allreduce_fn on RCCL: ncclAllReduce(max, uint32) in place over the communicator, on the stream of the
 * context the communicator was created with.  Use: lc_ctx_set_flag_allreduce(ctx, lc_comm_flag_allreduce, comm). */
int lc_comm_flag_allreduce(void *comm /* lc_comm* */, void *flags_dev, size_t count);

/* ---- one-call host entry point ----------------------------------------------
 * What a reference-side binding would call from LCS.__call__ (LCS/LCS.py:129-157):
 * host arrays in, host arrays out; upload, pack, advect, sigma, download, sync -- with the upload overlapped with the
 * kernels level chunk by level chunk and every transfer staged through pinned memory (lc_ctx_set_host_pipeline, the
 * default); keep the context across calls: it owns the staging ring.  The output buffers may be written (zeros) before the
 * results arrive.  Any of sigma_out / x_out / y_out / traj_x / traj_y may be NULL. */
int lc_lcs_host(lc_ctx *ctx, const void *u_host, const void *v_host, int dtype,
                int nt, int ny_f, int nx_f,
                const void *lat_f_host, const void *lon_f_host,
                const void *seed_lat_host, int ny, const void *seed_lon_host, int nx,
                double timestep, int settls_order, int interp_order, int cyclic_x,
                int t0, int nsteps, double gauss_sigma /* <=0: none */,
                int fd_fp32_cast, int tensor_layout,
                void *sigma_out, void *x_out, void *y_out, void *traj_x, void *traj_y);

/* ---- the reference's default global call form in one call, torch-free -------
 * LCS(...)(ds, isglobal=True) (LCS/LCS.py:105-157, examples/ideal_vortex.py:280-287) on host arrays:
 * [interp_to_common_grid: lc_regrid_common_grid of u and v onto lc_common_grid's grid, float64]
 * -> [truncation >= 0: lc_spectral_truncate at that wavenumber on the grid type lc_inspect_gridtype finds (equally
 * spaced global or Gaussian latitudes; anything else is refused as windspharm refuses it)] -> lc_field_pack -> lc_advect (cyclic, seeds = grid nodes, all nt-1 steps)
 * -> [gauss_sigma > 0: lc_gaussian_filter] -> lc_sigma.
 * Outputs [ny_o * nx_o] on the host, any may be NULL: ny_o x nx_o = 360 x 721 and float64 when
 * interp_to_common_grid, else ny_f x nx_f in `dtype`.  Synchronous. */
int lc_common_grid(int *ny_out, int *nx_out, double *lats_out /* NULL or [360] */, double *lons_out /* NULL or [721] */);
int lc_lcs_global_host(lc_ctx *ctx, const void *u_host, const void *v_host, int dtype,
                       int nt, int ny_f, int nx_f, const double *lat_f_host, const double *lon_f_host,
                       int interp_to_common_grid, int truncation /* < 0: none */,
                       double timestep, int settls_order, int interp_order,
                       double gauss_sigma /* <= 0: none */, int fd_fp32_cast, int tensor_layout,
                       void *sigma_out, void *x_out, void *y_out);

#ifdef __cplusplus
}
#endif
#endif /* LCS_HIP_H */
