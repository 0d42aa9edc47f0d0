// Internal header shared by the HIP translation units of liblcs_hip.so.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>

#include "../../include/lcs_hip.h"

struct lc_trunc_cache;                       // preprocess.hip: spectral-truncation operators of the last (nlat, nlon, T)
void lc_trunc_cache_free(lc_trunc_cache *c);

struct lc_ctx {
    int device;
    int n_cus;           // compute units of the device (the fused prefilter sizes its last round by it)
    hipStream_t own_stream;
    hipStream_t stream;  // the one work is enqueued on (own or borrowed)
    int lds_tiles;       // lc_advect float32 kernel choice: 3 LDS tiles, seeds per lane by size (default); 1 two seeds, 2 one seed per lane; 0 direct gathers (LCS_LDS_TILES at creation)
    int lds_tiles_init, sigma_march_init;  // what lc_ctx_create set (environment or built-in default): what -1 restores
    int sigma_march;     // lc_sigma float32 kernel choice: 2 by size (default: marching kernel from 2^23 cells), 1 marching kernel with wavefront shuffles, 0 LDS tiles (LCS_SIGMA_MARCH at creation)
    int xcd_split;       // lc_advect tile order: > 0 = a chunk is 1 / xcd_split of xcd_chunk_rows tile rows; -1 (default) = 8 when whole chunks would leave the XCDs > 15 % apart, else 0 (LCS_XCD_SPLIT at creation)
    int xcd_chunk_rows;  // lc_advect tile order: tile rows per chunk dealt to the XCDs cyclically; default 1; 0 = one contiguous band per XCD (LCS_XCD_CHUNK_ROWS at creation)
    int fir_prefilter;   // float32 order-3 pack: 1 one-pass truncated-convolution prefilter (default), 0 the recursive sweeps (LCS_FIR_PREFILTER at creation)
    int fused_prefilter; // float64 order-3 pack: 1 both sweeps in one pass (prefilter_fused_stream_kernel, default), 0 the two streaming sweeps (LCS_FUSED_PREFILTER at creation)
    int tile_order;      // lc_advect tile-row order: -1 per kernel (default), 0 as stored, 1 last row first, 2 poles inwards (LCS_TILE_ORDER at creation)
    int pole_blocks;     // lc_advect: 1 leading workgroups take the global pole rows (default), 0 the tiles do (LCS_POLE_BLOCKS at creation)
    int level_chunk;     // lc_advect: time levels per launch (-1 by size, the default: 32 from 2^18 seeds per call; 0 = the whole series in one launch); LCS_LEVEL_CHUNK at creation / lc_ctx_set_level_chunk
    int f64_fidelity;    // one-call host routes, float64: enum lc_f64_fidelity (LCS_F64_FIDELITY at creation / lc_ctx_set_f64_fidelity)
    int patch_mode;      // two-seed advect kernel, seeds of a wave / form of the trajectory stores: -1 by call (default: whole-line stores through LDS with trajectories, tall patches without, groups of members for lc_advect_batch), 0 tall, 1 wide, 2 lines, 3 two ensemble members per lane (LCS_PATCH_MODE at creation; advect.hip enum Patch)
    lc_flag_allreduce_fn flag_reduce;  // NULL, or the caller's max-all-reduce over the ranks of a row-sharded grid (lc_ctx_set_flag_allreduce)
    void *flag_reduce_user;
    int last_advect_launches;  // kernel launches the last lc_advect made (level chunks)
    const char *last_advect_kernel;
    const char *last_sigma_kernel;
    const char *last_pack_kernel;   // what the last lc_field_pack launched for the prefilter / interleave stage (lc_ctx_last_pack_kernel)
    unsigned *verify_dev;  // NULL, or 16 uint32 wave-state counters in device memory (lc_ctx_set_verify)
    lc_trunc_cache *trunc;
    struct lc_host_xfer *xfer;  // NULL until a one-call host route first needs it: the pinned staging ring + its threads (hostxfer.h)
    void *host_ws;              // lc_lcs_host's device buffers of the last call, kept for the next one (api.hip: HostWorkspace; lc_ctx_trim frees them)
    int host_cache;             // 1 (default): keep them; 0: hipMalloc / hipFree per call (LCS_HOST_CACHE at creation)
    double host_marks[4];       // the last lc_lcs_host call, ms since its entry: buffers allocated, uploads + launches issued, kernels done, results in the caller's buffers (lc_ctx_last_host_marks)
    int host_timing;            // lc_lcs_host: 1 = one line of wall-clock marks per call on stderr (LCS_HOST_TIMING at creation)
    int f64_wg_tile;            // float64 order 1, fused levels: 1 = one LDS tile per workgroup (advect_wg64_kernel; LCS_F64_WG_TILE at creation; measured, off)
    int host_threads;           // staging ring: worker threads beside the caller (-1: by the host's core count; LCS_HOST_THREADS at creation)
    int host_piece_mb;          // staging ring: piece size in MB (0: 32; LCS_HOST_PIECE_MB at creation)
    int host_pipeline;          // lc_lcs_host: 1 (default) staged transfers, upload cut into level chunks and overlapped with pack + advect; 0 the serial round-5 form (LCS_HOST_PIPELINE at creation)
};

void lc_set_error(const char *fmt, ...);

// (a failed runtime call also leaves its code in the thread's sticky "last error", which the NEXT lc_* call's
//  hipGetLastError() after its kernel launches would report as its own: consumed here, with the failure it belongs to --
//  found by tests/test_host_orchestration_asan.py's injected failures)
#define LC_HIP_CHECK(expr)                                                              \
    do {                                                                                \
        hipError_t _e = (expr);                                                         \
        if (_e != hipSuccess) {                                                         \
            lc_set_error("%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__, \
                         __LINE__);                                                     \
            (void)hipGetLastError();                                                    \
            return LC_EHIP;                                                             \
        }                                                                               \
    } while (0)

#define LC_REQUIRE(cond, ...)          \
    do {                               \
        if (!(cond)) {                 \
            lc_set_error(__VA_ARGS__); \
            return LC_EINVAL;          \
        }                              \
    } while (0)

// Padded gather image geometry (see lc_field_pack in lcs_hip.h).
constexpr int LC_PAD_LO = 1;
constexpr int LC_PAD_HI = 2;
constexpr int LC_PAD = LC_PAD_LO + LC_PAD_HI;

static inline size_t lc_level_elems(int ny_f, int nx_f) {
    return (size_t)(ny_f + LC_PAD) * (size_t)(nx_f + LC_PAD) * 2;
}

// kernel launchers implemented in the .hip files
int lc_launch_pack(lc_ctx *ctx, const void *u, const void *v, int dtype, int nt, int ny_f, int nx_f,
                   int order, void *packed, void *ext);
int lc_launch_extrapolate(lc_ctx *ctx, const void *img, int dtype, int nt, int ny_f, int nx_f, void *ext);
