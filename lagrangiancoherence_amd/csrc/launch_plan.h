// Launch bookkeeping of lc_advect as pure functions: no HIP call, no pointer, no context -- integers in, integers out.
// advect.hip's host launcher (advect_impl) and its kernels (tile order, member windows) call these; the CPU suite compiles
// the same header with g++ -fsanitize=address,undefined and checks the invariants the kernels rely on
// (tests/c/launch_plan_test.cpp, tests/test_launch_plan.py).  No reference counterpart: the reference's loop over time
// levels (LCS/trajectory.py:80-126) is one Python loop over whole arrays; this is how that loop is cut into launches.
#pragma once
#include <cstddef>

#if defined(__HIPCC__)
#define LCP_HD __host__ __device__ __forceinline__
#else
#define LCP_HD inline
#endif

namespace lcplan {

constexpr int XCDS = 8;  // MI355X: workgroups are dealt to the 8 XCDs round-robin (blockIdx % 8), each XCD has its own L2

LCP_HD int imin(int a, int b) { return a < b ? a : b; }
LCP_HD int imax(int a, int b) { return a > b ? a : b; }

// ---- XCD-aware block -> tile map --------------------------------------------------------------------------------------
// Blocks of a launch beyond the leading pole blocks: b = blockIdx.x - pole_blocks (pole_blocks is a multiple of 8, so
// b % 8 is still the XCD).  xcd_chunk = 0: XCD x takes the x-th contiguous eighth of the tiles.  xcd_chunk = C: chunks of
// C tiles (whole tile rows) go to the XCDs cyclically.  Returns the tile a block works on; a value >= ntiles means the
// block has no tile (the grid is rounded up).
// tile_order: 0 as stored; 1 the last tile row first, then 0, 1, 2, ...; 2 from the poles inwards (last, 0, last-1, 1, ...);
// 3 eight rows from the top, eight from the bottom, the next eight from the top, ... (snake_row) -- with whole tile rows dealt to
// the XCDs cyclically, XCD x then holds the x-th row from the top AND the x-th from the bottom: where the rows' cost rises
// towards ONE end (a row block of a sharded grid: one rank's latitudes) every XCD gets a dear row with a cheap one.  Order 2 is
// for grids with two dear ends (the whole globe): there order 3 would hand XCD 0 both polar-most rows.
LCP_HD int snake_row(int dr, int nty) {
    const int a = dr / XCDS, x = dr - a * XCDS, k = (a >> 1) * XCDS + x;
    return (a & 1) ? k : nty - 1 - k;
}
LCP_HD int tile_of_block(int b, int ntiles, int ntx, int xcd_chunk, int tile_order) {
    const int xcd = b % XCDS, j = b / XCDS;
    if (xcd_chunk <= 0) return xcd * ((ntiles + XCDS - 1) / XCDS) + j;
    const int cj = j / xcd_chunk, r = j - cj * xcd_chunk;
    const int d = (cj * XCDS + xcd) * xcd_chunk + r;  // position in dispatch order
    if (tile_order && d < ntiles) {
        const int dr = d / ntx, c = d - dr * ntx, nty = ntiles / ntx;
        const int row = tile_order == 3 ? snake_row(dr, nty)
                        : tile_order == 2 ? ((dr & 1) ? (dr >> 1) : nty - 1 - (dr >> 1)) : (dr == 0 ? nty - 1 : dr - 1);
        return row * ntx + c;
    }
    return d;
}
// Tiles per XCD chunk.  xcd_rows whole tile rows (0: contiguous bands) go to the XCDs cyclically -- which leaves one XCD with
// an extra chunk whenever the number of chunks is not a multiple of 8: 17 tile rows are 3 + 7 x 2, the launch ends when the
// XCD with 3 does (1084 x 8192 seeds: 7.19 ms against 5.06).  xcd_split > 0: a chunk is a 1 / xcd_split part of that (split
// 8: every tile row is dealt to all eight XCDs in ntx / 8-tile segments: equal shares whatever the row count, at ~4 % on
// launches that divide evenly anyway -- neighbouring tiles of a row share wind nodes, and eight times as many of them
// then sit on different L2s).  xcd_split < 0 (the default): split 8 exactly when whole chunks would leave the XCDs more
// than 15 % apart (profiles/r05/xcd_split_ab.txt).
// prefilter_fused_stream_kernel: waves per workgroup.  A workgroup of nw waves keeps 32 (nw - 2) columns (its first and last
// wave are halo); a line of nx columns takes ceil(nx / kept) workgroups.  The choice marches the fewest columns in all
// (ties: the larger workgroup -- fewer halo waves per kept column at the same cost).
LCP_HD int fused_prefilter_waves(int nx, int max_waves) {
    int best = 4, best_cost = 1 << 30;
    for (int nw = 4; nw <= max_waves; ++nw) {
        const int kept = 32 * (nw - 2), cost = (nx + kept - 1) / kept * nw;
        if (cost <= best_cost) best = nw, best_cost = cost;
    }
    return best;
}

// prefilter_fused_stream_kernel: row pieces of the workgroups that do not fill a round.  One workgroup per CU at a time, all
// the same length: `items` (level, x block) columns-of-workgroups on `cus` CUs run in ceil(items / cus) rounds, the last one
// with (items % cus) workgroups on an otherwise idle chip.  Those last ones are cut into row pieces so that the last round has
// up to `cus` shorter workgroups.  A piece starts the latitude march at its first row from the 64 rows above it
// (fused_causal_restart) -- and so that a level's bits do NOT depend on whether, or where, this launch happened to cut it
// (which follows from nt and the CU count: round 5's advisor finding), EVERY march restarts the same way at every row that
// is a multiple of FUSED_PIECE_ALIGN, cut there or not, and pieces begin on such rows only.  Returns the pieces per leftover
// item (1: no split).
constexpr int FUSED_PIECE_ALIGN = 256;
struct FusedSplit { int n_whole, pieces, piece_rows; };   // items run whole; pieces of each of the others; rows per piece (a multiple of FUSED_PIECE_ALIGN)
LCP_HD FusedSplit fused_prefilter_split(int items, int cus, int ny) {
    FusedSplit f;
    const int left = cus > 0 ? items % cus : 0;
    f.n_whole = items - left;
    f.pieces = 1;
    if (left > 0) f.pieces = imax(1, imin(cus / left, ny / FUSED_PIECE_ALIGN));
    f.piece_rows = ((ny + f.pieces - 1) / f.pieces + FUSED_PIECE_ALIGN - 1) / FUSED_PIECE_ALIGN * FUSED_PIECE_ALIGN;
    while (f.pieces > 1 && f.piece_rows * (f.pieces - 1) >= ny) --f.pieces;   // (rounding up may leave the last piece empty)
    return f;
}

LCP_HD int xcd_chunk_tiles(int ntx, int nty, int xcd_rows, int xcd_split) {
    if (xcd_rows <= 0) return 0;
    const int whole = xcd_rows * ntx;
    if (xcd_split < 0) {
        const int nch = (nty + xcd_rows - 1) / xcd_rows, full = (nch + XCDS - 1) / XCDS * XCDS;
        xcd_split = (full - nch) * 100 > 15 * nch ? XCDS : 0;
    }
    return xcd_split > 0 ? imax((whole + xcd_split - 1) / xcd_split, 1) : whole;
}
// Blocks to launch so that every tile has one (excluding the pole blocks).
LCP_HD int xcd_grid(int ntiles, int xcd_chunk) {
    if (xcd_chunk <= 0) return ((ntiles + XCDS - 1) / XCDS) * XCDS;
    const int nch = (ntiles + xcd_chunk - 1) / xcd_chunk;
    return ((nch + XCDS - 1) / XCDS) * XCDS * xcd_chunk;
}

// ---- pole rows -----------------------------------------------------------------------------------------------------------
// The first / last `order` GLOBAL seed rows take order 1 + 'constant' (LCS/tools.py:24-39, Q3).  Of the local block
// [row0, row0 + ny): lo of them at its start, hi at its end; `blocks` leading workgroups (a multiple of 8, `block` seeds
// each) take them one seed per thread; 0 blocks = the tiles do.
struct PoleRows {
    int lo, hi, blocks;
};
LCP_HD PoleRows pole_rows(int order, int row0, int ny, int ny_global, int nx, bool enabled, int block) {
    const int lo = imin(imax(order - row0, 0), ny), hi = imin(imax(row0 + ny - (ny_global - order), 0), ny);
    const long long npole = (long long)(lo + hi) * nx;
    const bool on = enabled && lo + hi <= ny && npole > 0 && npole < (1ll << 30);
    PoleRows p;
    p.lo = on ? lo : 0;
    p.hi = on ? hi : 0;
    p.blocks = on ? (int)(((npole + block - 1) / block + XCDS - 1) / XCDS * XCDS) : 0;
    return p;
}
// Local row of the k-th pole seed row (k < lo + hi).
LCP_HD int pole_row(int k, int lo, int hi, int ny) { return k < lo ? k : ny - hi + (k - lo); }

// ---- level chunks ---------------------------------------------------------------------------------------------------------
// A series of `total` levels runs as consecutive launches of at most `chunk` levels (lc_ctx_set_level_chunk).
// By SETTLS_order: 32 levels per launch for K >= 3, 64 for K = 2, one launch for K <= 1 (profiles/r03).
// float64 at order 1 with seeds on (or as dense as) the field's nodes -- BASELINE configs[1] -- wants HALF of that: its tiles are
// fetched for one patch each (nothing to share between workgroups), what a chunk buys is the launch's shorter tail
// (profiles/r06/level_chunk_sweep.txt: 8 / 12 / 16 / 20 / 24 / 32 levels: 3.02 / 2.94 / 2.91 / 2.92 / 2.95 / 3.09 ms;
// order 3 and the float32 kernels: flat or best at 32-48).
LCP_HD int chunk_for_k(int K, bool f64_order1 = false) { return K >= 3 ? (f64_order1 ? 16 : 32) : (K == 2 ? 64 : 0); }
constexpr long long CHUNK_FROM_SEEDS = 1ll << 18;  // seeds per call from which the default chunks (measured)
constexpr int OUTER_CHUNK = 16;                    // LC_X_CLAMP_REFERENCE_OUTER: levels between two reads of the clamp flag

// Levels per launch.  ctx_level_chunk: -1 by size (default), 0 one launch, n > 0 at most n levels.
// `outer` (LC_X_CLAMP_REFERENCE_OUTER): every chunk ends with one flag all-reduce over the ranks of a row-sharded grid, so
// the value MUST NOT depend on the local block (seeds_local differs between ranks whose row counts differ by one, or
// by the halo rows they advect redundantly): fixed 16 levels unless the caller set a value (which it sets on every rank).
// It also makes the fused-prefix / exact-suffix split of a sharded run the split of the unsharded run (float32 results
// stay bit-identical).
LCP_HD int level_chunk(int ctx_level_chunk, bool outer, long long seeds_local, int n_members, int K, int total, bool f64_order1 = false) {
    const int all = total > 0 ? total : 1;
    if (outer) return ctx_level_chunk > 0 ? ctx_level_chunk : OUTER_CHUNK;
    int want = ctx_level_chunk;
    if (want < 0) want = seeds_local * (n_members > 1 ? n_members : 1) >= CHUNK_FROM_SEEDS ? chunk_for_k(K, f64_order1) : 0;
    return want > 0 ? want : all;
}
// Number of launches and the i-th launch's first level / level count.  A call with total = 0 still makes one (empty)
// launch: it stores the start positions.
LCP_HD int n_chunks(int total, int chunk) { return total <= 0 ? 1 : (total + chunk - 1) / chunk; }
LCP_HD int chunk_first(int i, int chunk) { return i * chunk; }
LCP_HD int chunk_levels(int i, int total, int chunk) {
    const int left = total - i * chunk;
    return left < chunk ? imax(left, 0) : chunk;
}

// ---- member groups (lc_advect_batch, two members per lane) ---------------------------------------------------------------
// Members 2p and 2p+1 share a lane.  A group's LEVEL window is [0, nsteps + (g - 1) d): member q of the group steps at the
// window levels [q d, q d + nsteps) (d = t0_stride).  Grouping only pays (and is only correct as implemented) when the
// members overlap in time: nsteps > d.
struct Groups {
    int g;            // members per lane: 0 = no groups, else 2
    int total;        // levels the launches walk
    int n_groups;     // grid.y
    int last;         // members of the last group (1 .. g)
    int group_stride; // start-level distance of consecutive groups
};
LCP_HD Groups member_groups(int n_members, int t0_stride, int nsteps, bool eligible) {
    Groups G;
    G.g = (eligible && n_members > 1 && nsteps > t0_stride) ? 2 : 0;
    G.total = G.g ? nsteps + (G.g - 1) * t0_stride : nsteps;
    G.n_groups = G.g ? (n_members + G.g - 1) / G.g : n_members;
    G.last = G.g ? (n_members % G.g ? n_members % G.g : G.g) : 1;
    G.group_stride = G.g ? G.g * t0_stride : t0_stride;
    return G;
}
// Window levels [lo, hi) of member q's own steps inside a launch that covers the window levels [l0, l0 + n) (pair_n =
// the member's step count, d = t0_stride); empty when hi <= lo.
struct Window {
    int lo, hi;
};
LCP_HD Window member_window(int q, int l0, int n, int pair_n, int d) {
    Window w;
    w.lo = imax(l0, q * d);
    w.hi = imin(l0 + n, pair_n + q * d);
    return w;
}
// Does member q take a step at level s of a launch that starts at window level l0?
LCP_HD bool member_steps(int q, int s, int l0, int pair_n, int d) {
    const int l = l0 + s - q * d;
    return l >= 0 && l < pair_n;
}
// Levels of a launch in which ANY of the group's cnt members still steps (the kernel's loop count).
LCP_HD int group_levels(int n, int l0, int pair_n, int d, int cnt) { return imin(n, imax(pair_n + (cnt - 1) * d - l0, 0)); }

// ---- LC_X_CLAMP_REFERENCE_OUTER ------------------------------------------------------------------------------------------
// The fused kernel ran chunk [s0, ...) and a parcel left the box: the sub-step path restarts at s0 from the positions
// saved before that chunk, or from the seed grid when they could not be kept.
LCP_HD int outer_restart(int s0, bool have_saved) { return (s0 > 0 && !have_saved) ? 0 : s0; }

// ---- tile grids ----------------------------------------------------------------------------------------------------------
// Tiles of w x h seeds over an ny x nx block, XCD chunks of xcd_rows tile rows.
struct TileGrid {
    int ntx, nty, ntiles, xcd_chunk, grid;  // grid: blocks incl. the pole blocks
};
LCP_HD TileGrid tile_grid(int ny, int nx, int w, int h, int xcd_rows, int pole_blocks, int xcd_split = 0) {
    TileGrid t;
    t.ntx = (nx + w - 1) / w;
    t.nty = (ny + h - 1) / h;
    t.ntiles = t.ntx * t.nty;
    t.xcd_chunk = xcd_chunk_tiles(t.ntx, t.nty, xcd_rows, xcd_split);
    t.grid = xcd_grid(t.ntiles, t.xcd_chunk) + pole_blocks;
    return t;
}

}  // namespace lcplan
