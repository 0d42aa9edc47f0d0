// CPU unit test of lagrangiancoherence_amd/csrc/launch_plan.h -- the integer bookkeeping lc_advect's launcher and
// kernels share (level chunks, member-pair windows, XCD tile order, pole blocks, outer-clamp restart).
// Built by tests/test_launch_plan.py with  g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all
// (sanitizers on the host logic; the GPU has no sanitizer on this pool).
#include <cstdio>
#include <cstdlib>
#include <vector>

#include "../../lagrangiancoherence_amd/csrc/launch_plan.h"

static int g_fail = 0;
#define CHECK(cond, ...)                                          \
    do {                                                          \
        if (!(cond)) {                                            \
            if (++g_fail <= 20) {                                 \
                std::printf("FAIL %s:%d  %s  ", __FILE__, __LINE__, #cond); \
                std::printf(__VA_ARGS__);                         \
                std::printf("\n");                                \
            }                                                     \
        }                                                         \
    } while (0)

using namespace lcplan;

// Every tile gets exactly one block; a tile ROW's blocks sit on one XCD when whole rows are dealt (xcd_chunk = rows * ntx).
static void test_tile_order() {
    long cases = 0;
    for (int ntx = 1; ntx <= 40; ntx += (ntx < 8 ? 1 : 7))
        for (int nty : {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 16, 19, 28, 32, 37, 46, 48, 55, 64})
            for (int rows : {0, 1, 2, 4, 8})
                for (int order = 0; order <= 3; ++order) {
                    const int ntiles = ntx * nty, chunk = rows * ntx;
                    const int grid = xcd_grid(ntiles, chunk);
                    CHECK(grid % XCDS == 0 && grid >= ntiles, "grid %d ntiles %d", grid, ntiles);
                    std::vector<int> seen(ntiles, 0), xcd_of_row(nty, -1);
                    for (int b = 0; b < grid; ++b) {
                        const int t = tile_of_block(b, ntiles, ntx, chunk, order);
                        CHECK(t >= 0, "negative tile %d", t);
                        if (t < ntiles) {
                            ++seen[t];
                            if (chunk > 0) {
                                const int row = t / ntx;
                                if (xcd_of_row[row] < 0) xcd_of_row[row] = b % XCDS;
                                CHECK(xcd_of_row[row] == b % XCDS, "tile row %d on XCDs %d and %d (ntx %d nty %d rows %d order %d)",
                                      row, xcd_of_row[row], b % XCDS, ntx, nty, rows, order);
                            }
                        }
                    }
                    for (int t = 0; t < ntiles; ++t)
                        CHECK(seen[t] == 1, "tile %d visited %d times (ntx %d nty %d rows %d order %d)", t, seen[t], ntx, nty, rows, order);
                    // poles-inwards order: the first dispatched tile row is the last one, the second is row 0
                    if (chunk > 0 && order == 2 && nty >= 2) {
                        CHECK(tile_of_block(0, ntiles, ntx, chunk, 2) / ntx == nty - 1, "order 2 starts at row %d", tile_of_block(0, ntiles, ntx, chunk, 2) / ntx);
                    }
                    if (chunk > 0 && order == 1 && nty >= 2)
                        CHECK(tile_of_block(0, ntiles, ntx, chunk, 1) / ntx == nty - 1, "order 1 starts at the last row");
                    // snake order, whole rows per XCD: XCD x holds the x-th row from the top and the x-th from the bottom, ... --
                    // the rows of one XCD pair up across the middle of the block (their indices sum to nty - 1 wherever both
                    // rounds are whole)
                    if (rows == 1 && order == 3 && nty % (2 * XCDS) == 0)
                        for (int x = 0; x < XCDS; ++x) {
                            long sum = 0, n = 0;
                            for (int r = 0; r < nty; ++r)
                                if (xcd_of_row[r] == x) sum += r, ++n;
                            CHECK(n == nty / XCDS && 2 * sum == n * (nty - 1), "snake order: XCD %d holds %ld rows summing to %ld (nty %d)", x, n, sum, nty);
                        }
                    ++cases;
                }
    // chunks that are PARTS of a tile row (xcd_chunk_tiles with a split): still one block per tile, and the XCDs' shares of
    // the tiles differ by at most one chunk -- whatever the number of tile rows (whole rows: 17 rows are 3 + 7 x 2)
    for (int ntx = 1; ntx <= 1100; ntx = ntx * 3 + 1)
        for (int nty = 1; nty <= 70; nty += (nty < 20 ? 1 : 7))
            for (int split : {-1, 0, 4, 8})
                for (int order = 0; order <= 3; ++order) {
                    const int ntiles = ntx * nty, chunk = xcd_chunk_tiles(ntx, nty, 1, split);
                    CHECK(chunk >= 1 && chunk <= ntx, "chunk %d of a row of %d tiles", chunk, ntx);
                    if (split == 0) CHECK(chunk == ntx, "no split: whole rows");
                    if (split < 0) {   // the automatic rule: split exactly when whole rows would leave the XCDs > 15 % apart
                        const int full = (nty + XCDS - 1) / XCDS * XCDS;
                        CHECK((chunk < ntx || ntx < XCDS) == ((full - nty) * 100 > 15 * nty) || ntx < XCDS, "auto split: nty %d ntx %d chunk %d", nty, ntx, chunk);
                    }
                    const int grid = xcd_grid(ntiles, chunk);
                    CHECK(grid % XCDS == 0 && grid >= ntiles, "grid %d ntiles %d", grid, ntiles);
                    std::vector<int> seen(ntiles, 0);
                    long per_xcd[XCDS] = {0};
                    for (int b = 0; b < grid; ++b) {
                        const int t = tile_of_block(b, ntiles, ntx, chunk, order);
                        CHECK(t >= 0, "negative tile %d", t);
                        if (t < ntiles) {
                            ++seen[t];
                            ++per_xcd[b % XCDS];
                        }
                    }
                    for (int t = 0; t < ntiles; ++t) CHECK(seen[t] == 1, "tile %d visited %d times (ntx %d nty %d split %d)", t, seen[t], ntx, nty, split);
                    long lo = per_xcd[0], hi = per_xcd[0];
                    for (int x = 1; x < XCDS; ++x) lo = per_xcd[x] < lo ? per_xcd[x] : lo, hi = per_xcd[x] > hi ? per_xcd[x] : hi;
                    CHECK(hi - lo <= chunk, "XCD shares %ld .. %ld with chunks of %d (ntx %d nty %d split %d)", lo, hi, chunk, ntx, nty, split);
                    ++cases;
                }
    std::printf("tile order: %ld launch shapes\n", cases);
    // the fused float64 prefilter (pack.hip): waves per workgroup and the row pieces of the last round
    cases = 0;
    for (int nx = 64; nx <= 6000; nx += (nx < 200 ? 1 : 37)) {
        const int nw = fused_prefilter_waves(nx, 12), kept = 32 * (nw - 2), wgs = (nx + kept - 1) / kept;
        CHECK(nw >= 4 && nw <= 12 && wgs * kept >= nx && (wgs - 1) * kept < nx, "nx %d: %d waves", nx, nw);
        for (int other = 4; other <= 12; ++other)   // no other workgroup size marches fewer columns
            CHECK(wgs * nw <= (nx + 32 * (other - 2) - 1) / (32 * (other - 2)) * other, "nx %d: %d waves march more than %d", nx, nw, other);
        ++cases;
    }
    CHECK(fused_prefilter_waves(1024, 12) == 10 && fused_prefilter_waves(1440, 12) == 11, "BASELINE widths");
    for (int items = 1; items <= 3000; items += (items < 600 ? 1 : 13))
        for (int cus : {1, 8, 104, 256, 304})
            for (int ny : {64, 100, 128, 255, 256, 720, 1024, 4097}) {
                const FusedSplit f = fused_prefilter_split(items, cus, ny);
                CHECK(f.n_whole % cus == 0 && f.n_whole <= items && items - f.n_whole < cus, "items %d cus %d: %d whole", items, cus, f.n_whole);
                // pieces begin on the rows at which EVERY march restarts (a level's bits must not depend on the cut)
                CHECK(f.pieces >= 1 && f.piece_rows % FUSED_PIECE_ALIGN == 0, "pieces %d of %d rows", f.pieces, f.piece_rows);
                CHECK((long)f.pieces * f.piece_rows >= ny && (long)(f.pieces - 1) * f.piece_rows < ny, "pieces %d x %d rows cover %d", f.pieces, f.piece_rows, ny);
                CHECK(f.pieces == 1 || f.piece_rows >= 128, "piece of %d rows", f.piece_rows);     // (a piece restarts 64 rows above its own)
                CHECK((long)(items - f.n_whole) * f.pieces <= (cus > items - f.n_whole ? cus : items - f.n_whole), "last round: %d x %d on %d CUs", items - f.n_whole, f.pieces, cus);
                ++cases;
            }
    {
        const FusedSplit f = fused_prefilter_split(804, 256, 1024);   // BASELINE configs[1] at order 3: 201 levels x 4 workgroups
        CHECK(f.n_whole == 768 && f.pieces == 4 && f.piece_rows == 256, "configs[1]: %d whole, %d pieces of %d rows", f.n_whole, f.pieces, f.piece_rows);
    }
    std::printf("fused prefilter: %ld shapes\n", cases);
}

// Over any partition of the global grid into row blocks, the pole rows the blocks take are the global first / last
// `order` rows, each once; the leading workgroups cover them.
static void test_pole_rows() {
    const int BLOCK = 256;
    long cases = 0;
    for (int order = 1; order <= 5; ++order)
        for (int nyg : {8, 11, 64, 203, 1024})
            for (int world = 1; world <= 5; ++world)
                for (int nx : {1, 7, 320, 4096}) {
                    if (nyg < world * 2) continue;
                    std::vector<int> taken(nyg, 0);
                    for (int r = 0; r < world; ++r) {
                        const int base = nyg / world, rem = nyg % world;
                        const int lo = r * base + (r < rem ? r : rem), ny = base + (r < rem ? 1 : 0);
                        const PoleRows p = pole_rows(order, lo, ny, nyg, nx, true, BLOCK);
                        int want_lo = 0, want_hi = 0;
                        for (int i = 0; i < ny; ++i) {
                            const int g = lo + i;
                            if (g < order) ++want_lo;
                            else if (g >= nyg - order) ++want_hi;
                        }
                        // rows that are in both classes (grids thinner than 2 * order) count once, as "lo"
                        if (want_lo + want_hi <= ny && (long long)(want_lo + want_hi) * nx > 0) {
                            // the function counts a row of a thin grid in both classes; it then turns itself off
                            const int lo_f = imin(imax(order - lo, 0), ny), hi_f = imin(imax(lo + ny - (nyg - order), 0), ny);
                            if (lo_f + hi_f <= ny) {
                                CHECK(p.lo == lo_f && p.hi == hi_f, "pole rows %d/%d want %d/%d", p.lo, p.hi, lo_f, hi_f);
                                CHECK(p.blocks % XCDS == 0 && (long long)p.blocks * BLOCK >= (long long)(p.lo + p.hi) * nx, "blocks %d", p.blocks);
                                for (int k = 0; k < p.lo + p.hi; ++k) {
                                    const int iy = pole_row(k, p.lo, p.hi, ny);
                                    CHECK(iy >= 0 && iy < ny, "pole row %d outside the block", iy);
                                    const int g = lo + iy;
                                    CHECK(g < order || g >= nyg - order, "row %d is not a pole row", g);
                                    ++taken[g];
                                }
                            } else {
                                CHECK(p.blocks == 0 && p.lo == 0 && p.hi == 0, "overlapping classes must turn the pole blocks off");
                            }
                        }
                        const PoleRows off = pole_rows(order, lo, ny, nyg, nx, false, BLOCK);
                        CHECK(off.blocks == 0 && off.lo == 0 && off.hi == 0, "disabled");
                    }
                    if (nyg >= 2 * order * world + 2 * order)   // blocks thick enough that no block mixes the classes badly
                        for (int g = 0; g < nyg; ++g) {
                            const bool pole = g < order || g >= nyg - order;
                            CHECK(taken[g] == (pole ? 1 : 0), "global row %d taken %d times (order %d nyg %d world %d)", g, taken[g], order, nyg, world);
                        }
                    ++cases;
                }
    std::printf("pole rows: %ld partitions\n", cases);
}

static void test_level_chunks() {
    // by size: 2^18 seeds, SETTLS_order rule
    CHECK(level_chunk(-1, false, (1ll << 18) - 1, 1, 4, 200) == 200, "below 2^18 seeds: one launch");
    CHECK(level_chunk(-1, false, 1ll << 18, 1, 4, 200) == 32, "K = 4: 32");
    CHECK(level_chunk(-1, false, 1ll << 18, 1, 3, 200) == 32, "K = 3: 32");
    CHECK(level_chunk(-1, false, 1ll << 20, 1, 4, 200, true) == 16 && level_chunk(-1, false, 1ll << 20, 1, 2, 200, true) == 64, "float64 at order 1: 16 for K >= 3");
    CHECK(level_chunk(-1, true, 1ll << 20, 1, 4, 200, true) == 16 && level_chunk(24, false, 1ll << 20, 1, 4, 200, true) == 24, "outer / explicit values do not change");
    CHECK(level_chunk(-1, false, 1ll << 18, 1, 2, 200) == 64, "K = 2: 64");
    CHECK(level_chunk(-1, false, 1ll << 24, 1, 1, 200) == 200, "K = 1: one launch");
    CHECK(level_chunk(-1, false, 1ll << 24, 1, 0, 200) == 200, "K = 0: one launch");
    CHECK(level_chunk(-1, false, 1ll << 15, 8, 4, 200) == 32, "an ensemble counts its members' seeds");
    CHECK(level_chunk(0, false, 1ll << 24, 1, 4, 200) == 200, "0 = one launch");
    CHECK(level_chunk(16, false, 100, 1, 0, 200) == 16, "explicit");
    CHECK(level_chunk(-1, false, 100, 1, 4, 0) == 1, "empty series: chunk 1 (one empty launch)");
    // LC_X_CLAMP_REFERENCE_OUTER: rank-invariant.  511 x 1024 seeds over 2 ranks (round-3 advisor case): 256 and 255 rows
    // straddle 2^18 seeds; with redundant halos 258 and 257.  Every rank must choose the same chunk -> same number of
    // flag all-reduces.
    for (int K = 0; K <= 4; ++K)
        for (int set : {-1, 0, 8, 16, 32}) {
            const int a = level_chunk(set, true, 256ll * 1024, 1, K, 200), b = level_chunk(set, true, 255ll * 1024, 1, K, 200);
            const int c = level_chunk(set, true, 511ll * 1024, 1, K, 200), d = level_chunk(set, true, 4ll, 1, K, 200);
            CHECK(a == b && b == c && c == d, "outer chunk depends on the block: %d %d %d %d (K %d set %d)", a, b, c, d, K, set);
            CHECK(a == (set > 0 ? set : OUTER_CHUNK), "outer chunk %d", a);
            CHECK(n_chunks(200, a) == n_chunks(200, b), "ranks would issue different numbers of collectives");
        }
    // chunks tile [0, total) in order
    for (int total = 0; total <= 300; ++total)
        for (int chunk = 1; chunk <= 70; ++chunk) {
            const int n = n_chunks(total, chunk);
            CHECK(n >= 1, "at least one launch");
            int next = 0;
            for (int i = 0; i < n; ++i) {
                CHECK(chunk_first(i, chunk) == next, "gap before chunk %d", i);
                const int len = chunk_levels(i, total, chunk);
                CHECK(len <= chunk && len >= 0 && (len > 0 || total == 0), "chunk %d has %d levels", i, len);
                next += len;
            }
            CHECK(next == total, "chunks cover %d of %d levels", next, total);
        }
    std::printf("level chunks ok\n");
}

// lc_advect_batch with two members per lane: across the launches of a call every member takes its steps 0 .. nsteps-1
// exactly once and in order, at the field levels lc_advect(t0 + m * stride) would read; no launch walks a level at
// which no member of the group steps; no level beyond the series' bound is touched.
static void test_member_groups() {
    long cases = 0;
    for (int n_members = 1; n_members <= 9; ++n_members)
        for (int d = 0; d <= 3; ++d)
            for (int nsteps : {1, 2, 3, 5, 16, 17, 40})
                for (int chunk : {1, 3, 7, 16, 32, 1000}) {
                    const Groups G = member_groups(n_members, d, nsteps, true);
                    if (n_members == 1 || nsteps <= d) {
                        CHECK(G.g == 0 && G.total == nsteps && G.n_groups == n_members, "no groups expected");
                        continue;
                    }
                    CHECK(G.g == 2 && G.total == nsteps + d, "pairs");
                    CHECK(G.n_groups == (n_members + 1) / 2 && G.last == (n_members % 2 ? 1 : 2) && G.group_stride == 2 * d, "group geometry");
                    CHECK(!member_groups(n_members, d, nsteps, false).g, "not eligible: no groups");
                    const int t0 = 5, level_bound = t0 + (n_members - 1) * d + nsteps;   // lc_advect_batch: <= nt - 1
                    for (int grp = 0; grp < G.n_groups; ++grp) {
                        const int cnt = grp + 1 == G.n_groups ? G.last : G.g;
                        std::vector<int> next_step(cnt, 0);
                        const int nl = n_chunks(G.total, chunk);
                        for (int ci = 0; ci < nl; ++ci) {
                            const int l0 = chunk_first(ci, chunk), n = chunk_levels(ci, G.total, chunk);
                            const int launch_t0 = t0 + grp * G.group_stride + l0;   // for_member + chunk: field level of the launch's s = 0
                            const int nlev = group_levels(n, l0, nsteps, d, cnt);
                            CHECK(nlev >= 0 && nlev <= n, "nlev %d of %d", nlev, n);
                            for (int s = 0; s < n; ++s) {
                                bool any = false;
                                for (int q = 0; q < cnt; ++q) {
                                    const bool act = member_steps(q, s, l0, nsteps, d);
                                    const Window w = member_window(q, l0, n, nsteps, d);
                                    CHECK(act == (l0 + s >= w.lo && l0 + s < w.hi), "member_window and member_steps disagree");
                                    if (!act) continue;
                                    any = true;
                                    const int step = l0 + s - q * d;
                                    CHECK(step == next_step[q], "member %d takes step %d, expected %d", grp * 2 + q, step, next_step[q]);
                                    ++next_step[q];
                                    // the field level that step reads = what lc_advect(t0 + m * d) reads at its step
                                    const int m = grp * G.g + q;
                                    CHECK(launch_t0 + s == t0 + m * d + step, "member %d step %d reads level %d, lc_advect reads %d", m, step,
                                          launch_t0 + s, t0 + m * d + step);
                                    CHECK(launch_t0 + s + 1 <= level_bound, "level %d beyond the bound %d", launch_t0 + s + 1, level_bound);
                                }
                                CHECK(any == (s < nlev), "level %d of a launch: active %d, inside the kernel's loop %d", s, (int)any, (int)(s < nlev));
                            }
                        }
                        for (int q = 0; q < cnt; ++q) CHECK(next_step[q] == nsteps, "member %d took %d of %d steps", grp * 2 + q, next_step[q], nsteps);
                    }
                    ++cases;
                }
    std::printf("member groups: %ld ensembles\n", cases);
}

static void test_misc() {
    CHECK(outer_restart(0, false) == 0 && outer_restart(0, true) == 0, "first chunk: from the seed grid");
    CHECK(outer_restart(32, true) == 32, "saved positions: restart at the chunk");
    CHECK(outer_restart(32, false) == 0, "no saved positions: from the seed grid");
    for (int ny : {1, 8, 63, 64, 65, 4096})
        for (int nx : {1, 7, 8, 9, 4096})
            for (int rows : {0, 1, 2}) {
                const TileGrid t = tile_grid(ny, nx, 8, 32, rows, 16);
                CHECK(t.ntx * 8 >= nx && (t.ntx - 1) * 8 < nx && t.nty * 32 >= ny && (t.nty - 1) * 32 < ny, "tiles cover the block");
                CHECK(t.ntiles == t.ntx * t.nty && t.xcd_chunk == rows * t.ntx, "tile grid");
                CHECK(t.grid >= t.ntiles + 16 && (t.grid - 16) % XCDS == 0, "grid %d", t.grid);
            }
    std::printf("misc ok\n");
}

int main() {
    test_tile_order();
    test_pole_rows();
    test_level_chunks();
    test_member_groups();
    test_misc();
    if (g_fail) {
        std::printf("%d check(s) failed\n", g_fail);
        return 1;
    }
    std::printf("launch_plan: all checks passed\n");
    return 0;
}
