// K1 -- parcel advection: Euler + K accumulate-"SETTLS" sub-steps per time level,
// all time levels fused in one launch, one thread per seed, positions in registers.
//
// Restates trajectory.parcel_propagation (LCS/trajectory.py:55-126) and the
// tools.xr_map_coordinates calls inside it (LCS/tools.py:19-39), including the
// reference's quirks:
//   Q2  index scale n, not n-1                      tools.py:21-22
//   Q3  first/last `order` seed rows: order 1, 'constant'; others `order`, 'wrap'
//   Q4  every SETTLS iteration ADDS to the position  trajectory.py:110-112
//   Q5  conversion_x from the SEED latitude          trajectory.py:56-57
//   Q6  fields consumed in stored order for dt<0     trajectory.py:58-60,80
//   Q7  cyclic wrap hard-coded to +-180, floor-mod   trajectory.py:93-94
//   Q8  NaN latitude -> y_min                        trajectory.py:89-90
// Floating-point operation ORDER follows numpy/scipy where it is cheap to do so
// (contraction off in the position update and the double-precision tap sum), so
// the float64 path agrees with the CPU oracle to rounding, not merely to
// truncation error.
//
// Gather source: the padded interleaved image built by lc_field_pack (pack.hip):
// the 2x2 (order 1) or 4x4 (order 3) tap window of a wrapped coordinate is
// always in range and its (u,v) pairs are contiguous along x, so one sample
// position costs 2 (order 1, float) wide loads per level instead of 8 scalars.
#include <type_traits>

#include "lcs_common.h"
#include "launch_plan.h"

namespace {

// 8 x 32 seeds per workgroup: each wave owns an 8 x 8 patch.  Measured on C3 (ms, direct kernel):
// 64x4 17.9, 32x8 12.6, 16x16 11.2, 8x32 10.9, 4x64 10.85, 2x128 11.1, 1x256 13.5 -- compact patches
// touch the fewest distinct cache lines per gather.  Four waves stacked in latitude per workgroup (no
// barrier ties them; it decides which waves share a CU's vector L1): 1 / 4 / 8 / 16 waves measure
// 8.3 / 7.6 / 8.1 / 9.1 ms with the LDS kernel, 2x2 and 4x1 arrangements 7.7 and 8.4.
constexpr int TILE_W = 8;
constexpr int TILE_H = 32;
constexpr int BLOCK = TILE_W * TILE_H;

template <typename T>
struct AdvectArgs {
    const T *lin;  // order-1 image (raw values)
    const T *img;  // image for the interior rows (== lin for order 1, coefficients for order 3)
    const T *ext;  // null, or 2*img[t] - img[t+1] (lc_field_extrapolate)
    // The raw wind planes u, v [nt][ny_f][nx_f] as the ORDER-1 source instead of the lin image (lc_advect_ex; NULL: lin):
    // the pole seed rows' order-1 / 'constant' samples (Q3), and in float64 at order 1 the Euler sample (img == lin is then
    // not read at all).  Same node values, same arithmetic, bit-identical results -- the image the pack no longer writes.
    const T *u_raw, *v_raw;
    const float *lin32;  // LC_F64_WIND_F32_LIN32 (double instantiation, order 1): the order-1 image of the float32 wind AS float32; lin / img / raw planes unused
    const float *u_raw32, *v_raw32;  // LC_F64_WIND_F32_LIN32 at order 3: the float32 raw planes, the order-1 source of the pole rows (img = float64 coefficients)
    size_t raw_plane;  // ny_f * nx_f
    int ext_raw;       // float64, order 1, raw planes: the fused-level value 2 F[t] - F[t+1] is formed from the planes node by node
                       // (lc_advect_args.fuse_levels_raw): no packed image at all, ext == NULL
    int ext_cub;       // float64, order 3: the fused-level COEFFICIENTS 2 img[t] - img[t+1] are formed from the coefficient image
                       // node by node (lc_advect_args.fuse_levels_raw at order 3): no ext image, ext == NULL
    size_t level_elems;
    int pitch;  // nodes per padded row
    int ny_f, nx_f;
    T lat_min, lat_span, lon_min, lon_span;  // index transform (Q2)
    T sx, sy;                                // float path: c = (x - min) * (n / span)
    T y_min, y_max, x_min, x_max;            // clamp bounds = field coordinate extremes
    const T *seed_lat, *seed_lon;
    int ny, nx;          // local seed block
    int row0, ny_global; // pole rule uses the global row index
    T dt, half_dt;       // T(timestep), T(0.5*timestep)
    T dtcy, hdtcy;       // T(timestep*conversion_y), T((0.5*timestep)*conversion_y)
    int K, order, cyclic, t0, nsteps;
    int wind_f32;  // double instantiation only: the wind is float32-valued -> numpy's promotion rules (Q10)
    T *x_out, *y_out, *traj_x, *traj_y;
    const T *x_start, *y_start;  // NULL: start from the seed grid; else [ny*nx] positions to continue from (lc_advect_from)
    int n_members, member_t0_stride;  // lc_advect_batch: blockIdx.y = member; member m starts at time level t0 + m * stride ...
    size_t member_plane;         // ... and reads / writes positions at x_start / x_out + m * member_plane (elements)
    // Member groups (two-seed kernel on an ensemble, PATCH_PAIR): blockIdx.y = a GROUP of pair_g consecutive
    // members whose start levels lie pair_d apart; a launch walks nsteps LEVELS from t0 = level pair_l0 of the group's first
    // member; member q steps at the levels [q pair_d, q pair_d + pair_n) of the group's window; its planes lie q pair_plane on
    int pair_d, pair_l0, pair_n, pair_g, pair_last;  // pair_d < 0: no groups; pair_last: members of the last group (1 .. pair_g)
    size_t pair_plane;
    int traj_skip0;              // 1: traj entry 0 (the start positions) is already in place (a later chunk of one call)
    int traj_pair_ok, out_pair_ok;  // two-seed kernel, PATCH_WIDE: nx even and traj / out bases 8-byte aligned (paired stores)
    int traj_line_ok;            // two-seed kernel, PATCH_LINES: nx % 4 == 0 and traj bases 16-byte aligned (whole-line stores)
    int patch_mode;              // two-seed kernel: -1 by call (PATCH_LINES with trajectories, else PATCH_TALL), or a Patch value
    int xcd_rows;                // tile rows per XCD chunk (xcd_chunk = xcd_rows * ntx, recomputed when a launcher changes ntx)
    int xcd_split;               // > 0: a chunk is 1 / xcd_split of that (lcplan::xcd_chunk_tiles)
    int wg64;                    // float64, order 1, fused levels: 1 = one LDS tile per workgroup (advect_wg64_kernel; LCS_F64_WG_TILE at creation)
    int ntx, ntiles;
    int xcd_chunk;  // tiles per chunk of the XCD-cyclic tile order; 0: one contiguous band of tiles per XCD
    int tile_order;  // 0 as stored, 1 last tile row first, 2 from the poles inwards (xcd_tile_id)
    int tile_order_two_seed;  // host side only: what the two-seed kernel's launch puts into tile_order
    int pole_blocks, pole_lo, pole_hi;  // leading workgroups that take the pole rows (first pole_lo / last pole_hi local rows); 0: the tiles do
    unsigned *clamp_flag;  // NULL, or set to 1 when the non-cyclic longitude clamp moves any parcel (Q9)
    unsigned *verify;      // NULL, or the context's 16 wave-state counters (lc_ctx_set_verify: the one-seed order-1 LDS kernel's VERIFY instances)
};

// Tile of a workgroup.  Hardware deals workgroups to the 8 XCDs round-robin (blockIdx % 8), each with its own L2.
// xcd_chunk = 0: XCD x takes the x-th contiguous eighth of the tiles (neighbouring tiles share an L2).
// xcd_chunk = C: chunks of C tiles (whole tile rows) go to the XCDs cyclically, so every XCD sees every latitude
// band and they finish together even when the bands cost differently (redo rate, pole rows).
// (the arithmetic is lcplan::tile_of_block, launch_plan.h: unit-tested on the CPU as a bijection blocks <-> tiles)
template <typename T>
__device__ __forceinline__ int xcd_tile_id(const AdvectArgs<T> &A) {
    // pole_blocks is a multiple of 8: (blockIdx.x - pole_blocks) % 8 is still the XCD
    // Order of the tile rows.  Next to a pole 1 / cos(lat) makes a time step many cells long, the windows leave
    // their tiles at every sample and those workgroups live several times longer than the others: started last they
    // are the launch's tail, started first they hide behind it.  1: the last row first, then 0, 1, 2, ... (default:
    // order 3 17.69 -> 17.08 ms, float64 C2 and K = 0 1 % better than as stored); 2: from the poles inwards (last, 0,
    // last-1, 1, ...: the launch ends on the equatorial rows, whose patches stay coherent longest) -- the two-seed
    // order-1 kernel's default (C3 6.68 -> 6.40 ms like 1, but C5 426 -> 419 where 1 gives 440; order 3 17.54 and
    // float64 C2 +5 % with it, so not for them); 0: as stored.
    return lcplan::tile_of_block((int)blockIdx.x - A.pole_blocks, A.ntiles, A.ntx, A.xcd_chunk, A.tile_order);
}
template <typename T>
static inline unsigned nmem(const AdvectArgs<T> &A) { return A.n_members > 1 ? (unsigned)A.n_members : 1u; }  // grid.y of an advect launch
static inline int xcd_grid(int ntiles, int chunk) { return lcplan::xcd_grid(ntiles, chunk); }

template <typename T>
struct Pair {
    T u, v;
};

// Starting position of seed (iy, ix): its grid point (trajectory.py:68-70), or where an earlier call left it
// (lc_advect_from).  May alias x_out / y_out: every seed's start is read by the thread that later writes it.
template <typename T>
__device__ __forceinline__ T start_x(const AdvectArgs<T> &A, int iy, int ix) {
    return A.x_start ? A.x_start[(size_t)iy * A.nx + ix] : A.seed_lon[ix];
}
template <typename T>
__device__ __forceinline__ T start_y(const AdvectArgs<T> &A, int iy, int ix) {
    return A.y_start ? A.y_start[(size_t)iy * A.nx + ix] : A.seed_lat[iy];
}

// lc_advect_batch: the arguments as ensemble member blockIdx.y sees them (a no-op for every other call: gridDim.y == 1).
// Members share the seed grid and the wind series; member m integrates from time level t0 + m * member_t0_stride and
// keeps its positions in the m-th plane of x_start / x_out.  Wave-uniform scalar arithmetic, once per workgroup.
template <typename T>
__device__ __forceinline__ AdvectArgs<T> for_member(const AdvectArgs<T> &A0) {
    AdvectArgs<T> A = A0;
    if (A0.n_members > 1) {
        const size_t off = (size_t)blockIdx.y * A0.member_plane;
        A.t0 = A0.t0 + (int)blockIdx.y * A0.member_t0_stride;
        A.x_out = A0.x_out + off;
        A.y_out = A0.y_out + off;
        if (A0.x_start) {
            A.x_start = A0.x_start + off;
            A.y_start = A0.y_start + off;
        }
    }
    return A;
}

template <typename T>
void set_fast_transform(AdvectArgs<T> &A) {
    const double sx = (double)A.nx_f / (double)A.lon_span, sy = (double)A.ny_f / (double)A.lat_span;
    A.sx = (T)sx;
    A.sy = (T)sy;
}

// The float instantiation cannot be bit-identical to the reference anyway (scipy interpolates
// float32 fields in double), so it takes the cheap forms: FMA index map, FMA lerps, fmin/fmax
// clamps.  The double instantiation keeps numpy/scipy's operation order.
template <typename T>
struct Fast {
    static constexpr bool value = sizeof(T) == 4;
};

// scipy NI_EXTEND_WRAP coordinate map (ni_interpolation.c map_coordinate):
// identity on [0, n-1], otherwise periodic with period n-1.
template <typename T>
__device__ __forceinline__ T wrap_coord(T c, T sz) {
    if (c < T(0))
        c += sz * (trunc(-c / sz) + T(1));
    else if (c > sz)
        c -= sz * trunc(c / sz);
    return c;
}

__device__ __forceinline__ int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// python/numpy float floor-mod by 180 (npy_divmod): exact, fmod based.
template <typename T>
__device__ __forceinline__ T pymod180(T x) {
    T m = fmod(x, T(180));
    if (m != T(0)) {
        if (m < T(0)) m += T(180);
    } else {
        m = T(0);
    }
    return m;
}

// ---- generic sampling: locate once per position, fetch per time level ------------------------
// One sample position is used for time levels t and t+1 (trajectory.py:105-108), so the index map,
// the coordinate wrap, the floor and the spline weights are computed ONCE (locate) and each level only
// pays its loads and its tap sum (fetch).  The arithmetic is exactly what scipy does per call.
template <typename T>
struct Tap {
    unsigned off;     // element offset of the window origin inside one time level (orders 1, 3)
    int sy, sx;       // first tap's node index, before mirroring (orders 2, 4, 5)
    T wy[6], wx[6];   // scipy's per-axis weights (order+1 of them)
    T ty, tx;         // fractional parts (float order-1 lerp form)
    bool zero;        // 'constant' mode, coordinate outside [0, n-1]: the sample is exactly 0
};

template <typename T>
__device__ __forceinline__ void cubic_weights(T t, T w[4]) {
#pragma clang fp contract(off)
    // scipy get_spline_interpolation_weights, order 3
    const T y = t, z = T(1) - t;
    w[1] = (y * y * (y - T(2)) * T(3) + T(4)) / T(6);
    w[2] = (z * z * (z - T(2)) * T(3) + T(4)) / T(6);
    w[0] = z * z * z / T(6);
    w[3] = T(1) - w[0] - w[1] - w[2];
}


// ---- orders 2, 4, 5 (LCS/trajectory.py:16 and LCS/tools.py:26-30 pass any order to scipy) -----------------
// Generic form, direct kernels only: centred B-spline weights from the closed form
//   beta_n(t) = 1/n! sum_k (-1)^k C(n+1, k) (t + (n+1)/2 - k)_+^n,
// first tap floor(c) - n/2 (odd n) or floor(c + 1/2) - n/2 (even n) as in scipy's map_coordinates, every tap
// index mirrored into [0, n-1] (the pads of the image are not relied on), sums in double whatever T is (scipy
// evaluates in double and rounds to the field's dtype).  Agreement with scipy: 1e-14 (order 2), 1e-12 (orders
// 4, 5: the prefilter's pole constants differ from scipy's in the last bit and the gain amplifies it).
__device__ __forceinline__ int mirror_node(int i, int n) {
    const int s2 = 2 * n - 2;
    i = i < 0 ? -i : i;
    i %= s2;
    return i > n - 1 ? s2 - i : i;
}

template <int N>
__device__ __forceinline__ double bspline_centred(double t) {
    constexpr double half = 0.5 * (N + 1);
    constexpr int binom[6][7] = {{1, 1}, {1, 2, 1}, {1, 3, 3, 1}, {1, 4, 6, 4, 1}, {1, 5, 10, 10, 5, 1}, {1, 6, 15, 20, 15, 6, 1}};
    constexpr double fact[6] = {1, 1, 2, 6, 24, 120};
    double s = 0.0;
#pragma unroll
    for (int k = 0; k <= N + 1; ++k) {
        const double a = t + half - k;
        if (a > 0.0) {
            double p = a;
#pragma unroll
            for (int q = 1; q < N; ++q) p *= a;
            s += ((k & 1) ? -1.0 : 1.0) * binom[N][k] * p;
        }
    }
    return s / fact[N];
}

template <typename T, int ORDER>
__device__ __forceinline__ void locate_general(Tap<T> &t, T cy, T cx) {
    const double y = (double)cy, x = (double)cx;
    t.sy = (ORDER & 1) ? (int)floor(y) - ORDER / 2 : (int)floor(y + 0.5) - ORDER / 2;
    t.sx = (ORDER & 1) ? (int)floor(x) - ORDER / 2 : (int)floor(x + 0.5) - ORDER / 2;
#pragma unroll
    for (int k = 0; k <= ORDER; ++k) {
        t.wy[k] = (T)bspline_centred<ORDER>(y - (double)(t.sy + k));
        t.wx[k] = (T)bspline_centred<ORDER>(x - (double)(t.sx + k));
    }
}

template <typename T, int ORDER>
__device__ __forceinline__ Pair<T> fetch_general(const T *__restrict__ lvl, const AdvectArgs<T> &A, const Tap<T> &t) {
#pragma clang fp contract(off)
    double su = 0.0, sv = 0.0;
#pragma unroll
    for (int a = 0; a <= ORDER; ++a) {
        const T *row = lvl + ((size_t)(mirror_node(t.sy + a, A.ny_f) + LC_PAD_LO) * A.pitch + LC_PAD_LO) * 2;
#pragma unroll
        for (int b = 0; b <= ORDER; ++b) {
            const T *p = row + (size_t)mirror_node(t.sx + b, A.nx_f) * 2;
            su += ((double)p[0] * (double)t.wy[a]) * (double)t.wx[b];
            sv += ((double)p[1] * (double)t.wy[a]) * (double)t.wx[b];
        }
    }
    Pair<T> r;
    r.u = (T)su;
    r.v = (T)sv;
    return r;
}

template <typename T, int ORDER, bool WRAP>
__device__ __forceinline__ Tap<T> locate(const AdvectArgs<T> &A, T x, T y) {
#pragma clang fp contract(off)
    // tools.py:21-22: (n * (x - min)) / (max - min)
    T cx, cy;
    if (Fast<T>::value) {
        // subtract first: exact 0 at the grid origin, where 'constant' mode is discontinuous
        cx = (x - A.lon_min) * A.sx;
        cy = (y - A.lat_min) * A.sy;
    } else {
        cx = (T(A.nx_f) * (x - A.lon_min)) / A.lon_span;
        cy = (T(A.ny_f) * (y - A.lat_min)) / A.lat_span;
    }
    Tap<T> t;
    t.zero = false;
    if (WRAP) {
        cy = wrap_coord<T>(cy, T(A.ny_f - 1));
        cx = wrap_coord<T>(cx, T(A.nx_f - 1));
    } else if (cy < T(0) || cy > T(A.ny_f - 1) || cx < T(0) || cx > T(A.nx_f - 1)) {
        t.zero = true;  // 'constant': exactly cval=0 outside [0, n-1] (no interpolation towards cval)
    }
    if (ORDER != 1 && ORDER != 3) {
        t.ty = t.tx = T(0);
        t.off = 0;
        if (!(cy >= T(0) && cy <= T(A.ny_f - 1) && cx >= T(0) && cx <= T(A.nx_f - 1))) cy = cx = T(0);  // NaN / inf: memory safety
        locate_general<T, ORDER>(t, cy, cx);
        return t;
    }
    const T fy = floor(cy), fx = floor(cx);
    const int y0 = clampi((int)fy, 0, A.ny_f - 1);  // clamp: memory safety for NaN/inf/rounding
    const int x0 = clampi((int)fx, 0, A.nx_f - 1);
    t.ty = cy - fy;
    t.tx = cx - fx;
    if (ORDER == 3) {
        cubic_weights<T>(t.ty, t.wy);
        cubic_weights<T>(t.tx, t.wx);
        t.off = ((unsigned)y0 * (unsigned)A.pitch + (unsigned)x0) * 2u;  // window starts at padded (y0, x0)
        t.sy = y0;  // (an LDS tile addresses the window by its padded origin)
        t.sx = x0;
    } else {
        t.wy[0] = T(1) - t.ty;
        t.wy[1] = T(1) - t.wy[0];  // scipy: last weight = 1 - sum(others)
        t.wx[0] = T(1) - t.tx;
        t.wx[1] = T(1) - t.wx[0];
        t.off = ((unsigned)(y0 + LC_PAD_LO) * (unsigned)A.pitch + (unsigned)(x0 + LC_PAD_LO)) * 2u;
        t.sy = y0;  // (the raw-plane fetch addresses by node index)
        t.sx = x0;
    }
    return t;
}

// The order-1 window {u00, v00, u01, v01}, {u10, v10, u11, v11} of cell (y0, x0) from the RAW planes of one level
// (up = this level's u plane, vp = its v plane).  The image's pad nodes hold the mirrored interior value, so the
// neighbour of the last node is node n - 2 here too (its weight is 0 at c = n - 1 exactly; the value still matters for
// non-finite winds): the same eight numbers as a window of the lin image.
template <typename T>
__device__ __forceinline__ void raw_window(const T *__restrict__ up, const T *__restrict__ vp, const AdvectArgs<T> &A, int y0, int x0,
                                           T (&a)[4], T (&b)[4]) {
    const int x1 = x0 + 1 < A.nx_f ? x0 + 1 : A.nx_f - 2, y1 = y0 + 1 < A.ny_f ? y0 + 1 : A.ny_f - 2;
    const size_t r0 = (size_t)y0 * A.nx_f, r1 = (size_t)y1 * A.nx_f;
    a[0] = up[r0 + x0];
    a[1] = vp[r0 + x0];
    a[2] = up[r0 + x1];
    a[3] = vp[r0 + x1];
    b[0] = up[r1 + x0];
    b[1] = vp[r1 + x0];
    b[2] = up[r1 + x1];
    b[3] = vp[r1 + x1];
}

// The order-1 sample of a window {u00, v00, u01, v01}, {u10, v10, u11, v11} in scipy's operation order (tap order: last axis
// fastest; per tap ((value*wy)*wx), summed from 0).  ONE function for every source of the eight numbers -- packed image, raw
// planes, the float32 image of LC_F64_WIND_F32_LIN32, an LDS tile of it -- so a seed's bits do not depend on which served it.
template <typename T>
__device__ __forceinline__ Pair<T> tap_sum_order1(const T (&a)[4], const T (&b)[4], const Tap<T> &t) {
#pragma clang fp contract(off)
    T su = T(0), sv = T(0);
    su += (a[0] * t.wy[0]) * t.wx[0];
    sv += (a[1] * t.wy[0]) * t.wx[0];
    su += (a[2] * t.wy[0]) * t.wx[1];
    sv += (a[3] * t.wy[0]) * t.wx[1];
    su += (b[0] * t.wy[1]) * t.wx[0];
    sv += (b[1] * t.wy[1]) * t.wx[0];
    su += (b[2] * t.wy[1]) * t.wx[1];
    sv += (b[3] * t.wy[1]) * t.wx[1];
    Pair<T> r;
    r.u = su;
    r.v = sv;
    return r;
}

// ... and the order-3 sample of a 4 x 4 window (rows of four interleaved nodes), scipy's tap order: per tap ((value*wy)*wx),
// summed from 0, last axis fastest.  Shared by the global-memory window and the LDS tile of advect_lds64w_o3_kernel.
template <typename T>
__device__ __forceinline__ Pair<T> tap_sum_order3(const T (&q)[4][8], const Tap<T> &t) {
#pragma clang fp contract(off)
    T su = T(0), sv = T(0);
#pragma unroll
    for (int a = 0; a < 4; ++a) {
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            su += (q[a][2 * b] * t.wy[a]) * t.wx[b];
            sv += (q[a][2 * b + 1] * t.wy[a]) * t.wx[b];
        }
    }
    Pair<T> r;
    r.u = su;
    r.v = sv;
    return r;
}

// RAW (order 1 only): `lvl` is the level's raw u plane, the v plane lies A.v_raw - A.u_raw elements on
template <typename T, int ORDER, bool RAW = false>
__device__ __forceinline__ Pair<T> fetch(const T *__restrict__ lvl, const AdvectArgs<T> &A, const Tap<T> &t) {
#pragma clang fp contract(off)
    static_assert(!RAW || ORDER == 1, "the raw planes are an order-1 source");
    Pair<T> r;
    if (t.zero) {
        r.u = T(0);
        r.v = T(0);
        return r;
    }
    if (ORDER != 1 && ORDER != 3) return fetch_general<T, ORDER>(lvl, A, t);
    const T *p = RAW ? lvl : lvl + t.off;
    if (ORDER == 3) {
        T q[4][8];  // the 4 x 4 window, rows of {u0, v0, u1, v1, u2, v2, u3, v3}
#pragma unroll
        for (int a = 0; a < 4; ++a) __builtin_memcpy(q[a], p + (unsigned)a * (unsigned)A.pitch * 2u, sizeof(q[a]));
        return tap_sum_order3<T>(q, t);
    }
    T a[4], b[4];
    if (RAW) {
        raw_window<T>(p, p + (A.v_raw - A.u_raw), A, t.sy, t.sx, a, b);
    } else {
        __builtin_memcpy(a, p, sizeof(a));                            // {u00, v00, u01, v01}
        __builtin_memcpy(b, p + (unsigned)A.pitch * 2u, sizeof(b));   // {u10, v10, u11, v11}
    }
    if (Fast<T>::value) {
#pragma clang fp contract(fast)
        const T u0 = fma(t.tx, a[2] - a[0], a[0]), v0 = fma(t.tx, a[3] - a[1], a[1]);
        const T u1 = fma(t.tx, b[2] - b[0], b[0]), v1 = fma(t.tx, b[3] - b[1], b[1]);
        r.u = fma(t.ty, u1 - u0, u0);
        r.v = fma(t.ty, v1 - v0, v0);
        return r;
    }
    return tap_sum_order1<T>(a, b, t);
}

template <typename T, int ORDER, bool WRAP, bool RAW = false>
__device__ __forceinline__ Pair<T> sample(const T *__restrict__ lvl, const AdvectArgs<T> &A, T x, T y) {
    return fetch<T, ORDER, RAW>(lvl, A, locate<T, ORDER, WRAP>(A, x, y));
}

// y + a*x: fused for float, mul-then-add (numpy's two roundings) for double
template <typename T>
__device__ __forceinline__ T axpy(T a, T x, T y) {
    if (Fast<T>::value) {
        return fma(a, x, y);
    } else {
#pragma clang fp contract(off)
        const T m = a * x;
        return y + m;
    }
}

template <typename T>
__device__ __forceinline__ void clamp_position(const AdvectArgs<T> &A, T &x, T &y) {
    // trajectory.py:89-90 -- where(y > y_min, y, y_min): NaN -> y_min (Q8)
    if (Fast<T>::value) {
        y = fmin(fmax(y, A.y_min), A.y_max);  // same NaN rule: fmax(NaN, y_min) = y_min
    } else {
        y = (y > A.y_min) ? y : A.y_min;
        y = (y < A.y_max) ? y : A.y_max;
    }
    if (A.cyclic) {
        // trajectory.py:93-94 (Q7)
        if (!(x > T(-180))) x = pymod180<T>(x);
        if (!(x < T(180))) x = T(-180) + pymod180<T>(x);
    } else {
        // per-point clamp; the reference's outer-product assignment (Q9) is lc_advect's LC_X_CLAMP_REFERENCE_OUTER
        if (x < A.x_min || x > A.x_max) {
            if (A.clamp_flag) *A.clamp_flag = 1u;
            x = x < A.x_min ? A.x_min : A.x_max;
        }
    }
}

// float32 wind on float64 coordinates (the reference's behaviour for e.g. float32 reanalysis winds on
// float64 lat/lon): map_coordinates returns the FIELD's dtype, so samples are float32; python-float *
// float32-array stays float32, so the latitude increments are formed in float32 and only then added to
// the float64 position, while conversion_x (a float64 array) promotes the longitude increments to
// float64 (numpy promotion through trajectory.py:86-87,110-112).  No-ops unless A.wind_f32.
template <typename T>
__device__ __forceinline__ T round_sample(const AdvectArgs<T> &A, T v) {
    return (sizeof(T) == 8 && A.wind_f32) ? (T)(float)v : v;
}
template <typename T>
__device__ __forceinline__ T lat_increment(const AdvectArgs<T> &A, T scale, T vel_or_bracket) {
#pragma clang fp contract(off)
    if (sizeof(T) == 8 && A.wind_f32) return (T)((float)scale * (float)vel_or_bracket);
    return scale * vel_or_bracket;
}
template <typename T>
__device__ __forceinline__ T settls_bracket(const AdvectArgs<T> &A, T e, T c, T n) {
#pragma clang fp contract(off)
    if (sizeof(T) == 8 && A.wind_f32) return (T)(((float)e + 2.0f * (float)c) - (float)n);
    return (e + T(2) * c) - n;
}

// FUSED (opt-in, lc_advect with packed_ext in LC_F64): each SETTLS iteration samples the fused image
// ext[t] = 2 F[t] - F[t+1] once instead of F[t] and F[t+1] separately.  Interpolation is linear in the
// field, so the value differs from the reference's (e + 2c) - n only by rounding (~1 ulp of the wind).
// RAW: `image` is A.u_raw (order 1, two-sample form): levels are raw planes, raw_plane elements apart.
template <typename T, int ORDER, bool WRAP, bool FUSED = false, bool RAW = false>
__device__ void advect_seed(const AdvectArgs<T> &A, const T *__restrict__ image, int iy, int ix) {
#pragma clang fp contract(off)
    T x = start_x<T>(A, iy, ix);
    T y = start_y<T>(A, iy, ix);
    // trajectory.py:56 -- 180 / (pi * R * |cos(lat * pi / 180)|), seed latitude (Q5)
    const T ys = A.seed_lat[iy];
    const T cx_conv = T(180) / (T(3.141592653589793 * 6371000.0) * fabs(cos((ys * T(3.141592653589793)) / T(180))));
    const T dtcx = A.dt * cx_conv;        // timestep * conversion_x
    const T hdtcx = A.half_dt * cx_conv;  // (0.5 * timestep) * conversion_x
    const size_t idx = (size_t)iy * A.nx + ix;
    const size_t plane = (size_t)A.ny * A.nx;
    if (A.traj_x && !A.traj_skip0) {
        A.traj_x[idx] = x;
        A.traj_y[idx] = y;
    }
    const size_t lstride = RAW ? A.raw_plane : A.level_elems;
    const T *lvl = image + (size_t)A.t0 * lstride;
    const T *elv = FUSED ? A.ext + (size_t)A.t0 * A.level_elems : nullptr;
    for (int s = 0; s < A.nsteps; ++s) {
        const T *nxt = lvl + lstride;
        Pair<T> e = sample<T, ORDER, WRAP, RAW>(lvl, A, x, y);   // trajectory.py:82-84
        e.u = round_sample<T>(A, e.u);
        e.v = round_sample<T>(A, e.v);
        y = y + lat_increment<T>(A, A.dtcy, e.v);                // :86
        x = axpy<T>(dtcx, e.u, x);                               // :87
        clamp_position<T>(A, x, y);
        for (int k = 0; k < A.K; ++k) {                          // :100
            const Tap<T> tap = locate<T, ORDER, WRAP>(A, x, y);  // one position, two time levels
            if (FUSED) {
                const Pair<T> w = fetch<T, ORDER>(elv, A, tap);
                y = y + A.hdtcy * (e.v + w.v);
                x = axpy<T>(hdtcx, e.u + w.u, x);
                clamp_position<T>(A, x, y);
                continue;
            }
            Pair<T> c = fetch<T, ORDER, RAW>(lvl, A, tap);        // :105,107
            Pair<T> n = fetch<T, ORDER, RAW>(nxt, A, tap);        // :106,108
            c.u = round_sample<T>(A, c.u);
            c.v = round_sample<T>(A, c.v);
            n.u = round_sample<T>(A, n.u);
            n.v = round_sample<T>(A, n.v);
            y = y + lat_increment<T>(A, A.hdtcy, settls_bracket<T>(A, e.v, c.v, n.v));  // :110
            x = axpy<T>(hdtcx, settls_bracket<T>(A, e.u, c.u, n.u), x);                 // :112
            clamp_position<T>(A, x, y);
        }
        if (A.traj_x) {
            A.traj_x[(size_t)(s + 1) * plane + idx] = x;
            A.traj_y[(size_t)(s + 1) * plane + idx] = y;
        }
        lvl = nxt;
        if (FUSED) elv += A.level_elems;
    }
    A.x_out[idx] = x;
    A.y_out[idx] = y;
}

// The global pole rows (first / last `order` seed rows, Q3) take the generic per-seed path through the whole series.
// Inside a tile they would hold 8 lanes of a wave -- and with it the wave's workgroup -- several times a normal
// workgroup's life (every sample a dependent global gather), and the workgroups of the last tile row would be the
// launch's tail (measured on C3: 7.0 -> 6.7 ms without it).  So the launch starts with `pole_blocks` workgroups that
// take those rows one seed per thread, all lanes busy, and the tiles skip them.  Same function, same results.
// A pole-row seed: order 1 + 'constant' through the whole series, from the lin image or -- when the caller handed the raw
// planes instead (lc_advect_ex) -- from those.  SRC: 0 = the lin image (float32 order-1 kernels: lin is their interior
// image anyway, it always exists), 2 = the raw planes (kernel variants compiled for them), 1 = whichever the call has.
constexpr int POLE_LIN = 0, POLE_EITHER = 1, POLE_RAW = 2;
// float64 at order 1, where the samples come from: packed images (lin + ext) / raw planes for the Euler sample + ext image /
// raw planes for both (the fused-level value formed node by node)
constexpr int SRC_IMAGES = 0, SRC_RAW_EULER = 1, SRC_RAW_ALL = 2;
template <bool WRAP>
__device__ void advect_seed_w32(const AdvectArgs<double> &A, int iy, int ix);  // (below: the float32 wind's own order-1 path)
template <typename T, int SRC>
__device__ __forceinline__ void pole_seed(const AdvectArgs<T> &A, int iy, int ix) {
    if constexpr (sizeof(T) == 8 && SRC == POLE_EITHER) {
        if (A.u_raw32) {  // LC_F64_WIND_F32_LIN32 at order 3: the pole rows sample the float32 planes
            advect_seed_w32<false>(A, iy, ix);
            return;
        }
    }
    if (SRC == POLE_RAW || (SRC == POLE_EITHER && A.u_raw))
        advect_seed<T, 1, false, false, true>(A, A.u_raw, iy, ix);
    else
        advect_seed<T, 1, false>(A, A.lin, iy, ix);
}

template <typename T, int SRC = POLE_EITHER>
__device__ __forceinline__ bool pole_block(const AdvectArgs<T> &A) {
    if ((int)blockIdx.x >= A.pole_blocks) return false;
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i < (A.pole_lo + A.pole_hi) * A.nx) {
        const int k = i / A.nx, ix = i - k * A.nx;
        const int iy = lcplan::pole_row(k, A.pole_lo, A.pole_hi, A.ny);
        pole_seed<T, SRC>(A, iy, ix);
    }
    return true;
}

// ======================================================================================
// LC_F64_WIND_F32_LIN32 -- float32 wind on float64 coordinates with the wind KEPT float32 (order 1).
//
// The reference's behaviour for this mix (float32 reanalysis winds on float64 lat / lon; numpy's promotion through
// LCS/trajectory.py:86-87,110-112): scipy interpolates the float32 field in double and returns float32
// samples, the SETTLS bracket and the latitude increments are formed in float32, the longitude increments in float64.
// LC_F64_WIND_F32 does that on float64 IMAGES of the float32 values (twice the bytes, and a conversion pass before any
// advection); here the image stays float32 -- lc_field_pack(LC_F32, order 1) -- and a node is widened as it is read:
// the same eight doubles enter tap_sum_order1, so the results are those of LC_F64_WIND_F32 bit for bit.  Two samples per
// iteration (the bracket's float32 roundings do not commute with a fused level), numpy's exact index map (locate<double>).
// ======================================================================================
typedef float w4 __attribute__((ext_vector_type(4)));
typedef float w2 __attribute__((ext_vector_type(2)));
typedef double d2w __attribute__((ext_vector_type(2)));

// the window of cell (t.sy, t.sx) from one level of the float32 image (padded (row + 1, column + 1): pads hold the mirror)
// (no image -- order 3, where only the pole rows sample at order 1 --: `lvl` is the level's raw float32 u plane, the v plane
//  lies A.v_raw32 - A.u_raw32 on; the neighbour behind the last node is its mirror image, node n - 2, as in the image's pads)
__device__ __forceinline__ void window_w32(const float *__restrict__ lvl, const AdvectArgs<double> &A, const Tap<double> &t,
                                           double (&a)[4], double (&b)[4]) {
    if (!A.lin32) {
        const float *up = lvl, *vp = lvl + (A.v_raw32 - A.u_raw32);
        const int x1 = t.sx + 1 < A.nx_f ? t.sx + 1 : A.nx_f - 2, y1 = t.sy + 1 < A.ny_f ? t.sy + 1 : A.ny_f - 2;
        const size_t r0 = (size_t)t.sy * A.nx_f, r1 = (size_t)y1 * A.nx_f;
        a[0] = up[r0 + t.sx], a[1] = vp[r0 + t.sx], a[2] = up[r0 + x1], a[3] = vp[r0 + x1];
        b[0] = up[r1 + t.sx], b[1] = vp[r1 + t.sx], b[2] = up[r1 + x1], b[3] = vp[r1 + x1];
        return;
    }
    const float *p = lvl + ((size_t)(t.sy + LC_PAD_LO) * A.pitch + (t.sx + LC_PAD_LO)) * 2;
    w4 r0, r1;
    __builtin_memcpy(&r0, p, 16);                          // {u00, v00, u01, v01}
    __builtin_memcpy(&r1, p + (size_t)A.pitch * 2, 16);    // {u10, v10, u11, v11}
    a[0] = r0.x, a[1] = r0.y, a[2] = r0.z, a[3] = r0.w;
    b[0] = r1.x, b[1] = r1.y, b[2] = r1.z, b[3] = r1.w;
}
// one sample, rounded to float32 as map_coordinates returns it (Q10)
template <bool WRAP>
__device__ __forceinline__ Pair<double> sample_w32(const float *__restrict__ lvl, const AdvectArgs<double> &A, const Tap<double> &t) {
    Pair<double> r;
    if (!WRAP && t.zero) {
        r.u = r.v = 0.0;
        return r;
    }
    double a[4], b[4];
    window_w32(lvl, A, t, a, b);
    r = tap_sum_order1<double>(a, b, t);
    r.u = (double)(float)r.u;
    r.v = (double)(float)r.v;
    return r;
}

// trajectory.py:80-126 for one seed, every sample a global gather: the pole rows (WRAP = false: order 1 + 'constant', Q3),
// the direct kernel, and what the LDS kernel below must equal.  advect_seed<double, 1, WRAP>'s statements with A.wind_f32.
template <bool WRAP>
__device__ void advect_seed_w32(const AdvectArgs<double> &A, int iy, int ix) {
#pragma clang fp contract(off)
    typedef double T;
    T x = start_x<T>(A, iy, ix), y = start_y<T>(A, iy, ix);
    const T ys = A.seed_lat[iy];
    const T cx_conv = T(180) / (T(3.141592653589793 * 6371000.0) * fabs(cos((ys * T(3.141592653589793)) / T(180))));  // Q5
    const T dtcx = A.dt * cx_conv, hdtcx = A.half_dt * cx_conv;
    const size_t idx = (size_t)iy * A.nx + ix, plane = (size_t)A.ny * A.nx;
    if (A.traj_x && !A.traj_skip0) {
        A.traj_x[idx] = x;
        A.traj_y[idx] = y;
    }
    const size_t lstride = A.lin32 ? A.level_elems : A.raw_plane;   // image levels, or raw planes
    const float *lvl = (A.lin32 ? A.lin32 : A.u_raw32) + (size_t)A.t0 * lstride;
    for (int s = 0; s < A.nsteps; ++s) {
        const float *nxt = lvl + lstride;
        const Pair<T> e = sample_w32<WRAP>(lvl, A, locate<T, 1, WRAP>(A, x, y));     // trajectory.py:82-84
        y = y + lat_increment<T>(A, A.dtcy, e.v);                                     // :86
        x = axpy<T>(dtcx, e.u, x);                                                    // :87
        clamp_position<T>(A, x, y);
        for (int k = 0; k < A.K; ++k) {                                               // :100
            const Tap<T> tap = locate<T, 1, WRAP>(A, x, y);                           // one position, two time levels
            const Pair<T> c = sample_w32<WRAP>(lvl, A, tap), n = sample_w32<WRAP>(nxt, A, tap);   // :105-108
            y = y + lat_increment<T>(A, A.hdtcy, settls_bracket<T>(A, e.v, c.v, n.v));           // :110
            x = axpy<T>(hdtcx, settls_bracket<T>(A, e.u, c.u, n.u), x);                          // :112
            clamp_position<T>(A, x, y);
        }
        if (A.traj_x) {
            A.traj_x[(size_t)(s + 1) * plane + idx] = x;
            A.traj_y[(size_t)(s + 1) * plane + idx] = y;
        }
        lvl = nxt;
    }
    A.x_out[idx] = x;
    A.y_out[idx] = y;
}

__device__ __forceinline__ bool pole_block_w32(const AdvectArgs<double> &A) {   // pole_block with the float32 image as the source
    if ((int)blockIdx.x >= A.pole_blocks) return false;
    const int i = blockIdx.x * BLOCK + threadIdx.x;
    if (i < (A.pole_lo + A.pole_hi) * A.nx) {
        const int k = i / A.nx, ix = i - k * A.nx;
        advect_seed_w32<false>(A, lcplan::pole_row(k, A.pole_lo, A.pole_hi, A.ny), ix);
    }
    return true;
}

// direct kernel: fields smaller than a tile, SETTLS_order = 0, lc_ctx_set_lds_tiles(0)
__global__ void __launch_bounds__(BLOCK) advect_w32_kernel(const AdvectArgs<double> A0) {
    const AdvectArgs<double> A = for_member(A0);
    if (pole_block_w32(A)) return;
    const int tile = xcd_tile_id(A);
    if (tile >= A.ntiles) return;
    const int tyi = tile / A.ntx, txi = tile - tyi * A.ntx;
    const int ix = txi * TILE_W + (threadIdx.x % TILE_W), iy = tyi * TILE_H + (threadIdx.x / TILE_W);
    if (ix >= A.nx || iy >= A.ny) return;
    const int grow = A.row0 + iy;
    if (grow < A.order || grow >= A.ny_global - A.order) {  // tools.py:24-33 (Q3)
        if (!A.pole_blocks) advect_seed_w32<false>(A, iy, ix);
    } else
        advect_seed_w32<true>(A, iy, ix);
}

// Per-wave LDS tiles (advect_lds64_kernel's scheme): each wave stages a 16 x 12-node tile of levels t AND t + 1 (float32
// nodes of 8 bytes: 1.5 KB each) around the travel the previous level's displacement predicts, and the K iterations take
// both samples of a position out of LDS; a window that left the tile reads global memory.  Euler sample: direct gather.
constexpr int TW_COLS = 16, TW_ROWS = 12, TW_PITCH = 17;
template <int KFIX, bool CYCLIC>
__global__ void __launch_bounds__(BLOCK) advect_lds64w_kernel(const AdvectArgs<double> A0) {
#pragma clang fp contract(off)
    typedef double T;
    const AdvectArgs<double> A = for_member(A0);
    const int K = KFIX >= 0 ? KFIX : A.K;
    __shared__ __attribute__((aligned(16))) w2 s_tiles[BLOCK / 64][2][TW_ROWS * TW_PITCH];
    if (pole_block_w32(A)) return;
    const int tile_id = xcd_tile_id(A);
    if (tile_id >= A.ntiles) return;  // whole block
    const int tyi = tile_id / A.ntx, txi = tile_id - tyi * A.ntx;
    const int ix = txi * TILE_W + (threadIdx.x % TILE_W), iy = tyi * TILE_H + (threadIdx.x / TILE_W);
    const int lane = threadIdx.x & 63;
    w2 *tile0 = s_tiles[threadIdx.x >> 6][0], *tile1 = s_tiles[threadIdx.x >> 6][1];
    bool live = ix < A.nx && iy < A.ny;
    if (live) {
        const int grow = A.row0 + iy;
        if (grow < A.order || grow >= A.ny_global - A.order) {  // pole rows: generic path (Q3)
            if (!A.pole_blocks) advect_seed_w32<false>(A, iy, ix);
            live = false;
        }
    }
    if (__ballot(live) == 0ull) return;  // whole wave (no workgroup barrier anywhere below)
    const int sx_i = min(ix, A.nx - 1), sy_i = min(iy, A.ny - 1);  // lanes without a seed shadow a neighbour; stores masked
    T x = start_x<T>(A, sy_i, sx_i), y = start_y<T>(A, sy_i, sx_i);
    const T ys = A.seed_lat[sy_i];
    const T cx_conv = T(180) / (T(3.141592653589793 * 6371000.0) * fabs(cos((ys * T(3.141592653589793)) / T(180))));  // Q5
    const T dtcx = A.dt * cx_conv, hdtcx = A.half_dt * cx_conv;
    const size_t idx = live ? (size_t)iy * A.nx + ix : 0, plane = (size_t)A.ny * A.nx;
    if (live && A.traj_x && !A.traj_skip0) {
        A.traj_x[idx] = x;
        A.traj_y[idx] = y;
    }
    const float *lvl = A.lin32 + (size_t)A.t0 * A.level_elems;
    const int pad_cols = A.pitch, pad_rows = A.ny_f + LC_PAD;
    constexpr int CENTRE = TILE_W / 2 + TILE_W * 4;  // middle seed of the wave's 8 x 8 patch
    const int st_row = lane >> 4, st_col = lane & 15;  // staging: one node (8 bytes) per lane and level, 4 rows per pass, 3 passes
    const unsigned st_off = ((unsigned)st_row * (unsigned)pad_cols + (unsigned)st_col) * 8u;
    double dprev_x = 0.0, dprev_y = 0.0;  // previous level's Euler displacement in index space: predicts this level's travel
    const double kpred = 0.5 * (double)(K > 0 ? K - 1 : 0);
    for (int s = 0; s < A.nsteps; ++s) {
        const float *nxt = lvl + A.level_elems;
        // ---- 1. anchor the tiles on the centre lane's predicted travel, issue their loads ----------------------------
        int ox = 0, oy = 0;
        w2 st0[TW_ROWS / 4], st1[TW_ROWS / 4];
        if (K > 0) {
            const double cax = (x - A.lon_min) * A.sx + dprev_x * (1.0 + kpred), cay = (y - A.lat_min) * A.sy + dprev_y * (1.0 + kpred);
            const int rxm = __builtin_amdgcn_readlane((int)floor(fmin(fmax(cax, -4.0), 1.0e9)), CENTRE);
            const int rym = __builtin_amdgcn_readlane((int)floor(fmin(fmax(cay, -4.0), 1.0e9)), CENTRE);
            ox = min(max(rxm + LC_PAD_LO - (TW_COLS - 2) / 2, 0), pad_cols - TW_COLS);
            oy = min(max(rym + LC_PAD_LO - (TW_ROWS - 2) / 2, 0), pad_rows - TW_ROWS);
            const char *src = (const char *)lvl + ((size_t)oy * pad_cols + ox) * 8, *srcn = src + A.level_elems * sizeof(float);
#pragma unroll
            for (int r = 0; r < TW_ROWS / 4; ++r) {
                __builtin_memcpy(&st0[r], src + (size_t)(r * 4) * pad_cols * 8 + st_off, 8);
                __builtin_memcpy(&st1[r], srcn + (size_t)(r * 4) * pad_cols * 8 + st_off, 8);
            }
        }
        // ---- 2. Euler sample: direct gather from level t ------------------------------------------------------------------
        const double x0p = x, y0p = y;
        const Pair<T> e = sample_w32<true>(lvl, A, locate<T, 1, true>(A, x, y));   // trajectory.py:82-84
        y = y + lat_increment<T>(A, A.dtcy, e.v);                                   // :86
        x = axpy<T>(dtcx, e.u, x);                                                  // :87
        clamp_position<T>(A, x, y);
        dprev_x = (x - x0p) * A.sx;
        dprev_y = (y - y0p) * A.sy;
        // ---- 3. tiles into LDS ----------------------------------------------------------------------------------------------
        if (K > 0) {
            __builtin_amdgcn_wave_barrier();  // the previous level's reads are done (LDS ops of a wave are in order)
#pragma unroll
            for (int r = 0; r < TW_ROWS / 4; ++r) {
                tile0[(r * 4 + st_row) * TW_PITCH + st_col] = st0[r];
                tile1[(r * 4 + st_row) * TW_PITCH + st_col] = st1[r];
            }
            __builtin_amdgcn_wave_barrier();
        }
        // ---- 4. K iterations, both levels' windows out of LDS -------------------------------------------------------------
        for (int k = 0; k < K; ++k) {
            const Tap<T> tap = locate<T, 1, true>(A, x, y);
            const int rx = tap.sx + LC_PAD_LO - ox, ry = tap.sy + LC_PAD_LO - oy;   // window origin inside the tile (padded coordinates)
            Pair<T> c, n;
            if ((unsigned)rx <= (unsigned)(TW_COLS - 2) && (unsigned)ry <= (unsigned)(TW_ROWS - 2)) {
                double a[4], b[4];
                const w2 *w = tile0 + ry * TW_PITCH + rx;
                w2 n00 = w[0], n01 = w[1], n10 = w[TW_PITCH], n11 = w[TW_PITCH + 1];
                a[0] = n00.x, a[1] = n00.y, a[2] = n01.x, a[3] = n01.y;
                b[0] = n10.x, b[1] = n10.y, b[2] = n11.x, b[3] = n11.y;
                c = tap_sum_order1<T>(a, b, tap);
                w = tile1 + ry * TW_PITCH + rx;
                n00 = w[0], n01 = w[1], n10 = w[TW_PITCH], n11 = w[TW_PITCH + 1];
                a[0] = n00.x, a[1] = n00.y, a[2] = n01.x, a[3] = n01.y;
                b[0] = n10.x, b[1] = n10.y, b[2] = n11.x, b[3] = n11.y;
                n = tap_sum_order1<T>(a, b, tap);
                c.u = (double)(float)c.u, c.v = (double)(float)c.v;   // map_coordinates returns the field's dtype (Q10)
                n.u = (double)(float)n.u, n.v = (double)(float)n.v;
            } else {  // the window left the tile: the same nodes from global memory
                c = sample_w32<true>(lvl, A, tap);
                n = sample_w32<true>(nxt, A, tap);
            }
            y = y + lat_increment<T>(A, A.hdtcy, settls_bracket<T>(A, e.v, c.v, n.v));   // :110
            x = axpy<T>(hdtcx, settls_bracket<T>(A, e.u, c.u, n.u), x);                  // :112
            clamp_position<T>(A, x, y);
        }
        if (live && A.traj_x) {
            A.traj_x[(size_t)(s + 1) * plane + idx] = x;
            A.traj_y[(size_t)(s + 1) * plane + idx] = y;
        }
        lvl = nxt;
    }
    if (live) {
        A.x_out[idx] = x;
        A.y_out[idx] = y;
    }
}

// ORDER 3 with such a wind (the reference's default interpolation on float32 reanalysis winds): scipy forms the spline
// coefficients in double (spline_filter(output=float64) inside map_coordinates), samples them in double and returns float32.
// img = the float64 coefficient image of the float32 planes (lc_field_pack(LC_F64_WIND_F32, order 3)); per wave two 16 x 16-node
// tiles (levels t and t + 1, one anchor on the middle of the level's travel) serve the Euler sample and both samples of every
// iteration; scipy's weights (cubic_weights), tap order (tap_sum_order3) and numpy's index map (locate<double, 3>), so the
// result is advect_seed<double, 3, true> with A.wind_f32 bit for bit.  A window outside the tiles reads global memory.
#ifndef LCS_LDS64W_O3_MINWAVES
#define LCS_LDS64W_O3_MINWAVES 2   // 200 vector registers, no spills (a 4 x 4 window of float64 nodes is 64 of them): config 2's shape 15.8 ms;
                                   // 3 waves (168 registers, 34 spilled) 17.4; 4 waves (128, 138 spilled) not run
#endif
constexpr int TW3 = 16, TW3_PITCH = 20;
template <int KFIX, bool CYCLIC>
__global__ void __launch_bounds__(BLOCK, LCS_LDS64W_O3_MINWAVES) advect_lds64w_o3_kernel(const AdvectArgs<double> A0) {
#pragma clang fp contract(off)
    typedef double T;
    const AdvectArgs<double> A = for_member(A0);
    const int K = KFIX >= 0 ? KFIX : A.K;
    __shared__ __attribute__((aligned(16))) d2w s_tiles[BLOCK / 64][2][TW3 * TW3_PITCH];
    if (pole_block<double, POLE_EITHER>(A)) return;
    const int tile_id = xcd_tile_id(A);
    if (tile_id >= A.ntiles) return;  // whole block
    const int tyi = tile_id / A.ntx, txi = tile_id - tyi * A.ntx;
    const int ix = txi * TILE_W + (threadIdx.x % TILE_W), iy = tyi * TILE_H + (threadIdx.x / TILE_W);
    const int lane = threadIdx.x & 63;
    d2w *tile0 = s_tiles[threadIdx.x >> 6][0], *tile1 = s_tiles[threadIdx.x >> 6][1];
    bool live = ix < A.nx && iy < A.ny;
    if (live) {
        const int grow = A.row0 + iy;
        if (grow < A.order || grow >= A.ny_global - A.order) {  // pole rows: generic path (Q3)
            if (!A.pole_blocks) pole_seed<double, POLE_EITHER>(A, iy, ix);
            live = false;
        }
    }
    if (__ballot(live) == 0ull) return;  // whole wave (no workgroup barrier anywhere below)
    const int sx_i = min(ix, A.nx - 1), sy_i = min(iy, A.ny - 1);  // lanes without a seed shadow a neighbour; stores masked
    T x = start_x<T>(A, sy_i, sx_i), y = start_y<T>(A, sy_i, sx_i);
    const T ys = A.seed_lat[sy_i];
    const T cx_conv = T(180) / (T(3.141592653589793 * 6371000.0) * fabs(cos((ys * T(3.141592653589793)) / T(180))));  // Q5
    const T dtcx = A.dt * cx_conv, hdtcx = A.half_dt * cx_conv;
    const size_t idx = live ? (size_t)iy * A.nx + ix : 0, plane = (size_t)A.ny * A.nx;
    if (live && A.traj_x && !A.traj_skip0) {
        A.traj_x[idx] = x;
        A.traj_y[idx] = y;
    }
    const double *lvl = A.img + (size_t)A.t0 * A.level_elems;
    const int pad_cols = A.pitch, pad_rows = A.ny_f + LC_PAD;
    constexpr int CENTRE = TILE_W / 2 + TILE_W * 4;  // middle seed of the wave's 8 x 8 patch
    const int st_row = lane >> 4, st_col = lane & 15;  // staging: one node (16 bytes) per lane and level, 4 rows per pass, 4 passes
    const unsigned st_off = ((unsigned)st_row * (unsigned)pad_cols + (unsigned)st_col) * 16u;
    double dprev_x = 0.0, dprev_y = 0.0;  // previous level's Euler displacement in index space: predicts this level's travel
    // one sample at a located window: out of the tile when the window (padded origin (sy, sx), 4 x 4) lies inside it
    auto sample = [&](const double *level, const d2w *tile, const Tap<T> &tap, int ox, int oy) {
        const int rx = tap.sx - ox, ry = tap.sy - oy;
        Pair<T> r;
        if ((unsigned)rx <= (unsigned)(TW3 - 4) && (unsigned)ry <= (unsigned)(TW3 - 4)) {
            T q[4][8];
#pragma unroll
            for (int a = 0; a < 4; ++a)
#pragma unroll
                for (int b = 0; b < 4; ++b) {
                    const d2w n = tile[(ry + a) * TW3_PITCH + rx + b];
                    q[a][2 * b] = n.x;
                    q[a][2 * b + 1] = n.y;
                }
            r = tap_sum_order3<T>(q, tap);
        } else {
            r = fetch<T, 3>(level, A, tap);
        }
        r.u = (double)(float)r.u;   // map_coordinates returns the field's dtype (Q10)
        r.v = (double)(float)r.v;
        return r;
    };
    for (int s = 0; s < A.nsteps; ++s) {
        const double *nxt = lvl + A.level_elems;
        // ---- 1. both tiles, anchored on the middle of the travel [0, K] Euler displacements the previous level predicts ------
        const double cax = (x - A.lon_min) * A.sx + dprev_x * (0.5 * K), cay = (y - A.lat_min) * A.sy + dprev_y * (0.5 * K);
        const int rxm = __builtin_amdgcn_readlane((int)floor(fmin(fmax(cax, -4.0), 1.0e9)), CENTRE);
        const int rym = __builtin_amdgcn_readlane((int)floor(fmin(fmax(cay, -4.0), 1.0e9)), CENTRE);
        const int ox = min(max(rxm - (TW3 - 4) / 2, 0), pad_cols - TW3), oy = min(max(rym - (TW3 - 4) / 2, 0), pad_rows - TW3);
        {
            d2w st0[TW3 / 4], st1[TW3 / 4];
            const char *src = (const char *)lvl + ((size_t)oy * pad_cols + ox) * 16, *srcn = src + A.level_elems * sizeof(double);
#pragma unroll
            for (int r = 0; r < TW3 / 4; ++r) {
                __builtin_memcpy(&st0[r], src + (size_t)(r * 4) * pad_cols * 16 + st_off, 16);
                if (K > 0) __builtin_memcpy(&st1[r], srcn + (size_t)(r * 4) * pad_cols * 16 + st_off, 16);
            }
            __builtin_amdgcn_wave_barrier();  // the previous level's reads are done (LDS ops of a wave are in order)
#pragma unroll
            for (int r = 0; r < TW3 / 4; ++r) {
                tile0[(r * 4 + st_row) * TW3_PITCH + st_col] = st0[r];
                if (K > 0) tile1[(r * 4 + st_row) * TW3_PITCH + st_col] = st1[r];
            }
            __builtin_amdgcn_wave_barrier();
        }
        // ---- 2. Euler sample ------------------------------------------------------------------------------------------------
        const double x0p = x, y0p = y;
        const Pair<T> e = sample(lvl, tile0, locate<T, 3, true>(A, x, y), ox, oy);   // trajectory.py:82-84
        y = y + lat_increment<T>(A, A.dtcy, e.v);                                     // :86
        x = axpy<T>(dtcx, e.u, x);                                                    // :87
        clamp_position<T>(A, x, y);
        dprev_x = (x - x0p) * A.sx;
        dprev_y = (y - y0p) * A.sy;
        // ---- 3. K iterations, one located window, both levels ------------------------------------------------------------------
#pragma unroll 1
        for (int k = 0; k < K; ++k) {
            const Tap<T> tap = locate<T, 3, true>(A, x, y);
            const Pair<T> c = sample(lvl, tile0, tap, ox, oy);   // :105,107
            __builtin_amdgcn_sched_barrier(0);                  // (one 64-register window at a time)
            const Pair<T> n = sample(nxt, tile1, tap, ox, oy);   // :106,108
            y = y + lat_increment<T>(A, A.hdtcy, settls_bracket<T>(A, e.v, c.v, n.v));               // :110
            x = axpy<T>(hdtcx, settls_bracket<T>(A, e.u, c.u, n.u), x);                              // :112
            clamp_position<T>(A, x, y);
        }
        if (live && A.traj_x) {
            A.traj_x[(size_t)(s + 1) * plane + idx] = x;
            A.traj_y[(size_t)(s + 1) * plane + idx] = y;
        }
        lvl = nxt;
    }
    if (live) {
        A.x_out[idx] = x;
        A.y_out[idx] = y;
    }
}

// Member groups (PATCH_PAIR): the arguments of member q of this workgroup's group for the launch's level window
// -- its own steps only (the generic per-seed path of the pole rows runs member by member).
template <typename T>
__device__ __forceinline__ AdvectArgs<T> group_member(const AdvectArgs<T> &A, int q) {
    AdvectArgs<T> M = A;
    const lcplan::Window w = lcplan::member_window(q, A.pair_l0, A.nsteps, A.pair_n, A.pair_d);
    M.t0 = A.t0 + (w.lo - A.pair_l0);
    M.nsteps = max(w.hi - w.lo, 0);
    const size_t off = (size_t)q * A.pair_plane;
    M.x_out = A.x_out + off;
    M.y_out = A.y_out + off;
    if (A.x_start) {
        M.x_start = A.x_start + off;
        M.y_start = A.y_start + off;
    }
    return M;
}
template <typename T>
__device__ __forceinline__ int group_count(const AdvectArgs<T> &A) { return blockIdx.y + 1 == gridDim.y ? A.pair_last : A.pair_g; }
template <typename T>
__device__ __forceinline__ bool pole_block_group(const AdvectArgs<T> &A) {
    if ((int)blockIdx.x >= A.pole_blocks) return false;
    const int cnt = group_count(A);
    for (int q = 0; q < cnt; ++q) pole_block<T, POLE_LIN>(group_member(A, q));  // (float32 order 1: the lin image is the interior image too)
    return true;
}

// ======================================================================================
// float fast path (interior rows).  Same algorithm as advect_seed<float,...>; arranged so
// that one sample position costs the VALU as little as possible, because rocprof shows
// the fused kernel is VALU-issue bound (SQ_ACTIVE_INST_VALU ~ 94 % of SIMD cycles):
//   * the tap window is located ONCE per position (shared by time levels t and t+1);
//   * in-range test = one unsigned compare per axis on the float's bits (catches c<0,
//     c>n-1 and NaN), the exact scipy wrap runs in the rare out-of-range branch;
//   * index clamp is a float med3, the address is a 24-bit mad off a uniform level base;
//   * (u,v) pairs are processed as 2-vectors (v_pk_fma_f32).
// ======================================================================================
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));

struct TapF {
    unsigned byte_off;  // of the window origin inside one time level
    float tx, ty;
};

// floor(x) as an integer in ONE instruction (the compiler expands __float2int_rd to v_floor + v_cvt)
__device__ __forceinline__ unsigned floor_to_uint(float x) {
    int r;
    asm("v_cvt_flr_i32_f32 %0, %1" : "=v"(r) : "v"(x));
    return (unsigned)r;
}

__device__ __forceinline__ TapF locate_wrap_f(const AdvectArgs<float> &A, float x, float y, unsigned origin_bytes) {
    float cx = (x - A.lon_min) * A.sx;  // subtract first: exact at the grid origin
    float cy = (y - A.lat_min) * A.sy;
    const float szx = (float)(A.nx_f - 1), szy = (float)(A.ny_f - 1);
    if ((unsigned)(__float_as_uint(cx) > __float_as_uint(szx)) | (unsigned)(__float_as_uint(cy) > __float_as_uint(szy))) {
        cx = wrap_coord<float>(cx, szx);
        cy = wrap_coord<float>(cy, szy);
    }
    // floor-convert + fract: one instruction each (v_cvt_flr_i32_f32, v_fract_f32).  The coordinate is
    // non-negative here; the unsigned min is memory safety (NaN converts to 0, garbage saturates).
    TapF t;
    t.tx = __builtin_amdgcn_fractf(cx);
    t.ty = __builtin_amdgcn_fractf(cy);
    const unsigned x0 = min(floor_to_uint(cx), (unsigned)(A.nx_f - 1));
    const unsigned y0 = min(floor_to_uint(cy), (unsigned)(A.ny_f - 1));
    t.byte_off = (__umul24(y0, (unsigned)A.pitch) + x0) * 8u + origin_bytes;
    return t;
}

__device__ __forceinline__ f2 fetch1_f(const float *__restrict__ lvl, const TapF &t, unsigned row_bytes) {
    // two uniform bases + one 32-bit lane offset: both loads take the SGPR-base addressing form
    const char *row0 = (const char *)lvl, *row1 = row0 + row_bytes;
    f4 a, b;
    __builtin_memcpy(&a, row0 + t.byte_off, 16);  // {u00, v00, u01, v01}
    __builtin_memcpy(&b, row1 + t.byte_off, 16);  // {u10, v10, u11, v11}
    const f2 r0 = a.xy + t.tx * (a.zw - a.xy);
    const f2 r1 = b.xy + t.tx * (b.zw - b.xy);
    return r0 + t.ty * (r1 - r0);
}

__device__ __forceinline__ void cubic_weights_f(float t, float w[4]) {
    const float z = 1.0f - t, s = 1.0f / 6.0f;
    w[0] = z * z * z * s;
    w[1] = (t * t * (t - 2.0f) * 3.0f + 4.0f) * s;
    w[2] = (z * z * (z - 2.0f) * 3.0f + 4.0f) * s;
    w[3] = 1.0f - w[0] - w[1] - w[2];
}

__device__ __forceinline__ f2 fetch3_f(const float *__restrict__ lvl, const TapF &t, unsigned row_bytes,
                                       const float wx[4], const float wy[4]) {
    const char *p = (const char *)lvl + t.byte_off;
    f2 acc = {0.0f, 0.0f};
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        f4 lo, hi;
        __builtin_memcpy(&lo, p + a * row_bytes, 16);
        __builtin_memcpy(&hi, p + a * row_bytes + 16, 16);
        const f2 r = wx[0] * lo.xy + wx[1] * lo.zw + wx[2] * hi.xy + wx[3] * hi.zw;
        acc += wy[a] * r;
    }
    return acc;
}

template <int ORDER>
__device__ __forceinline__ f2 fetch_f(const float *__restrict__ lvl, const TapF &t, unsigned row_bytes,
                                      const float wx[4], const float wy[4]) {
    if (ORDER == 3) return fetch3_f(lvl, t, row_bytes, wx, wy);
    return fetch1_f(lvl, t, row_bytes);
}

__device__ __forceinline__ void clamp_position_f(const AdvectArgs<float> &A, float &x, float &y) {
    // trajectory.py:89-90 in one v_med3_f32; a NaN input makes med3 return min3 = y_min, which is Q8's rule
    y = __builtin_amdgcn_fmed3f(y, A.y_min, A.y_max);
    if (A.cyclic) {
        if (!(fabsf(x) < 180.0f)) {  // rare: the exact reference sequence, trajectory.py:93-94 (Q7)
            if (!(x > -180.0f)) x = pymod180<float>(x);
            if (!(x < 180.0f)) x = -180.0f + pymod180<float>(x);
        }
    } else if (x < A.x_min || x > A.x_max) {  // NaN stays NaN, as in the reference
        if (A.clamp_flag) *A.clamp_flag = 1u;
        x = x < A.x_min ? A.x_min : A.x_max;
    }
}

template <int ORDER>
__device__ void advect_seed_f32(const AdvectArgs<float> &A, int iy, int ix) {
#pragma clang fp contract(fast)
    static_assert(ORDER >= 1 && ORDER <= 5, "interp_order");
    if (ORDER != 1 && ORDER != 3) return;  // general orders go through advect_seed (InteriorPath)
    float x = start_x<float>(A, iy, ix);
    float y = start_y<float>(A, iy, ix);
    const float ys = A.seed_lat[iy];
    const float cx_conv =
        180.0f / ((float)(3.141592653589793 * 6371000.0) * fabsf(cosf((ys * (float)3.141592653589793) / 180.0f)));
    const float dtcx = A.dt * cx_conv, hdtcx = A.half_dt * cx_conv;
    const size_t idx = (size_t)iy * A.nx + ix;
    const size_t plane = (size_t)A.ny * A.nx;
    const unsigned row_bytes = (unsigned)A.pitch * 8u;
    const unsigned origin = ORDER == 3 ? 0u : row_bytes + 8u;  // order 3 window starts one node up/left
    if (A.traj_x && !A.traj_skip0) {
        A.traj_x[idx] = x;
        A.traj_y[idx] = y;
    }
    const float *lvl = A.img + (size_t)A.t0 * A.level_elems;
    const float *elv = A.ext ? A.ext + (size_t)A.t0 * A.level_elems : nullptr;
    float wx[4], wy[4];
    for (int s = 0; s < A.nsteps; ++s) {
        const float *nxt = lvl + A.level_elems;
        TapF t = locate_wrap_f(A, x, y, origin);
        if (ORDER == 3) {
            cubic_weights_f(t.tx, wx);
            cubic_weights_f(t.ty, wy);
        }
        const f2 e = fetch_f<ORDER>(lvl, t, row_bytes, wx, wy);
        y = fmaf(A.dtcy, e.y, y);
        x = fmaf(dtcx, e.x, x);
        clamp_position_f(A, x, y);
        for (int k = 0; k < A.K; ++k) {
            t = locate_wrap_f(A, x, y, origin);
            if (ORDER == 3) {
                cubic_weights_f(t.tx, wx);
                cubic_weights_f(t.ty, wy);
            }
            f2 d;
            if (elv) {  // uniform branch: one gather of (2 F[t] - F[t+1])
                d = e + fetch_f<ORDER>(elv, t, row_bytes, wx, wy);
            } else {
                const f2 c = fetch_f<ORDER>(lvl, t, row_bytes, wx, wy);
                const f2 n = fetch_f<ORDER>(nxt, t, row_bytes, wx, wy);
                d = (e + 2.0f * c) - n;
            }
            y = fmaf(A.hdtcy, d.y, y);
            x = fmaf(hdtcx, d.x, x);
            clamp_position_f(A, x, y);
        }
        if (A.traj_x) {
            A.traj_x[(size_t)(s + 1) * plane + idx] = x;
            A.traj_y[(size_t)(s + 1) * plane + idx] = y;
        }
        lvl = nxt;
        if (elv) elv += A.level_elems;
    }
    A.x_out[idx] = x;
    A.y_out[idx] = y;
}

// ======================================================================================
// LDS-staged variant of the float path (used when the fused-level image ext is given).
//
// rocprof on the direct-gather kernel: TCP_TOTAL_CACHE_ACCESSES ~ 1 per CU-cycle, ~27 tag
// lookups per 16-byte gather instruction -- the vector L1's lookup rate, not HBM, bounds it.
// A wave's 64 seeds (8 x 8 patch) sit within a few field cells of each other and a SETTLS
// sub-step moves them a fraction of a cell, so the K iterations of one time level read a
// window of ext[t] a few nodes wide.  Per time level each WAVE:
//   1. anchors a fixed-size tile (TileGeom<ORDER> nodes) on the patch's centre lane, shifted to the
//      middle of that lane's predicted travel -- two v_readlane, no reduction -- and issues the
//      tile's coalesced 16-byte row loads.  Order 1 predicts the travel from the PREVIOUS level's
//      Euler displacement, so these loads fly together with step 2's gather; order 3 (4 loads per
//      lane) anchors on this level's displacement after step 2;
//   2. takes the Euler sample with direct gathers from img[t];
//   3. writes the tile of ext[t] into its own LDS region;
//   4. runs the K iterations reading windows with ds_read2_b64; a lane whose window falls
//      outside the tile (jets, polar rows, the +-180 seam, patches the flow has stretched) redoes
//      that sample with the exact sequence and a global gather.
// Tiles are per wave, so there is no workgroup barrier anywhere (LDS operations of one wave
// execute in order) and waves of a block drift freely.  Same arithmetic as the direct-gather
// float path; only the memory the window is read from differs.
// ======================================================================================
// Tile geometry per interpolation order (nodes).  Measured on C3 with 8x8-seed waves (ms):
//   order 1: 16x4 11.9, 16x8 10.2, 16x16 10.2, 32x8 10.6     order 3: 16x8 22.7, 16x16 21.2, 32x8 24.3, 32x16 20.4
template <int ORDER>
struct TileGeom {
#ifndef LCS_O3_COLS
#define LCS_O3_COLS 32
#endif
#ifndef LCS_O3_ROWS
#define LCS_O3_ROWS 16
#endif
    static constexpr int COLS = ORDER == 3 ? LCS_O3_COLS : 16;   // one row = COLS/2 lanes x 16 B
    static constexpr int ROWS = ORDER == 3 ? LCS_O3_ROWS : 8;
#ifndef LCS_O3_PITCH
#define LCS_O3_PITCH (LCS_O3_COLS + 4)  // 36 nodes: the 4 x 4 window reads of neighbouring rows fall into different banks (two-seed kernel on C3: 15.45 against 15.6-15.9 ms with 34)
#endif
    static constexpr int PITCH = ORDER == 3 ? LCS_O3_PITCH : COLS + 2;  // rows stay 16-byte aligned
    static constexpr int LANES_PER_ROW = COLS / 2;
    static constexpr int ROWS_PER_PASS = 64 / LANES_PER_ROW;
};

// Order 3 also takes the Euler sample out of LDS: its direct form is 8 sixteen-byte gathers per lane and level
// (~27 vector-L1 tag lookups each: 13 of the kernel's 20 ms of TCP time on C3, next to 13.5 ms of VALU work),
// while a small tile of img[t] around the patch's CURRENT position is one coalesced 16-byte load per lane.
// (Order 1 gathers 2 x 16 B per lane; there the tile was measured slower, 8.2 vs 7.6 ms.)
#ifndef LCS_E_ROWS
#define LCS_E_ROWS 8
#endif
#ifndef LCS_E_PITCH
#define LCS_E_PITCH 18
#endif
template <int ORDER>
struct EulerGeom {
    static constexpr bool ON = ORDER == 3;
    static constexpr int COLS = 16, ROWS = LCS_E_ROWS, PITCH = LCS_E_PITCH;
    static constexpr int LANES_PER_ROW = COLS / 2, ROWS_PER_PASS = 64 / LANES_PER_ROW, NPASS = ROWS / ROWS_PER_PASS;
    static constexpr int ELEMS = ON ? ROWS * PITCH : 0;
};

struct TapL {
    int x0, y0;  // floor of the index-space coordinate (the window origin follows from ORDER)
    float tx, ty;
};

// Index-space coordinate of a position (tools.py:21-22 with the float path's multiply form) after scipy's
// 'wrap' map.  Packed: one v_pk_add_f32 + one v_pk_mul_f32.  The in-range test is one unsigned compare
// per axis on the float's bits (catches c < 0, c > n-1 and NaN); the exact wrap runs in the rare branch.
__device__ __forceinline__ f2 index_coords(const AdvectArgs<float> &A, f2 p) {
    f2 c = (p - (f2){A.lon_min, A.lat_min}) * (f2){A.sx, A.sy};
    const float szx = (float)(A.nx_f - 1), szy = (float)(A.ny_f - 1);
    if ((unsigned)(__float_as_uint(c.x) > __float_as_uint(szx)) | (unsigned)(__float_as_uint(c.y) > __float_as_uint(szy))) {
        c.x = wrap_coord<float>(c.x, szx);
        c.y = wrap_coord<float>(c.y, szy);
    }
    return c;
}

// Window origin and fractions of a wrapped coordinate.  No index clamp here: a wrapped finite coordinate
// lies in [0, n-1] (+ rounding < 1), NaN converts to 0; the LDS path range-checks the tile-relative
// index anyway and the global path clamps (window_global).
__device__ __forceinline__ TapL tap_of(f2 c) {
    TapL t;
    t.tx = __builtin_amdgcn_fractf(c.x);
    t.ty = __builtin_amdgcn_fractf(c.y);
    t.x0 = (int)floor_to_uint(c.x);
    t.y0 = (int)floor_to_uint(c.y);
    return t;
}

// start + the sample of one time level at a located window, gathered from global memory
template <int ORDER>
__device__ __forceinline__ f2 window_global(const float *__restrict__ lvl, const AdvectArgs<float> &A, const TapL &t, f2 start);

// ``tile_addr``: LDS byte address of the wave's tile.  The window address is one 24-bit mad + one
// shift-add (left to itself the compiler picks the quarter-rate v_mul_lo_u32 here).
typedef __attribute__((address_space(3))) const f2 lds_f2;

__device__ __forceinline__ unsigned lds_address(const void *shared_ptr) {
    return (unsigned)(size_t)(__attribute__((address_space(3))) const char *)shared_ptr;
}

// One ds_read_b64 per node, not ds_read2_b64 pairs: on gfx950 a wave's ds_read_b64 takes 2 LDS cycles (64
// banks, 32-lane groups) while ds_read2_b64 takes 8 for twice the bytes (32 banks, 16-lane groups) -- half
// the bandwidth, and conflicts between a node and its lower-left neighbour at these pitches (order 3 on C3:
// LDS busy 81 % -> 52 % of the cycles).  The load/store optimiser pairs plain loads; volatile ones are left alone.
typedef volatile lds_f2 lds_node;

// NOPS: two wait states in front of the multiply-add.  Its SGPR operand is loop-invariant, but under register pressure hipcc
// parks it in a VGPR lane and reloads it (v_readlane_b32) right before the use -- a VALU-written SGPR read by a VALU
// instruction needs 2 wait states on gfx950, and hipcc does not insert them in front of asm (tools/asm_hazards.py found
// exactly that in the order-3 verify instance, the one kernel with the pressure; the product instances carry no such
// reload, which tests/test_asm_hazards.py checks on every build).
template <int LT_PITCH, bool NOPS = false>
__device__ __forceinline__ lds_node *window_origin(unsigned tile_addr, unsigned pitch_bytes, int rx, int ry) {
    unsigned row_addr;
    if (NOPS)
        asm("s_nop 1\n\tv_mad_u32_u24 %0, %1, %2, %3" : "=v"(row_addr) : "v"(ry), "s"(pitch_bytes), "v"(tile_addr));
    else
        asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(row_addr) : "v"(ry), "s"(pitch_bytes), "v"(tile_addr));
    return (lds_node *)(size_t)(row_addr + ((unsigned)rx << 3));
}

// order 1: the 2x2 window and its two lerps
template <int LT_PITCH, bool NOPS = false>
__device__ __forceinline__ f2 window_lds1(unsigned tile_addr, unsigned pitch_bytes, int rx, int ry, const TapL &t) {
    lds_node *p = window_origin<LT_PITCH, NOPS>(tile_addr, pitch_bytes, rx, ry);
    const f2 n00 = p[0], n01 = p[1], n10 = p[LT_PITCH], n11 = p[LT_PITCH + 1];
    const f2 r0 = n00 + t.tx * (n01 - n00);
    const f2 r1 = n10 + t.tx * (n11 - n10);
    return r0 + t.ty * (r1 - r0);
}

// order 3: the 16 window reads are issued back to back BEFORE the weights are formed, so the LDS latency is
// covered by the wave's own arithmetic (left alone, the compiler forms the weights first and then waits for
// each row in turn).  Weights as (x, y) pairs: one packed polynomial for both axes.
struct CubicW {
    f2 w0, w1, w2, w3;  // {wx_k, wy_k}
};
// explicit fused multiply-adds: the rounding of these two functions must not depend on the call site
__device__ __forceinline__ f2 pkfma(f2 a, f2 b, f2 c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f2 splat(float v) { return (f2){v, v}; }

__device__ __forceinline__ CubicW cubic_weights_p(float tx, float ty) {
    // scipy's cubic B-spline weights (get_spline_interpolation_weights) as polynomials in t, 11 packed ops:
    //   w3 = t^3/6,  w0 = (1-t)^3/6 = (1/6 - t/2 + t^2/2) - w3,  w1 = 2/3 + t^2 (t/2 - 1),
    //   w2 = 1/6 + t/2 + t^2 (1/2 - t/2)                                   (the four sum to 1)
#pragma clang fp contract(off)
    const f2 t = {tx, ty};
    const f2 tt = t * t;
    CubicW w;
    w.w3 = tt * (t * splat(1.0f / 6.0f));
    w.w0 = pkfma(tt, splat(0.5f), pkfma(t, splat(-0.5f), splat(1.0f / 6.0f))) - w.w3;
    w.w1 = pkfma(tt, pkfma(t, splat(0.5f), splat(-1.0f)), splat(2.0f / 3.0f));
    w.w2 = pkfma(tt, pkfma(t, splat(-0.5f), splat(0.5f)), pkfma(t, splat(0.5f), splat(1.0f / 6.0f)));
    return w;
}
// ``start`` + the 16-tap sum (the caller's Euler velocity rides in on the first row's fma).  ONE function for
// the LDS window and for the global-gather fallback: a lane's result must not depend on which of the two
// served it (tile placement differs between sharded and unsharded runs, results must not).
__device__ __forceinline__ f2 cubic_apply(const f2 (&q)[4][4], const TapL &t, f2 start) {
#pragma clang fp contract(off)
    const CubicW w = cubic_weights_p(t.tx, t.ty);
    const f2 wx[4] = {splat(w.w0.x), splat(w.w1.x), splat(w.w2.x), splat(w.w3.x)};
    const f2 wy[4] = {splat(w.w0.y), splat(w.w1.y), splat(w.w2.y), splat(w.w3.y)};
    f2 acc = start;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const f2 r = pkfma(wx[3], q[a][3], pkfma(wx[2], q[a][2], pkfma(wx[1], q[a][1], wx[0] * q[a][0])));
        acc = pkfma(wy[a], r, acc);
    }
    return acc;
}
template <int LT_PITCH, bool NOPS = false>
__device__ __forceinline__ f2 window_lds3(unsigned tile_addr, unsigned pitch_bytes, int rx, int ry, const TapL &t, f2 start) {
    lds_node *p = window_origin<LT_PITCH, NOPS>(tile_addr, pitch_bytes, rx, ry);
    f2 q[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) q[a][b] = p[a * LT_PITCH + b];
    __builtin_amdgcn_sched_barrier(0);  // all 16 reads in flight before the weights are formed
    return cubic_apply(q, t, start);
}

// returns start + interpolated (u, v)
template <int ORDER, int LT_PITCH, bool NOPS = false>
__device__ __forceinline__ f2 window_lds(unsigned tile_addr, unsigned pitch_bytes, int rx, int ry, const TapL &t, f2 start) {
    if (ORDER == 3) return window_lds3<LT_PITCH, NOPS>(tile_addr, pitch_bytes, rx, ry, t, start);
    return start + window_lds1<LT_PITCH, NOPS>(tile_addr, pitch_bytes, rx, ry, t);
}

template <int ORDER>
__device__ __forceinline__ f2 window_global(const float *__restrict__ lvl, const AdvectArgs<float> &A, const TapL &t, f2 start) {
    // order 1 window starts at padded (y0+1, x0+1); order 3 one node up/left of that, i.e. padded (y0, x0).
    // The unsigned min is memory safety (garbage coordinates saturate).
    const unsigned x0 = min((unsigned)t.x0, (unsigned)(A.nx_f - 1)), y0 = min((unsigned)t.y0, (unsigned)(A.ny_f - 1));
    const unsigned row_bytes = (unsigned)A.pitch * 8u;
    const unsigned byte_off = (__umul24(y0, (unsigned)A.pitch) + x0) * 8u + (ORDER == 3 ? 0u : ((unsigned)A.pitch + 1u) * 8u);
    if (ORDER == 3) {
        const char *p = (const char *)lvl + byte_off;
        f2 q[4][4];
#pragma unroll
        for (int a = 0; a < 4; ++a) {
            f4 lo, hi;
            __builtin_memcpy(&lo, p + a * row_bytes, 16);
            __builtin_memcpy(&hi, p + a * row_bytes + 16, 16);
            q[a][0] = lo.xy;
            q[a][1] = lo.zw;
            q[a][2] = hi.xy;
            q[a][3] = hi.zw;
        }
        return cubic_apply(q, t, start);
    }
    TapF g;
    g.byte_off = byte_off;
    g.tx = t.tx;
    g.ty = t.ty;
    return start + fetch1_f(lvl, g, row_bytes);
}

// The Euler sample: the sample alone (order 1: no `0 + ...` in front of it -- the compiler may not drop an addition of +0,
// x + 0 differs from x for x = -0 -- one packed add per seed and level less; the value is the direct-gather kernel's).
template <int ORDER>
__device__ __forceinline__ f2 euler_global(const float *__restrict__ lvl, const AdvectArgs<float> &A, const TapL &t) {
    if (ORDER == 3) return window_global<3>(lvl, A, t, (f2){0.0f, 0.0f});
    const unsigned x0 = min((unsigned)t.x0, (unsigned)(A.nx_f - 1)), y0 = min((unsigned)t.y0, (unsigned)(A.ny_f - 1));
    TapF g;
    g.byte_off = (__umul24(y0, (unsigned)A.pitch) + x0) * 8u + ((unsigned)A.pitch + 1u) * 8u;
    g.tx = t.tx;
    g.ty = t.ty;
    return fetch1_f(lvl, g, (unsigned)A.pitch * 8u);
}

// trajectory.py:89-94 on a packed position: latitude clamp in one v_med3_f32 (a NaN input makes med3
// return min3 = y_min, which is Q8's rule; ``ymax_v`` lives in a VGPR because a VOP3 takes one SGPR);
// the cyclic wrap is the exact reference sequence in a rare branch.
__device__ __forceinline__ void clamp_position_p(const AdvectArgs<float> &A, f2 &p, float ymax_v) {
    p.y = __builtin_amdgcn_fmed3f(p.y, A.y_min, ymax_v);
    if (A.cyclic) {
        if (!(fabsf(p.x) < 180.0f)) {  // rare (Q7)
            float x = p.x;
            if (!(x > -180.0f)) x = pymod180<float>(x);
            if (!(x < 180.0f)) x = -180.0f + pymod180<float>(x);
            p.x = x;
        }
    } else if (p.x < A.x_min || p.x > A.x_max) {  // NaN stays NaN, as in the reference
        if (A.clamp_flag) *A.clamp_flag = 1u;
        p.x = p.x < A.x_min ? A.x_min : A.x_max;
    }
}

// ... with the boundary kind known at compile time (the kernels are instantiated per kind: no uniform branch on A.cyclic,
// one v_cmp + one exec-masked rare block per call)
template <bool CYCLIC>
__device__ __forceinline__ void clamp_position_c(const AdvectArgs<float> &A, f2 &p, float ymax_v) {
    p.y = __builtin_amdgcn_fmed3f(p.y, A.y_min, ymax_v);
    if (CYCLIC) {
        if (!(fabsf(p.x) < 180.0f)) {  // rare (Q7)
            float x = p.x;
            if (!(x > -180.0f)) x = pymod180<float>(x);
            if (!(x < 180.0f)) x = -180.0f + pymod180<float>(x);
            p.x = x;
        }
    } else if (p.x < A.x_min || p.x > A.x_max) {  // NaN stays NaN, as in the reference
        if (A.clamp_flag) *A.clamp_flag = 1u;
        p.x = p.x < A.x_min ? A.x_min : A.x_max;
    }
}

// DEFER_X (round 6): the longitude wrap / clamp of trajectory.py:93-97,119-123 is deferred exactly as the latitude clamp
// already was.  A longitude the reference would touch -- x <= -180 or x >= 180 (cyclic), x < x_min or x > x_max (not) -- maps
// to an index outside [0, n-1), which the NEXT sample's window test flags by itself, and the exact-redo path starts by
// applying both clamps to the position it was handed; the level ends with one clamp of each kind (stores, next Euler
// sample).  Same values as clamping after every update (the clamps are pure functions of the position), one v_cmp + one
// s_or fewer per sample: 10 VALU and 10 SALU of the 260 / 134 per wave-level.  -DLCS_LDS2_DEFER_X=0 is the round-5 form.
#ifndef LCS_LDS2_DEFER_X
#define LCS_LDS2_DEFER_X 1
#endif
// Control flow: rocprof shows the VALU (~80 % of issue slots) and the CU's scalar pipe (SALU + branches,
// ~75 %) saturating together, so the loop is written with ONE rare branch per sample instead of one per
// special case.  Every lane first runs the common case unconditionally -- coordinate already in
// [0, n-1), window inside the wave's tile, new longitude strictly inside the wrap/clamp bounds -- and
// records in ``bad`` whether any assumption failed; only then the exact sequence (scipy's wrap map, a
// global gather, the reference's clamps) is re-run for those lanes from the saved position.  The common
// case never faults on garbage: LDS reads cannot fault (out-of-range DS reads return 0) and the global
// gather clamps its indices.  Lanes without a seed (grid edge, pole rows) shadow a neighbouring seed so
// that they follow the same path; only their stores are masked.
#ifndef LCS_O3_MINWAVES
#define LCS_O3_MINWAVES 1
#endif
#ifndef LCS_LDS_NUM_SGPR
#define LCS_LDS_NUM_SGPR 0
#endif
enum Patch { PATCH_TALL = 0, PATCH_WIDE = 1, PATCH_LINES = 2, PATCH_PAIR = 3 };
// The whole-line trajectory store.  Measured on C3 with return_traj (advect ms, profiles/r03): plain 9.24-9.35, nontemporal
// (`nt`: the written lines do not push the wind tiles out of the XCD's L2) 8.27-8.29, write-through sc1 9.16-9.29,
// sc0 sc1 9.25-9.36.  -DLCS_TRAJ_STORE_KIND=0 plain, 1 nt (default), 2 sc1, 3 sc0 sc1.
#ifndef LCS_TRAJ_STORE_KIND
#define LCS_TRAJ_STORE_KIND 1
#endif
__device__ __forceinline__ void traj_store_line(float *dst, f4 v) {
#if LCS_TRAJ_STORE_KIND == 1
    __builtin_nontemporal_store(v, (f4 *)dst);
#elif LCS_TRAJ_STORE_KIND == 2
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" : : "v"(dst), "v"(v) : "memory");  // (s_nop: the >8-byte store-data hazard the compiler cannot see inside asm)
#elif LCS_TRAJ_STORE_KIND == 3
    asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1\n\ts_nop 1" : : "v"(dst), "v"(v) : "memory");
#else
    *(f4 *)dst = v;
#endif
}
#define LCS_TRAJ_STORE(dst, v) traj_store_line(dst, v)

// LINES (with return_traj on grids whose rows are 16-byte aligned): the four waves of a workgroup sit side by side in
// longitude (32 x 8 seeds) instead of stacked (8 x 32), put their positions into an LDS slab after each time level, meet at
// one workgroup barrier, and waves 0 and 1 write the longitudes' and the latitudes' 8 rows x 32 columns as whole 128-byte
// lines, non-temporal -- the two-seed kernels' PATCH_LINES (see there for the counters) for launches below their size.
// VERIFY (lc_ctx_set_verify; without LINES only): the same kernel with a wave-state audit after each level's iterations
// -- does the tile in LDS still hold what this wave staged, is the wave still in the hardware slot it started in -- counted
// into A.verify.  A wave's tile and registers are its own for the whole launch (no other wave writes them), so a count can only
// come from outside the kernel: the wave's context having been saved and restored by the driver (two processes time-sharing
// a GPU), or hardware.  Results are bit-identical to the plain instance; DESIGN.md section 8 says what it is for.
// A.verify[LC_VERIFY_WORDS]: [0] wave-levels whose tile changed, [1] 16-byte entries changed, [2] wave slot changes, [3] wave-levels audited,
// [4..11] first event: block, tile, wave, level, HW_ID before / after, lane mask lo / hi, [12] first-event latch, [15] 0xBAD: inject one (test hook)
static_assert(LC_VERIFY_WORDS >= 16, "lc_ctx_read_verify's counter block");
template <int ORDER, int KFIX, bool CYCLIC, bool LINES, bool VERIFY = false>
__global__ void __launch_bounds__(BLOCK, ORDER == 3 ? LCS_O3_MINWAVES : 1)
#if LCS_LDS_NUM_SGPR > 0
    __attribute__((amdgpu_num_sgpr(LCS_LDS_NUM_SGPR)))
#endif
    advect_lds_kernel(const AdvectArgs<float> A0) {
#pragma clang fp contract(fast)
    const AdvectArgs<float> A = for_member(A0);
    constexpr int SLAB1_PITCH = 36;  // floats per slab row (32 + 4)
    __shared__ __attribute__((aligned(16))) float s_slab1[2][2][LINES ? 8 * SLAB1_PITCH : 4];
    const int K = KFIX >= 0 ? KFIX : A.K;  // KFIX: SETTLS_order known at compile time (the iteration loop unrolls)
    // the longitude wrap / clamp deferred like the latitude clamp (the two-seed kernel's DEFER_X, see there)
    constexpr bool DEFER_X = LCS_LDS2_DEFER_X != 0;
    typedef TileGeom<ORDER> G;
    constexpr int LT_COLS = G::COLS, LT_ROWS = G::ROWS, LT_PITCH = G::PITCH;
    constexpr int WIN = ORDER + 1;  // window edge in nodes
    constexpr int WOFF = ORDER == 3 ? 0 : LC_PAD_LO;  // padded window origin = (y0 + WOFF, x0 + WOFF)
    typedef EulerGeom<ORDER> E;
    __shared__ __attribute__((aligned(16))) f2 s_tiles[BLOCK / 64][LT_ROWS * LT_PITCH + E::ELEMS];
    if (pole_block<float, ORDER != 1 ? POLE_EITHER : POLE_LIN>(A)) return;
    const int tile_id = xcd_tile_id(A);
    if (tile_id >= A.ntiles) return;  // whole block
    const int tyi = tile_id / A.ntx, txi = tile_id - tyi * A.ntx;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ix = LINES ? txi * (TILE_W * 4) + wave * TILE_W + (lane % TILE_W) : txi * TILE_W + (threadIdx.x % TILE_W);
    const int iy = LINES ? tyi * 8 + lane / TILE_W : tyi * TILE_H + (threadIdx.x / TILE_W);
    f2 *tile = s_tiles[wave];

    bool live = ix < A.nx && iy < A.ny;
    if (live) {
        const int grow = A.row0 + iy;
        if (grow < A.order || grow >= A.ny_global - A.order) {  // pole rows: generic path, whole integration (Q3)
            if (!A.pole_blocks) pole_seed<float, ORDER != 1 ? POLE_EITHER : POLE_LIN>(A, iy, ix);  // (else the leading workgroups did them)
            live = false;
        }
    }
    // LINES: whole-line stores for workgroups whose 32 columns are all inside the grid; workgroup-uniform, so either all
    // four waves meet at the level's barrier or none does (a wave without seeds then stays in the loop as a shadow)
    const bool lines = LINES && A.traj_x && A.traj_line_ok && (txi + 1) * (TILE_W * 4) <= A.nx;
    if (!lines && __ballot(live) == 0ull) return;  // whole wave (no workgroup barrier below unless `lines`)
    bool sl_ok = false;
    size_t sl_idx = 0;
    if (lines) {
        const int iyr = tyi * 8 + (lane >> 3), grow = A.row0 + iyr;
        sl_ok = wave < 2 && iyr < A.ny && grow >= A.order && grow < A.ny_global - A.order;  // (pole rows: written by their own threads)
        sl_idx = (size_t)min(iyr, A.ny - 1) * A.nx + (size_t)txi * (TILE_W * 4) + (lane & 7) * 4;
    }
    // (longitude, latitude) in adjacent registers: index map and position update are packed operations
    const int sx_i = min(ix, A.nx - 1), sy_i = min(iy, A.ny - 1);
    f2 p = {start_x<float>(A, sy_i, sx_i), start_y<float>(A, sy_i, sx_i)};
    const float ys = A.seed_lat[sy_i];  // conversion_x is a function of the SEED latitude (Q5), wherever the parcel is now
    const float cx_conv =
        180.0f / ((float)(3.141592653589793 * 6371000.0) * fabsf(cosf((ys * (float)3.141592653589793) / 180.0f)));
    const f2 dd = {A.dt * cx_conv, A.dtcy};        // degrees per (m/s) over a full step (trajectory.py:55-57,86-87)
    const f2 hd = {A.half_dt * cx_conv, A.hdtcy};  // ... over a SETTLS half step (trajectory.py:110-112)
    const size_t idx = live ? (size_t)iy * A.nx + ix : 0;
    const size_t plane = (size_t)A.ny * A.nx;
    if (live && A.traj_x && !A.traj_skip0) {
        A.traj_x[idx] = p.x;
        A.traj_y[idx] = p.y;
    }
    float ymax_v = A.y_max;
    asm volatile("" : "+v"(ymax_v));  // keep it in a VGPR (a VOP3 takes one SGPR; see clamp_position_p)
    const unsigned tile_addr = lds_address(tile);
    unsigned pitch_bytes = (unsigned)LT_PITCH * 8u;
    asm volatile("" : "+s"(pitch_bytes));  // one SGPR for the whole kernel (a VOP3 literal is not encodable)
    f2 *etile = tile + LT_ROWS * LT_PITCH;  // Euler tile of img[t] (order 3)
    const unsigned etile_addr = lds_address(etile);
    unsigned epitch_bytes = (unsigned)E::PITCH * 8u;
    asm volatile("" : "+s"(epitch_bytes));
    const int e_row = lane / E::LANES_PER_ROW, e_col = (lane % E::LANES_PER_ROW) * 2;
    const f2 pmin = {A.lon_min, A.lat_min}, sc = {A.sx, A.sy};
    // Index-space coordinate, subtract first (tools.py:21-22): a seed sitting exactly on the grid origin must map
    // to exactly 0 -- the one-fma form p*sc - (pmin*sc) leaves a residual of up to +-3e-5 there, and a negative
    // one wraps to the far end of the periodic axis (measured 2 % faster, rejected).  Same expression as
    // index_coords(), so the exact-redo path sees the same coordinate as the common path.
    auto to_index = [&](f2 q) { return (q - pmin) * sc; };
    // the common case needs the new longitude strictly inside these bounds (Q7 wrap / Q9 clamp otherwise)
    // (cyclic: |x| < 180 is one compare with a source modifier; hence the template parameter)
    const float xlo = A.x_min, xhi = A.x_max;
    auto x_needs_care = [&](float x) { return CYCLIC ? !(fabsf(x) < 180.0f) : !((x > xlo) & (x < xhi)); };
    const float *lvl = A.img + (size_t)A.t0 * A.level_elems;
    const float *elv = A.ext + (size_t)A.t0 * A.level_elems;
    const int pad_cols = A.pitch, pad_rows = A.ny_f + LC_PAD;  // >= LT_COLS, LT_ROWS (checked by the launcher)
    const float kpred = 0.5f * (float)(K > 0 ? K - 1 : 0);  // half of the predicted travel, in Euler displacements
    // staging geometry of this lane: ROWS_PER_PASS tile rows per pass, 16 B (2 nodes) per lane
    const int st_row = lane / G::LANES_PER_ROW, st_col = (lane % G::LANES_PER_ROW) * 2;
    const unsigned st_off = ((unsigned)st_row * (unsigned)pad_cols + (unsigned)st_col) * 8u;  // bytes inside a level
    f2 dprev = {0.0f, 0.0f};  // previous level's Euler displacement in index space: predicts this level's travel
    constexpr int NPASS = LT_ROWS / G::ROWS_PER_PASS;
    // With the Euler sample gathered directly (order 3 without its Euler tile) the tile loads stay behind it:
    // 16 more live VGPRs across the 8 gathers cost two waves per SIMD (measured: 21.7 vs 21.0-21.2 ms)
    // (order 3 with its Euler tile: prefetching costs 14 VGPRs = one wave per SIMD, 18.1 vs 17.84 ms without)
    constexpr bool PREFETCH = ORDER == 1;
    constexpr int CENTRE = TILE_W / 2 + TILE_W * ((64 / TILE_W) / 2);  // middle seed of the wave's patch
    // VERIFY: where this wave runs -- HW_REG_HW_ID (wave slot, SIMD, CU, SE, queue, VMID) and HW_REG_XCC_ID
    unsigned hw_id = 0, xcc_id = 0;
    if constexpr (VERIFY) {
        hw_id = __builtin_amdgcn_s_getreg((31 << 11) | 4);
        xcc_id = __builtin_amdgcn_s_getreg((31 << 11) | 20);
    }
    for (int s = 0; s < A.nsteps; ++s) {
        f2 c0 = to_index(p);
        // ---- 1. anchor the tile and issue its loads ------------------------------------------------
        // Tile origin (padded coordinates): centre of the windows' predicted travel.  The Euler step and
        // every iteration move a parcel by about one Euler displacement (Q4), and that displacement
        // changes little from one 15-minute level to the next, so the previous level's stands in for this
        // one's -- which lets the tile loads fly together with the Euler gather instead of after it (one
        // exposed global round trip per level, not two: 8.1 -> 7.65 ms on C3).  Heuristic only: results do
        // not depend on where the tile sits.  Two v_readlane, no reduction.
        int ox = 0, oy = 0;
        f4 stage[NPASS];
        auto anchor_and_load = [&](f2 ca) {
            const int rxm = __builtin_amdgcn_readlane((int)floor_to_uint(ca.x), CENTRE);
            const int rym = __builtin_amdgcn_readlane((int)floor_to_uint(ca.y), CENTRE);
            ox = min(max(rxm + WOFF - (LT_COLS - WIN) / 2, 0), pad_cols - LT_COLS);
            oy = min(max(rym + WOFF - (LT_ROWS - WIN) / 2, 0), pad_rows - LT_ROWS);
            // uniform base (SGPR pair) + this lane's constant 32-bit offset: no per-level address arithmetic
            const char *src = (const char *)elv + ((size_t)__umul24((unsigned)oy, (unsigned)pad_cols) + (unsigned)ox) * 8;
#pragma unroll
            for (int r = 0; r < NPASS; ++r)
                __builtin_memcpy(&stage[r], src + (size_t)(r * G::ROWS_PER_PASS) * pad_cols * 8 + st_off, 16);
        };
        if (PREFETCH && K > 0) anchor_and_load(dprev * (1.0f + kpred) + c0);
        // ---- 2. Euler sample: global gather (order 1) / LDS tile of img[t] (order 3) ------------------
        f2 e;
        {
            TapL t = tap_of(c0);
            // common case: 0 <= floor(c) <= n-2, i.e. c in [0, n-1) -- no wrap needed
            bool bad = ((unsigned)t.x0 > (unsigned)(A.nx_f - 2)) | ((unsigned)t.y0 > (unsigned)(A.ny_f - 2));
            const f2 zero = {0.0f, 0.0f};
            if (E::ON) {
                // tile origin: the centre lane's window in the middle of the tile
                const int exm = __builtin_amdgcn_readlane(t.x0, CENTRE), eym = __builtin_amdgcn_readlane(t.y0, CENTRE);
                const int eox = min(max(exm + WOFF - (E::COLS - WIN) / 2, 0), pad_cols - E::COLS);
                const int eoy = min(max(eym + WOFF - (E::ROWS - WIN) / 2, 0), pad_rows - E::ROWS);
                const char *src = (const char *)lvl + ((size_t)__umul24((unsigned)eoy, (unsigned)pad_cols) + (unsigned)eox) * 8;
                const unsigned e_off = ((unsigned)e_row * (unsigned)pad_cols + (unsigned)e_col) * 8u;
                f4 es[E::NPASS > 0 ? E::NPASS : 1];
#pragma unroll
                for (int r = 0; r < E::NPASS; ++r)
                    __builtin_memcpy(&es[r], src + (size_t)(r * E::ROWS_PER_PASS) * pad_cols * 8 + e_off, 16);
                __builtin_amdgcn_wave_barrier();  // the previous level's reads of this region are done
#pragma unroll
                for (int r = 0; r < E::NPASS; ++r) *(f4 *)(etile + (r * E::ROWS_PER_PASS + e_row) * E::PITCH + e_col) = es[r];
                __builtin_amdgcn_wave_barrier();
                // window origins the tile serves: inside it AND in [0, n-2] (the same rule as for the ext tile)
                const int sox = eox - WOFF, soy = eoy - WOFF;
                const int hx = min(sox + E::COLS - WIN, A.nx_f - 2), hy = min(soy + E::ROWS - WIN, A.ny_f - 2);
                const int lx = max(sox, 0), ly = max(soy, 0);
                const int rx = t.x0 - lx, ry = t.y0 - ly;
                bad |= ((unsigned)rx > (unsigned)(hx - lx)) | ((unsigned)ry > (unsigned)(hy - ly)) | (hx < lx) | (hy < ly);
                const unsigned ebase = etile_addr + (unsigned)(lx - sox) * 8u + (unsigned)(ly - soy) * ((unsigned)E::PITCH * 8u);
                e = window_lds<ORDER, E::PITCH, VERIFY>(ebase, epitch_bytes, rx, ry, t, zero);
            } else {
                e = euler_global<ORDER>(lvl, A, t);
            }
            f2 pn = dd * e + p;
            if (!DEFER_X) bad |= x_needs_care(pn.x);
            if (bad) {  // exact sequence
                c0 = index_coords(A, p);
                t = tap_of(c0);
                e = euler_global<ORDER>(lvl, A, t);
                pn = dd * e + p;
                if (!DEFER_X) clamp_position_p(A, pn, ymax_v);
            }
            dprev = (pn - p) * sc;  // Euler displacement in index space
            p = pn;
        }
        if (!PREFETCH && K > 0) anchor_and_load(dprev * (1.0f + kpred) + c0);  // this level's own displacement
        // ---- 3. tile into LDS: ext[t][oy .. oy+LT_ROWS) x [ox .. ox+LT_COLS) ----------------------
        int lo_x = 0x40000000, lo_y = 0x40000000, lim_x = 0, lim_y = 0;  // no tile: nothing is "inside"
        unsigned base_addr = tile_addr;
        if (K > 0) {
            __builtin_amdgcn_wave_barrier();  // the previous level's reads are done (LDS ops of a wave are in order)
#pragma unroll
            for (int r = 0; r < NPASS; ++r) *(f4 *)(tile + (r * G::ROWS_PER_PASS + st_row) * LT_PITCH + st_col) = stage[r];
            __builtin_amdgcn_wave_barrier();
            if constexpr (VERIFY) {  // test hook: one entry of one tile overwritten behind the wave's back
                if (A.verify && A.verify[15] == 0xBADu && tile_id == 5 && wave == 1 && s == 1 && lane == 5)
                    *(volatile f2 *)(tile + ((LT_ROWS - WIN) / 2) * LT_PITCH + (LT_COLS - WIN) / 2) = (f2){1.0e3f, -1.0e3f};  // the centre lane's window origin
                __builtin_amdgcn_wave_barrier();
            }
            // window origins floor(c) the common case accepts: inside the tile AND in [0, n-2] (no wrap)
            const int sox = ox - WOFF, soy = oy - WOFF;
            const int hx = min(sox + LT_COLS - WIN, A.nx_f - 2), hy = min(soy + LT_ROWS - WIN, A.ny_f - 2);
            const int lx = max(sox, DEFER_X ? 1 : 0), ly = max(soy, DEFER_X ? 1 : 0);  // (DEFER_X: origin 0 to the exact path, as in the two-seed kernel)
            if (hx >= lx && hy >= ly) {
                lo_x = lx;
                lo_y = ly;
                lim_x = hx - lx;
                lim_y = hy - ly;
                base_addr = tile_addr + (unsigned)(lx - sox) * 8u + (unsigned)(ly - soy) * ((unsigned)LT_PITCH * 8u);
            }
        }
        // ---- 4. K iterations out of LDS ----------------------------------------------------------
        // The latitude clamp (trajectory.py:89-90) is DEFERRED: a latitude outside [y_min, y_max] maps to an index
        // outside [0, n-1), which the next sample's window test flags, and every exact-redo path starts by
        // clamping the position it was handed -- the same values as clamping after each update, one v_med3
        // per level instead of one per sample.
#pragma unroll
        for (int k = 0; k < K; ++k) {
            // absolute index coordinate: its rounding must not depend on where the tile sits (row-sharded runs
            // place tiles differently and must stay bit-identical to unsharded ones)
            TapL t = tap_of(to_index(p));
            const int rx = t.x0 - lo_x, ry = t.y0 - lo_y;  // the subtrahends are wave-uniform (SGPRs)
            bool bad = ((unsigned)rx > (unsigned)lim_x) | ((unsigned)ry > (unsigned)lim_y);
            const f2 ew = window_lds<ORDER, LT_PITCH, VERIFY>(base_addr, pitch_bytes, rx, ry, t, e);  // e + sample of ext[t]
            f2 pn = hd * ew + p;
            if (!DEFER_X) bad |= x_needs_care(pn.x);
            if (bad) {  // exact sequence, global gather
                f2 pc = p;
                if (DEFER_X)
                    clamp_position_c<CYCLIC>(A, pc, ymax_v);  // the deferred clamps of the previous update (Q7 / Q8 / Q9)
                else
                    pc.y = __builtin_amdgcn_fmed3f(pc.y, A.y_min, ymax_v);  // the deferred clamp (Q8)
                t = tap_of(index_coords(A, pc));
                pn = hd * window_global<ORDER>(elv, A, t, e) + pc;
                if (!DEFER_X) clamp_position_p(A, pn, ymax_v);
            }
            p = pn;
        }
        if (DEFER_X)
            clamp_position_c<CYCLIC>(A, p, ymax_v);  // the level's one clamp of either kind (stores, next Euler sample)
        else
            p.y = __builtin_amdgcn_fmed3f(p.y, A.y_min, ymax_v);  // the level's one latitude clamp (stores, next Euler sample)
        if constexpr (VERIFY) {
            if (A.verify && K > 0) {
                // the audit: every lane reads back the 16 bytes it staged for this level, after the last window read
                bool changed = false;
#pragma unroll
                for (int r = 0; r < NPASS; ++r) {
                    typedef unsigned u4 __attribute__((ext_vector_type(4)));
                    const u4 back = *(volatile u4 *)(tile + (r * G::ROWS_PER_PASS + st_row) * LT_PITCH + st_col);
                    u4 was;
                    __builtin_memcpy(&was, &stage[r], 16);
                    changed |= (back.x != was.x) | (back.y != was.y) | (back.z != was.z) | (back.w != was.w);
                }
                const unsigned long long m = __ballot(changed);
                const unsigned hw = __builtin_amdgcn_s_getreg((31 << 11) | 4), xc = __builtin_amdgcn_s_getreg((31 << 11) | 20);
                const bool moved = hw != hw_id || xc != xcc_id;
                if (lane == 0) {
                    atomicAdd(&A.verify[3], 1u);
                    if (moved) atomicAdd(&A.verify[2], 1u);
                    if (m) {
                        atomicAdd(&A.verify[0], 1u);
                        atomicAdd(&A.verify[1], (unsigned)__popcll(m));
                    }
                    if ((m || moved) && atomicCAS(&A.verify[12], 0u, 1u) == 0u) {
                        A.verify[4] = blockIdx.x;
                        A.verify[5] = (unsigned)tile_id;
                        A.verify[6] = (unsigned)wave;
                        A.verify[7] = (unsigned)(A.t0 + s);
                        A.verify[8] = hw_id;
                        A.verify[9] = hw;
                        A.verify[10] = (unsigned)m;
                        A.verify[11] = (unsigned)(m >> 32);
                    }
                }
                hw_id = hw;
                xcc_id = xc;
            }
        }
        if (LINES && lines) {
            float *sx = s_slab1[s & 1][0], *sy = s_slab1[s & 1][1];
            const int o = (lane / TILE_W) * SLAB1_PITCH + wave * TILE_W + (lane % TILE_W);
            sx[o] = p.x;
            sy[o] = p.y;
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // one barrier per level: two slabs alternate
            const f4 line = *(const f4 *)(s_slab1[s & 1][wave & 1] + (lane >> 3) * SLAB1_PITCH + (lane & 7) * 4);
            if (sl_ok) LCS_TRAJ_STORE((wave ? A.traj_y : A.traj_x) + (size_t)(s + 1) * plane + sl_idx, line);
        } else if (live && A.traj_x) {
            A.traj_x[(size_t)(s + 1) * plane + idx] = p.x;
            A.traj_y[(size_t)(s + 1) * plane + idx] = p.y;
        }
        lvl += A.level_elems;
        elv += A.level_elems;
    }
    if (live) {
        A.x_out[idx] = p.x;
        A.y_out[idx] = p.y;
    }
}


// ======================================================================================
// Order 1, TWO seeds per lane (the default for float32 + fused levels at order 1).
//
// The one-seed kernel above is neither VALU- nor scalar-throughput bound: adding 14 % more (independent) vector
// instructions to its loop costs 5 % of time, 37 % more scalar instructions 3 % (tools/ab.sh, LCS_EXP_*): what
// bounds it is the dependent chain position -> index -> LDS window -> lerps -> position of each sample, with at
// most 8 waves per SIMD to cover it.  A second, independent seed per lane doubles the work in flight per
// wave at the same occupancy (44 -> ~70 VGPRs), and the per-wave bookkeeping (tile anchor, staging, loop,
// level pointers, one rare branch per sample) is shared by 128 seeds instead of 64.
// A wave's patch is 8 x 16 seeds (lane = column + 8 * row, seeds (row, col) and (row + 8, col)); on C3 that is
// 2.8 x 2.8 field cells, which the same 16 x 8-node tile covers.  Arithmetic per seed is exactly the one-seed
// kernel's (results are bit-identical to it and to the direct-gather kernel).
// ======================================================================================
constexpr int SPL = 2;  // seeds per lane

// Tile geometry of the two-seed kernel (nodes).  Measured (advect ms on C3 / on C5's 64 members): 16x8 7.15 / 462,
// 16x16 7.44 / 477, 32x8 7.85 / 495, 32x16 - / 479 (round 3, member pairs on C5: 16x8 6.31 / 277.5, 8x16 6.63 / 297).  Where the windows leave the tile (-DLCS_STAMPS counters, C3):
// 2.3 % of the seed-samples in x, 4.6 % in y, evenly above and below -- patches the flow has deformed; a 16-row
// tile brings y down to 1.5 % and the wave-samples with a redo from 19 % to 11 %, but its second staging pass
// costs what the redos saved (7.19 against 7.04 ms).  C5 (seeds half as dense, 200 steps) is slower per particle
// whatever the tile: its patches are pulled apart by the flow over the longer integration, not merely too wide.
#ifndef LCS_LDS2_ROWS
#define LCS_LDS2_ROWS 8
#endif
#ifndef LCS_LDS2_COLS
#define LCS_LDS2_COLS 16
#endif
#ifndef LCS_LDS2_PITCH
#define LCS_LDS2_PITCH (LCS_LDS2_COLS + 4)
#endif
struct Lds2Geom {  // staging: a tile row = COLS/2 lanes x 16 bytes
    static constexpr int COLS = LCS_LDS2_COLS, LANES_PER_ROW = COLS / 2, ROWS_PER_PASS = 64 / LANES_PER_ROW;
    // one pass of 64 lanes covers the tile, or the tile is a whole number of full passes (a row count that 64 lanes do not
    // divide -- 12 x 10 nodes: 6 lanes per row, 10 rows, lanes 60..63 over -- has the spare lanes repeat the last row)
    static_assert(LCS_LDS2_ROWS % ROWS_PER_PASS == 0 || LCS_LDS2_ROWS < ROWS_PER_PASS + 1, "tile rows per staging pass");
};

#ifdef LCS_STAMPS  // diagnostic build only: where a wave's cycles go (s_memtime), summed over waves and levels
__device__ unsigned long long g_stamps[8];
__device__ unsigned long long g_cause[4];  // seed-samples outside the tile in x, in y, of those below (x < lo), (y < lo)
__device__ unsigned long long g_redo[3][3][3];  // [latitude band 0-30 / 30-60 / 60-90][third of the levels][samples, with redo, seeds redone]
__device__ unsigned long long g_hist[2][5];     // wave-levels by the number of their K = 4 iterations with a redo: [0] all, [1] those whose PREVIOUS level had >= 3
#define LCS_STAMP(i)                                           \
    {                                                          \
        const long long _t = __builtin_amdgcn_s_memtime();     \
        acc_t[i] += _t - last_t;                               \
        last_t = _t;                                           \
    }
#else
#define LCS_STAMP(i)
#endif

// 98 SGPRs as compiled admit 6 workgroups per CU (MI355X_MICROARCH.md: 97-112 -> 6); capped at 96 -> 7
// (7.55 -> 7.31 ms on C3; 80 -> 8 workgroups measures the same as 96).
#ifndef LCS_LDS2_NUM_SGPR
#define LCS_LDS2_NUM_SGPR 96
#endif
// Which seeds a wave holds, and how return_traj's positions reach memory (template parameter MODE).  Per-seed arithmetic
// does not depend on it: results are bit-identical in every mode.
//   PATCH_TALL   lane = column + 8 * row, the lane's second seed 8 rows further down: 8 x 16 seeds per wave, the four waves
//                of a workgroup stacked in latitude (8 x 64).  The default without trajectories (fewest distinct lines per
//                Euler gather: 6.47 ms on C3 against 6.85 for PATCH_WIDE).  With trajectories every store instruction
//                writes 8 rows x 32 bytes: quarter-line partial writes, for each of which the L2 FETCHES the rest of the
//                128-byte line (profiles/r03: +9.7 GB of fetches = 3/4 of the 12.9 GB stored, TCC_MISS x 7, advect
//                6.5 -> 11.8 ms).
//   PATCH_WIDE   the lane's second seed one column on: 16 x 8 seeds per wave, one 8-byte store per lane = 8 rows x 64
//                bytes per instruction.  11.8 -> 11.2 ms; half lines still make the L2 fetch the other half.  Kept for A/B.
//   PATCH_LINES  tall patches, the four waves SIDE BY SIDE in longitude (32 x 16 seeds per workgroup); after each time level
//                the waves put their positions into an LDS slab, meet at ONE workgroup barrier (two slabs alternate, so one
//                barrier per level is enough), and every wave writes a plane of 8 rows x 32 columns with one
//                global_store_dwordx4 per lane: whole 128-byte lines.  The default with trajectories when nx % 4 == 0.
//   PATCH_PAIR   lc_advect_batch only: the lane's seeds are the SAME grid point in two consecutive ensemble members whose
//                start levels lie d apart (AdvectArgs::pair_*).  A wave holds 8 x 8 grid points (the one-seed kernel's
//                patch: a sparse seed grid -- config 5, 0.7 field cells per seed -- fits the 16 x 8-node tile where the 16
//                rows of a tall patch do not) and walks LEVELS: at level l member q takes its step l - q d, all from the
//                same tile of ext[l] and neighbouring lines of img[l]; at the ends of the group's window only some members
//                move (the others' step is computed and dropped: bits of the kept positions do not change).
// (enum Patch: declared above the one-seed kernel, which shares PATCH_LINES' store form)
constexpr int SLAB_PITCH = 36;  // floats per slab row (32 + 4: rows stay 16-byte aligned)

// (65 VGPRs = 7 waves per SIMD.  Asked for 8 -- -DLCS_LDS2_MINWAVES=8 -DLCS_LDS2_NUM_SGPR=80: 57 VGPRs, 78 SGPRs, no
// spills -- C3 measures 6.58-6.73 ms against 6.48-6.52 and config 5 302 against 299 ms: occupancy is not what either lacks.)
#ifndef LCS_LDS2_MINWAVES
#define LCS_LDS2_MINWAVES 1
#endif
// DIRECT LEVELS (round 6).  Which waves leave their tiles is not spread thin: on C3, 77 % of the wave-levels have no redo in
// any of their four iterations and 17 % have one in ALL four (2.5 / 1.7 / 1.3 % in one / two / three), and a wave that had
// three or more is at four again in the next level 92 times in 100 (-DLCS_STAMPS, g_hist: patches the flow has stretched
// beyond the tile, polar rows whose zonal travel is longer than the tile -- they stay that way).  Such a wave pays for the
// tile (staging, 2 x 17 instructions of window arithmetic per iteration) and then for the exact sequence on top (2 x 27) in
// every iteration.  So a wave whose level had >= 3 iterations with a redo takes the next LCS_LDS2_DIRECT_LEVELS levels
// DIRECT: no tile, every lane through the exact sequence (the redo block, unmasked); then one level with a tile again, which
// decides anew.  Same functions, same values (a seed's bits do not depend on which path served it).  0 = never.
// MEASURED AND NOT ADOPTED (profiles/r06/direct_levels_ab.txt): C3 advect 5.99-6.03 ms without, 6.13-6.15 / 6.15-6.20 /
// 6.23-6.25 ms with 3 / 7 / 15 direct levels -- the instructions saved (~12 % of the VALU stream) are vector-L1 lookups
// spent: a direct level gathers 2 x 16 bytes per sample for ALL 64 lanes of the wave where the redo path gathers for the
// ~29 % that left the tile, and the vector L1 already answers 0.73 lookups per CU-cycle.  (A launch made of polar rows alone
// does gain: rank 7 of 8 of C4, 14.2 -> 13.0 ms.)  Kept as a compile-time option, off.
#ifndef LCS_LDS2_DPP_NEXT
#define LCS_LDS2_DPP_NEXT 1   // node c+2 of the staging from the neighbouring lane (two DPP moves) instead of an 8-byte load: C3 5.98 -> 5.91 ms
                             // (one vector-memory instruction of eight per wave-level less; profiles/r06/headline_kernel_ab.txt section 9).  0: the load
#endif
#ifndef LCS_LDS2_DIRECT_LEVELS
#define LCS_LDS2_DIRECT_LEVELS 0
#endif
template <int KFIX, bool CYCLIC, int MODE>
__global__ void __launch_bounds__(BLOCK, LCS_LDS2_MINWAVES) __attribute__((amdgpu_num_sgpr(LCS_LDS2_NUM_SGPR)))
    advect_lds2_kernel(const AdvectArgs<float> A0) {
#pragma clang fp contract(fast)
    const AdvectArgs<float> A = for_member(A0);
    constexpr bool WIDE = MODE == PATCH_WIDE, LINES = MODE == PATCH_LINES, GROUP = MODE == PATCH_PAIR;
    constexpr int NS = SPL;  // seeds per lane (member groups: members per lane)
    constexpr int ORDER = 1;
    constexpr bool DEFER_X = LCS_LDS2_DEFER_X != 0;
    constexpr int DIRECT_LEVELS = DEFER_X ? LCS_LDS2_DIRECT_LEVELS : 0;
    // (a DPP row is 16 lanes: the neighbouring lane holds the next nodes of the same tile row only when tile rows do not straddle DPP rows)
    constexpr bool DPP_NEXT = LCS_LDS2_DPP_NEXT != 0 && 16 % Lds2Geom::LANES_PER_ROW == 0;
    const int K = KFIX >= 0 ? KFIX : A.K;
    typedef Lds2Geom G;
    constexpr int LT_COLS = G::COLS, LT_ROWS = LCS_LDS2_ROWS;
    // LDS tile of 16-byte entries {u, v, u[x+1] - u, v[x+1] - v}: the x-differences of the two lerps are formed
    // ONCE per node when the tile is staged (2 packed subtractions + one more 8-byte load per lane and level) instead of
    // once per sample (2 per sample and seed), and a window is two 16-byte reads instead of four 8-byte ones.
    // Same values as n00 + tx * (n01 - n00): results stay bit-identical to the other float kernels.
    // Pitch 20 entries: rows shift 16 banks, no conflicts between neighbouring rows for ds_read_b128.
    constexpr int LT_PITCH = LCS_LDS2_PITCH;
    constexpr int WIN = 2, WOFF = LC_PAD_LO;
    __shared__ __attribute__((aligned(16))) f4 s_tiles[BLOCK / 64][LT_ROWS * LT_PITCH];
    // PATCH_LINES: two slabs (alternating by level) of the workgroup's 16 x 32 longitudes and latitudes
    __shared__ __attribute__((aligned(16))) float s_slab[2][2][LINES ? 16 * SLAB_PITCH : 4];
    if (GROUP ? pole_block_group(A) : pole_block<float, POLE_LIN>(A)) return;
    const int tile_id = xcd_tile_id(A);
    if (tile_id >= A.ntiles) return;  // whole block
    const int tyi = tile_id / A.ntx, txi = tile_id - tyi * A.ntx;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // workgroup origin and the lane's first seed (see enum Patch)
    const int ix0 = WIDE ? txi * (TILE_W * SPL) + SPL * (lane % TILE_W)
                  : LINES ? txi * (TILE_W * 4) + wave * TILE_W + (lane % TILE_W) : txi * TILE_W + (lane % TILE_W);
    const int iy0 = (WIDE || GROUP) ? tyi * TILE_H + wave * 8 + lane / TILE_W
                  : LINES ? tyi * (8 * SPL) + lane / TILE_W : tyi * (TILE_H * SPL) + wave * (8 * SPL) + lane / TILE_W;
    f4 *tile = s_tiles[wave];

    bool live[NS];
    f2 p[NS], dd[NS], hd[NS];
    size_t idx[NS];
    const size_t plane = (size_t)A.ny * A.nx;
    bool any = false;
    const int cnt = GROUP ? group_count(A) : NS;  // member groups: members of this workgroup's group (the last one may be short)
#pragma unroll
    for (int q = 0; q < NS; ++q) {
        const int ix = ix0 + (WIDE ? q : 0), iy = iy0 + ((WIDE || GROUP) ? 0 : 8 * q);
        live[q] = ix < A.nx && iy < A.ny && q < cnt;
        if (live[q]) {
            const int grow = A.row0 + iy;
            if (grow < A.order || grow >= A.ny_global - A.order) {  // pole rows: generic path, whole integration (Q3)
                if (!A.pole_blocks) {  // (else the leading workgroups did them)
                    if (GROUP)
                        advect_seed<float, 1, false>(group_member(A, q), A.lin, iy, ix);
                    else
                        advect_seed<float, 1, false>(A, A.lin, iy, ix);
                }
                live[q] = false;
            }
        }
        any |= live[q];
        // lanes without a seed shadow a neighbouring one so that they follow the same path; only their stores are masked
        const int sx_i = min(ix, A.nx - 1), sy_i = min(iy, A.ny - 1);
        const size_t moff = (GROUP && q < cnt) ? (size_t)q * A.pair_plane : 0;  // member q's planes
        p[q] = (f2){A.x_start ? A.x_start[moff + (size_t)sy_i * A.nx + sx_i] : A.seed_lon[sx_i],
                    A.y_start ? A.y_start[moff + (size_t)sy_i * A.nx + sx_i] : A.seed_lat[sy_i]};
        const float ys = A.seed_lat[sy_i];  // conversion_x is a function of the SEED latitude (Q5), wherever the parcel is now
        const float cx_conv =
            180.0f / ((float)(3.141592653589793 * 6371000.0) * fabsf(cosf((ys * (float)3.141592653589793) / 180.0f)));
        dd[q] = (f2){A.dt * cx_conv, A.dtcy};        // trajectory.py:55-57,86-87
        hd[q] = (f2){A.half_dt * cx_conv, A.hdtcy};  // trajectory.py:110-112
        idx[q] = live[q] ? moff + (size_t)iy * A.nx + ix : 0;
    }
    // WIDE: the lane's two positions are neighbours in memory -- one 8-byte store when both are live and 8-byte aligned
    // (nx even and an aligned base: the launcher checks, A.traj_pair_ok)
    const bool pair = WIDE && live[0] && live[1] && A.traj_pair_ok;
    auto store_pair = [&](float *dx, float *dy, size_t off) {
        if (pair) {
            *(f2 *)(dx + off + idx[0]) = (f2){p[0].x, p[1].x};
            *(f2 *)(dy + off + idx[0]) = (f2){p[0].y, p[1].y};
        } else {
#pragma unroll
            for (int q = 0; q < NS; ++q)
                if (live[q]) {
                    dx[off + idx[q]] = p[q].x;
                    dy[off + idx[q]] = p[q].y;
                }
        }
    };
    if (A.traj_x && !A.traj_skip0) store_pair(A.traj_x, A.traj_y, 0);
    // PATCH_LINES: whole-line stores for workgroups whose 32 columns are all inside the grid (the others, and grids whose
    // rows are not 16-byte aligned, store per lane as above).  Workgroup-uniform: either all four waves meet at the
    // level's barrier or none does -- so a wave without seeds stays in the loop (it shadows its neighbours' seeds).
    const bool lines = LINES && A.traj_x && A.traj_line_ok && (txi + 1) * (TILE_W * 4) <= A.nx;
    if (!lines && __ballot(any) == 0ull) return;  // whole wave (no workgroup barrier below unless `lines`)
    // the 8 rows x 32 columns this wave writes per level: plane (wave >> 1: longitudes, latitudes), rows (wave & 1) * 8 ...
    const int sl_row = (wave & 1) * 8 + (lane >> 3), sl_col = (lane & 7) * 4;
    bool sl_ok = false;
    size_t sl_idx = 0;
    if (lines) {
        const int iyr = tyi * (8 * SPL) + sl_row, grow = A.row0 + iyr;
        sl_ok = iyr < A.ny && (A.pole_blocks == 0 || (grow >= A.order && grow < A.ny_global - A.order));  // pole rows: written by their own workgroups
        if (!A.pole_blocks && iyr < A.ny && (grow < A.order || grow >= A.ny_global - A.order)) sl_ok = false;  // ... or by advect_seed above
        sl_idx = (size_t)min(iyr, A.ny - 1) * A.nx + (size_t)txi * (TILE_W * 4) + sl_col;
    }
    float ymax_v = A.y_max;
    asm volatile("" : "+v"(ymax_v));
    const unsigned tile_addr = lds_address(tile);
    unsigned pitch_bytes = (unsigned)LT_PITCH * 16u;
    asm volatile("" : "+s"(pitch_bytes));
    const f2 pmin = {A.lon_min, A.lat_min}, sc = {A.sx, A.sy};
    auto to_index = [&](f2 v) { return (v - pmin) * sc; };  // subtract first: exact 0 at the grid origin
    const float xlo = A.x_min, xhi = A.x_max;
    auto x_needs_care = [&](float x) { return CYCLIC ? !(fabsf(x) < 180.0f) : !((x > xlo) & (x < xhi)); };
    const float *lvl = A.img + (size_t)A.t0 * A.level_elems;
    const float *elv = A.ext + (size_t)A.t0 * A.level_elems;
    const int pad_cols = A.pitch, pad_rows = A.ny_f + LC_PAD;
    const float kpred = 0.5f * (float)(K > 0 ? K - 1 : 0);
    const int st_row = min(lane / G::LANES_PER_ROW, LT_ROWS - 1), st_col = (lane % G::LANES_PER_ROW) * 2;
    const unsigned st_off = ((unsigned)st_row * (unsigned)pad_cols + (unsigned)st_col) * 8u;
    constexpr int NPASS = (LT_ROWS + G::ROWS_PER_PASS - 1) / G::ROWS_PER_PASS;
    // node c+2 of the lane's tile row (the x-difference of node c+1 needs it): loaded with the tile, 8 more bytes
    // per lane; the last lane of a row re-reads its own node instead (its entry c+1 is never a window origin)
    const unsigned st_next = st_off + (st_col + 2 < LT_COLS ? 16u : 0u);
    // seed 0 of the lane in the patch's middle: row 8 of 16 (tall patches), row 4 of 8 and column 8 of 16 (PATCH_WIDE)
    constexpr int CENTRE = (WIDE || GROUP) ? TILE_W / 2 + TILE_W * 4 : TILE_W / 2 + TILE_W * 7;
    f2 dprev = {0.0f, 0.0f};
    // member groups: level iterations of this workgroup (a short last group stops with its last member's steps)
    const int nlev = GROUP ? lcplan::group_levels(A.nsteps, A.pair_l0, A.pair_n, A.pair_d, cnt) : A.nsteps;
#ifdef LCS_STAMPS
    long long acc_t[5] = {0, 0, 0, 0, 0}, last_t = __builtin_amdgcn_s_memtime();
    unsigned long long acc_n[3] = {0, 0, 0};
    int prev_redos = 0;
#endif
    int direct_left = 0;  // wave-uniform: levels this wave still takes without a tile
    for (int s = 0; s < nlev; ++s) {
        const bool direct = DIRECT_LEVELS > 0 && direct_left > 0;
        if (direct) --direct_left;
#ifdef LCS_STAMPS
        int level_redos = 0;
#endif
        f2 c0[NS];
#pragma unroll
        for (int q = 0; q < NS; ++q) c0[q] = to_index(p[q]);
        // member groups: which members step at this level (wave-uniform: member q's own steps are levels [q d, q d + n) of
        // the group's window); the others' steps are dropped below.  The tile follows the middle one of those that move.
        bool act[NS];
        f2 keep[NS];
        int qa = 0;
        if (GROUP) {
            int qlo = NS, qhi = 0;
#pragma unroll
            for (int q = 0; q < NS; ++q) {
                act[q] = lcplan::member_steps(q, s, A.pair_l0, A.pair_n, A.pair_d) && q < cnt;
                keep[q] = p[q];
                if (act[q]) {
                    qlo = min(qlo, q);
                    qhi = q;
                }
            }
            qa = min(max(NS / 2, qlo), qhi);
        }
        auto of_anchor = [&](const f2 (&v)[NS]) {  // v[qa], qa wave-uniform
            f2 r = v[0];
#pragma unroll
            for (int q = 1; q < NS; ++q) r = (GROUP && qa == q) ? v[q] : r;
            return r;
        };
        // ---- 1. anchor the tile on the centre lane's predicted travel and issue its loads ------------
        int ox = 0, oy = 0;
        f4 stage[NPASS];
        f2 stage_next[NPASS];
        if (K > 0 && !direct) {
            // (anchoring a pair's tile half way between its two members instead: 276.4 against 276.4 ms on config 5)
            const f2 ca = dprev * (1.0f + kpred) + of_anchor(c0);
            const int rxm = __builtin_amdgcn_readlane((int)floor_to_uint(ca.x), CENTRE);
            // (the patch's middle lies 0.09 cells above this lane on C3: no shift.  A tile anchored one row higher
            // measured 23 % instead of 19 % of wave-samples with a redo, 7.19 against 7.04 ms)
            const int rym = __builtin_amdgcn_readlane((int)floor_to_uint(ca.y), CENTRE);
            ox = min(max(rxm + WOFF - (LT_COLS - WIN) / 2, 0), pad_cols - LT_COLS);
            oy = min(max(rym + WOFF - (LT_ROWS - WIN) / 2, 0), pad_rows - LT_ROWS);
            const char *src = (const char *)elv + ((size_t)__umul24((unsigned)oy, (unsigned)pad_cols) + (unsigned)ox) * 8;
#pragma unroll
            for (int r = 0; r < NPASS; ++r) {
                __builtin_memcpy(&stage[r], src + (size_t)(r * G::ROWS_PER_PASS) * pad_cols * 8 + st_off, 16);
                if (!DPP_NEXT) __builtin_memcpy(&stage_next[r], src + (size_t)(r * G::ROWS_PER_PASS) * pad_cols * 8 + st_next, 8);
            }
        }
        LCS_STAMP(0)  // anchor + tile load issue
        // ---- 2. Euler samples (global gathers) ----------------------------------------------------
        f2 e[NS], pn[NS];
        bool bad[NS], anybad = false;
#pragma unroll
        for (int q = 0; q < NS; ++q) {
            const TapL t = tap_of(c0[q]);
            bad[q] = ((unsigned)t.x0 > (unsigned)(A.nx_f - 2)) | ((unsigned)t.y0 > (unsigned)(A.ny_f - 2));
            e[q] = euler_global<ORDER>(lvl, A, t);
            pn[q] = dd[q] * e[q] + p[q];
            if (!DEFER_X) bad[q] |= x_needs_care(pn[q].x);
            anybad |= bad[q];
        }
        if (anybad) {  // exact sequence for the lanes / seeds that need it
#pragma unroll
            for (int q = 0; q < NS; ++q) {
                if (bad[q]) {
                    const TapL t = tap_of(index_coords(A, p[q]));
                    e[q] = euler_global<ORDER>(lvl, A, t);
                    pn[q] = dd[q] * e[q] + p[q];
                    if (!DEFER_X) clamp_position_p(A, pn[q], ymax_v);
                }
            }
        }
        dprev = (of_anchor(pn) - of_anchor(p)) * sc;  // Euler displacement in index space (predicts the next level's travel)
#pragma unroll
        for (int q = 0; q < NS; ++q) p[q] = pn[q];
#ifdef LCS_STAMPS
        asm volatile("" : : "v"(p[0].x), "v"(p[1].x));
#endif
        LCS_STAMP(1)  // Euler sample
        // ---- 3. tile into LDS ------------------------------------------------------------------------
        int lo_x = 0x40000000, lo_y = 0x40000000, lim_x = 0, lim_y = 0;  // no tile: nothing is "inside"
        unsigned base_addr = tile_addr;
        if (K > 0 && !direct) {
            __builtin_amdgcn_wave_barrier();  // the previous level's reads are done (LDS ops of a wave are in order)
#pragma unroll
            for (int r = 0; r < NPASS; ++r) {
                // this lane holds nodes (c, c+1) of a tile row; node c+2 is the next lane's first node.  Two cross-lane moves of the
                // .x and .y of one register pair written with the builtins (update_dpp or ds_bpermute alike) came out of hipcc 7.2 as
                // ONE move feeding both halves (seen in the ISA and in wrong results: rounds 2-5 loaded the node instead, 8 more
                // bytes per lane) -- so the two moves are spelled out: row_shl:1 within the 16 lanes of a DPP row; the last lane of a
                // TILE row receives the next tile row's first node or zero, and its entry c+1 is never a window origin.  (s_nop 4:
                // the five wait states a DPP instruction needs after a VALU write of EXEC, two after one of its source, whatever
                // precedes the statement: tools/asm_hazards.py audits the sites.)
                f2 nxt;
                if constexpr (DPP_NEXT) {
                    float nx0, nx1;
                    const float s0 = stage[r].x, s1 = stage[r].y;
                    asm volatile("s_nop 4\n\tv_mov_b32_dpp %0, %2 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0\n\t"
                                 "v_mov_b32_dpp %1, %3 row_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0"
                                 : "=&v"(nx0), "=&v"(nx1) : "v"(s0), "v"(s1));
                    nxt = (f2){nx0, nx1};
                } else {
                    nxt = stage_next[r];
                }
                f4 *dst = tile + (r * G::ROWS_PER_PASS + st_row) * LT_PITCH + st_col;
                const f2 n0 = stage[r].xy, n1 = stage[r].zw, d0 = n1 - n0, d1 = nxt - n1;
                dst[0] = (f4){n0.x, n0.y, d0.x, d0.y};
                dst[1] = (f4){n1.x, n1.y, d1.x, d1.y};
            }
            __builtin_amdgcn_wave_barrier();
            const int sox = ox - WOFF, soy = oy - WOFF;
            const int hx = min(sox + LT_COLS - WIN, A.nx_f - 2), hy = min(soy + LT_ROWS - WIN, A.ny_f - 2);
            // DEFER_X: column 0 and row 0 are left to the exact path.  A parcel sitting EXACTLY on lon_min = -180 has index
            // 0.0 and must still meet Q7's `x > -180` test (every other longitude the reference would wrap or clamp maps
            // to an index outside [0, n-1) and fails the window test by itself); and a NaN coordinate converts to index 0,
            // so with origin 0 excluded it fails the window test on either axis and is clamped first (Q8: a NaN latitude
            // becomes y_min while the longitude lives on) -- the round-5 form caught it through x_needs_care(NaN).
            const int lx = max(sox, DEFER_X ? 1 : 0), ly = max(soy, DEFER_X ? 1 : 0);
            if (hx >= lx && hy >= ly) {
                lo_x = lx;
                lo_y = ly;
                lim_x = hx - lx;
                lim_y = hy - ly;
                base_addr = tile_addr + (unsigned)(lx - sox) * 16u + (unsigned)(ly - soy) * ((unsigned)LT_PITCH * 16u);
            }
        }
#ifdef LCS_STAMPS
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
#endif
        LCS_STAMP(2)  // wait for the tile + LDS write
        // ---- 4. K iterations out of LDS (latitude clamp deferred to the redo path / the level's end) -----
        int redos = 0;  // iterations of this level in which some lane needed the exact sequence (wave-uniform)
        if (direct) {   // (see LCS_LDS2_DIRECT_LEVELS) the exact sequence for every lane, no tile
#pragma unroll 1
            for (int k = 0; k < K; ++k) {
#pragma unroll
                for (int q = 0; q < NS; ++q) {
                    f2 pc = p[q];
                    clamp_position_c<CYCLIC>(A, pc, ymax_v);
                    const TapL t = tap_of(index_coords(A, pc));
                    p[q] = hd[q] * window_global<ORDER>(elv, A, t, e[q]) + pc;
                }
            }
        } else
#pragma unroll
        for (int k = 0; k < K; ++k) {
            anybad = false;
#pragma unroll
            for (int q = 0; q < NS; ++q) {
                typedef __attribute__((address_space(3))) const f4 lds_f4;
                const TapL t = tap_of(to_index(p[q]));
                const int rx = t.x0 - lo_x, ry = t.y0 - lo_y;
                bad[q] = ((unsigned)rx > (unsigned)lim_x) | ((unsigned)ry > (unsigned)lim_y);
#ifdef LCS_STAMPS
                {
                    const unsigned long long bx = __ballot((unsigned)rx > (unsigned)lim_x), by = __ballot((unsigned)ry > (unsigned)lim_y);
                    const unsigned long long bxl = __ballot(rx < 0), byl = __ballot(ry < 0);
                    if (lane == 0) {
                        atomicAdd(&g_cause[0], (unsigned long long)__popcll(bx));
                        atomicAdd(&g_cause[1], (unsigned long long)__popcll(by));
                        atomicAdd(&g_cause[2], (unsigned long long)__popcll(bxl));
                        atomicAdd(&g_cause[3], (unsigned long long)__popcll(byl));
                    }
                }
#endif
                unsigned row_addr;
                asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(row_addr) : "v"(ry), "s"(pitch_bytes), "v"(base_addr));
                lds_f4 *cell = (lds_f4 *)(size_t)(row_addr + ((unsigned)rx << 4));
                const f4 c0 = cell[0], c1 = cell[LT_PITCH];
                const f2 r0 = c0.xy + t.tx * c0.zw;   // n00 + tx (n01 - n00)
                const f2 r1 = c1.xy + t.tx * c1.zw;   // n10 + tx (n11 - n10)
                const f2 ew = e[q] + (r0 + t.ty * (r1 - r0));  // e + sample of ext[t]
                pn[q] = hd[q] * ew + p[q];
                if (!DEFER_X) bad[q] |= x_needs_care(pn[q].x);
                anybad |= bad[q];
            }
#ifdef LCS_STAMPS
            {   // g_stamps[4]: wave-samples, [5]: wave-samples with a redo, [6]: lane-seed-samples redone
                const unsigned long long m0 = __ballot(bad[0]), m1 = __ballot(bad[1]);
                if (lane == 0) {
                    acc_n[0] += 1;
                    acc_n[1] += (m0 | m1) ? 1 : 0;
                    level_redos += (m0 | m1) ? 1 : 0;
                    acc_n[2] += __popcll(m0) + __popcll(m1);
                    const int band = min(2, (int)(fabsf(A.seed_lat[min(iy0, A.ny - 1)]) / 30.0f)), third = min(2, 3 * s / A.nsteps);
                    atomicAdd(&g_redo[band][third][0], 1ull);
                    atomicAdd(&g_redo[band][third][1], (m0 | m1) ? 1ull : 0ull);
                    atomicAdd(&g_redo[band][third][2], (unsigned long long)(__popcll(m0) + __popcll(m1)));
                }
            }
#endif
            if (DIRECT_LEVELS > 0) redos += __ballot(anybad) != 0ull;
            if (anybad) {
#pragma unroll
                for (int q = 0; q < NS; ++q) {
                    if (bad[q]) {  // exact sequence, global gather
                        f2 pc = p[q];
                        if (DEFER_X)
                            clamp_position_c<CYCLIC>(A, pc, ymax_v);  // the deferred clamps of the previous update (Q7 / Q8 / Q9)
                        else
                            pc.y = __builtin_amdgcn_fmed3f(pc.y, A.y_min, ymax_v);  // the deferred clamp (Q8)
                        const TapL t = tap_of(index_coords(A, pc));
                        pn[q] = hd[q] * window_global<ORDER>(elv, A, t, e[q]) + pc;
                        if (!DEFER_X) clamp_position_p(A, pn[q], ymax_v);
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < NS; ++q) p[q] = pn[q];
        }
        if (DIRECT_LEVELS > 0 && redos >= 3) direct_left = DIRECT_LEVELS;
#pragma unroll
        for (int q = 0; q < NS; ++q) {  // the level's one latitude clamp -- and, with DEFER_X, its one longitude test
            if (DEFER_X)
                clamp_position_c<CYCLIC>(A, p[q], ymax_v);
            else
                p[q].y = __builtin_amdgcn_fmed3f(p[q].y, A.y_min, ymax_v);
        }
        if (GROUP) {  // a member outside its own steps keeps its position, bit for bit
#pragma unroll
            for (int q = 0; q < NS; ++q)
                if (!act[q]) p[q] = keep[q];
        }
        if (LINES && lines) {
            float *sx = s_slab[s & 1][0], *sy = s_slab[s & 1][1];
#pragma unroll
            for (int q = 0; q < NS; ++q) {
                const int o = (lane / TILE_W + 8 * q) * SLAB_PITCH + wave * TILE_W + (lane % TILE_W);
                sx[o] = p[q].x;
                sy[o] = p[q].y;
            }
            // LDS writes done, then the level's one barrier.  The slab written at level s is read below by other waves;
            // it is written again at level s + 2, which every wave reaches only after the barrier of level s + 1, i.e.
            // after every wave has finished these reads.
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            const f4 line = *(const f4 *)(s_slab[s & 1][wave >> 1] + sl_row * SLAB_PITCH + sl_col);
            if (sl_ok) {
                float *dst = ((wave >> 1) ? A.traj_y : A.traj_x) + (size_t)(s + 1) * plane + sl_idx;
                LCS_TRAJ_STORE(dst, line);
            }
        } else if (A.traj_x) {
            store_pair(A.traj_x, A.traj_y, (size_t)(s + 1) * plane);
        }
#ifdef LCS_STAMPS
        asm volatile("" : : "v"(p[0].x), "v"(p[1].x));
#endif
        LCS_STAMP(3)  // K iterations
#ifdef LCS_STAMPS
        if (lane == 0) {
            atomicAdd(&g_hist[0][min(level_redos, 4)], 1ull);
            if (prev_redos >= 3) atomicAdd(&g_hist[1][min(level_redos, 4)], 1ull);
            prev_redos = level_redos;
        }
#endif
        lvl += A.level_elems;
        elv += A.level_elems;
    }
#ifdef LCS_STAMPS
    if (lane == 0) {
        for (int i = 0; i < 4; ++i) atomicAdd(&g_stamps[i], (unsigned long long)acc_t[i]);
        for (int i = 0; i < 3; ++i) atomicAdd(&g_stamps[4 + i], acc_n[i]);
    }
#endif
    if (WIDE && pair && A.out_pair_ok) {
        *(f2 *)(A.x_out + idx[0]) = (f2){p[0].x, p[1].x};
        *(f2 *)(A.y_out + idx[0]) = (f2){p[0].y, p[1].y};
    } else {
#pragma unroll
        for (int q = 0; q < NS; ++q) {
            if (live[q]) {
                A.x_out[idx[q]] = p[q].x;
                A.y_out[idx[q]] = p[q].y;
            }
        }
    }
}

// ======================================================================================
// Order 3 (the reference's default, LCS/trajectory.py:16), TWO seeds per lane.
//
// advect_lds_kernel<3> with the seed patches, the patch modes and the trajectory stores of advect_lds2_kernel: a wave
// advects 8 x 16 seeds, so the per-wave work of a time level -- anchoring and staging the 32 x 16-node tile of ext[t]
// and the 16 x 8-node tile of img[t], the scalar bookkeeping, one rare branch per sample -- is shared by 128 seeds
// instead of 64, and two independent 16-tap windows are in flight per lane.  Arithmetic per seed is exactly the
// one-seed kernel's (cubic_apply on the same taps): results are bit-identical to it and to the direct-gather kernel.
// ======================================================================================
template <int MODE>
struct PatchLanes {  // the lane's SPL seeds: where they are in the grid, their state, how they are stored (enum Patch)
    static constexpr bool WIDE = MODE == PATCH_WIDE, LINES = MODE == PATCH_LINES;
    bool live[SPL], any, pair, lines, sl_ok;
    f2 p[SPL], dd[SPL], hd[SPL];
    size_t idx[SPL], sl_idx, plane;
    int lane, wave, sl_row, sl_col;

    __device__ __forceinline__ void init(const AdvectArgs<float> &A, int txi, int tyi) {
        lane = threadIdx.x & 63;
        wave = threadIdx.x >> 6;
        plane = (size_t)A.ny * A.nx;
        const int ix0 = WIDE ? txi * (TILE_W * SPL) + SPL * (lane % TILE_W)
                      : LINES ? txi * (TILE_W * 4) + wave * TILE_W + (lane % TILE_W) : txi * TILE_W + (lane % TILE_W);
        const int iy0 = WIDE ? tyi * TILE_H + wave * 8 + lane / TILE_W
                      : LINES ? tyi * (8 * SPL) + lane / TILE_W : tyi * (TILE_H * SPL) + wave * (8 * SPL) + lane / TILE_W;
        any = false;
#pragma unroll
        for (int q = 0; q < SPL; ++q) {
            const int ix = ix0 + (WIDE ? q : 0), iy = iy0 + (WIDE ? 0 : 8 * q);
            live[q] = ix < A.nx && iy < A.ny;
            if (live[q]) {
                const int grow = A.row0 + iy;
                if (grow < A.order || grow >= A.ny_global - A.order) {  // pole rows: generic path, whole integration (Q3)
                    if (!A.pole_blocks) pole_seed<float, POLE_EITHER>(A, iy, ix);  // (else the leading workgroups did them)
                    live[q] = false;
                }
            }
            any |= live[q];
            // lanes without a seed shadow a neighbouring one so that they follow the same path; only their stores are masked
            const int sx_i = min(ix, A.nx - 1), sy_i = min(iy, A.ny - 1);
            p[q] = (f2){start_x<float>(A, sy_i, sx_i), start_y<float>(A, sy_i, sx_i)};
            const float ys = A.seed_lat[sy_i];  // conversion_x is a function of the SEED latitude (Q5)
            const float cx_conv =
                180.0f / ((float)(3.141592653589793 * 6371000.0) * fabsf(cosf((ys * (float)3.141592653589793) / 180.0f)));
            dd[q] = (f2){A.dt * cx_conv, A.dtcy};        // trajectory.py:55-57,86-87
            hd[q] = (f2){A.half_dt * cx_conv, A.hdtcy};  // trajectory.py:110-112
            idx[q] = live[q] ? (size_t)iy * A.nx + ix : 0;
        }
        pair = WIDE && live[0] && live[1] && A.traj_pair_ok;
        // PATCH_LINES: whole-line stores for workgroups whose 32 columns are all inside the grid; workgroup-uniform
        lines = LINES && A.traj_x && A.traj_line_ok && (txi + 1) * (TILE_W * 4) <= A.nx;
        sl_row = (wave & 1) * 8 + (lane >> 3);
        sl_col = (lane & 7) * 4;
        sl_ok = false;
        sl_idx = 0;
        if (lines) {
            const int iyr = tyi * (8 * SPL) + sl_row, grow = A.row0 + iyr;
            sl_ok = iyr < A.ny && grow >= A.order && grow < A.ny_global - A.order;  // (pole rows are written by their own threads)
            sl_idx = (size_t)min(iyr, A.ny - 1) * A.nx + (size_t)txi * (TILE_W * 4) + sl_col;
        }
    }
    __device__ __forceinline__ void store_lanes(float *dx, float *dy, size_t off) const {
        if (pair) {
            *(f2 *)(dx + off + idx[0]) = (f2){p[0].x, p[1].x};
            *(f2 *)(dy + off + idx[0]) = (f2){p[0].y, p[1].y};
        } else {
#pragma unroll
            for (int q = 0; q < SPL; ++q)
                if (live[q]) {
                    dx[off + idx[q]] = p[q].x;
                    dy[off + idx[q]] = p[q].y;
                }
        }
    }
    // positions after time level s -> traj entry s + 1 (slab: the workgroup's [2 slabs][2 planes][16 * SLAB_PITCH] floats)
    __device__ __forceinline__ void store_level(const AdvectArgs<float> &A, float *slab, int s) const {
        if (LINES && lines) {
            float *sx = slab + (size_t)(s & 1) * 2 * 16 * SLAB_PITCH, *sy = sx + 16 * SLAB_PITCH;
#pragma unroll
            for (int q = 0; q < SPL; ++q) {
                const int o = (lane / TILE_W + 8 * q) * SLAB_PITCH + wave * TILE_W + (lane % TILE_W);
                sx[o] = p[q].x;
                sy[o] = p[q].y;
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");  // (one barrier per level: see advect_lds2_kernel)
            const f4 line = *(const f4 *)(((wave >> 1) ? sy : sx) + sl_row * SLAB_PITCH + sl_col);
            if (sl_ok) LCS_TRAJ_STORE(((wave >> 1) ? A.traj_y : A.traj_x) + (size_t)(s + 1) * plane + sl_idx, line);
        } else if (A.traj_x) {
            store_lanes(A.traj_x, A.traj_y, (size_t)(s + 1) * plane);
        }
    }
    __device__ __forceinline__ void store_final(const AdvectArgs<float> &A) const {
        if (WIDE && pair && A.out_pair_ok) {
            *(f2 *)(A.x_out + idx[0]) = (f2){p[0].x, p[1].x};
            *(f2 *)(A.y_out + idx[0]) = (f2){p[0].y, p[1].y};
        } else {
#pragma unroll
            for (int q = 0; q < SPL; ++q)
                if (live[q]) {
                    A.x_out[idx[q]] = p[q].x;
                    A.y_out[idx[q]] = p[q].y;
                }
        }
    }
};

// 97 VGPRs as compiled = 4 waves per SIMD; asked for 5 (96 VGPRs) it measures 16.05-16.1 ms on C3 against 16.6-16.7, asked
// for 6 (spills) 17.4; the one-seed kernel 17.2 on the same box (profiles/r03).
#ifndef LCS_LDS2_O3_MINWAVES
#define LCS_LDS2_O3_MINWAVES 5
#endif
template <int KFIX, bool CYCLIC, int MODE>
__global__ void __launch_bounds__(BLOCK, LCS_LDS2_O3_MINWAVES) advect_lds2_o3_kernel(const AdvectArgs<float> A0) {
#pragma clang fp contract(fast)
    const AdvectArgs<float> A = for_member(A0);
    constexpr int ORDER = 3;
    constexpr bool WIDE = MODE == PATCH_WIDE, LINES = MODE == PATCH_LINES;
    constexpr bool DEFER_X = LCS_LDS2_DEFER_X != 0;  // the longitude wrap / clamp deferred like the latitude clamp (see the order-1 kernel)
    const int K = KFIX >= 0 ? KFIX : A.K;
    typedef TileGeom<ORDER> G;
    typedef EulerGeom<ORDER> E;
    static_assert(E::ON, "the two-seed order-3 kernel takes its Euler sample from the LDS tile of img[t]");
    constexpr int LT_COLS = G::COLS, LT_ROWS = G::ROWS, LT_PITCH = G::PITCH;
    constexpr int WIN = ORDER + 1, WOFF = 0;  // padded window origin = (y0, x0): one node up / left of the cell
    __shared__ __attribute__((aligned(16))) f2 s_tiles[BLOCK / 64][LT_ROWS * LT_PITCH + E::ELEMS];
    __shared__ __attribute__((aligned(16))) float s_slab[2][2][LINES ? 16 * SLAB_PITCH : 4];
    if (pole_block(A)) return;
    const int tile_id = xcd_tile_id(A);
    if (tile_id >= A.ntiles) return;  // whole block
    const int tyi = tile_id / A.ntx, txi = tile_id - tyi * A.ntx;
    PatchLanes<MODE> L;
    L.init(A, txi, tyi);
    if (A.traj_x && !A.traj_skip0) L.store_lanes(A.traj_x, A.traj_y, 0);
    if (!L.lines && __ballot(L.any) == 0ull) return;  // whole wave (no workgroup barrier below unless `lines`)
    const int lane = L.lane;
    f2 *tile = s_tiles[L.wave], *etile = tile + LT_ROWS * LT_PITCH;
    float ymax_v = A.y_max;
    asm volatile("" : "+v"(ymax_v));
    const unsigned tile_addr = lds_address(tile), etile_addr = lds_address(etile);
    unsigned pitch_bytes = (unsigned)LT_PITCH * 8u, epitch_bytes = (unsigned)E::PITCH * 8u;
    asm volatile("" : "+s"(pitch_bytes));
    asm volatile("" : "+s"(epitch_bytes));
    const f2 pmin = {A.lon_min, A.lat_min}, sc = {A.sx, A.sy};
    auto to_index = [&](f2 v) { return (v - pmin) * sc; };  // subtract first: exact 0 at the grid origin
    const float xlo = A.x_min, xhi = A.x_max;
    auto x_needs_care = [&](float x) { return CYCLIC ? !(fabsf(x) < 180.0f) : !((x > xlo) & (x < xhi)); };
    const float *lvl = A.img + (size_t)A.t0 * A.level_elems;
    const float *elv = A.ext + (size_t)A.t0 * A.level_elems;
    const int pad_cols = A.pitch, pad_rows = A.ny_f + LC_PAD;
    const float kpred = 0.5f * (float)(K > 0 ? K - 1 : 0);
    const int st_row = lane / G::LANES_PER_ROW, st_col = (lane % G::LANES_PER_ROW) * 2;
    const unsigned st_off = ((unsigned)st_row * (unsigned)pad_cols + (unsigned)st_col) * 8u;
    constexpr int NPASS = LT_ROWS / G::ROWS_PER_PASS;
    const int e_row = lane / E::LANES_PER_ROW, e_col = (lane % E::LANES_PER_ROW) * 2;
    const unsigned e_off = ((unsigned)e_row * (unsigned)pad_cols + (unsigned)e_col) * 8u;
    constexpr int CENTRE = WIDE ? TILE_W / 2 + TILE_W * 4 : TILE_W / 2 + TILE_W * 7;  // seed 0 of the lane in the patch's middle
    const f2 zero = {0.0f, 0.0f};
    f2 (&p)[SPL] = L.p;
    for (int s = 0; s < A.nsteps; ++s) {
        // ---- 1. Euler samples out of a 16 x 8-node tile of img[t] around the patch's current position -----------
        f2 c0[SPL], e[SPL], pn[SPL];
        TapL t0[SPL];
        bool bad[SPL], anybad = false;
#pragma unroll
        for (int q = 0; q < SPL; ++q) {
            c0[q] = to_index(p[q]);
            t0[q] = tap_of(c0[q]);
            bad[q] = ((unsigned)t0[q].x0 > (unsigned)(A.nx_f - 2)) | ((unsigned)t0[q].y0 > (unsigned)(A.ny_f - 2));
        }
        {
            const int exm = __builtin_amdgcn_readlane(t0[0].x0, CENTRE), eym = __builtin_amdgcn_readlane(t0[0].y0, CENTRE);
            const int eox = min(max(exm + WOFF - (E::COLS - WIN) / 2, 0), pad_cols - E::COLS);
            const int eoy = min(max(eym + WOFF - (E::ROWS - WIN) / 2, 0), pad_rows - E::ROWS);
            const char *src = (const char *)lvl + ((size_t)__umul24((unsigned)eoy, (unsigned)pad_cols) + (unsigned)eox) * 8;
            f4 es[E::NPASS];
#pragma unroll
            for (int r = 0; r < E::NPASS; ++r)
                __builtin_memcpy(&es[r], src + (size_t)(r * E::ROWS_PER_PASS) * pad_cols * 8 + e_off, 16);
            __builtin_amdgcn_wave_barrier();  // the previous level's reads of this region are done
#pragma unroll
            for (int r = 0; r < E::NPASS; ++r) *(f4 *)(etile + (r * E::ROWS_PER_PASS + e_row) * E::PITCH + e_col) = es[r];
            __builtin_amdgcn_wave_barrier();
            // window origins the tile serves: inside it AND in [0, n-2]
            const int sox = eox - WOFF, soy = eoy - WOFF;
            const int hx = min(sox + E::COLS - WIN, A.nx_f - 2), hy = min(soy + E::ROWS - WIN, A.ny_f - 2);
            const int lx = max(sox, 0), ly = max(soy, 0);
            const unsigned ebase = etile_addr + (unsigned)(lx - sox) * 8u + (unsigned)(ly - soy) * ((unsigned)E::PITCH * 8u);
#pragma unroll
            for (int q = 0; q < SPL; ++q) {
                const int rx = t0[q].x0 - lx, ry = t0[q].y0 - ly;
                bad[q] |= ((unsigned)rx > (unsigned)(hx - lx)) | ((unsigned)ry > (unsigned)(hy - ly)) | (hx < lx) | (hy < ly);
                e[q] = window_lds<ORDER, E::PITCH>(ebase, epitch_bytes, rx, ry, t0[q], zero);
                pn[q] = L.dd[q] * e[q] + p[q];
                if (!DEFER_X) bad[q] |= x_needs_care(pn[q].x);
                anybad |= bad[q];
            }
        }
        if (anybad) {  // exact sequence for the lanes / seeds that need it
#pragma unroll
            for (int q = 0; q < SPL; ++q) {
                if (bad[q]) {
                    const TapL t = tap_of(index_coords(A, p[q]));
                    e[q] = euler_global<ORDER>(lvl, A, t);
                    pn[q] = L.dd[q] * e[q] + p[q];
                    if (!DEFER_X) clamp_position_p(A, pn[q], ymax_v);
                }
            }
        }
        const f2 dnow = (pn[0] - p[0]) * sc;  // this level's Euler displacement in index space
#pragma unroll
        for (int q = 0; q < SPL; ++q) p[q] = pn[q];
        // ---- 2. tile of ext[t], anchored on the centre lane's travel as this level's displacement predicts it --------
        int lo_x = 0x40000000, lo_y = 0x40000000, lim_x = 0, lim_y = 0;  // no tile: nothing is "inside"
        unsigned base_addr = tile_addr;
        if (K > 0) {
            const f2 ca = dnow * (1.0f + kpred) + c0[0];
            const int rxm = __builtin_amdgcn_readlane((int)floor_to_uint(ca.x), CENTRE);
            const int rym = __builtin_amdgcn_readlane((int)floor_to_uint(ca.y), CENTRE);
            const int ox = min(max(rxm + WOFF - (LT_COLS - WIN) / 2, 0), pad_cols - LT_COLS);
            const int oy = min(max(rym + WOFF - (LT_ROWS - WIN) / 2, 0), pad_rows - LT_ROWS);
            const char *src = (const char *)elv + ((size_t)__umul24((unsigned)oy, (unsigned)pad_cols) + (unsigned)ox) * 8;
            f4 stage[NPASS];
#pragma unroll
            for (int r = 0; r < NPASS; ++r)
                __builtin_memcpy(&stage[r], src + (size_t)(r * G::ROWS_PER_PASS) * pad_cols * 8 + st_off, 16);
            __builtin_amdgcn_wave_barrier();  // the previous level's reads are done (LDS ops of a wave are in order)
#pragma unroll
            for (int r = 0; r < NPASS; ++r) *(f4 *)(tile + (r * G::ROWS_PER_PASS + st_row) * LT_PITCH + st_col) = stage[r];
            __builtin_amdgcn_wave_barrier();
            const int sox = ox - WOFF, soy = oy - WOFF;
            const int hx = min(sox + LT_COLS - WIN, A.nx_f - 2), hy = min(soy + LT_ROWS - WIN, A.ny_f - 2);
            const int lx = max(sox, DEFER_X ? 1 : 0), ly = max(soy, DEFER_X ? 1 : 0);  // (DEFER_X: origin 0 to the exact path, as at order 1)
            if (hx >= lx && hy >= ly) {
                lo_x = lx;
                lo_y = ly;
                lim_x = hx - lx;
                lim_y = hy - ly;
                base_addr = tile_addr + (unsigned)(lx - sox) * 8u + (unsigned)(ly - soy) * ((unsigned)LT_PITCH * 8u);
            }
        }
        // ---- 3. K iterations out of LDS (latitude clamp deferred to the redo path / the level's end) ---------------
#pragma unroll
        for (int k = 0; k < K; ++k) {
            anybad = false;
#pragma unroll
            for (int q = 0; q < SPL; ++q) {
                const TapL t = tap_of(to_index(p[q]));  // absolute coordinate: its rounding must not depend on the tile
                const int rx = t.x0 - lo_x, ry = t.y0 - lo_y;
                bad[q] = ((unsigned)rx > (unsigned)lim_x) | ((unsigned)ry > (unsigned)lim_y);
                const f2 ew = window_lds<ORDER, LT_PITCH>(base_addr, pitch_bytes, rx, ry, t, e[q]);  // e + sample of ext[t]
                pn[q] = L.hd[q] * ew + p[q];
                if (!DEFER_X) bad[q] |= x_needs_care(pn[q].x);
                anybad |= bad[q];
            }
            if (anybad) {
#pragma unroll
                for (int q = 0; q < SPL; ++q) {
                    if (bad[q]) {  // exact sequence, global gather
                        f2 pc = p[q];
                        if (DEFER_X)
                            clamp_position_c<CYCLIC>(A, pc, ymax_v);  // the deferred clamps of the previous update (Q7 / Q8 / Q9)
                        else
                            pc.y = __builtin_amdgcn_fmed3f(pc.y, A.y_min, ymax_v);  // the deferred clamp (Q8)
                        const TapL t = tap_of(index_coords(A, pc));
                        pn[q] = L.hd[q] * window_global<ORDER>(elv, A, t, e[q]) + pc;
                        if (!DEFER_X) clamp_position_p(A, pn[q], ymax_v);
                    }
                }
            }
#pragma unroll
            for (int q = 0; q < SPL; ++q) p[q] = pn[q];
        }
#pragma unroll
        for (int q = 0; q < SPL; ++q) {  // the level's one clamp of either kind
            if (DEFER_X)
                clamp_position_c<CYCLIC>(A, p[q], ymax_v);
            else
                p[q].y = __builtin_amdgcn_fmed3f(p[q].y, A.y_min, ymax_v);
        }
        L.store_level(A, &s_slab[0][0][0], s);
        lvl += A.level_elems;
        elv += A.level_elems;
    }
    L.store_final(A);
}

// Kernel family of a float32 LDS-tile launch.  Two seeds per lane pays once the launch is many rounds of workgroups deep;
// mode (lc_ctx_set_lds_tiles) 1 / 2 force either, 3 = by size (see LdsLaunch).
static inline bool two_seeds_per_lane(const AdvectArgs<float> &A, int mode) {
    return mode == 1 || (mode == 3 && (long long)A.nx * A.ny * (long long)nmem(A) >= (1ll << 23));
}
static inline bool order1_two_seed_applies(const AdvectArgs<float> &A, int mode) {
    // SETTLS_order = 0 (the library default, LCS/trajectory.py:14) stages no tile and needs no ext image; two seeds per lane and
    // packed position arithmetic still pay (round 6: C3 at K = 0, 2.21 -> 1.72 ms against the one-seed direct kernel)
    return two_seeds_per_lane(A, mode) && (A.K == 0 || A.ext) && A.nx_f + LC_PAD >= 32 && A.ny_f + LC_PAD >= 16;
}
template <typename T>
static inline bool order1_two_seed_applies(const AdvectArgs<T> &, int) { return false; }

template <typename T, int ORDER>
struct LdsLaunch {
    static const char *launch(const AdvectArgs<T> &, int, hipStream_t, int) { return nullptr; }
};
template <int ORDER>
struct LdsLaunch<float, ORDER> {
    // returns the launched kernel's name, or NULL when the LDS kernel does not apply
    static const char *launch(const AdvectArgs<float> &A0, int grid, hipStream_t st, int mode) {
        AdvectArgs<float> A = A0;
        // two seeds per lane pays once the launch is many rounds of workgroups deep; below ~8 M seeds the one-seed kernel's
        // twice as many waves fill the machine better (4096 x 512 seeds, one GPU's share of C3 split 8 ways: +24 %;
        // measured cross-over between 2896^2 and 3500^2).  mode 1 / 2 force either (tests, A/B); 3 = by size.
        // (with trajectories too: both families store whole lines; 2048^2 seeds x 96 levels, one seed / two seeds per lane:
        // order 1 3.26 / 3.24 ms, order 3 5.94 / 6.42; 1024^2: 1.41 / 1.60 and 2.66 / 3.39)
        const bool two_seed = two_seeds_per_lane(A, mode);
        if (ORDER == 1 && A.pair_d >= 0) {
            // an ensemble, two MEMBERS per lane (advect_impl checked order1_two_seed_applies): the one-seed kernel's 8 x 32-seed workgroups
            A.tile_order = A.tile_order_two_seed;
#define LC_LDS2P(KF, CY, MD, NAME)                                                                          \
    {                                                                                                       \
        hipLaunchKernelGGL((advect_lds2_kernel<KF, CY, MD>), dim3(grid, nmem(A)), dim3(BLOCK), 0, st, A);   \
        return NAME;                                                                                        \
    }
            if (A.K == 4 && A.cyclic) LC_LDS2P(4, true, PATCH_PAIR, "advect_lds2_kernel<4, true, 3>")
            if (A.K == 4) LC_LDS2P(4, false, PATCH_PAIR, "advect_lds2_kernel<4, false, 3>")
            if (A.cyclic) LC_LDS2P(-1, true, PATCH_PAIR, "advect_lds2_kernel<-1, true, 3>")
            LC_LDS2P(-1, false, PATCH_PAIR, "advect_lds2_kernel<-1, false, 3>")
#undef LC_LDS2P
        }
        if (ORDER == 1 && order1_two_seed_applies(A, mode)) {
            // two seeds per lane; a workgroup covers 8 x 64 seeds (PATCH_TALL), 16 x 32 (PATCH_WIDE) or 32 x 16 (PATCH_LINES)
            const int mode = (A.patch_mode >= 0 && A.patch_mode < PATCH_PAIR) ? A.patch_mode
                                                                               : (A.traj_x && A.traj_line_ok && A.nx >= TILE_W * 4 ? PATCH_LINES : PATCH_TALL);
            int nty = (A.ny + TILE_H * SPL - 1) / (TILE_H * SPL);
            if (mode == PATCH_WIDE) {
                A.ntx = (A.nx + TILE_W * SPL - 1) / (TILE_W * SPL);
                nty = (A.ny + TILE_H - 1) / TILE_H;
            } else if (mode == PATCH_LINES) {
                A.ntx = (A.nx + TILE_W * 4 - 1) / (TILE_W * 4);
                nty = (A.ny + 8 * SPL - 1) / (8 * SPL);
            }
            A.xcd_chunk = lcplan::xcd_chunk_tiles(A.ntx, nty, A.xcd_rows, A.xcd_split);
            A.ntiles = A.ntx * nty;
            A.tile_order = A.tile_order_two_seed;
            const int g2 = xcd_grid(A.ntiles, A.xcd_chunk) + A.pole_blocks;
#define LC_LDS2(KF, CY, MD, NAME)                                                                      \
    {                                                                                                  \
        hipLaunchKernelGGL((advect_lds2_kernel<KF, CY, MD>), dim3(g2, nmem(A)), dim3(BLOCK), 0, st, A);         \
        return NAME;                                                                                   \
    }
            if (mode == PATCH_LINES) {
                if (A.K == 4 && A.cyclic) LC_LDS2(4, true, PATCH_LINES, "advect_lds2_kernel<4, true, 2>")
                if (A.K == 4) LC_LDS2(4, false, PATCH_LINES, "advect_lds2_kernel<4, false, 2>")
                if (A.cyclic) LC_LDS2(-1, true, PATCH_LINES, "advect_lds2_kernel<-1, true, 2>")
                LC_LDS2(-1, false, PATCH_LINES, "advect_lds2_kernel<-1, false, 2>")
            }
            if (mode == PATCH_WIDE) {
                if (A.K == 4 && A.cyclic) LC_LDS2(4, true, PATCH_WIDE, "advect_lds2_kernel<4, true, 1>")
                if (A.K == 4) LC_LDS2(4, false, PATCH_WIDE, "advect_lds2_kernel<4, false, 1>")
                if (A.cyclic) LC_LDS2(-1, true, PATCH_WIDE, "advect_lds2_kernel<-1, true, 1>")
                LC_LDS2(-1, false, PATCH_WIDE, "advect_lds2_kernel<-1, false, 1>")
            }
            // (SETTLS_order = 0, the library default, compiled as such: no tile, no iteration blocks -- C3 1.73 -> 1.69 ms against the run-time-K instance)
            if (A.K == 0 && A.cyclic) LC_LDS2(0, true, PATCH_TALL, "advect_lds2_kernel<0, true, 0>")
            if (A.K == 0) LC_LDS2(0, false, PATCH_TALL, "advect_lds2_kernel<0, false, 0>")   // (cyclic_xboundary=False is the reference's default too)
            // (SETTLS_order 1, 2, 3 compiled as such too: the iteration loop unrolls and the kernel keeps the K = 4 instance's 59 registers
            // instead of the run-time-K instance's 71 -- C3 at K = 1: 3.28 -> 2.91 ms, K = 2: 4.26 -> 3.93)
            if (A.K == 1 && A.cyclic) LC_LDS2(1, true, PATCH_TALL, "advect_lds2_kernel<1, true, 0>")
            if (A.K == 2 && A.cyclic) LC_LDS2(2, true, PATCH_TALL, "advect_lds2_kernel<2, true, 0>")
            if (A.K == 3 && A.cyclic) LC_LDS2(3, true, PATCH_TALL, "advect_lds2_kernel<3, true, 0>")
            if (A.K == 4 && A.cyclic) LC_LDS2(4, true, PATCH_TALL, "advect_lds2_kernel<4, true, 0>")
            if (A.K == 4) LC_LDS2(4, false, PATCH_TALL, "advect_lds2_kernel<4, false, 0>")
            if (A.cyclic) LC_LDS2(-1, true, PATCH_TALL, "advect_lds2_kernel<-1, true, 0>")
            LC_LDS2(-1, false, PATCH_TALL, "advect_lds2_kernel<-1, false, 0>")
#undef LC_LDS2
        }
        // (order 3 also with SETTLS_order = 0, the reference's default: the Euler sample alone already gains from its LDS tile)
        if (ORDER == 3 && two_seed && (A.ext || A.K == 0) && A.nx_f + LC_PAD >= TileGeom<3>::COLS && A.ny_f + LC_PAD >= TileGeom<3>::ROWS) {
            // order 3, two seeds per lane: the same patches and patch modes as above
            const int mode = A.patch_mode >= 0 ? A.patch_mode : (A.traj_x && A.traj_line_ok && A.nx >= TILE_W * 4 ? PATCH_LINES : PATCH_TALL);
            int nty = (A.ny + TILE_H * SPL - 1) / (TILE_H * SPL);
            if (mode == PATCH_WIDE) {
                A.ntx = (A.nx + TILE_W * SPL - 1) / (TILE_W * SPL);
                nty = (A.ny + TILE_H - 1) / TILE_H;
            } else if (mode == PATCH_LINES) {
                A.ntx = (A.nx + TILE_W * 4 - 1) / (TILE_W * 4);
                nty = (A.ny + 8 * SPL - 1) / (8 * SPL);
            }
            A.xcd_chunk = lcplan::xcd_chunk_tiles(A.ntx, nty, A.xcd_rows, A.xcd_split);
            A.ntiles = A.ntx * nty;
            const int g2 = xcd_grid(A.ntiles, A.xcd_chunk) + A.pole_blocks;
#define LC_LDS2O3(KF, CY, MD, NAME)                                                                       \
    {                                                                                                     \
        hipLaunchKernelGGL((advect_lds2_o3_kernel<KF, CY, MD>), dim3(g2, nmem(A)), dim3(BLOCK), 0, st, A);         \
        return NAME;                                                                                      \
    }
            if (mode == PATCH_LINES) {
                if (A.K == 4 && A.cyclic) LC_LDS2O3(4, true, PATCH_LINES, "advect_lds2_o3_kernel<4, true, 2>")
                if (A.K == 4) LC_LDS2O3(4, false, PATCH_LINES, "advect_lds2_o3_kernel<4, false, 2>")
                if (A.cyclic) LC_LDS2O3(-1, true, PATCH_LINES, "advect_lds2_o3_kernel<-1, true, 2>")
                LC_LDS2O3(-1, false, PATCH_LINES, "advect_lds2_o3_kernel<-1, false, 2>")
            }
            if (mode == PATCH_WIDE) {
                if (A.K == 4 && A.cyclic) LC_LDS2O3(4, true, PATCH_WIDE, "advect_lds2_o3_kernel<4, true, 1>")
                if (A.K == 4) LC_LDS2O3(4, false, PATCH_WIDE, "advect_lds2_o3_kernel<4, false, 1>")
                if (A.cyclic) LC_LDS2O3(-1, true, PATCH_WIDE, "advect_lds2_o3_kernel<-1, true, 1>")
                LC_LDS2O3(-1, false, PATCH_WIDE, "advect_lds2_o3_kernel<-1, false, 1>")
            }
            // (interp_order = 3 with SETTLS_order = 0 are the reference's DEFAULT arguments: compiled as such, 85 registers instead of
            // 94 + scratch and no iteration blocks -- C3 3.26 -> 3.10 ms against the run-time-K instance)
            if (A.K == 0 && A.cyclic) LC_LDS2O3(0, true, PATCH_TALL, "advect_lds2_o3_kernel<0, true, 0>")
            if (A.K == 0) LC_LDS2O3(0, false, PATCH_TALL, "advect_lds2_o3_kernel<0, false, 0>")   // (... with cyclic_xboundary=False, its default)
            // (order-3 instances for K = 1, 2: measured, < 1 %)
            if (A.K == 4 && A.cyclic) LC_LDS2O3(4, true, PATCH_TALL, "advect_lds2_o3_kernel<4, true, 0>")
            if (A.K == 4) LC_LDS2O3(4, false, PATCH_TALL, "advect_lds2_o3_kernel<4, false, 0>")
            if (A.cyclic) LC_LDS2O3(-1, true, PATCH_TALL, "advect_lds2_o3_kernel<-1, true, 0>")
            LC_LDS2O3(-1, false, PATCH_TALL, "advect_lds2_o3_kernel<-1, false, 0>")
#undef LC_LDS2O3
        }
        // the fixed-size tile must fit inside one padded time level
        if ((!A.ext && !(ORDER == 3 && A.K == 0)) || A.nx_f + LC_PAD < TileGeom<ORDER>::COLS || A.ny_f + LC_PAD < TileGeom<ORDER>::ROWS) return nullptr;
        // SETTLS_order = 0 (the library default) at order 1: one Euler sample per level and nothing to stage a tile
        // for -- the direct-gather kernel is the faster one (2.2 vs 2.45 ms on C3; an Euler-from-LDS variant
        // with its tile loaded a level ahead measured 2.5 ms).  Order 3 keeps its LDS kernels: their Euler sample comes
        // from a tile of img[t] (one coalesced load per lane instead of eight gathers).
        if (A.K == 0 && ORDER != 3) return nullptr;
        // with trajectories on 16-byte aligned rows: whole-line stores (waves side by side: 32 x 8 seeds per workgroup)
        const bool lines = A.traj_x && A.traj_line_ok && A.nx >= TILE_W * 4 && A.patch_mode != PATCH_TALL && A.patch_mode != PATCH_WIDE;
        int g1 = grid;
        if (lines) {
            A.ntx = (A.nx + TILE_W * 4 - 1) / (TILE_W * 4);
            const int nty = (A.ny + 7) / 8;
            A.ntiles = A.ntx * nty;
            A.xcd_chunk = lcplan::xcd_chunk_tiles(A.ntx, nty, A.xcd_rows, A.xcd_split);
            g1 = xcd_grid(A.ntiles, A.xcd_chunk) + A.pole_blocks;
        }
#define LC_LDS1(KF, CY, LN, NAME)                                                                                       \
    {                                                                                                                   \
        hipLaunchKernelGGL((advect_lds_kernel<ORDER, KF, CY, LN>), dim3(g1, nmem(A)), dim3(BLOCK), 0, st, A);           \
        return NAME;                                                                                                    \
    }
        // K = 4 is the setting the reference's example and drivers use
        if (lines) {
            if (A.K == 4 && A.cyclic) LC_LDS1(4, true, true, ORDER == 3 ? "advect_lds_kernel<3, 4, true, lines>" : "advect_lds_kernel<1, 4, true, lines>")
            if (A.K == 4) LC_LDS1(4, false, true, ORDER == 3 ? "advect_lds_kernel<3, 4, false, lines>" : "advect_lds_kernel<1, 4, false, lines>")
            if (A.cyclic) LC_LDS1(-1, true, true, ORDER == 3 ? "advect_lds_kernel<3, -1, true, lines>" : "advect_lds_kernel<1, -1, true, lines>")
            LC_LDS1(-1, false, true, ORDER == 3 ? "advect_lds_kernel<3, -1, false, lines>" : "advect_lds_kernel<1, -1, false, lines>")
        }
        if (A.verify) {  // lc_ctx_set_verify: the instances that audit each wave's tile and slot level by level
#define LC_LDS1V(KF, CY, NAME)                                                                                           \
    {                                                                                                                    \
        hipLaunchKernelGGL((advect_lds_kernel<ORDER, KF, CY, false, true>), dim3(g1, nmem(A)), dim3(BLOCK), 0, st, A);   \
        return NAME;                                                                                                     \
    }
            if (A.K == 4 && A.cyclic) LC_LDS1V(4, true, ORDER == 3 ? "advect_lds_kernel<3, 4, true, verify>" : "advect_lds_kernel<1, 4, true, verify>")
            if (A.K == 4) LC_LDS1V(4, false, ORDER == 3 ? "advect_lds_kernel<3, 4, false, verify>" : "advect_lds_kernel<1, 4, false, verify>")
            if (A.cyclic) LC_LDS1V(-1, true, ORDER == 3 ? "advect_lds_kernel<3, -1, true, verify>" : "advect_lds_kernel<1, -1, true, verify>")
            LC_LDS1V(-1, false, ORDER == 3 ? "advect_lds_kernel<3, -1, false, verify>" : "advect_lds_kernel<1, -1, false, verify>")
#undef LC_LDS1V
        }
        if (A.K == 4 && A.cyclic) LC_LDS1(4, true, false, ORDER == 3 ? "advect_lds_kernel<3, 4, true>" : "advect_lds_kernel<1, 4, true>")
        if (A.K == 4) LC_LDS1(4, false, false, ORDER == 3 ? "advect_lds_kernel<3, 4, false>" : "advect_lds_kernel<1, 4, false>")
        if (A.cyclic) LC_LDS1(-1, true, false, ORDER == 3 ? "advect_lds_kernel<3, -1, true>" : "advect_lds_kernel<1, -1, true>")
        LC_LDS1(-1, false, false, ORDER == 3 ? "advect_lds_kernel<3, -1, false>" : "advect_lds_kernel<1, -1, false>")
#undef LC_LDS1
    }
};

// ======================================================================================
// float64, order 1, fused levels (the float64 default since round 3): the float path's arithmetic in double.
//
// The exact-order kernel above follows numpy / scipy operation by operation (two true divisions in the index map, scipy's
// 8-term tap sum, two samples per SETTLS iteration) and is bound by vector-L1 tag lookups and 650 float64 instructions
// per wave-level.  This form takes one sample of ext[t] = 2 F[t] - F[t+1] per iteration, maps the index by
// one subtraction and one multiplication with n / span, interpolates as three fused lerps and updates with fmas.  Every
// change is a rounding-level one (a few 1e-16 relative per operation); measured against the exact-order result on
// config 2 (1024^2 x 200 steps) the positions move by <= 1e-10 degrees, inside the 1e-9 degrees / 1e-7 sigma the float64
// parity tests state -- they run on this path unchanged.  fuse_levels=False selects the exact-order kernel.
// ======================================================================================
typedef double d2 __attribute__((ext_vector_type(2)));
typedef double d4 __attribute__((ext_vector_type(4)));

// trajectory.py:89-97 for the fast float64 family (direct and LDS-tile kernels alike: one function, so a seed's bits do not
// depend on which served it): the values of clamp_position<double> -- fmax(NaN, y_min) = y_min is Q8's rule, and a finite y
// comes out as the two selects give it -- in v_max_f64 + v_min_f64 instead of two compares and four 32-bit selects, and the
// cyclic wrap's two rare cases behind ONE test of |x| (round 5: the float64 order-1 kernel issues VALU in 0.67 of its
// cycles, 65 instructions per sample; these were 8 of them).
__device__ __forceinline__ void clamp_position_fast64(const AdvectArgs<double> &A, double &x, double &y) {
    y = fmin(fmax(y, A.y_min), A.y_max);
    if (A.cyclic) {
        if (!(fabs(x) < 180.0)) {  // rare: the exact reference sequence, trajectory.py:93-94 (Q7)
            if (!(x > -180.0)) x = pymod180<double>(x);
            if (!(x < 180.0)) x = -180.0 + pymod180<double>(x);
        }
    } else if (x < A.x_min || x > A.x_max) {
        if (A.clamp_flag) *A.clamp_flag = 1u;
        x = x < A.x_min ? A.x_min : A.x_max;
    }
}

// The two halves of a fast float64 sample, shared by the direct-gather kernel and the LDS-tile kernel below so that a
// seed's result never depends on which of the two served it: explicit operations, no contraction left to the compiler.
struct Loc64 {
    double tx, ty;
    int x0, y0;  // floor of the wrapped index coordinate, clamped into the grid (memory safety for NaN / inf)
};
__device__ __forceinline__ Loc64 locate_fast64(const AdvectArgs<double> &A, double x, double y) {
#pragma clang fp contract(off)
    double cx = (x - A.lon_min) * A.sx;  // subtract first: exact 0 at the grid origin
    double cy = (y - A.lat_min) * A.sy;
    const double szx = (double)(A.nx_f - 1), szy = (double)(A.ny_f - 1);
    // scipy's 'wrap' map is discontinuous at the last node (c = n - 1 stays, anything above lands next to node 0), and within
    // a rounding of it this multiply form and numpy's (n (x - min)) / span (tools.py:21-22) can fall on different sides -- a
    // seed row that sits exactly on the last node row of a coarser field does.  So the common-case test stops 1e-12 short of
    // n - 1, and in that band the coordinate IS numpy's expression: the decision and the value are the reference's there, and
    // the two forms meet at the band's edges to rounding.  (At c = 0 both forms are exact: the subtraction comes first.)
    const double gx = szx * (1.0 - 1e-12), gy = szy * (1.0 - 1e-12);
    if (!(cx >= 0.0 && cx < gx && cy >= 0.0 && cy < gy)) {  // rare: the band, or scipy's 'wrap' map (NaN falls through to the clamp)
        if (cx >= gx && cx <= szx * (1.0 + 1e-12)) cx = ((double)A.nx_f * (x - A.lon_min)) / A.lon_span;
        if (cy >= gy && cy <= szy * (1.0 + 1e-12)) cy = ((double)A.ny_f * (y - A.lat_min)) / A.lat_span;
        cx = wrap_coord<double>(cx, szx);
        cy = wrap_coord<double>(cy, szy);
    }
    const double fx = floor(cx), fy = floor(cy);
    Loc64 t;
    t.tx = cx - fx;
    t.ty = cy - fy;
    t.x0 = clampi((int)fx, 0, A.nx_f - 1);
    t.y0 = clampi((int)fy, 0, A.ny_f - 1);
    return t;
}
// a = {u00, v00, u01, v01}, b = {u10, v10, u11, v11}: three fused lerps on (u, v)
__device__ __forceinline__ d2 lerp_fast64(d4 a, d4 b, double tx, double ty) {
#pragma clang fp contract(off)
    d2 r0, r1, r;
    r0.x = fma(tx, a.z - a.x, a.x);
    r0.y = fma(tx, a.w - a.y, a.y);
    r1.x = fma(tx, b.z - b.x, b.x);
    r1.y = fma(tx, b.w - b.y, b.y);
    r.x = fma(ty, r1.x - r0.x, r0.x);
    r.y = fma(ty, r1.y - r0.y, r0.y);
    return r;
}
__device__ __forceinline__ d2 sample_fast64(const double *__restrict__ lvl, const AdvectArgs<double> &A, double x, double y) {
    const Loc64 t = locate_fast64(A, x, y);
    const double *p = lvl + ((size_t)(t.y0 + LC_PAD_LO) * A.pitch + (t.x0 + LC_PAD_LO)) * 2;
    d4 a, b;
    __builtin_memcpy(&a, p, 32);                           // {u00, v00, u01, v01}
    __builtin_memcpy(&b, p + (size_t)A.pitch * 2, 32);     // {u10, v10, u11, v11}
    return lerp_fast64(a, b, t.tx, t.ty);
}

// The same sample with the RAW planes of the level as the source (up = its u plane; lc_advect_ex): the window's rows are
// two 16-byte loads per plane instead of one 32-byte load of interleaved nodes.  The image's pad node behind the last
// node holds node n - 2 (mirror); the pair is loaded one node to the left there and swapped -- the same eight numbers.
__device__ __forceinline__ void window_fast64_raw(const double *__restrict__ up, const AdvectArgs<double> &A, const Loc64 &t, d4 &a, d4 &b) {
    const double *vp = up + (A.v_raw - A.u_raw);
    const int xa = min(t.x0, A.nx_f - 2), y1 = t.y0 + 1 < A.ny_f ? t.y0 + 1 : A.ny_f - 2;
    const size_t r0 = (size_t)t.y0 * A.nx_f + xa, r1 = (size_t)y1 * A.nx_f + xa;
    d2 u0, v0, u1, v1;
    __builtin_memcpy(&u0, up + r0, 16);
    __builtin_memcpy(&v0, vp + r0, 16);
    __builtin_memcpy(&u1, up + r1, 16);
    __builtin_memcpy(&v1, vp + r1, 16);
    const bool last = t.x0 > xa;  // x0 == nx_f - 1: window = {node n - 1, node n - 2}
    a = last ? (d4){u0.y, v0.y, u0.x, v0.x} : (d4){u0.x, v0.x, u0.y, v0.y};
    b = last ? (d4){u1.y, v1.y, u1.x, v1.x} : (d4){u1.x, v1.x, u1.y, v1.y};
}
__device__ __forceinline__ d2 sample_fast64_raw(const double *__restrict__ up, const AdvectArgs<double> &A, double x, double y) {
    const Loc64 t = locate_fast64(A, x, y);
    d4 a, b;
    window_fast64_raw(up, A, t, a, b);
    return lerp_fast64(a, b, t.tx, t.ty);
}

// ... and one sample of the fused-level field 2 F[t] - F[t+1] formed from the raw planes of levels t (up) and t + 1 node by
// node: the numbers the ext image holds (lc_field_pack: T(2) * a - b, one rounding: 2 a is exact), mirrored neighbour
// included, so the result is the ext-image sample bit for bit -- without the image (lc_advect_args.fuse_levels_raw).
__device__ __forceinline__ d2 sample_ext_fast64_raw(const double *__restrict__ up, const AdvectArgs<double> &A, double x, double y) {
#pragma clang fp contract(off)
    const Loc64 t = locate_fast64(A, x, y);
    const double *vp = up + (A.v_raw - A.u_raw);
    const int xa = min(t.x0, A.nx_f - 2), y1 = t.y0 + 1 < A.ny_f ? t.y0 + 1 : A.ny_f - 2;
    const size_t r0 = (size_t)t.y0 * A.nx_f + xa, r1 = (size_t)y1 * A.nx_f + xa, lp = A.raw_plane;
    d2 u0, v0, u1, v1, u0n, v0n, u1n, v1n;
    __builtin_memcpy(&u0, up + r0, 16);
    __builtin_memcpy(&v0, vp + r0, 16);
    __builtin_memcpy(&u1, up + r1, 16);
    __builtin_memcpy(&v1, vp + r1, 16);
    __builtin_memcpy(&u0n, up + lp + r0, 16);
    __builtin_memcpy(&v0n, vp + lp + r0, 16);
    __builtin_memcpy(&u1n, up + lp + r1, 16);
    __builtin_memcpy(&v1n, vp + lp + r1, 16);
    u0 = 2.0 * u0 - u0n;
    v0 = 2.0 * v0 - v0n;
    u1 = 2.0 * u1 - u1n;
    v1 = 2.0 * v1 - v1n;
    const bool last = t.x0 > xa;
    const d4 a = last ? (d4){u0.y, v0.y, u0.x, v0.x} : (d4){u0.x, v0.x, u0.y, v0.y};
    const d4 b = last ? (d4){u1.y, v1.y, u1.x, v1.x} : (d4){u1.x, v1.x, u1.y, v1.y};
    return lerp_fast64(a, b, t.tx, t.ty);
}

// Order 3 in the fast float64 form: scipy's cubic B-spline weights as polynomials in t (cubic_weights_p in double) and the
// 16 taps as four fused row sums combined by a fused column sum.  Shared by the direct and the LDS-tile kernel (explicit
// operations: a seed's result must not depend on which of the two served it).  `w` points at the window's first node
// (padded (y0, x0): one node up / left of the cell), `rs` is the row stride in nodes.
__device__ __forceinline__ void cubic_weights_fast64(double t, double (&w)[4]) {
#pragma clang fp contract(off)
    const double tt = t * t;
    w[3] = tt * (t * (1.0 / 6.0));
    w[0] = fma(tt, 0.5, fma(t, -0.5, 1.0 / 6.0)) - w[3];
    w[1] = fma(tt, fma(t, 0.5, -1.0), 2.0 / 3.0);
    w[2] = fma(tt, fma(t, -0.5, 0.5), fma(t, 0.5, 1.0 / 6.0));
}
// start + the 16-tap sum of a 4 x 4 window already in registers.  ONE function for the LDS window and for the
// global-memory window: a lane's result must not depend on which of the two served it.
__device__ __forceinline__ d2 cubic_apply_fast64(const d2 (&q)[4][4], const double (&wx)[4], const double (&wy)[4], d2 start) {
#pragma clang fp contract(off)
    d2 acc = start;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        const double rx = fma(wx[3], q[a][3].x, fma(wx[2], q[a][2].x, fma(wx[1], q[a][1].x, wx[0] * q[a][0].x)));
        const double ry = fma(wx[3], q[a][3].y, fma(wx[2], q[a][2].y, fma(wx[1], q[a][1].y, wx[0] * q[a][0].y)));
        acc.x = fma(wy[a], rx, acc.x);
        acc.y = fma(wy[a], ry, acc.y);
    }
    return acc;
}
// the window from global memory (row stride rs nodes) ...
__device__ __forceinline__ d2 cubic_taps_fast64(const d2 *w, size_t rs, const double (&wx)[4], const double (&wy)[4], d2 start) {
    d2 q[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) q[a][b] = w[(size_t)a * rs + b];
    return cubic_apply_fast64(q, wx, wy, start);
}
// ... the same window of the fused-level coefficients 2 c[t] - c[t+1] formed node by node from the coefficient images of
// levels t (w) and t + 1 (wn): the numbers the ext image holds (lc_field_pack: T(2) * a - b, one rounding since 2 a is
// exact), so the sum is the ext-image sample bit for bit -- without the image (AdvectArgs::ext_cub).  Row by row, in
// cubic_apply_fast64's operation order (32 loads: eight in flight at a time keep the registers of the kernel's cap).
__device__ __forceinline__ d2 cubic_taps_fast64_fused(const d2 *w, const d2 *wn, size_t rs, const double (&wx)[4], const double (&wy)[4], d2 start) {
#pragma clang fp contract(off)
    d2 acc = start;
#pragma unroll
    for (int a = 0; a < 4; ++a) {
        d2 q[4];
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            const d2 c0 = w[(size_t)a * rs + b], c1 = wn[(size_t)a * rs + b];
            q[b] = (d2){2.0 * c0.x - c1.x, 2.0 * c0.y - c1.y};
        }
        const double rx = fma(wx[3], q[3].x, fma(wx[2], q[2].x, fma(wx[1], q[1].x, wx[0] * q[0].x)));
        const double ry = fma(wx[3], q[3].y, fma(wx[2], q[2].y, fma(wx[1], q[1].y, wx[0] * q[0].y)));
        acc.x = fma(wy[a], rx, acc.x);
        acc.y = fma(wy[a], ry, acc.y);
    }
    return acc;
}
// ... and from an LDS tile (byte address of the window origin; PITCH nodes per row): ds_read_b128 with immediate
// offsets.  (Through one generic pointer for both, hipcc 7.2 merged the tails of the two branches and read the
// window's last row with flat_load_dwordx4 -- LDS through the flat path.)
template <int PITCH>
__device__ __forceinline__ d2 cubic_taps_lds64(unsigned lds_addr, const double (&wx)[4], const double (&wy)[4], d2 start) {
    typedef __attribute__((address_space(3))) const d2 lds_d2;
    lds_d2 *w = (lds_d2 *)(size_t)lds_addr;
    d2 q[4][4];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) q[a][b] = w[a * PITCH + b];
    // (all 16 reads forced in flight before the first use -- sched_barrier -- measures the same: 6.49 against 6.45 ms)
    return cubic_apply_fast64(q, wx, wy, start);
}
__device__ __forceinline__ d2 sample_fast64_o3(const double *__restrict__ lvl, const AdvectArgs<double> &A, double x, double y, d2 start) {
    const Loc64 t = locate_fast64(A, x, y);
    double wx[4], wy[4];
    cubic_weights_fast64(t.tx, wx);
    cubic_weights_fast64(t.ty, wy);
    return cubic_taps_fast64((const d2 *)lvl + ((size_t)t.y0 * A.pitch + t.x0), (size_t)A.pitch, wx, wy, start);
}

// one sample of the fused-level coefficients formed from img[t], img[t+1] (lvl = img[t]; AdvectArgs::ext_cub)
__device__ __forceinline__ d2 sample_fast64_o3_fused(const double *__restrict__ lvl, const AdvectArgs<double> &A, double x, double y, d2 start) {
    const Loc64 t = locate_fast64(A, x, y);
    double wx[4], wy[4];
    cubic_weights_fast64(t.tx, wx);
    cubic_weights_fast64(t.ty, wy);
    const d2 *w = (const d2 *)lvl + ((size_t)t.y0 * A.pitch + t.x0);
    return cubic_taps_fast64_fused(w, w + A.level_elems / 2, (size_t)A.pitch, wx, wy, start);
}

__device__ void advect_seed_fast64_o3(const AdvectArgs<double> &A, int iy, int ix) {
#pragma clang fp contract(off)
    double x = start_x<double>(A, iy, ix), y = start_y<double>(A, iy, ix);
    const double ys = A.seed_lat[iy];
    const double cx_conv = 180.0 / ((3.141592653589793 * 6371000.0) * fabs(cos((ys * 3.141592653589793) / 180.0)));  // Q5
    const double dtcx = A.dt * cx_conv, hdtcx = A.half_dt * cx_conv;
    const size_t idx = (size_t)iy * A.nx + ix, plane = (size_t)A.ny * A.nx;
    if (A.traj_x && !A.traj_skip0) {
        A.traj_x[idx] = x;
        A.traj_y[idx] = y;
    }
    const double *lvl = A.img + (size_t)A.t0 * A.level_elems;
    const double *elv = A.ext_cub ? nullptr : A.ext + (size_t)A.t0 * A.level_elems;
    const d2 zero = {0.0, 0.0};
    for (int s = 0; s < A.nsteps; ++s) {
        const d2 e = sample_fast64_o3(lvl, A, x, y, zero);   // trajectory.py:82-84
        y = fma(A.dtcy, e.y, y);                             // :86
        x = fma(dtcx, e.x, x);                               // :87
        clamp_position_fast64(A, x, y);                     // :89-97
        for (int k = 0; k < A.K; ++k) {                      // :100
            const d2 d = A.ext_cub ? sample_fast64_o3_fused(lvl, A, x, y, e) : sample_fast64_o3(elv, A, x, y, e);  // e + (2 F[t] - F[t+1])(x, y): :105-112 in one sample
            y = fma(A.hdtcy, d.y, y);
            x = fma(hdtcx, d.x, x);
            clamp_position_fast64(A, x, y);
        }
        if (A.traj_x) {
            A.traj_x[(size_t)(s + 1) * plane + idx] = x;
            A.traj_y[(size_t)(s + 1) * plane + idx] = y;
        }
        lvl += A.level_elems;
        if (elv) elv += A.level_elems;
    }
    A.x_out[idx] = x;
    A.y_out[idx] = y;
}

// SRC: 0 = lin image + ext image; 1 = raw planes for the Euler sample + ext image; 2 = raw planes for both (the fused-level
// value formed node by node: no image at all)
template <int SRC>
__device__ void advect_seed_fast64(const AdvectArgs<double> &A, int iy, int ix) {
#pragma clang fp contract(fast)
    constexpr bool RAW = SRC != SRC_IMAGES, EXTRAW = SRC == SRC_RAW_ALL;
    double x = start_x<double>(A, iy, ix), y = start_y<double>(A, iy, ix);
    const double ys = A.seed_lat[iy];
    const double cx_conv = 180.0 / ((3.141592653589793 * 6371000.0) * fabs(cos((ys * 3.141592653589793) / 180.0)));  // Q5
    const double dtcx = A.dt * cx_conv, hdtcx = A.half_dt * cx_conv;
    const size_t idx = (size_t)iy * A.nx + ix, plane = (size_t)A.ny * A.nx;
    if (A.traj_x && !A.traj_skip0) {
        A.traj_x[idx] = x;
        A.traj_y[idx] = y;
    }
    // the Euler sample's source: the lin image, or the raw planes (then img is not read at all)
    const size_t lstride = RAW ? A.raw_plane : A.level_elems;
    const double *lvl = (RAW ? A.u_raw : A.img) + (size_t)A.t0 * lstride;
    const double *elv = EXTRAW ? nullptr : A.ext + (size_t)A.t0 * A.level_elems;
    for (int s = 0; s < A.nsteps; ++s) {
        const d2 e = RAW ? sample_fast64_raw(lvl, A, x, y) : sample_fast64(lvl, A, x, y);   // trajectory.py:82-84
        y = fma(A.dtcy, e.y, y);                        // :86
        x = fma(dtcx, e.x, x);                          // :87
        clamp_position_fast64(A, x, y);                // :89-97
        for (int k = 0; k < A.K; ++k) {                 // :100
            const d2 d = e + (EXTRAW ? sample_ext_fast64_raw(lvl, A, x, y) : sample_fast64(elv, A, x, y));   // e + (2 F[t] - F[t+1])(x, y): :105-112 in one sample
            y = fma(A.hdtcy, d.y, y);
            x = fma(hdtcx, d.x, x);
            clamp_position_fast64(A, x, y);
        }
        if (A.traj_x) {
            A.traj_x[(size_t)(s + 1) * plane + idx] = x;
            A.traj_y[(size_t)(s + 1) * plane + idx] = y;
        }
        lvl += lstride;
        if (!EXTRAW) elv += A.level_elems;
    }
    A.x_out[idx] = x;
    A.y_out[idx] = y;
}

// ======================================================================================
// float64, order 1, fused levels, per-wave LDS tiles (config 2's size class).
//
// The direct kernel above issues 20 sixteen-byte gathers per wave-level and is bound by the vector L1's handling of them
// (0.79 tag lookups per CU-cycle, VALU issuing in 43 % of the slots: profiles/r03/c2_*).  Here each WAVE stages a
// 16 x 12-node tile of ext[t] (16-byte {u, v} nodes; 16 / 12 / 8 rows measure 3.73 / 3.33 / 3.48 ms on config 2 against
// 4.0 for the direct kernel; 8 waves per SIMD or fewer SGPRs: no gain) with three coalesced loads per lane, anchored like the float
// kernels' tiles, and takes the K iteration samples out of LDS with locate_fast64 / lerp_fast64 -- the same two
// functions as the direct kernel, so results are bit-identical to it; the Euler sample stays a direct gather (img[t]
// is another image) and a lane whose window left the tile re-samples from global memory.  8 x 8 seeds per wave, four
// waves stacked per workgroup, no workgroup barrier.
// ======================================================================================
#ifndef LCS_T64_ROWS
#define LCS_T64_ROWS 12
#endif
#ifndef LCS_LDS64_MINWAVES
#define LCS_LDS64_MINWAVES 1
#endif
#ifndef LCS_LDS64_NUM_SGPR
#define LCS_LDS64_NUM_SGPR 0
#endif
#ifndef LCS_T64_PITCH
#define LCS_T64_PITCH 17
#endif
constexpr int T64_COLS = 16, T64_ROWS = LCS_T64_ROWS, T64_PITCH = LCS_T64_PITCH;  // nodes
// RAW: the Euler sample (and the pole rows) from the raw planes of the level instead of the lin image (lc_advect_ex) -- a
// compile-time variant: as a run-time choice the extra uniform state cost the kernel 9 vector registers and a wave per SIMD.
// SRC = SRC_RAW_ALL: the tile of ext[t] itself is formed while it is staged -- each lane loads its nodes of levels t and
// t + 1 from the raw planes (mirrored where the image has pads) and stores 2 F[t] - F[t+1]: no packed image exists at all.
template <int KFIX, bool CYCLIC, int SRC>
__global__ void __launch_bounds__(BLOCK, LCS_LDS64_MINWAVES)
#if LCS_LDS64_NUM_SGPR > 0
    __attribute__((amdgpu_num_sgpr(LCS_LDS64_NUM_SGPR)))
#endif
    advect_lds64_kernel(const AdvectArgs<double> A0) {
#pragma clang fp contract(off)
    constexpr bool RAW = SRC != SRC_IMAGES, EXTRAW = SRC == SRC_RAW_ALL;
    const AdvectArgs<double> A = for_member(A0);
    const int K = KFIX >= 0 ? KFIX : A.K;
    __shared__ __attribute__((aligned(16))) d2 s_tiles[BLOCK / 64][T64_ROWS * T64_PITCH];
    if (pole_block<double, RAW ? POLE_RAW : POLE_LIN>(A)) return;
    const int tile_id = xcd_tile_id(A);
    if (tile_id >= A.ntiles) return;  // whole block
    const int tyi = tile_id / A.ntx, txi = tile_id - tyi * A.ntx;
    const int ix = txi * TILE_W + (threadIdx.x % TILE_W);
    const int iy = tyi * TILE_H + (threadIdx.x / TILE_W);
    const int lane = threadIdx.x & 63;
    d2 *tile = s_tiles[threadIdx.x >> 6];
    bool live = ix < A.nx && iy < A.ny;
    if (live) {
        const int grow = A.row0 + iy;
        if (grow < A.order || grow >= A.ny_global - A.order) {  // pole rows: generic path (Q3)
            if (!A.pole_blocks) pole_seed<double, RAW ? POLE_RAW : POLE_LIN>(A, iy, ix);
            live = false;
        }
    }
    if (__ballot(live) == 0ull) return;  // whole wave (no workgroup barrier anywhere below)
    const int sx_i = min(ix, A.nx - 1), sy_i = min(iy, A.ny - 1);  // lanes without a seed shadow a neighbour; stores masked
    double x = start_x<double>(A, sy_i, sx_i), y = start_y<double>(A, sy_i, sx_i);
    const double ys = A.seed_lat[sy_i];
    const double cx_conv = 180.0 / ((3.141592653589793 * 6371000.0) * fabs(cos((ys * 3.141592653589793) / 180.0)));  // Q5
    const double dtcx = A.dt * cx_conv, hdtcx = A.half_dt * cx_conv;
    const size_t idx = live ? (size_t)iy * A.nx + ix : 0, plane = (size_t)A.ny * A.nx;
    if (live && A.traj_x && !A.traj_skip0) {
        A.traj_x[idx] = x;
        A.traj_y[idx] = y;
    }
    // the Euler sample's source: the lin image, or the raw planes (lc_advect_ex: then img is not read at all)
    const size_t lstride = RAW ? A.raw_plane : A.level_elems;
    const double *lvl = (RAW ? A.u_raw : A.img) + (size_t)A.t0 * lstride;
    const double *elv = EXTRAW ? nullptr : A.ext + (size_t)A.t0 * A.level_elems;
    const int pad_cols = A.pitch, pad_rows = A.ny_f + LC_PAD;
    constexpr int CENTRE = TILE_W / 2 + TILE_W * 4;  // middle seed of the wave's 8 x 8 patch
    // staging: one node (16 bytes) per lane, 16 lanes per tile row, 4 rows per pass, 4 passes
    const int st_row = lane >> 4, st_col = lane & 15;
    const unsigned st_off = ((unsigned)st_row * (unsigned)pad_cols + (unsigned)st_col) * 16u;
    double dprev_x = 0.0, dprev_y = 0.0;  // previous level's Euler displacement in index space: predicts this level's travel
    const double kpred = 0.5 * (double)(K > 0 ? K - 1 : 0);
    for (int s = 0; s < A.nsteps; ++s) {
        // ---- 1. anchor the tile on the centre lane's predicted travel, issue its loads ----------------------------
        int ox = 0, oy = 0;
        d2 stage[T64_ROWS / 4];
        if (K > 0) {
            const double cax = (x - A.lon_min) * A.sx + dprev_x * (1.0 + kpred), cay = (y - A.lat_min) * A.sy + dprev_y * (1.0 + kpred);
            const int rxm = __builtin_amdgcn_readlane((int)floor(fmin(fmax(cax, -4.0), 1.0e9)), CENTRE);
            const int rym = __builtin_amdgcn_readlane((int)floor(fmin(fmax(cay, -4.0), 1.0e9)), CENTRE);
            ox = min(max(rxm + LC_PAD_LO - (T64_COLS - 2) / 2, 0), pad_cols - T64_COLS);
            oy = min(max(rym + LC_PAD_LO - (T64_ROWS - 2) / 2, 0), pad_rows - T64_ROWS);
            if (EXTRAW) {
                // the lane's nodes of levels t and t + 1 from the raw planes; padded (row, column) -> node index, mirrored
                // where the image has pads (lc_field_pack's rule), and 2 F[t] - F[t+1] as the pack forms it
                const ptrdiff_t dv = A.v_raw - A.u_raw;
                int cx = ox + st_col - LC_PAD_LO;
                cx = cx < 0 ? -cx : (cx > A.nx_f - 1 ? 2 * (A.nx_f - 1) - cx : cx);
                double ut[T64_ROWS / 4], vt[T64_ROWS / 4], un[T64_ROWS / 4], vn[T64_ROWS / 4];
#pragma unroll
                for (int r = 0; r < T64_ROWS / 4; ++r) {
                    int cy = oy + r * 4 + st_row - LC_PAD_LO;
                    cy = cy < 0 ? -cy : (cy > A.ny_f - 1 ? 2 * (A.ny_f - 1) - cy : cy);
                    const double *p = lvl + (size_t)cy * A.nx_f + cx;
                    ut[r] = p[0];
                    vt[r] = p[dv];
                    un[r] = p[A.raw_plane];
                    vn[r] = p[A.raw_plane + dv];
                }
#pragma unroll
                for (int r = 0; r < T64_ROWS / 4; ++r) stage[r] = (d2){2.0 * ut[r] - un[r], 2.0 * vt[r] - vn[r]};
            } else {
                const char *src = (const char *)elv + ((size_t)oy * pad_cols + ox) * 16;
#pragma unroll
                for (int r = 0; r < T64_ROWS / 4; ++r) __builtin_memcpy(&stage[r], src + (size_t)(r * 4) * pad_cols * 16 + st_off, 16);
            }
        }
        // ---- 2. Euler sample: direct gather from img[t] (or the raw planes of level t) ---------------------------
        const double x0p = x, y0p = y;
        const d2 e = RAW ? sample_fast64_raw(lvl, A, x, y) : sample_fast64(lvl, A, x, y);   // trajectory.py:82-84
        y = fma(A.dtcy, e.y, y);                        // :86
        x = fma(dtcx, e.x, x);                          // :87
        clamp_position_fast64(A, x, y);                // :89-97
        dprev_x = (x - x0p) * A.sx;
        dprev_y = (y - y0p) * A.sy;
        // ---- 3. tile into LDS ------------------------------------------------------------------------------------
        // window origins (x0, y0) the tile serves: [acc_x, acc_x + acc_w] x [acc_y, acc_y + acc_h] (one unsigned compare per axis);
        // tile-relative index = x0 - lo_x.  No tile: nothing is "inside" (x0 - 2^30 wraps to a huge unsigned number)
        int lo_x = 0, lo_y = 0, acc_x = 0x40000000, acc_y = 0x40000000, acc_w = 0, acc_h = 0;
        if (K > 0) {
            __builtin_amdgcn_wave_barrier();  // the previous level's reads are done (LDS ops of a wave are in order)
#pragma unroll
            for (int r = 0; r < T64_ROWS / 4; ++r) tile[(r * 4 + st_row) * T64_PITCH + st_col] = stage[r];
            __builtin_amdgcn_wave_barrier();
            // inside the tile (padded origin = (y0 + 1, x0 + 1)) and in [0, n - 2]
            const int sox = ox - LC_PAD_LO, soy = oy - LC_PAD_LO;
            const int hx = min(sox + T64_COLS - 2, A.nx_f - 2), hy = min(soy + T64_ROWS - 2, A.ny_f - 2);
            const int lx = max(sox, 0), ly = max(soy, 0);
            if (hx >= lx && hy >= ly) {
                lo_x = sox;
                lo_y = soy;
                acc_x = lx;
                acc_y = ly;
                acc_w = hx - lx;
                acc_h = hy - ly;
            }
        }
        // ---- 4. K iterations out of LDS ---------------------------------------------------------------------------
        for (int k = 0; k < K; ++k) {
            const Loc64 t = locate_fast64(A, x, y);
            const bool in_tile = ((unsigned)(t.x0 - acc_x) <= (unsigned)acc_w) & ((unsigned)(t.y0 - acc_y) <= (unsigned)acc_h);
#ifdef LCS_STAMPS
            {   // g_stamps[4]: wave-samples, [5]: those with a lane outside the tile, [6]: lanes outside
                const unsigned long long m = __ballot(!in_tile);
                if ((threadIdx.x & 63) == 0) {
                    atomicAdd(&g_stamps[4], 1ull);
                    atomicAdd(&g_stamps[5], m ? 1ull : 0ull);
                    atomicAdd(&g_stamps[6], (unsigned long long)__popcll(m));
                }
            }
#endif
            d4 a, b;
            if (in_tile) {
                const d2 *w = tile + (t.y0 - lo_y) * T64_PITCH + (t.x0 - lo_x);
                const d2 n00 = w[0], n01 = w[1], n10 = w[T64_PITCH], n11 = w[T64_PITCH + 1];
                a = (d4){n00.x, n00.y, n01.x, n01.y};
                b = (d4){n10.x, n10.y, n11.x, n11.y};
            } else if (EXTRAW) {  // the window left the tile: the same nodes from the raw planes of levels t, t + 1
                const double *up = lvl, *vp = lvl + (A.v_raw - A.u_raw);
                const int xa = min(t.x0, A.nx_f - 2), y1 = t.y0 + 1 < A.ny_f ? t.y0 + 1 : A.ny_f - 2;
                const size_t r0 = (size_t)t.y0 * A.nx_f + xa, r1 = (size_t)y1 * A.nx_f + xa, lp = A.raw_plane;
                d2 u0, v0, u1, v1, u0n, v0n, u1n, v1n;
                __builtin_memcpy(&u0, up + r0, 16);
                __builtin_memcpy(&v0, vp + r0, 16);
                __builtin_memcpy(&u1, up + r1, 16);
                __builtin_memcpy(&v1, vp + r1, 16);
                __builtin_memcpy(&u0n, up + lp + r0, 16);
                __builtin_memcpy(&v0n, vp + lp + r0, 16);
                __builtin_memcpy(&u1n, up + lp + r1, 16);
                __builtin_memcpy(&v1n, vp + lp + r1, 16);
                u0 = 2.0 * u0 - u0n;
                v0 = 2.0 * v0 - v0n;
                u1 = 2.0 * u1 - u1n;
                v1 = 2.0 * v1 - v1n;
                const bool last = t.x0 > xa;
                a = last ? (d4){u0.y, v0.y, u0.x, v0.x} : (d4){u0.x, v0.x, u0.y, v0.y};
                b = last ? (d4){u1.y, v1.y, u1.x, v1.x} : (d4){u1.x, v1.x, u1.y, v1.y};
            } else {  // the window left the tile: the same taps from global memory
                const double *p = elv + ((size_t)(t.y0 + LC_PAD_LO) * A.pitch + (t.x0 + LC_PAD_LO)) * 2;
                __builtin_memcpy(&a, p, 32);
                __builtin_memcpy(&b, p + (size_t)A.pitch * 2, 32);
            }
            const d2 d = e + lerp_fast64(a, b, t.tx, t.ty);   // e + (2 F[t] - F[t+1])(x, y)
            y = fma(A.hdtcy, d.y, y);
            x = fma(hdtcx, d.x, x);
            clamp_position_fast64(A, x, y);
        }
        if (live && A.traj_x) {
            A.traj_x[(size_t)(s + 1) * plane + idx] = x;
            A.traj_y[(size_t)(s + 1) * plane + idx] = y;
        }
        lvl += lstride;
        if (!EXTRAW) elv += A.level_elems;
    }
    if (live) {
        A.x_out[idx] = x;
        A.y_out[idx] = y;
    }
}

// float64, order 1, fused levels, ONE tile per WORKGROUP (round 6: the structural attempt the round-5 review allowed one of;
// LCS_F64_WG_TILE=1 at context creation, off by default -- profiles/r06/c2_wg_tile_ab.txt says why).
//
// With seeds = nodes (config 2) the per-wave tile above stages 192 nodes for 64 seeds: 3 nodes per seed, 1.67 x the compulsory
// bytes from HBM.  Here the four waves sit 2 x 2 (16 x 16 seeds per workgroup) and share one W64_COLS x W64_ROWS-node tile
// of ext[t] (24 x 20 = 480 nodes: 1.9 per seed, the same +-3.5 / +-1.5 cells of margin around the patch), two tiles
// alternating by level so that ONE workgroup barrier per level is enough (the tile of level s is written again at level
// s + 2, which a wave reaches only through the barrier of level s + 1, i.e. after every wave's reads of level s: the scheme
// of PATCH_LINES' slabs).  The anchor must be the workgroup's: the centre seed's thread predicts it ONE LEVEL AHEAD -- from
// its position after the Euler step of level s it writes level s + 1's anchor before the barrier of level s (the K iterations
// that follow move the parcel by about K Euler displacements: trajectory.py:100-120 accumulates) -- so the anchor costs no
// second barrier.  Same locate / lerp / clamp functions, same fallback for a window outside the tile: bit-identical to
// advect_lds64_kernel and to the direct kernel whatever the tile holds.
#ifndef LCS_W64_COLS
#define LCS_W64_COLS 24
#endif
#ifndef LCS_W64_ROWS
#define LCS_W64_ROWS 20
#endif
#ifndef LCS_W64_PITCH
#define LCS_W64_PITCH (LCS_W64_COLS + 1)
#endif
#ifndef LCS_W64_NEXT
#define LCS_W64_NEXT 1   // raw planes only: keep the level-(t + 1) nodes of the staging as the next level's Euler tile
#endif
#ifndef LCS_W64_MINWAVES
#define LCS_W64_MINWAVES 1
#endif
constexpr int W64_COLS = LCS_W64_COLS, W64_ROWS = LCS_W64_ROWS, W64_PITCH = LCS_W64_PITCH, W64_SIDE = 16;  // nodes; seeds per side
template <int KFIX, bool CYCLIC, int SRC>
__global__ void __launch_bounds__(BLOCK, LCS_W64_MINWAVES) advect_wg64_kernel(const AdvectArgs<double> A0) {
#pragma clang fp contract(off)
    constexpr bool RAW = SRC != SRC_IMAGES, EXTRAW = SRC == SRC_RAW_ALL, NEXT = EXTRAW && LCS_W64_NEXT != 0;
    constexpr int NODES = W64_COLS * W64_ROWS;
    const AdvectArgs<double> A = for_member(A0);
    const int K = KFIX >= 0 ? KFIX : A.K;
    __shared__ __attribute__((aligned(16))) d2 s_tile[2][W64_ROWS * W64_PITCH];
    // raw planes only: the nodes of level t + 1 the staging loads anyway (ext = 2 F[t] - F[t+1]) are kept as a tile of their own --
    // it is the NEXT level's Euler sample's field around where the parcels will be, so that sample is four LDS reads, not four gathers
    __shared__ __attribute__((aligned(16))) d2 s_next[NEXT ? 2 : 1][NEXT ? W64_ROWS * W64_PITCH : 1];
    __shared__ int s_anchor[2][2];
    if (pole_block<double, RAW ? POLE_RAW : POLE_LIN>(A)) return;
    const int tile_id = xcd_tile_id(A);
    if (tile_id >= A.ntiles) return;  // whole block
    const int tyi = tile_id / A.ntx, txi = tile_id - tyi * A.ntx;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ix = txi * W64_SIDE + (wave & 1) * 8 + (lane & 7);
    const int iy = tyi * W64_SIDE + (wave >> 1) * 8 + (lane >> 3);
    bool live = ix < A.nx && iy < A.ny;
    if (live) {
        const int grow = A.row0 + iy;
        if (grow < A.order || grow >= A.ny_global - A.order) {  // pole rows: generic path (Q3)
            if (!A.pole_blocks) pole_seed<double, RAW ? POLE_RAW : POLE_LIN>(A, iy, ix);
            live = false;
        }
    }
    if (!__syncthreads_or(live)) return;  // whole WORKGROUP (every wave meets the level barriers below, seeds or none)
    const int sx_i = min(ix, A.nx - 1), sy_i = min(iy, A.ny - 1);  // threads without a seed shadow a neighbour; stores masked
    double x = start_x<double>(A, sy_i, sx_i), y = start_y<double>(A, sy_i, sx_i);
    const double ys = A.seed_lat[sy_i];
    const double cx_conv = 180.0 / ((3.141592653589793 * 6371000.0) * fabs(cos((ys * 3.141592653589793) / 180.0)));  // Q5
    const double dtcx = A.dt * cx_conv, hdtcx = A.half_dt * cx_conv;
    const size_t idx = live ? (size_t)iy * A.nx + ix : 0, plane = (size_t)A.ny * A.nx;
    if (live && A.traj_x && !A.traj_skip0) {
        A.traj_x[idx] = x;
        A.traj_y[idx] = y;
    }
    const size_t lstride = RAW ? A.raw_plane : A.level_elems;
    const double *lvl = (RAW ? A.u_raw : A.img) + (size_t)A.t0 * lstride;
    const double *elv = EXTRAW ? nullptr : A.ext + (size_t)A.t0 * A.level_elems;
    const int pad_cols = A.pitch, pad_rows = A.ny_f + LC_PAD;
    // staging: thread t holds nodes t, t + 256, ... of the tile in row-major order (raw planes: the node PAIRS 2 t, 2 t + 1, ...:
    // two neighbours of a plane are one 16-byte load)
    static_assert(W64_COLS % 2 == 0, "a node pair lies in one tile row");
    constexpr int PER = EXTRAW ? 2 : 1, NSLOT = (NODES / PER + BLOCK - 1) / BLOCK;
    int st_row[NSLOT], st_col[NSLOT];
    bool st_ok[NSLOT];
#pragma unroll
    for (int r = 0; r < NSLOT; ++r) {
        const int n = ((int)threadIdx.x + r * BLOCK) * PER;
        st_ok[r] = n < NODES;
        st_row[r] = min(n, NODES - PER) / W64_COLS;
        st_col[r] = min(n, NODES - PER) % W64_COLS;
    }
    int prev_ox = 0, prev_oy = 0;  // ... and its origin (padded)
    bool has_prev = false;
    int plo_x = 0, plo_y = 0, pacc_x = 0x40000000, pacc_y = 0x40000000, pacc_w = 0, pacc_h = 0;  // the previous level's tile window
    const bool anchor_thread = threadIdx.x == 3 * 64;  // seed (8, 8) of the 16 x 16 patch: wave 3's first lane
    const double kpred = 0.5 * (double)(K > 0 ? K - 1 : 0);
    auto anchor_of = [&](double cax, double cay, int slot) {
        s_anchor[slot][0] = (int)floor(fmin(fmax(cax, -4.0), 1.0e9));
        s_anchor[slot][1] = (int)floor(fmin(fmax(cay, -4.0), 1.0e9));
    };
    if (anchor_thread) anchor_of((x - A.lon_min) * A.sx, (y - A.lat_min) * A.sy, 0);
    __syncthreads();
    for (int s = 0; s < A.nsteps; ++s) {
        // ---- 1. the workgroup's anchor for this level (written one level ago), the tile's loads ---------------------
        int ox = 0, oy = 0;
        d2 stage[NSLOT * PER], keep[NEXT ? NSLOT * PER : 1];
        d2 *tile = s_tile[s & 1];
        if (K > 0) {
            const int rxm = __builtin_amdgcn_readfirstlane(s_anchor[s & 1][0]);
            const int rym = __builtin_amdgcn_readfirstlane(s_anchor[s & 1][1]);
            // the anchor seed is number 8 of 16 per side: its window's origin goes to the tile's middle column / row
            ox = min(max(rxm + LC_PAD_LO - W64_COLS / 2, 0), pad_cols - W64_COLS);
            oy = min(max(rym + LC_PAD_LO - W64_ROWS / 2, 0), pad_rows - W64_ROWS);
            if constexpr (EXTRAW) {
                // the thread's node pair of levels t and t + 1 from the raw planes: padded (row, column) -> node index, mirrored
                // where the image has pads (lc_field_pack's rule); 2 F[t] - F[t+1] as the pack forms it; F[t+1] kept
                const ptrdiff_t dv = A.v_raw - A.u_raw;
#pragma unroll
                for (int r = 0; r < NSLOT; ++r) {
                    int ca = ox + st_col[r] - LC_PAD_LO, cb = ca + 1, cy = oy + st_row[r] - LC_PAD_LO;
                    ca = ca < 0 ? -ca : (ca > A.nx_f - 1 ? 2 * (A.nx_f - 1) - ca : ca);
                    cb = cb < 0 ? -cb : (cb > A.nx_f - 1 ? 2 * (A.nx_f - 1) - cb : cb);
                    cy = cy < 0 ? -cy : (cy > A.ny_f - 1 ? 2 * (A.ny_f - 1) - cy : cy);
                    const double *row = lvl + (size_t)cy * A.nx_f;
                    d2 ut, vt, un, vn;
                    const bool whole = cb == ca + 1;  // (all but the pairs that straddle a mirrored edge)
                    if (whole) {
                        __builtin_memcpy(&un, row + A.raw_plane + ca, 16);
                        __builtin_memcpy(&vn, row + A.raw_plane + dv + ca, 16);
                    } else {
                        un = (d2){row[A.raw_plane + ca], row[A.raw_plane + cb]};
                        vn = (d2){row[A.raw_plane + dv + ca], row[A.raw_plane + dv + cb]};
                    }
                    // level t's nodes: the previous level kept them (as ITS level t + 1) where the two tiles overlap -- same padded
                    // coordinates, same numbers; only the fringe the anchor moved onto comes from memory
                    const int pr = st_row[r] + (oy - prev_oy), pc = st_col[r] + (ox - prev_ox);
                    if (NEXT && has_prev && (unsigned)pr < (unsigned)W64_ROWS && pc >= 0 && pc + 1 < W64_COLS) {
                        const d2 *w = s_next[(s + 1) & 1] + pr * W64_PITCH + pc;
                        const d2 n0 = w[0], n1 = w[1];
                        ut = (d2){n0.x, n1.x};
                        vt = (d2){n0.y, n1.y};
                    } else if (whole) {
                        __builtin_memcpy(&ut, row + ca, 16);
                        __builtin_memcpy(&vt, row + dv + ca, 16);
                    } else {
                        ut = (d2){row[ca], row[cb]};
                        vt = (d2){row[dv + ca], row[dv + cb]};
                    }
                    stage[2 * r] = (d2){2.0 * ut.x - un.x, 2.0 * vt.x - vn.x};
                    stage[2 * r + 1] = (d2){2.0 * ut.y - un.y, 2.0 * vt.y - vn.y};
                    if constexpr (NEXT) {
                        keep[2 * r] = (d2){un.x, vn.x};
                        keep[2 * r + 1] = (d2){un.y, vn.y};
                    }
                }
            } else {
                const char *src = (const char *)elv + ((size_t)oy * pad_cols + ox) * 16;
#pragma unroll
                for (int r = 0; r < NSLOT; ++r)
                    __builtin_memcpy(&stage[r], src + ((size_t)st_row[r] * pad_cols + st_col[r]) * 16, 16);
            }
        }
        // ---- 2. Euler sample: direct gather; raw planes only: out of the tile of this level's nodes the previous level kept ----
        const double x0p = x, y0p = y;
        d2 e;
        if constexpr (NEXT) {
            const Loc64 t = locate_fast64(A, x, y);
            const bool in_prev = ((unsigned)(t.x0 - pacc_x) <= (unsigned)pacc_w) & ((unsigned)(t.y0 - pacc_y) <= (unsigned)pacc_h);
#ifdef LCS_STAMPS
            {   // g_cause[0]: Euler wave-samples, [1]: those with a lane outside the kept tile, [2]: lanes outside
                const unsigned long long m = __ballot(!in_prev);
                if (lane == 0) {
                    atomicAdd(&g_cause[0], 1ull);
                    atomicAdd(&g_cause[1], m ? 1ull : 0ull);
                    atomicAdd(&g_cause[2], (unsigned long long)__popcll(m));
                }
            }
#endif
            d4 a, b;
            if (in_prev) {
                const d2 *w = s_next[(s + 1) & 1] + (t.y0 - plo_y) * W64_PITCH + (t.x0 - plo_x);
                const d2 n00 = w[0], n01 = w[1], n10 = w[W64_PITCH], n11 = w[W64_PITCH + 1];
                a = (d4){n00.x, n00.y, n01.x, n01.y};
                b = (d4){n10.x, n10.y, n11.x, n11.y};
            } else {
                window_fast64_raw(lvl, A, t, a, b);
            }
            e = lerp_fast64(a, b, t.tx, t.ty);
        } else {
            e = RAW ? sample_fast64_raw(lvl, A, x, y) : sample_fast64(lvl, A, x, y);   // trajectory.py:82-84
        }
        y = fma(A.dtcy, e.y, y);                        // :86
        x = fma(dtcx, e.x, x);                          // :87
        clamp_position_fast64(A, x, y);                // :89-97
        // ---- 3. tile into LDS, the next level's anchor, the level's one barrier ---------------------------------------
        int lo_x = 0, lo_y = 0, acc_x = 0x40000000, acc_y = 0x40000000, acc_w = 0, acc_h = 0;
        if (K > 0) {
#pragma unroll
            for (int r = 0; r < NSLOT; ++r)
                if (st_ok[r]) {
#pragma unroll
                    for (int q = 0; q < PER; ++q) {
                        tile[st_row[r] * W64_PITCH + st_col[r] + q] = stage[PER * r + q];
                        if constexpr (NEXT) s_next[s & 1][st_row[r] * W64_PITCH + st_col[r] + q] = keep[PER * r + q];
                    }
                }
            if (anchor_thread) {
                // level s + 1 starts about K Euler displacements further on; its samples centre (1 + kpred) beyond that
                const double ahead = (double)K + 1.0 + kpred;
                anchor_of((x - A.lon_min) * A.sx + (x - x0p) * A.sx * ahead, (y - A.lat_min) * A.sy + (y - y0p) * A.sy * ahead, (s + 1) & 1);
            }
            asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
            const int sox = ox - LC_PAD_LO, soy = oy - LC_PAD_LO;
            const int hx = min(sox + W64_COLS - 2, A.nx_f - 2), hy = min(soy + W64_ROWS - 2, A.ny_f - 2);
            const int lx = max(sox, 0), ly = max(soy, 0);
            if (hx >= lx && hy >= ly) {
                lo_x = sox;
                lo_y = soy;
                acc_x = lx;
                acc_y = ly;
                acc_w = hx - lx;
                acc_h = hy - ly;
            }
        }
        // ---- 4. K iterations out of LDS ---------------------------------------------------------------------------
        for (int k = 0; k < K; ++k) {
            const Loc64 t = locate_fast64(A, x, y);
            const bool in_tile = ((unsigned)(t.x0 - acc_x) <= (unsigned)acc_w) & ((unsigned)(t.y0 - acc_y) <= (unsigned)acc_h);
#ifdef LCS_STAMPS
            {   // g_stamps[4]: wave-samples, [5]: those with a lane outside the tile, [6]: lanes outside
                const unsigned long long m = __ballot(!in_tile);
                if ((threadIdx.x & 63) == 0) {
                    atomicAdd(&g_stamps[4], 1ull);
                    atomicAdd(&g_stamps[5], m ? 1ull : 0ull);
                    atomicAdd(&g_stamps[6], (unsigned long long)__popcll(m));
                }
            }
#endif
            d4 a, b;
            if (in_tile) {
                const d2 *w = tile + (t.y0 - lo_y) * W64_PITCH + (t.x0 - lo_x);
                const d2 n00 = w[0], n01 = w[1], n10 = w[W64_PITCH], n11 = w[W64_PITCH + 1];
                a = (d4){n00.x, n00.y, n01.x, n01.y};
                b = (d4){n10.x, n10.y, n11.x, n11.y};
            } else if (EXTRAW) {  // the window left the tile: the same nodes from the raw planes of levels t, t + 1
                const double *up = lvl, *vp = lvl + (A.v_raw - A.u_raw);
                const int xa = min(t.x0, A.nx_f - 2), y1 = t.y0 + 1 < A.ny_f ? t.y0 + 1 : A.ny_f - 2;
                const size_t r0 = (size_t)t.y0 * A.nx_f + xa, r1 = (size_t)y1 * A.nx_f + xa, lp = A.raw_plane;
                d2 u0, v0, u1, v1, u0n, v0n, u1n, v1n;
                __builtin_memcpy(&u0, up + r0, 16);
                __builtin_memcpy(&v0, vp + r0, 16);
                __builtin_memcpy(&u1, up + r1, 16);
                __builtin_memcpy(&v1, vp + r1, 16);
                __builtin_memcpy(&u0n, up + lp + r0, 16);
                __builtin_memcpy(&v0n, vp + lp + r0, 16);
                __builtin_memcpy(&u1n, up + lp + r1, 16);
                __builtin_memcpy(&v1n, vp + lp + r1, 16);
                u0 = 2.0 * u0 - u0n;
                v0 = 2.0 * v0 - v0n;
                u1 = 2.0 * u1 - u1n;
                v1 = 2.0 * v1 - v1n;
                const bool last = t.x0 > xa;
                a = last ? (d4){u0.y, v0.y, u0.x, v0.x} : (d4){u0.x, v0.x, u0.y, v0.y};
                b = last ? (d4){u1.y, v1.y, u1.x, v1.x} : (d4){u1.x, v1.x, u1.y, v1.y};
            } else {  // the window left the tile: the same taps from global memory
                const double *p = elv + ((size_t)(t.y0 + LC_PAD_LO) * A.pitch + (t.x0 + LC_PAD_LO)) * 2;
                __builtin_memcpy(&a, p, 32);
                __builtin_memcpy(&b, p + (size_t)A.pitch * 2, 32);
            }
            const d2 d = e + lerp_fast64(a, b, t.tx, t.ty);   // e + (2 F[t] - F[t+1])(x, y)
            y = fma(A.hdtcy, d.y, y);
            x = fma(hdtcx, d.x, x);
            clamp_position_fast64(A, x, y);
        }
        prev_ox = ox, prev_oy = oy, has_prev = true;
        plo_x = lo_x, plo_y = lo_y, pacc_x = acc_x, pacc_y = acc_y, pacc_w = acc_w, pacc_h = acc_h;
        if (live && A.traj_x) {
            A.traj_x[(size_t)(s + 1) * plane + idx] = x;
            A.traj_y[(size_t)(s + 1) * plane + idx] = y;
        }
        lvl += lstride;
        if (!EXTRAW) elv += A.level_elems;
    }
    if (live) {
        A.x_out[idx] = x;
        A.y_out[idx] = y;
    }
}

// float64, ORDER 3 (the reference's default interpolation on the reference's default dtype), fused levels, per-wave LDS
// tiles: a 16 x 16-node tile of img[t] around the patch's current position serves the Euler sample, one of ext[t] anchored on
// the predicted travel the K iterations; 4 x 4 windows read row by row (cubic_taps_fast64, the function the direct kernel
// uses: bit-identical); lanes whose window left a tile take the same taps from global memory.  8 x 8 seeds per wave.
// Tile shapes (nodes).  Iteration tile of ext[t]: 16 x 16 at pitch 24 -- 384-byte rows put the four patch rows of a
// ds_read_b128 lane group into different bank quarters (conflicts 161 M -> 96 M per launch on config 2, the wave's LDS wait
// halved).  Euler tile of img[t]: 12 rows (the patch where it is: 8 rows + the window) at pitch 20, so that both fit 39 KB
// per workgroup = four workgroups per CU.  Measured on config 2 at order 3 (profiles/r04/c2_o3_lds_pitch_ab.txt): pitch 20
// for both 16-row tiles 6.39 ms, this 6.24-6.33; pitch 24 for both 16-row tiles (48 KB, three workgroups per CU) 6.79
// although each wave runs 18 % shorter; round 2-3: pitch 17 8.46, 20 7.90, 24 (48 KB) 8.25.
#ifndef LCS_T64O3_PITCH
#define LCS_T64O3_PITCH 24
#endif
#ifndef LCS_T64O3_EROWS
#define LCS_T64O3_EROWS 12   // rows of the Euler tile (img[t]: the patch where it is, 8 rows + the window = 11 at least)
#endif
#ifndef LCS_T64O3_EPITCH
#define LCS_T64O3_EPITCH 20
#endif
constexpr int T64O3 = 16, T64O3_PITCH = LCS_T64O3_PITCH, T64O3_EROWS = LCS_T64O3_EROWS, T64O3_EPITCH = LCS_T64O3_EPITCH;
static_assert(T64O3_EROWS % 4 == 0 && T64O3_EROWS >= 12 && T64O3_EROWS <= T64O3, "Euler tile rows");
#ifndef LCS_LDS64_O3_MINWAVES
#define LCS_LDS64_O3_MINWAVES 4   // 128 vector registers: the kernel compiles to 126-129 depending on the tile shapes, and 129 is a wave per SIMD less
#endif
// EXTCUB: no ext image -- the iteration tile is staged from the coefficient images of levels t and t + 1 as
// 2 img[t] - img[t+1] node by node (the pack's own expression: bit-identical), and a window that left the tile takes its
// taps from both levels.  The pack then neither reads the coefficients back nor writes a second image (float64 config 2:
// 8.5 GB of 22 per step), and the advect kernel streams ONE image series from HBM instead of two.
template <int KFIX, bool CYCLIC, bool EXTCUB = false>
__global__ void __launch_bounds__(BLOCK, LCS_LDS64_O3_MINWAVES) advect_lds64_o3_kernel(const AdvectArgs<double> A0) {
#pragma clang fp contract(off)
    const AdvectArgs<double> A = for_member(A0);
    const int K = KFIX >= 0 ? KFIX : A.K;
    __shared__ __attribute__((aligned(16))) d2 s_tiles[BLOCK / 64][T64O3_EROWS * T64O3_EPITCH + T64O3 * T64O3_PITCH];
    if (pole_block(A)) return;
    const int tile_id = xcd_tile_id(A);
    if (tile_id >= A.ntiles) return;  // whole block
    const int tyi = tile_id / A.ntx, txi = tile_id - tyi * A.ntx;
    const int ix = txi * TILE_W + (threadIdx.x % TILE_W);
    const int iy = tyi * TILE_H + (threadIdx.x / TILE_W);
    const int lane = threadIdx.x & 63;
    d2 *etile = s_tiles[threadIdx.x >> 6], *gtile = s_tiles[threadIdx.x >> 6] + T64O3_EROWS * T64O3_EPITCH;
    bool live = ix < A.nx && iy < A.ny;
    if (live) {
        const int grow = A.row0 + iy;
        if (grow < A.order || grow >= A.ny_global - A.order) {  // pole rows: generic path (Q3)
            if (!A.pole_blocks) pole_seed<double, POLE_EITHER>(A, iy, ix);
            live = false;
        }
    }
    if (__ballot(live) == 0ull) return;  // whole wave (no workgroup barrier anywhere below)
    const int sx_i = min(ix, A.nx - 1), sy_i = min(iy, A.ny - 1);  // lanes without a seed shadow a neighbour; stores masked
    double x = start_x<double>(A, sy_i, sx_i), y = start_y<double>(A, sy_i, sx_i);
    const double ys = A.seed_lat[sy_i];
    const double cx_conv = 180.0 / ((3.141592653589793 * 6371000.0) * fabs(cos((ys * 3.141592653589793) / 180.0)));  // Q5
    const double dtcx = A.dt * cx_conv, hdtcx = A.half_dt * cx_conv;
    const size_t idx = live ? (size_t)iy * A.nx + ix : 0, plane = (size_t)A.ny * A.nx;
    if (live && A.traj_x && !A.traj_skip0) {
        A.traj_x[idx] = x;
        A.traj_y[idx] = y;
    }
    const double *lvl = A.img + (size_t)A.t0 * A.level_elems;
    const double *elv = EXTCUB ? lvl : A.ext + (size_t)A.t0 * A.level_elems;  // EXTCUB: "ext[t]" is formed from lvl and lvl + level_elems
    const int pad_cols = A.pitch, pad_rows = A.ny_f + LC_PAD;
    constexpr int CENTRE = TILE_W / 2 + TILE_W * 4;  // middle seed of the wave's 8 x 8 patch
    const int st_row = lane >> 4, st_col = lane & 15;  // staging: one node per lane, 16 lanes per tile row, 4 rows per pass
    const unsigned st_off = ((unsigned)st_row * (unsigned)pad_cols + (unsigned)st_col) * 16u;
    const double kpred = 0.5 * (double)(K > 0 ? K - 1 : 0);
    const d2 zero = {0.0, 0.0};
    // window origins (padded (y0, x0)) a tile with padded origin (oy, ox) serves: [ox, ox + 16 - 4] x [oy, oy + 16 - 4]
    // (ROWS, PITCH: the tile's shape -- the Euler tile may be lower than the iteration tile: it serves the patch where it is)
    auto stage_tile = [&](const double *level, int ox, int oy, d2 *tile, auto rows_c, auto pitch_c, auto fused_c) {
        constexpr int ROWS = decltype(rows_c)::value, PITCH = decltype(pitch_c)::value;
        constexpr bool FUSED2 = decltype(fused_c)::value;  // 2 level[t] - level[t+1] while staging (EXTCUB's iteration tile)
        d2 st[ROWS / 4];
        const char *src = (const char *)level + ((size_t)oy * pad_cols + ox) * 16;
#pragma unroll
        for (int r = 0; r < ROWS / 4; ++r) __builtin_memcpy(&st[r], src + (size_t)(r * 4) * pad_cols * 16 + st_off, 16);
        if constexpr (FUSED2) {
            d2 nx[ROWS / 4];
            const char *srcn = src + A.level_elems * sizeof(double);
#pragma unroll
            for (int r = 0; r < ROWS / 4; ++r) __builtin_memcpy(&nx[r], srcn + (size_t)(r * 4) * pad_cols * 16 + st_off, 16);
#pragma unroll
            for (int r = 0; r < ROWS / 4; ++r) st[r] = (d2){2.0 * st[r].x - nx[r].x, 2.0 * st[r].y - nx[r].y};
        }
        __builtin_amdgcn_wave_barrier();  // the previous level's reads of this tile are done (LDS ops of a wave are in order)
#pragma unroll
        for (int r = 0; r < ROWS / 4; ++r) tile[(r * 4 + st_row) * PITCH + st_col] = st[r];
        __builtin_amdgcn_wave_barrier();
    };
    auto sample = [&](const double *level, const d2 *tile, int ox, int oy, bool have, double px, double py, d2 start, auto rows_c, auto pitch_c, auto fused_c) {
        constexpr int ROWS = decltype(rows_c)::value, PITCH = decltype(pitch_c)::value;
        constexpr bool FUSED2 = decltype(fused_c)::value;
        const Loc64 t = locate_fast64(A, px, py);
        double wx[4], wy[4];
        cubic_weights_fast64(t.tx, wx);
        cubic_weights_fast64(t.ty, wy);
        const int rx = t.x0 - ox, ry = t.y0 - oy;
        if (have && (unsigned)rx <= (unsigned)(T64O3 - 4) && (unsigned)ry <= (unsigned)(ROWS - 4))
            return cubic_taps_lds64<PITCH>(lds_address(tile) + (unsigned)(ry * PITCH + rx) * 16u, wx, wy, start);
        const d2 *w = (const d2 *)level + ((size_t)t.y0 * A.pitch + t.x0);
        if constexpr (FUSED2) return cubic_taps_fast64_fused(w, w + A.level_elems / 2, (size_t)A.pitch, wx, wy, start);
        return cubic_taps_fast64(w, (size_t)A.pitch, wx, wy, start);
    };
    typedef std::integral_constant<bool, false> OneLevel;
    typedef std::integral_constant<bool, EXTCUB> IterLevels;
    typedef std::integral_constant<int, T64O3_EROWS> ERows;
    typedef std::integral_constant<int, T64O3_EPITCH> EPitch;
    typedef std::integral_constant<int, T64O3> GRows;
    typedef std::integral_constant<int, T64O3_PITCH> GPitch;
    for (int s = 0; s < A.nsteps; ++s) {
        // ---- 1. Euler sample out of a tile of img[t] centred on the patch's current position -------------------------
        const double c0x = (x - A.lon_min) * A.sx, c0y = (y - A.lat_min) * A.sy;
        const int exm = __builtin_amdgcn_readlane((int)floor(fmin(fmax(c0x, -4.0), 1.0e9)), CENTRE);
        const int eym = __builtin_amdgcn_readlane((int)floor(fmin(fmax(c0y, -4.0), 1.0e9)), CENTRE);
        const int eox = min(max(exm - (T64O3 - 4) / 2, 0), pad_cols - T64O3), eoy = min(max(eym - (T64O3_EROWS - 4) / 2, 0), pad_rows - T64O3_EROWS);
        stage_tile(lvl, eox, eoy, etile, ERows(), EPitch(), OneLevel());
        const double x0p = x, y0p = y;
        const d2 e = sample(lvl, etile, eox, eoy, true, x, y, zero, ERows(), EPitch(), OneLevel());   // trajectory.py:82-84
        y = fma(A.dtcy, e.y, y);                                        // :86
        x = fma(dtcx, e.x, x);                                          // :87
        clamp_position_fast64(A, x, y);                                // :89-97
        // ---- 2. tile of ext[t] anchored on the travel this level's Euler displacement predicts -----------------------
        int ox = 0, oy = 0;
        if (K > 0) {
            const double cax = c0x + (x - x0p) * A.sx * (1.0 + kpred), cay = c0y + (y - y0p) * A.sy * (1.0 + kpred);
            const int rxm = __builtin_amdgcn_readlane((int)floor(fmin(fmax(cax, -4.0), 1.0e9)), CENTRE);
            const int rym = __builtin_amdgcn_readlane((int)floor(fmin(fmax(cay, -4.0), 1.0e9)), CENTRE);
            ox = min(max(rxm - (T64O3 - 4) / 2, 0), pad_cols - T64O3);
            oy = min(max(rym - (T64O3 - 4) / 2, 0), pad_rows - T64O3);
            stage_tile(elv, ox, oy, gtile, GRows(), GPitch(), IterLevels());
        }
        // ---- 3. K iterations out of LDS ---------------------------------------------------------------------------------
        for (int k = 0; k < K; ++k) {
            const d2 d = sample(elv, gtile, ox, oy, true, x, y, e, GRows(), GPitch(), IterLevels());   // e + (2 F[t] - F[t+1])(x, y): :105-112 in one sample
            y = fma(A.hdtcy, d.y, y);
            x = fma(hdtcx, d.x, x);
            clamp_position_fast64(A, x, y);
        }
        if (live && A.traj_x) {
            A.traj_x[(size_t)(s + 1) * plane + idx] = x;
            A.traj_y[(size_t)(s + 1) * plane + idx] = y;
        }
        lvl += A.level_elems;
        elv += A.level_elems;
    }
    if (live) {
        A.x_out[idx] = x;
        A.y_out[idx] = y;
    }
}

// SRC != SRC_IMAGES (float64 at order 1 only): the order-1 source is the raw planes (lc_advect_ex; img == lin is not read)
template <typename T, int ORDER, bool FUSED, int SRC = SRC_IMAGES>
struct InteriorPath {
    static __device__ __forceinline__ void run(const AdvectArgs<T> &A, int iy, int ix) {
        if constexpr (SRC != SRC_IMAGES)
            advect_seed<T, 1, true, false, true>(A, A.u_raw, iy, ix);   // exact order, two samples per iteration
        else
            advect_seed<T, ORDER, true, FUSED>(A, A.img, iy, ix);
    }
};
template <>
struct InteriorPath<double, 3, true, SRC_IMAGES> {
    static __device__ __forceinline__ void run(const AdvectArgs<double> &A, int iy, int ix) {
        if (A.wind_f32)
            advect_seed<double, 3, true, true>(A, A.img, iy, ix);  // (never launched: LC_F64_WIND_F32 takes no ext)
        else
            advect_seed_fast64_o3(A, iy, ix);
    }
};
template <int SRC>
struct InteriorPath<double, 1, true, SRC> {
    static __device__ __forceinline__ void run(const AdvectArgs<double> &A, int iy, int ix) {
        advect_seed_fast64<SRC>(A, iy, ix);  // (LC_F64_WIND_F32 takes no ext: never launched in this form)
    }
};
template <int ORDER, bool FUSED>
struct InteriorPath<float, ORDER, FUSED, SRC_IMAGES> {
    static __device__ __forceinline__ void run(const AdvectArgs<float> &A, int iy, int ix) {
        if (ORDER == 1 || ORDER == 3)
            advect_seed_f32<ORDER>(A, iy, ix);  // looks at A.ext itself
        else
            advect_seed<float, ORDER, true, false>(A, A.img, iy, ix);
    }
};

// double, order 1 sits at 69 VGPRs (7 waves per SIMD); asking for 8 costs nothing measurable per wave and
// lets BASELINE config 2 (1024^2 seeds = 16 workgroups per CU) run in two full rounds instead of 7 + 7 + 2.
template <typename T, int ORDER, bool FUSED, int SRC = SRC_IMAGES>
__device__ __forceinline__ void advect_kernel_body(const AdvectArgs<T> &A0) {
    const AdvectArgs<T> A = for_member(A0);
    constexpr bool RAW = SRC != SRC_IMAGES;
    static_assert(!RAW || (sizeof(T) == 8 && ORDER == 1), "raw-plane variants exist for float64 at order 1 (the others choose per call)");
    // order-1 source of the pole rows: float64 at order 1 has its own kernel variants for the raw planes (held to 64
    // registers, a run-time choice spills); float32 at order 1 always has the lin image; everything else: per call
    constexpr int PSRC = (sizeof(T) == 8 && ORDER == 1) ? (RAW ? POLE_RAW : POLE_LIN) : ((sizeof(T) == 8 || ORDER != 1) ? POLE_EITHER : POLE_LIN);
    if (pole_block<T, PSRC>(A)) return;
    const int tile = xcd_tile_id(A);  // tile rows dealt to the XCDs cyclically (see xcd_tile_id)
    if (tile >= A.ntiles) return;
    const int tyi = tile / A.ntx, txi = tile - tyi * A.ntx;
    const int ix = txi * TILE_W + (threadIdx.x % TILE_W);
    const int iy = tyi * TILE_H + (threadIdx.x / TILE_W);
    if (ix >= A.nx || iy >= A.ny) return;
    const int grow = A.row0 + iy;
    const bool pole = grow < A.order || grow >= A.ny_global - A.order;  // tools.py:24-33 (Q3)
    if (pole) {
        if (!A.pole_blocks) pole_seed<T, PSRC>(A, iy, ix);  // (else the leading workgroups did them)
    } else
        InteriorPath<T, ORDER, FUSED, SRC>::run(A, iy, ix);
}

// double, order 1 sits at 69 VGPRs (7 waves per SIMD); asking for 8 costs nothing measurable per wave and
// lets BASELINE config 2 (1024^2 seeds = 16 workgroups per CU) run in two full rounds instead of 7 + 7 + 2.
template <typename T, int ORDER, bool FUSED = false, int SRC = SRC_IMAGES>
__global__ void __launch_bounds__(BLOCK, (sizeof(T) == 8 && ORDER == 1) ? 8 : 1) advect_kernel(const AdvectArgs<T> A) {
    advect_kernel_body<T, ORDER, FUSED, SRC>(A);
}

// float: 98 SGPRs as compiled would admit 6 workgroups per CU instead of 7 (MI355X_MICROARCH.md: 97-112 -> 6);
// capped at 96 (K = 0 on C3: 2.5 -> 2.2 ms).  (The attribute takes no template-dependent value: own kernel.)
template <int ORDER>
__global__ void __launch_bounds__(BLOCK) __attribute__((amdgpu_num_sgpr(96))) advect_kernel_f32(const AdvectArgs<float> A) {
    advect_kernel_body<float, ORDER, false>(A);
}
// Orders above 1 without the cap: held to 96 SGPRs the order-3 kernel spills scalars into vector registers (85 VGPRs = 5
// waves per SIMD instead of 65 = 6 with 106 SGPRs); it is the kernel of the reference's DEFAULT arguments on float32 data
// (SETTLS_order=0, interp_order=3).
template <int ORDER>
__global__ void __launch_bounds__(BLOCK) advect_kernel_f32_wide(const AdvectArgs<float> A) {
    advect_kernel_body<float, ORDER, false>(A);
}

template <typename T>
struct Lds64Launch {
    static const char *launch(const AdvectArgs<T> &, int, hipStream_t, int) { return nullptr; }
    static const char *launch_o3(const AdvectArgs<T> &, int, hipStream_t, int) { return nullptr; }
};
template <>
struct Lds64Launch<double> {
    // float64, order 1, fused levels: per-wave LDS tiles unless direct gathers are forced (lc_ctx_set_lds_tiles(0)),
    // the wind is float32-valued (numpy promotion path), K = 0, or the field is smaller than a tile
    static const char *launch(const AdvectArgs<double> &A, int grid, hipStream_t st, int mode) {
        if (mode == 0 || A.wind_f32 || A.K == 0 || !(A.ext || A.ext_raw) || A.nx_f + LC_PAD < T64_COLS || A.ny_f + LC_PAD < T64_ROWS) return nullptr;
        if (A.wg64 && A.nx_f + LC_PAD >= W64_COLS && A.ny_f + LC_PAD >= W64_ROWS) {   // one tile per workgroup (LCS_F64_WG_TILE=1: measured, off by default)
            AdvectArgs<double> B = A;
            B.ntx = (A.nx + W64_SIDE - 1) / W64_SIDE;
            const int nty = (A.ny + W64_SIDE - 1) / W64_SIDE;
            B.ntiles = B.ntx * nty;
            B.xcd_chunk = lcplan::xcd_chunk_tiles(B.ntx, nty, A.xcd_rows, A.xcd_split);
            const int gw = xcd_grid(B.ntiles, B.xcd_chunk) + B.pole_blocks;
#define LC_WG64(KF, CY, SR, NAME)                                                                                   \
    {                                                                                                               \
        hipLaunchKernelGGL((advect_wg64_kernel<KF, CY, SR>), dim3(gw, nmem(B)), dim3(BLOCK), 0, st, B);             \
        return NAME;                                                                                                \
    }
            const int sr = A.ext_raw ? 2 : (A.u_raw ? 1 : 0);
            if (A.K == 4 && A.cyclic && sr == 2) LC_WG64(4, true, 2, "advect_wg64_kernel<4, true, 2>")
            if (A.K == 4 && A.cyclic && sr == 1) LC_WG64(4, true, 1, "advect_wg64_kernel<4, true, 1>")
            if (A.K == 4 && A.cyclic) LC_WG64(4, true, 0, "advect_wg64_kernel<4, true, 0>")
            if (A.cyclic && sr == 2) LC_WG64(-1, true, 2, "advect_wg64_kernel<-1, true, 2>")
            if (A.cyclic && sr == 1) LC_WG64(-1, true, 1, "advect_wg64_kernel<-1, true, 1>")
            if (A.cyclic) LC_WG64(-1, true, 0, "advect_wg64_kernel<-1, true, 0>")
            if (sr == 2) LC_WG64(-1, false, 2, "advect_wg64_kernel<-1, false, 2>")
            if (sr == 1) LC_WG64(-1, false, 1, "advect_wg64_kernel<-1, false, 1>")
            LC_WG64(-1, false, 0, "advect_wg64_kernel<-1, false, 0>")
#undef LC_WG64
        }
        // (names as a profiler prints them; the last argument: 0 = lin + ext images, 1 = raw planes for the Euler sample + ext
        // image, 2 = raw planes for both, the fused-level value formed node by node: lc_advect_ex)
#define LC_LDS64(KF, CY, SR, NAME)                                                                                  \
    {                                                                                                               \
        hipLaunchKernelGGL((advect_lds64_kernel<KF, CY, SR>), dim3(grid, nmem(A)), dim3(BLOCK), 0, st, A);          \
        return NAME;                                                                                                \
    }
        if (A.ext_raw) {
            if (A.K == 4 && A.cyclic) LC_LDS64(4, true, 2, "advect_lds64_kernel<4, true, 2>")
            if (A.K == 4) LC_LDS64(4, false, 2, "advect_lds64_kernel<4, false, 2>")
            if (A.cyclic) LC_LDS64(-1, true, 2, "advect_lds64_kernel<-1, true, 2>")
            LC_LDS64(-1, false, 2, "advect_lds64_kernel<-1, false, 2>")
        }
        if (A.u_raw) {
            if (A.K == 4 && A.cyclic) LC_LDS64(4, true, 1, "advect_lds64_kernel<4, true, 1>")
            if (A.K == 4) LC_LDS64(4, false, 1, "advect_lds64_kernel<4, false, 1>")
            if (A.cyclic) LC_LDS64(-1, true, 1, "advect_lds64_kernel<-1, true, 1>")
            LC_LDS64(-1, false, 1, "advect_lds64_kernel<-1, false, 1>")
        }
        if (A.K == 4 && A.cyclic) LC_LDS64(4, true, 0, "advect_lds64_kernel<4, true, 0>")
        if (A.K == 4) LC_LDS64(4, false, 0, "advect_lds64_kernel<4, false, 0>")
        if (A.cyclic) LC_LDS64(-1, true, 0, "advect_lds64_kernel<-1, true, 0>")
        LC_LDS64(-1, false, 0, "advect_lds64_kernel<-1, false, 0>")
#undef LC_LDS64
    }
    // order 3 (SETTLS_order = 0 included: the Euler sample has its own tile)
    static const char *launch_o3(const AdvectArgs<double> &A, int grid, hipStream_t st, int mode) {
        if (mode == 0 || A.wind_f32 || !(A.ext || A.ext_cub) || A.nx_f + LC_PAD < T64O3 || A.ny_f + LC_PAD < T64O3) return nullptr;
        if (A.ext_cub) {  // no ext image: the iteration tile is formed from img[t], img[t+1] while it is staged
            if (A.K == 4 && A.cyclic) {
                hipLaunchKernelGGL((advect_lds64_o3_kernel<4, true, true>), dim3(grid, nmem(A)), dim3(BLOCK), 0, st, A);
                return "advect_lds64_o3_kernel<4, true, cub>";
            } else if (A.K == 4) {
                hipLaunchKernelGGL((advect_lds64_o3_kernel<4, false, true>), dim3(grid, nmem(A)), dim3(BLOCK), 0, st, A);
                return "advect_lds64_o3_kernel<4, false, cub>";
            } else if (A.cyclic) {
                hipLaunchKernelGGL((advect_lds64_o3_kernel<-1, true, true>), dim3(grid, nmem(A)), dim3(BLOCK), 0, st, A);
                return "advect_lds64_o3_kernel<-1, true, cub>";
            }
            hipLaunchKernelGGL((advect_lds64_o3_kernel<-1, false, true>), dim3(grid, nmem(A)), dim3(BLOCK), 0, st, A);
            return "advect_lds64_o3_kernel<-1, false, cub>";
        }
        if (A.K == 4 && A.cyclic) {
            hipLaunchKernelGGL((advect_lds64_o3_kernel<4, true>), dim3(grid, nmem(A)), dim3(BLOCK), 0, st, A);
            return "advect_lds64_o3_kernel<4, true>";
        } else if (A.K == 4) {
            hipLaunchKernelGGL((advect_lds64_o3_kernel<4, false>), dim3(grid, nmem(A)), dim3(BLOCK), 0, st, A);
            return "advect_lds64_o3_kernel<4, false>";
        } else if (A.cyclic) {
            hipLaunchKernelGGL((advect_lds64_o3_kernel<-1, true>), dim3(grid, nmem(A)), dim3(BLOCK), 0, st, A);
            return "advect_lds64_o3_kernel<-1, true>";
        }
        hipLaunchKernelGGL((advect_lds64_o3_kernel<-1, false>), dim3(grid, nmem(A)), dim3(BLOCK), 0, st, A);
        return "advect_lds64_o3_kernel<-1, false>";
    }
};

template <typename T, int ORDER>
struct DirectLaunch {
    static const char *launch(const AdvectArgs<T> &A, int grid, hipStream_t st) {
        if constexpr (ORDER == 1 && sizeof(T) == 8) {
            if (A.u_raw) {  // exact order with the raw planes as the order-1 source (lc_advect_ex)
                hipLaunchKernelGGL((advect_kernel<T, 1, false, SRC_RAW_EULER>), dim3(grid, nmem(A)), dim3(BLOCK), 0, st, A);
                return "advect_kernel<double, 1, false, 1>";
            }
        }
        hipLaunchKernelGGL((advect_kernel<T, ORDER>), dim3(grid, nmem(A)), dim3(BLOCK), 0, st, A);
        // (as a profiler prints them: all four template arguments)
        return ORDER == 1 ? "advect_kernel<double, 1, false, 0>" : ORDER == 2 ? "advect_kernel<double, 2, false, 0>"
             : ORDER == 3 ? "advect_kernel<double, 3, false, 0>" : ORDER == 4 ? "advect_kernel<double, 4, false, 0>"
                                                                              : "advect_kernel<double, 5, false, 0>";
    }
};
template <int ORDER>
struct DirectLaunch<float, ORDER> {
    static const char *launch(const AdvectArgs<float> &A, int grid, hipStream_t st) {
        if (ORDER == 1)
            hipLaunchKernelGGL((advect_kernel_f32<ORDER>), dim3(grid, nmem(A)), dim3(BLOCK), 0, st, A);
        else
            hipLaunchKernelGGL((advect_kernel_f32_wide<ORDER>), dim3(grid, nmem(A)), dim3(BLOCK), 0, st, A);
        return ORDER == 1 ? "advect_kernel_f32<1>" : ORDER == 2 ? "advect_kernel_f32_wide<2>" : ORDER == 3 ? "advect_kernel_f32_wide<3>"
             : ORDER == 4 ? "advect_kernel_f32_wide<4>" : "advect_kernel_f32_wide<5>";
    }
};

// ======================================================================================
// LC_X_CLAMP_REFERENCE_OUTER -- the reference's non-cyclic longitude clamp, as written (Q9):
//     positions_x[np.where(positions_x < x_min)] = x_min          (LCS/trajectory.py:96-97, 122-123)
// on a DataArray is ORTHOGONAL indexing: every (row, col) in the cross product of the rows and the columns
// that hold an offending parcel is set.  That couples all seeds after every sub-step, so this path runs one
// sub-step per launch with the positions in global memory: [sample + update + flag rows/cols below x_min]
// -> [flag rows/cols above x_max, on the array as it stands after the first assignment] -> next sub-step,
// which applies both cross products while loading.  Two-sample form and operation order of advect_seed.
// Only reached when a parcel really left the box (lc_advect tries the fused kernel first).
// Row-sharded grids: the row flags are local to a rank; the COLUMN flags of both assignments are OR-ed over the ranks
// (lc_ctx_set_flag_allreduce) before they are used -- the offending columns of the first assignment before the second
// looks at the array it leaves, those of the second before the next sub-step applies both.
// ======================================================================================
template <typename T>
struct OuterArgs {
    T *x, *y, *eu, *ev;                    // positions (= x_out, y_out) and the level's Euler sample
    unsigned *rlo, *clo, *rhi, *chi;       // this sub-step's flags
    const unsigned *p_rlo, *p_clo, *p_rhi, *p_chi;  // the previous sub-step's (NULL before the first)
};

template <typename T>
__device__ __forceinline__ T outer_applied(const AdvectArgs<T> &A, const unsigned *rlo, const unsigned *clo,
                                           const unsigned *rhi, const unsigned *chi, int iy, int ix, T x) {
    if (rlo) {
        if (rlo[iy] && clo[ix]) x = A.x_min;  // first assignment ...
        if (rhi[iy] && chi[ix]) x = A.x_max;  // ... then the second, on top of it
    }
    return x;
}

template <typename T, int ORDER>
__global__ void outer_substep_kernel(const AdvectArgs<T> A, const OuterArgs<T> O, int level, int is_iter) {
#pragma clang fp contract(off)
    const size_t n = (size_t)A.ny * A.nx;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int iy = (int)(i / A.nx), ix = (int)(i - (size_t)iy * A.nx);
        T x = outer_applied<T>(A, O.p_rlo, O.p_clo, O.p_rhi, O.p_chi, iy, ix, O.x[i]);
        T y = O.y[i];
        const T ys = A.seed_lat[iy];
        const T cx_conv = T(180) / (T(3.141592653589793 * 6371000.0) * fabs(cos((ys * T(3.141592653589793)) / T(180))));
        const int grow = A.row0 + iy;
        const bool pole = grow < A.order || grow >= A.ny_global - A.order;  // tools.py:24-33 (Q3), global row index
        // order-1 source: the raw planes when the caller handed those instead of the lin image (lc_advect_ex)
        const bool raw = A.u_raw && (pole || ORDER == 1);
        const size_t lstride = raw ? A.raw_plane : A.level_elems;
        const T *lvl = (raw ? A.u_raw : (pole ? A.lin : A.img)) + (size_t)level * lstride;
        if (!is_iter) {
            Pair<T> e = raw ? (pole ? sample<T, 1, false, true>(lvl, A, x, y) : sample<T, 1, true, true>(lvl, A, x, y))
                            : (pole ? sample<T, 1, false>(lvl, A, x, y) : sample<T, ORDER, true>(lvl, A, x, y));
            e.u = round_sample<T>(A, e.u);
            e.v = round_sample<T>(A, e.v);
            O.eu[i] = e.u;
            O.ev[i] = e.v;
            y = y + lat_increment<T>(A, A.dtcy, e.v);          // trajectory.py:86
            x = axpy<T>(A.dt * cx_conv, e.u, x);               // :87
        } else {
            const T *nxt = lvl + lstride;
            Pair<T> c, nn;
            if (raw) {
                const Tap<T> tap = pole ? locate<T, 1, false>(A, x, y) : locate<T, 1, true>(A, x, y);
                c = fetch<T, 1, true>(lvl, A, tap);
                nn = fetch<T, 1, true>(nxt, A, tap);
            } else if (pole) {
                const Tap<T> tap = locate<T, 1, false>(A, x, y);
                c = fetch<T, 1>(lvl, A, tap);
                nn = fetch<T, 1>(nxt, A, tap);
            } else {
                const Tap<T> tap = locate<T, ORDER, true>(A, x, y);
                c = fetch<T, ORDER>(lvl, A, tap);
                nn = fetch<T, ORDER>(nxt, A, tap);
            }
            c.u = round_sample<T>(A, c.u);
            c.v = round_sample<T>(A, c.v);
            nn.u = round_sample<T>(A, nn.u);
            nn.v = round_sample<T>(A, nn.v);
            const T eu = O.eu[i], ev = O.ev[i];
            y = y + lat_increment<T>(A, A.hdtcy, settls_bracket<T>(A, ev, c.v, nn.v));   // :110
            x = axpy<T>(A.half_dt * cx_conv, settls_bracket<T>(A, eu, c.u, nn.u), x);     // :112
        }
        y = (y > A.y_min) ? y : A.y_min;  // :89-90 / 115-116 (Q8)
        y = (y < A.y_max) ? y : A.y_max;
        O.x[i] = x;
        O.y[i] = y;
        if (x < A.x_min) {                // rows / columns of np.where(positions_x < x_min)
            O.rlo[iy] = 1u;
            O.clo[ix] = 1u;
        }
    }
}

template <typename T>
__global__ void outer_hi_kernel(const AdvectArgs<T> A, const OuterArgs<T> O) {
    const size_t n = (size_t)A.ny * A.nx;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int iy = (int)(i / A.nx), ix = (int)(i - (size_t)iy * A.nx);
        T x = O.x[i];
        if (O.rlo[iy] && O.clo[ix]) x = A.x_min;  // the array np.where(positions_x > x_max) looks at
        if (x > A.x_max) {
            O.rhi[iy] = 1u;
            O.chi[ix] = 1u;
        }
    }
}

template <typename T>
__global__ void outer_store_kernel(const AdvectArgs<T> A, const OuterArgs<T> O, T *dst_x, T *dst_y, int seeds_only) {
    const size_t n = (size_t)A.ny * A.nx;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int iy = (int)(i / A.nx), ix = (int)(i - (size_t)iy * A.nx);
        if (seeds_only) {  // trajectory.py:68-70: the seed grid
            dst_x[i] = A.seed_lon[ix];
            dst_y[i] = A.seed_lat[iy];
        } else {
            dst_x[i] = outer_applied<T>(A, O.rlo, O.clo, O.rhi, O.chi, iy, ix, O.x[i]);
            dst_y[i] = O.y[i];
        }
    }
}

template <typename T>
int advect_outer_impl(lc_ctx *ctx, AdvectArgs<T> A, int s_begin = 0, const T *x0 = nullptr, const T *y0 = nullptr) {
    // s_begin, x0, y0: restart at step s_begin from these positions (the state the fused kernel saved before the chunk
    // of levels in which a parcel first left the box); 0 / NULL: from the seed grid
    hipStream_t st = ctx->stream;
    const size_t n = (size_t)A.ny * A.nx;
    const size_t nflag = 2 * ((size_t)A.ny + A.nx);
    T *e = nullptr;
    unsigned *flags = nullptr;
    LC_HIP_CHECK(hipMallocAsync((void **)&e, 2 * n * sizeof(T), st));
    hipError_t er = hipMallocAsync((void **)&flags, 2 * nflag * sizeof(unsigned), st);
    if (er != hipSuccess) {
        (void)hipFreeAsync(e, st);
        LC_HIP_CHECK(er);
    }
    const int blocks = (int)((n + 255) / 256 < 16384 ? (n + 255) / 256 : 16384);
    OuterArgs<T> O = {};
    O.x = A.x_out;
    O.y = A.y_out;
    O.eu = e;
    O.ev = e + n;
    if (x0) {
        (void)hipMemcpyAsync(A.x_out, x0, n * sizeof(T), hipMemcpyDeviceToDevice, st);
        (void)hipMemcpyAsync(A.y_out, y0, n * sizeof(T), hipMemcpyDeviceToDevice, st);
    } else {
        hipLaunchKernelGGL((outer_store_kernel<T>), dim3(blocks), dim3(256), 0, st, A, O, A.x_out, A.y_out, 1);
        if (A.traj_x) hipLaunchKernelGGL((outer_store_kernel<T>), dim3(blocks), dim3(256), 0, st, A, O, A.traj_x, A.traj_y, 1);
    }
    int sub = 0;
    for (int s = s_begin; s < A.nsteps; ++s) {
        for (int k = 0; k <= A.K; ++k, ++sub) {
            unsigned *cur = flags + (size_t)(sub & 1) * nflag, *prev = flags + (size_t)((sub & 1) ^ 1) * nflag;
            (void)hipMemsetAsync(cur, 0, nflag * sizeof(unsigned), st);
            O.rlo = cur;
            O.clo = cur + A.ny;
            O.rhi = O.clo + A.nx;
            O.chi = O.rhi + A.ny;
            O.p_rlo = sub ? prev : nullptr;
            O.p_clo = prev + A.ny;
            O.p_rhi = O.p_clo + A.nx;
            O.p_chi = O.p_rhi + A.ny;
#define LC_OUTER(ORD) hipLaunchKernelGGL((outer_substep_kernel<T, ORD>), dim3(blocks), dim3(256), 0, st, A, O, A.t0 + s, k > 0)
            switch (A.order) {
                case 2: LC_OUTER(2); break;
                case 3: LC_OUTER(3); break;
                case 4: LC_OUTER(4); break;
                case 5: LC_OUTER(5); break;
                default: LC_OUTER(1); break;
            }
#undef LC_OUTER
            if (ctx->flag_reduce && ctx->flag_reduce(ctx->flag_reduce_user, O.clo, (size_t)A.nx) != 0) goto reduce_failed;
            hipLaunchKernelGGL((outer_hi_kernel<T>), dim3(blocks), dim3(256), 0, st, A, O);
            if (ctx->flag_reduce && ctx->flag_reduce(ctx->flag_reduce_user, O.chi, (size_t)A.nx) != 0) goto reduce_failed;
        }
        if (A.traj_x)
            hipLaunchKernelGGL((outer_store_kernel<T>), dim3(blocks), dim3(256), 0, st, A, O,
                               A.traj_x + (size_t)(s + 1) * n, A.traj_y + (size_t)(s + 1) * n, 0);
    }
    if (sub) hipLaunchKernelGGL((outer_store_kernel<T>), dim3(blocks), dim3(256), 0, st, A, O, A.x_out, A.y_out, 0);  // the last sub-step's flags applied
    {
        const hipError_t le = hipGetLastError();
        (void)hipFreeAsync(e, st);
        (void)hipFreeAsync(flags, st);
        LC_HIP_CHECK(le);
    }
    ctx->last_advect_kernel = "outer_substep_kernel";
    return LC_OK;
reduce_failed:
    (void)hipFreeAsync(e, st);
    (void)hipFreeAsync(flags, st);
    lc_set_error("lc_advect: the flag all-reduce of LC_X_CLAMP_REFERENCE_OUTER failed (lc_ctx_set_flag_allreduce callback returned non-zero)");
    return LC_ERCCL;
}

template <typename T>
int advect_impl(lc_ctx *ctx, const void *packed_lin, const void *packed_cub, const void *packed_ext, const void *u_raw,
                const void *v_raw, int nt, int ny_f, int nx_f,
                double lat_min, double lat_max, double lon_min, double lon_max, const void *seed_lat, int ny,
                const void *seed_lon, int nx, int row0, int ny_global, double timestep, int K, int order, int cyclic,
                int t0, int nsteps, void *x_out, void *y_out, void *traj_x, void *traj_y, const void *x_start,
                const void *y_start, int wind_f32 = 0, int n_members = 1, int t0_stride = 0, int fuse_levels_raw = 0,
                const void *lin32 = nullptr, const void *lin32_v = nullptr) {
    AdvectArgs<T> A{};
    A.wind_f32 = wind_f32;
    A.n_members = n_members;
    A.member_t0_stride = t0_stride;
    A.member_plane = (size_t)ny * nx;
    A.x_start = (const T *)x_start;
    A.y_start = (const T *)y_start;
    A.traj_skip0 = 0;
    A.pair_d = -1;
    A.traj_pair_ok = (nx % 2 == 0) && ((size_t)traj_x % (2 * sizeof(T)) == 0) && ((size_t)traj_y % (2 * sizeof(T)) == 0);
    A.out_pair_ok = (nx % 2 == 0) && ((size_t)x_out % (2 * sizeof(T)) == 0) && ((size_t)y_out % (2 * sizeof(T)) == 0);
    A.traj_line_ok = (nx % 4 == 0) && ((size_t)traj_x % 16 == 0) && ((size_t)traj_y % 16 == 0) && sizeof(T) == 4;
    A.patch_mode = ctx->patch_mode;
    A.xcd_rows = ctx->xcd_chunk_rows;
    A.xcd_split = ctx->xcd_split;
    A.lin = (const T *)packed_lin;
    A.img = (order != 1) ? (const T *)packed_cub : (const T *)packed_lin;
    A.ext = (order == 1 || order == 3) ? (const T *)packed_ext : nullptr;  // general orders: two-sample form
    A.lin32 = (sizeof(T) == 8 && order == 1) ? (const float *)lin32 : nullptr;  // LC_F64_WIND_F32_LIN32: the float32 order-1 image (then lin == img == NULL)
    A.u_raw32 = (sizeof(T) == 8 && order == 3) ? (const float *)lin32 : nullptr;  // ... at order 3: the float32 raw planes (u, then v given as u_raw / v_raw)
    A.v_raw32 = (sizeof(T) == 8 && order == 3) ? (const float *)lin32_v : nullptr;
    A.u_raw = (const T *)u_raw;  // (lc_advect_ex validated: only where a kernel reads them)
    A.v_raw = (const T *)v_raw;
    A.raw_plane = (size_t)ny_f * nx_f;
    A.ext_raw = fuse_levels_raw && sizeof(T) == 8 && order == 1 && u_raw && !wind_f32 && !packed_ext;
    A.ext_cub = fuse_levels_raw && sizeof(T) == 8 && order == 3 && packed_cub && !wind_f32 && !packed_ext;
    A.level_elems = lc_level_elems(ny_f, nx_f);
    A.pitch = nx_f + LC_PAD;
    A.ny_f = ny_f;
    A.nx_f = nx_f;
    A.lat_min = (T)lat_min;
    A.lon_min = (T)lon_min;
    A.lat_span = (T)lat_max - (T)lat_min;
    A.lon_span = (T)lon_max - (T)lon_min;
    set_fast_transform(A);
    A.y_min = (T)lat_min;
    A.y_max = (T)lat_max;
    A.x_min = (T)lon_min;
    A.x_max = (T)lon_max;
    A.seed_lat = (const T *)seed_lat;
    A.seed_lon = (const T *)seed_lon;
    A.ny = ny;
    A.nx = nx;
    A.row0 = row0;
    A.ny_global = ny_global;
    const double conv_y = 180.0 / (6371000.0 * 3.141592653589793);  // trajectory.py:55
    A.dt = (T)timestep;
    A.half_dt = (T)(0.5 * timestep);
    A.dtcy = (T)(timestep * conv_y);
    A.hdtcy = (T)((0.5 * timestep) * conv_y);
    A.K = K;
    A.order = order;
    const bool outer = cyclic == LC_X_CLAMP_REFERENCE_OUTER;
    A.cyclic = cyclic == LC_X_CYCLIC;
    A.clamp_flag = nullptr;
    A.verify = sizeof(T) == 4 ? ctx->verify_dev : nullptr;
    unsigned *clamp_flag = nullptr;
    if (outer) {
        // fused kernel first, with a flag that says whether the clamp ever moved a parcel; if not, per-point and
        // outer-product clamps coincide (both are no-ops) and the fused result IS the reference's
        LC_HIP_CHECK(hipMallocAsync((void **)&clamp_flag, sizeof(unsigned), ctx->stream));
        (void)hipMemsetAsync(clamp_flag, 0, sizeof(unsigned), ctx->stream);
        A.clamp_flag = clamp_flag;
    }
    A.t0 = t0;
    A.nsteps = nsteps;
    A.x_out = (T *)x_out;
    A.y_out = (T *)y_out;
    A.traj_x = (T *)traj_x;
    A.traj_y = (T *)traj_y;
    A.ntx = (nx + TILE_W - 1) / TILE_W;
    const int nty = (ny + TILE_H - 1) / TILE_H;
    A.ntiles = A.ntx * nty;
    A.xcd_chunk = lcplan::xcd_chunk_tiles(A.ntx, nty, ctx->xcd_chunk_rows, ctx->xcd_split);
    A.wg64 = ctx->f64_wg_tile;
    A.tile_order = ctx->tile_order >= 0 ? ctx->tile_order : 1;
    A.tile_order_two_seed = ctx->tile_order >= 0 ? ctx->tile_order : 2;
    {   // leading workgroups for the global pole rows present in this block of seed rows
        const lcplan::PoleRows pr = lcplan::pole_rows(A.order, A.row0, ny, A.ny_global, nx, ctx->pole_blocks != 0, BLOCK);
        A.pole_lo = pr.lo;
        A.pole_hi = pr.hi;
        A.pole_blocks = pr.blocks;
    }
    const int grid = xcd_grid(A.ntiles, A.xcd_chunk) + A.pole_blocks;
    // Kernel choice (float + fused levels only; measured on MI355X, 4096^2 seeds, 96 steps, K=4, 8x8-seed waves):
    //   order 1: direct gather 10.9 ms (vector-L1 lookup bound), LDS tiles 10.2 ms (VALU-issue bound);
    //   order 3: direct gather 37.8 ms, LDS tiles 20.4 ms.
    // With the wind scaled x4 / x10 (patches stretched far beyond a tile) the LDS kernel degrades to
    // 10.7-11.9 ms against 11.0 for direct gathers: its global-gather fallback is per lane, so the worst
    // case costs the tile bookkeeping only.  An adaptive "skip the tile when most lanes miss" vote was
    // measured and dropped (it costs 5 % everywhere to save 8 % in that extreme).
    // lc_ctx_set_lds_tiles / LCS_LDS_TILES (read once at context creation) override (profiling).
    const bool use_lds = ctx->lds_tiles != 0;
    const bool fused64 = sizeof(T) == 8 && (A.ext != nullptr || A.ext_raw || A.ext_cub);  // single-sample iterations in float64
    const char *name = nullptr;
    auto launch = [&](const AdvectArgs<T> &A) {
        if constexpr (sizeof(T) == 8) {
            if (A.lin32) {  // LC_F64_WIND_F32_LIN32: per-wave LDS tiles of levels t and t + 1, or direct gathers
                const bool tiles = use_lds && A.K > 0 && A.nx_f + LC_PAD >= TW_COLS && A.ny_f + LC_PAD >= TW_ROWS;
#define LC_W32(KF, CY, NAME)                                                                                       \
    {                                                                                                              \
        hipLaunchKernelGGL((advect_lds64w_kernel<KF, CY>), dim3(grid, nmem(A)), dim3(BLOCK), 0, ctx->stream, A);   \
        name = NAME;                                                                                               \
        return;                                                                                                    \
    }
                if (tiles && A.K == 4 && A.cyclic) LC_W32(4, true, "advect_lds64w_kernel<4, true>")
                if (tiles && A.K == 4) LC_W32(4, false, "advect_lds64w_kernel<4, false>")
                if (tiles && A.cyclic) LC_W32(-1, true, "advect_lds64w_kernel<-1, true>")
                if (tiles) LC_W32(-1, false, "advect_lds64w_kernel<-1, false>")
#undef LC_W32
                hipLaunchKernelGGL(advect_w32_kernel, dim3(grid, nmem(A)), dim3(BLOCK), 0, ctx->stream, A);
                name = "advect_w32_kernel";
                return;
            }
            if (A.u_raw32 && order == 3 && use_lds && A.nx_f + LC_PAD >= TW3 && A.ny_f + LC_PAD >= TW3) {
                // LC_F64_WIND_F32_LIN32 at order 3: tiles of the float64 coefficients of levels t and t + 1 (else: the generic kernel below,
                // whose pole rows read the float32 planes)
#define LC_W32O3(KF, CY, NAME)                                                                                        \
    {                                                                                                                 \
        hipLaunchKernelGGL((advect_lds64w_o3_kernel<KF, CY>), dim3(grid, nmem(A)), dim3(BLOCK), 0, ctx->stream, A);   \
        name = NAME;                                                                                                  \
        return;                                                                                                       \
    }
                if (A.K == 4 && A.cyclic) LC_W32O3(4, true, "advect_lds64w_o3_kernel<4, true>")
                if (A.K == 4) LC_W32O3(4, false, "advect_lds64w_o3_kernel<4, false>")
                if (A.cyclic) LC_W32O3(-1, true, "advect_lds64w_o3_kernel<-1, true>")
                LC_W32O3(-1, false, "advect_lds64w_o3_kernel<-1, false>")
#undef LC_W32O3
            }
        }
        if (order == 2 || order == 4 || order == 5) {  // generic direct kernel, any dtype
            name = order == 2 ? DirectLaunch<T, 2>::launch(A, grid, ctx->stream)
                 : order == 4 ? DirectLaunch<T, 4>::launch(A, grid, ctx->stream) : DirectLaunch<T, 5>::launch(A, grid, ctx->stream);
        } else if (order == 3) {
            if (fused64) {
                name = Lds64Launch<T>::launch_o3(A, grid, ctx->stream, ctx->lds_tiles);
                if (!name) {  // (with ext_cub too: advect_seed_fast64_o3 looks at it)
                    hipLaunchKernelGGL((advect_kernel<T, 3, sizeof(T) == 8>), dim3(grid, nmem(A)), dim3(BLOCK), 0, ctx->stream, A);
                    name = "advect_kernel<double, 3, true, 0>";
                }
            } else if (!(use_lds && (name = LdsLaunch<T, 3>::launch(A, grid, ctx->stream, ctx->lds_tiles)))) {
                name = DirectLaunch<T, 3>::launch(A, grid, ctx->stream);
            }
        } else {
            if (fused64) {
                name = Lds64Launch<T>::launch(A, grid, ctx->stream, ctx->lds_tiles);
                if (!name) {
                    if constexpr (sizeof(T) == 8) {
                        if (A.ext_raw) {
                            hipLaunchKernelGGL((advect_kernel<T, 1, true, SRC_RAW_ALL>), dim3(grid, nmem(A)), dim3(BLOCK), 0, ctx->stream, A);
                            name = "advect_kernel<double, 1, true, 2>";
                        } else if (A.u_raw) {
                            hipLaunchKernelGGL((advect_kernel<T, 1, true, SRC_RAW_EULER>), dim3(grid, nmem(A)), dim3(BLOCK), 0, ctx->stream, A);
                            name = "advect_kernel<double, 1, true, 1>";
                        } else {
                            hipLaunchKernelGGL((advect_kernel<T, 1, true>), dim3(grid, nmem(A)), dim3(BLOCK), 0, ctx->stream, A);
                            name = "advect_kernel<double, 1, true, 0>";
                        }
                    }
                }
            } else if (!(use_lds && (name = LdsLaunch<T, 1>::launch(A, grid, ctx->stream, ctx->lds_tiles)))) {
                name = DirectLaunch<T, 1>::launch(A, grid, ctx->stream);
            }
        }
    };
    // Level chunks (lc_ctx_set_level_chunk): the series runs as consecutive launches of at most `chunk` time levels,
    // each continuing from the positions the previous one left in x_out / y_out (a seed's start is read by the thread
    // that writes its result, so in place is safe).  A launch's workgroups then stay within `chunk` levels of each
    // other -- their tiles of the wind images meet in L2 / the Infinity Cache instead of being spread over the whole
    // series (sparse seed grids: DESIGN 4) -- and the launch's tail is one chunk long.  Positions at a level's end are
    // the kernels' whole state (the latitude clamp is applied before they are stored), so chunked == unchunked, bit
    // for bit.
    // Default (level_chunk < 0): 32 levels per launch -- measured on MI355X against one launch (profiles/r03): C3 96 steps
    // 6.55 -> 6.46 ms, 200 steps 15.0 -> 13.9, C4 (8192^2 x 384) 100.2 -> 93.0, order 3 16.1 -> 15.8, one member of C5
    // (2048^2 x 200) +17 %; chunks of 16 / 24 / 48 / 64 within 1 % of 32.  Small grids too (below): only launches of
    // less than one workgroup per compute unit keep the single launch.
    // By SETTLS_order too (a level costs 1 + K samples; 200 steps on C3): K = 4 chunks of 32 13.9 ms against 15.0 in one
    // launch, K = 2 chunks of 64 9.66 against 9.86 (32) / 9.91 (one launch), K = 1 one launch 7.21 against 7.59 (32),
    // K = 0 one launch 4.77 against 5.41 (32): the lighter the level, the less a launch's spread over the levels costs.
    // (lcplan::chunk_for_k)
    // From 2^18 seeds per call (measured, chunks of 32 against one launch, 96 levels of the 720 x 1440 float32 series unless
    // noted: 512^2 seeds 1.256 -> 1.23 ms, 724^2 1.25 -> 1.05, 1024^2 1.43 -> 1.37, 1024^2 x 200 levels 3.16 -> 2.89, order 3
    // 3.27 -> 3.15, 2048^2 2.67 -> 2.60; float64 config 2, 1024^2 x 200: order 1 3.30 -> 3.06, order 3 7.66 -> 6.97 with 25-32
    // levels, 8 / 16 / 50 / 100 within 2-5 % of that).
    // (lcplan::CHUNK_FROM_SEEDS)
    // LC_X_CLAMP_REFERENCE_OUTER: chunks of 16 levels WHATEVER the size (lcplan::OUTER_CHUNK), the clamp flag read back after
    // each, the positions before each chunk kept -- so the sub-step path restarts at the chunk in which a parcel first left
    // the box instead of at t0 (regional domains: parcels leave routinely; the fused work thrown away is one chunk).  The
    // value must not depend on the local block: each chunk ends in one flag all-reduce over the ranks of a row-sharded
    // grid, whose blocks differ in size (a by-size rule sent ranks on either side of 2^18 seeds into different numbers
    // of collectives).
    const size_t plane_elems = (size_t)ny * nx;
    // An ensemble through the float32 two-seed order-1 kernel: consecutive MEMBERS share a lane (PATCH_PAIR).
    // Launches walk the group's level window [0, nsteps + (g - 1) d): member q steps at levels [q d, q d + nsteps)
    // (d = t0_stride), so the group shares its tiles for nsteps - (g - 1) d of them; continuation in place as for any chunk.
    // (four members per lane -- 99 VGPRs, the members of a lane up to 3 d levels of travel apart -- measured 480 ms on
    // config 5 against 280 for pairs: in the jets three steps are 6 cells, nearly every wave-sample has a lane outside the tile)
    const bool pairs_ok = n_members > 1 && order == 1 && !outer && !traj_x && use_lds &&
                          (ctx->patch_mode < 0 || ctx->patch_mode == PATCH_PAIR) && K > 0 && order1_two_seed_applies(A, ctx->lds_tiles);
    const lcplan::Groups G = lcplan::member_groups(n_members, t0_stride, nsteps, pairs_ok);
    const int total = G.total;
    if (G.g) {
        A.pair_d = t0_stride;
        A.pair_n = nsteps;
        A.pair_plane = plane_elems;
        A.pair_g = G.g;
        A.pair_last = G.last;
        A.n_members = G.n_groups;
        A.member_t0_stride = G.group_stride;
        A.member_plane = (size_t)G.g * plane_elems;
    }
    int n_launches = 0;
    T *saved = nullptr;   // [2][ny*nx]: positions at the start of the current chunk (outer mode, from the second chunk on)
    int restart = -1;
    auto flag_error = [&]() {
        lc_set_error("lc_advect: the flag all-reduce of LC_X_CLAMP_REFERENCE_OUTER failed (lc_ctx_set_flag_allreduce callback returned non-zero)");
        return LC_ERCCL;
    };
    const int chunk = lcplan::level_chunk(ctx->level_chunk, outer, (long long)ny * nx, n_members, K, total, sizeof(T) == 8 && order == 1);
    for (int ci = 0, nci = lcplan::n_chunks(total, chunk); ci < nci; ++ci) {
        const int s0 = lcplan::chunk_first(ci, chunk);
        AdvectArgs<T> C = A;
        C.t0 = t0 + s0;
        C.nsteps = lcplan::chunk_levels(ci, total, chunk);
        C.pair_l0 = s0;
        if (s0 > 0) {
            C.x_start = A.x_out;
            C.y_start = A.y_out;
            C.traj_skip0 = 1;
            if (A.traj_x) {
                C.traj_x = A.traj_x + (size_t)s0 * plane_elems;
                C.traj_y = A.traj_y + (size_t)s0 * plane_elems;
            }
            if (outer) {
                if (!saved && hipMallocAsync((void **)&saved, 2 * plane_elems * sizeof(T), ctx->stream) != hipSuccess) {
                    saved = nullptr;          // no room for the saved positions: the restart is from the seed grid, as before
                    (void)hipGetLastError();  // ... and the refusal is not this call's error
                }
                if (saved) {
                    (void)hipMemcpyAsync(saved, A.x_out, plane_elems * sizeof(T), hipMemcpyDeviceToDevice, ctx->stream);
                    (void)hipMemcpyAsync(saved + plane_elems, A.y_out, plane_elems * sizeof(T), hipMemcpyDeviceToDevice, ctx->stream);
                }
            }
        }
        launch(C);
        ++n_launches;
        if (outer) {
            unsigned moved = 0;
            hipError_t e1 = hipGetLastError();
            // row-sharded: "did a parcel leave the box ANYWHERE" -- every rank must take the same path below
            const bool red_fail = e1 == hipSuccess && ctx->flag_reduce && ctx->flag_reduce(ctx->flag_reduce_user, clamp_flag, 1) != 0;
            if (e1 == hipSuccess && !red_fail) e1 = hipMemcpyAsync(&moved, clamp_flag, sizeof(unsigned), hipMemcpyDeviceToHost, ctx->stream);
            if (e1 == hipSuccess && !red_fail) e1 = hipStreamSynchronize(ctx->stream);
            if (red_fail || e1 != hipSuccess) {
                (void)hipFreeAsync(clamp_flag, ctx->stream);
                if (saved) (void)hipFreeAsync(saved, ctx->stream);
                if (red_fail) return flag_error();
                LC_HIP_CHECK(e1);
            }
            if (moved) {
                restart = lcplan::outer_restart(s0, saved != nullptr);   // (no room for the saved positions: from the seed grid, as before)
                break;
            }
        }
    }
    ctx->last_advect_kernel = name;
    ctx->last_advect_launches = n_launches;
    if (outer) {
        (void)hipFreeAsync(clamp_flag, ctx->stream);
        A.clamp_flag = nullptr;
        int rc = LC_OK;
        if (restart >= 0) {
            A.ext = nullptr;  // the exact path keeps the reference's two-sample form
            A.ext_raw = 0;
            A.ext_cub = 0;
            rc = advect_outer_impl<T>(ctx, A, restart, restart > 0 ? saved : (const T *)A.x_start,
                                      restart > 0 ? saved + plane_elems : (const T *)A.y_start);
        }
        if (saved) (void)hipFreeAsync(saved, ctx->stream);
        if (rc != LC_OK) return rc;
    }
    LC_HIP_CHECK(hipGetLastError());
    return LC_OK;
}

// ---- a2 on its own: one interpolation pass at given positions (tools.py:11-41) ----
template <typename T, int ORDER>
__global__ void sample_kernel(const AdvectArgs<T> A, const T *__restrict__ px, const T *__restrict__ py, int level,
                              T *__restrict__ out_u, T *__restrict__ out_v) {
    const size_t n = (size_t)A.ny * A.nx;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int iy = (int)(i / A.nx);
        const int grow = A.row0 + iy;
        const bool pole = grow < A.order || grow >= A.ny_global - A.order;
        Pair<T> r;
        if (A.u_raw && (pole || ORDER == 1)) {  // order-1 source: the raw planes (lc_sample_ex)
            const T *up = A.u_raw + (size_t)level * A.raw_plane;
            r = pole ? sample<T, 1, false, true>(up, A, px[i], py[i]) : sample<T, 1, true, true>(up, A, px[i], py[i]);
        } else if (pole)
            r = sample<T, 1, false>(A.lin + (size_t)level * A.level_elems, A, px[i], py[i]);
        else
            r = sample<T, ORDER, true>(A.img + (size_t)level * A.level_elems, A, px[i], py[i]);
        out_u[i] = r.u;
        out_v[i] = r.v;
    }
}

template <typename T>
int sample_impl(lc_ctx *ctx, const void *packed_lin, const void *packed_cub, const void *u_raw, const void *v_raw, int ny_f, int nx_f, double lat_min,
                double lat_max, double lon_min, double lon_max, int level, const void *px, const void *py, int ny,
                int nx, int row0, int ny_global, int order, void *out_u, void *out_v) {
    AdvectArgs<T> A = {};
    A.lin = (const T *)packed_lin;
    A.img = (order != 1) ? (const T *)packed_cub : (const T *)packed_lin;
    A.u_raw = (const T *)u_raw;
    A.v_raw = (const T *)v_raw;
    A.raw_plane = (size_t)ny_f * nx_f;
    A.level_elems = lc_level_elems(ny_f, nx_f);
    A.pitch = nx_f + LC_PAD;
    A.ny_f = ny_f;
    A.nx_f = nx_f;
    A.lat_min = (T)lat_min;
    A.lon_min = (T)lon_min;
    A.lat_span = (T)lat_max - (T)lat_min;
    A.lon_span = (T)lon_max - (T)lon_min;
    set_fast_transform(A);
    A.ny = ny;
    A.nx = nx;
    A.row0 = row0;
    A.ny_global = ny_global;
    A.order = order;
    const size_t n = (size_t)ny * nx;
    const int blocks = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
#define LC_SAMPLE(ORD)                                                                                                 \
    hipLaunchKernelGGL((sample_kernel<T, ORD>), dim3(blocks), dim3(256), 0, ctx->stream, A, (const T *)px, (const T *)py, \
                       level, (T *)out_u, (T *)out_v)
    switch (order) {
        case 2: LC_SAMPLE(2); break;
        case 3: LC_SAMPLE(3); break;
        case 4: LC_SAMPLE(4); break;
        case 5: LC_SAMPLE(5); break;
        default: LC_SAMPLE(1); break;
    }
#undef LC_SAMPLE
    LC_HIP_CHECK(hipGetLastError());
    return LC_OK;
}

}  // namespace

#ifdef LCS_STAMPS
extern "C" int lc_debug_read_cause(unsigned long long *out4, int reset) {
    if (hipMemcpyFromSymbol(out4, HIP_SYMBOL(g_cause), sizeof(g_cause)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[4] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_cause), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
extern "C" int lc_debug_read_hist(unsigned long long *out10, int reset) {
    if (hipMemcpyFromSymbol(out10, HIP_SYMBOL(g_hist), sizeof(g_hist)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[10] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_hist), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
extern "C" int lc_debug_read_redo(unsigned long long *out27, int reset) {
    if (hipMemcpyFromSymbol(out27, HIP_SYMBOL(g_redo), sizeof(g_redo)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[27] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_redo), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
extern "C" int lc_debug_read_stamps(unsigned long long *out8, int reset) {
    if (hipMemcpyFromSymbol(out8, HIP_SYMBOL(g_stamps), sizeof(g_stamps)) != hipSuccess) return -1;
    if (reset) {
        unsigned long long z[8] = {};
        if (hipMemcpyToSymbol(HIP_SYMBOL(g_stamps), z, sizeof(z)) != hipSuccess) return -1;
    }
    return 0;
}
#endif

// order-1 source of a call: may the lin image be absent?  (raw planes given, and a kernel family that reads them)
static bool raw_replaces_lin(const void *u_raw, const void *v_raw, int dtype, int interp_order) {
    return u_raw && v_raw && (interp_order != 1 || dtype != LC_F32);
}

extern "C" int lc_sample_raw(lc_ctx *ctx, const void *packed_lin, const void *packed_cub, const void *u_raw, const void *v_raw,
                             int dtype, int nt, int ny_f, int nx_f, double lat_min, double lat_max, double lon_min,
                             double lon_max, int level, const void *pos_x_dev, const void *pos_y_dev, int ny, int nx, int row0,
                             int ny_global, int interp_order, void *out_u, void *out_v) {
    LC_REQUIRE(ctx, "lc_sample: null context");
    LC_REQUIRE(dtype == LC_F32 || dtype == LC_F64, "lc_sample: bad dtype %d", dtype);
    if (interp_order < 1 || interp_order > 5) {
        lc_set_error("lc_sample: interp_order %d unsupported (scipy's spline orders 1..5; 0 fails in the reference)", interp_order);
        return LC_EUNSUPPORTED;
    }
    LC_REQUIRE((u_raw == nullptr) == (v_raw == nullptr), "lc_sample_raw: u_raw and v_raw must both be set or both NULL");
    LC_REQUIRE((packed_lin || raw_replaces_lin(u_raw, v_raw, dtype, interp_order)) && (interp_order == 1 || packed_cub),
               "lc_sample: missing field image");
    LC_REQUIRE(pos_x_dev && pos_y_dev && out_u && out_v, "lc_sample: null pointer");
    LC_REQUIRE(level >= 0 && level < nt && ny_f >= 4 && nx_f >= 4 && ny >= 1 && nx >= 1, "lc_sample: bad sizes");
    LC_REQUIRE(row0 >= 0 && row0 + ny <= ny_global, "lc_sample: rows outside the global grid");
    LC_REQUIRE(lat_max > lat_min && lon_max > lon_min, "lc_sample: field coordinates must be ascending");
    LC_HIP_CHECK(hipSetDevice(ctx->device));
    if (!raw_replaces_lin(u_raw, v_raw, dtype, interp_order)) u_raw = v_raw = nullptr;
    if (dtype == LC_F32)
        return sample_impl<float>(ctx, packed_lin, packed_cub, u_raw, v_raw, ny_f, nx_f, lat_min, lat_max, lon_min, lon_max, level,
                                  pos_x_dev, pos_y_dev, ny, nx, row0, ny_global, interp_order, out_u, out_v);
    return sample_impl<double>(ctx, packed_lin, packed_cub, u_raw, v_raw, ny_f, nx_f, lat_min, lat_max, lon_min, lon_max, level,
                               pos_x_dev, pos_y_dev, ny, nx, row0, ny_global, interp_order, out_u, out_v);
}

extern "C" int lc_sample(lc_ctx *ctx, const void *packed_lin, const void *packed_cub, int dtype, int nt, int ny_f,
                         int nx_f, double lat_min, double lat_max, double lon_min, double lon_max, int level,
                         const void *pos_x_dev, const void *pos_y_dev, int ny, int nx, int row0, int ny_global,
                         int interp_order, void *out_u, void *out_v) {
    return lc_sample_raw(ctx, packed_lin, packed_cub, nullptr, nullptr, dtype, nt, ny_f, nx_f, lat_min, lat_max, lon_min, lon_max,
                         level, pos_x_dev, pos_y_dev, ny, nx, row0, ny_global, interp_order, out_u, out_v);
}

extern "C" int lc_advect(lc_ctx *ctx, const void *packed_lin, const void *packed_cub, const void *packed_ext,
                         int dtype, int nt, int ny_f, int nx_f, double lat_min, double lat_max, double lon_min, double lon_max,
                         const void *seed_lat_dev, int ny, const void *seed_lon_dev, int nx, int row0, int ny_global,
                         double timestep, int settls_order, int interp_order, int cyclic_x, int t0, int nsteps,
                         void *x_out, void *y_out, void *traj_x, void *traj_y) {
    return lc_advect_from(ctx, packed_lin, packed_cub, packed_ext, dtype, nt, ny_f, nx_f, lat_min, lat_max, lon_min, lon_max,
                          seed_lat_dev, ny, seed_lon_dev, nx, row0, ny_global, nullptr, nullptr, timestep, settls_order,
                          interp_order, cyclic_x, t0, nsteps, x_out, y_out, traj_x, traj_y);
}

extern "C" int lc_advect_from(lc_ctx *ctx, const void *packed_lin, const void *packed_cub, const void *packed_ext,
                              int dtype, int nt, int ny_f, int nx_f, double lat_min, double lat_max, double lon_min,
                              double lon_max, const void *seed_lat_dev, int ny, const void *seed_lon_dev, int nx, int row0,
                              int ny_global, const void *x_start, const void *y_start, double timestep, int settls_order,
                              int interp_order, int cyclic_x, int t0, int nsteps, void *x_out, void *y_out, void *traj_x,
                              void *traj_y) {
    return lc_advect_batch(ctx, packed_lin, packed_cub, packed_ext, dtype, nt, ny_f, nx_f, lat_min, lat_max, lon_min, lon_max,
                           seed_lat_dev, ny, seed_lon_dev, nx, row0, ny_global, x_start, y_start, timestep, settls_order,
                           interp_order, cyclic_x, t0, nsteps, 1, 0, x_out, y_out, traj_x, traj_y);
}

extern "C" int lc_advect_batch(lc_ctx *ctx, const void *packed_lin, const void *packed_cub, const void *packed_ext,
                               int dtype, int nt, int ny_f, int nx_f, double lat_min, double lat_max, double lon_min,
                               double lon_max, const void *seed_lat_dev, int ny, const void *seed_lon_dev, int nx, int row0,
                               int ny_global, const void *x_start, const void *y_start, double timestep, int settls_order,
                               int interp_order, int cyclic_x, int t0, int nsteps, int n_members, int t0_stride, void *x_out,
                               void *y_out, void *traj_x, void *traj_y) {
    lc_advect_args a = {};
    a.struct_size = sizeof(a);
    a.packed_lin = packed_lin;
    a.packed_cub = packed_cub;
    a.packed_ext = packed_ext;
    a.dtype = dtype;
    a.nt = nt;
    a.ny_f = ny_f;
    a.nx_f = nx_f;
    a.lat_min = lat_min;
    a.lat_max = lat_max;
    a.lon_min = lon_min;
    a.lon_max = lon_max;
    a.seed_lat_dev = seed_lat_dev;
    a.ny = ny;
    a.seed_lon_dev = seed_lon_dev;
    a.nx = nx;
    a.row0 = row0;
    a.ny_global = ny_global;
    a.x_start = x_start;
    a.y_start = y_start;
    a.timestep = timestep;
    a.settls_order = settls_order;
    a.interp_order = interp_order;
    a.cyclic_x = cyclic_x;
    a.t0 = t0;
    a.nsteps = nsteps;
    a.n_members = n_members;
    a.t0_stride = t0_stride;
    a.x_out = x_out;
    a.y_out = y_out;
    a.traj_x = traj_x;
    a.traj_y = traj_y;
    return lc_advect_ex(ctx, &a);
}

extern "C" int lc_advect_ex(lc_ctx *ctx, const lc_advect_args *args) {
    LC_REQUIRE(args, "lc_advect_ex: null arguments");
    LC_REQUIRE(args->struct_size == sizeof(lc_advect_args), "lc_advect_ex: struct_size %zu, this library's lc_advect_args has %zu bytes",
               args->struct_size, sizeof(lc_advect_args));
    LC_REQUIRE(ctx, "lc_advect: null context");
    const lc_advect_args &a = *args;
    const void *packed_lin = a.packed_lin, *packed_cub = a.packed_cub, *packed_ext = a.packed_ext, *u_raw = a.u_raw, *v_raw = a.v_raw;
    const int dtype = a.dtype, nt = a.nt, ny_f = a.ny_f, nx_f = a.nx_f, ny = a.ny, nx = a.nx, row0 = a.row0, ny_global = a.ny_global;
    const int settls_order = a.settls_order, interp_order = a.interp_order, cyclic_x = a.cyclic_x, t0 = a.t0, nsteps = a.nsteps;
    const int n_members = a.n_members, t0_stride = a.t0_stride;
    const void *x_start = a.x_start, *y_start = a.y_start, *seed_lat_dev = a.seed_lat_dev, *seed_lon_dev = a.seed_lon_dev;
    void *x_out = a.x_out, *y_out = a.y_out, *traj_x = a.traj_x, *traj_y = a.traj_y;
    const double lat_min = a.lat_min, lat_max = a.lat_max, lon_min = a.lon_min, lon_max = a.lon_max, timestep = a.timestep;
    LC_REQUIRE(n_members >= 1 && n_members <= 65535 && t0_stride >= 0, "lc_advect_batch: bad n_members %d / t0_stride %d", n_members,
               t0_stride);
    if (n_members > 1) {
        LC_REQUIRE(!traj_x && !traj_y, "lc_advect_batch: trajectories are per member: call lc_advect for each");
        if (cyclic_x == LC_X_CLAMP_REFERENCE_OUTER) {
            lc_set_error("lc_advect_batch: LC_X_CLAMP_REFERENCE_OUTER is decided per member: call lc_advect for each");
            return LC_EUNSUPPORTED;
        }
    }
    LC_REQUIRE((x_start == nullptr) == (y_start == nullptr), "lc_advect_from: x_start and y_start must both be set or both NULL");
    if (x_start && cyclic_x == LC_X_CLAMP_REFERENCE_OUTER) {
        lc_set_error("lc_advect_from: LC_X_CLAMP_REFERENCE_OUTER restarts from the seed grid when a parcel leaves the box "
                     "and cannot continue from given positions");
        return LC_EUNSUPPORTED;
    }
    LC_REQUIRE(dtype == LC_F32 || dtype == LC_F64 || dtype == LC_F64_WIND_F32 || dtype == LC_F64_WIND_F32_LIN32, "lc_advect: bad dtype %d", dtype);
    LC_REQUIRE((dtype != LC_F64_WIND_F32 && dtype != LC_F64_WIND_F32_LIN32) || !packed_ext, "lc_advect: LC_F64_WIND_F32 keeps the two-sample form (no ext)");
    if (dtype == LC_F64_WIND_F32_LIN32) {
        if ((interp_order != 1 && interp_order != 3) || cyclic_x == LC_X_CLAMP_REFERENCE_OUTER) {
            lc_set_error("lc_advect: LC_F64_WIND_F32_LIN32 serves interp_order 1 and 3 with cyclic / per-point boundaries; interp_order %d or "
                         "LC_X_CLAMP_REFERENCE_OUTER take LC_F64_WIND_F32 (float64 images of the float32 wind)", interp_order);
            return LC_EUNSUPPORTED;
        }
        if (interp_order == 1)
            LC_REQUIRE(packed_lin && !packed_cub && !u_raw && !v_raw, "lc_advect: LC_F64_WIND_F32_LIN32 at order 1 takes packed_lin (the float32 order-1 image) and nothing else");
        else
            LC_REQUIRE(packed_cub && u_raw && v_raw && !packed_lin, "lc_advect: LC_F64_WIND_F32_LIN32 at order 3 takes packed_cub (float64 coefficients: "
                       "lc_field_pack with LC_F64_WIND_F32) and the float32 planes as u_raw / v_raw");
    }
    if (interp_order < 1 || interp_order > 5) {
        lc_set_error("lc_advect: interp_order %d unsupported (scipy's spline orders 1..5; 0 fails in the reference too)",
                     interp_order);
        return LC_EUNSUPPORTED;
    }
    LC_REQUIRE((u_raw == nullptr) == (v_raw == nullptr), "lc_advect_ex: u_raw and v_raw must both be set or both NULL");
    const bool raw_ok = raw_replaces_lin(u_raw, v_raw, dtype, interp_order) || dtype == LC_F64_WIND_F32_LIN32;
    LC_REQUIRE(packed_lin || raw_ok, "lc_advect: packed_lin is required (the order-1 image: pole rows use order 1) unless lc_advect_ex is "
               "given the raw planes u_raw / v_raw -- and in LC_F32 at interp_order 1 always");
    if (!raw_ok) u_raw = v_raw = nullptr;  // (float32 at order 1 reads the lin image's 16-byte node pairs)
    LC_REQUIRE(interp_order == 1 || packed_cub, "lc_advect: interp_order > 1 needs packed_cub (lc_field_pack of that order)");
    LC_REQUIRE(interp_order == 1 || interp_order == 3 || !packed_ext, "lc_advect: orders 2, 4, 5 take no packed_ext");
    LC_REQUIRE(nt >= 2 && ny_f >= 4 && nx_f >= 4, "lc_advect: field too small (nt=%d ny_f=%d nx_f=%d)", nt, ny_f, nx_f);
    LC_REQUIRE(ny >= 1 && nx >= 1 && seed_lat_dev && seed_lon_dev, "lc_advect: bad seed grid");
    LC_REQUIRE(row0 >= 0 && row0 + ny <= ny_global, "lc_advect: rows [%d,%d) outside global grid of %d rows", row0,
               row0 + ny, ny_global);
    LC_REQUIRE(settls_order >= 0, "lc_advect: SETTLS_order must be >= 0");
    LC_REQUIRE(cyclic_x >= LC_X_CLAMP_POINT && cyclic_x <= LC_X_CLAMP_REFERENCE_OUTER, "lc_advect: bad cyclic_x %d", cyclic_x);
    if (cyclic_x == LC_X_CLAMP_REFERENCE_OUTER && (row0 != 0 || ny != ny_global) && !ctx->flag_reduce) {
        lc_set_error("lc_advect: LC_X_CLAMP_REFERENCE_OUTER couples every seed row through the offending columns: a row "
                     "block (rows [%d,%d) of %d) needs lc_ctx_set_flag_allreduce", row0, row0 + ny, ny_global);
        return LC_EUNSUPPORTED;
    }
    LC_REQUIRE(t0 >= 0 && nsteps >= 0 && t0 + (n_members - 1) * t0_stride + nsteps <= nt - 1,
               "lc_advect: steps [%d,%d) need levels up to %d, have %d", t0, t0 + (n_members - 1) * t0_stride + nsteps,
               t0 + (n_members - 1) * t0_stride + nsteps, nt);
    LC_REQUIRE(x_out && y_out, "lc_advect: null output");
    LC_REQUIRE((traj_x == nullptr) == (traj_y == nullptr), "lc_advect: traj_x and traj_y must both be set or both NULL");
    LC_REQUIRE(lat_max > lat_min && lon_max > lon_min, "lc_advect: field coordinates must be ascending");
    // the kernels address a tap inside one time level with 32-bit offsets (24-bit row multiply)
    if (lc_level_elems(ny_f, nx_f) * ((dtype == LC_F32 || dtype == LC_F64_WIND_F32_LIN32) ? 4 : 1) >= (size_t)1 << 32 || nx_f + LC_PAD >= (1 << 24) ||
        ny_f + LC_PAD >= (1 << 24)) {
        lc_set_error("lc_advect: a %dx%d time level is too large for 32-bit tap offsets", ny_f, nx_f);
        return LC_EUNSUPPORTED;
    }
    LC_HIP_CHECK(hipSetDevice(ctx->device));
    if (dtype == LC_F32)
        return advect_impl<float>(ctx, packed_lin, packed_cub, packed_ext, u_raw, v_raw, nt, ny_f, nx_f, lat_min, lat_max, lon_min,
                                  lon_max, seed_lat_dev, ny, seed_lon_dev, nx, row0, ny_global, timestep, settls_order,
                                  interp_order, cyclic_x, t0, nsteps, x_out, y_out, traj_x, traj_y, x_start, y_start, 0,
                                  n_members, t0_stride, 0);
    if (dtype == LC_F64_WIND_F32_LIN32)   // (order 1: the float32 image; order 3: the float64 coefficients + the float32 planes)
        return advect_impl<double>(ctx, nullptr, interp_order == 3 ? packed_cub : nullptr, nullptr, nullptr, nullptr, nt, ny_f, nx_f, lat_min, lat_max, lon_min,
                                   lon_max, seed_lat_dev, ny, seed_lon_dev, nx, row0, ny_global, timestep, settls_order,
                                   interp_order, cyclic_x, t0, nsteps, x_out, y_out, traj_x, traj_y, x_start, y_start,
                                   1, n_members, t0_stride, 0, interp_order == 1 ? packed_lin : u_raw, interp_order == 3 ? v_raw : nullptr);
    return advect_impl<double>(ctx, packed_lin, packed_cub, packed_ext, u_raw, v_raw, nt, ny_f, nx_f, lat_min, lat_max, lon_min,
                               lon_max, seed_lat_dev, ny, seed_lon_dev, nx, row0, ny_global, timestep, settls_order,
                               interp_order, cyclic_x, t0, nsteps, x_out, y_out, traj_x, traj_y, x_start, y_start,
                               dtype == LC_F64_WIND_F32, n_members, t0_stride, a.fuse_levels_raw);
}
