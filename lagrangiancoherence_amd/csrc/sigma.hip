// K3 -- flow-map gradient + largest singular value, fused.
//
// Restates, per seed:
//   LCS.flowmap_gradient            LCS/LCS.py:195-208   lon/lat -> X,Y,Z on the sphere
//   tools.derivative_spherical_coords  LCS/tools.py:254-264  metric dx, dy
//   tools.fourth_order_derivative   LCS/tools.py:202-228   5-point stencil, cyclic in
//                                   longitude, one-sided/2 on the 2 first/last rows (Q12)
//   eigen step of LCS.__call__      LCS/LCS.py:152-154   ||M||_2 of the 3x3 built by a
//                                   row-major reshape of the 9 components (Q13)
// The reference materialises X,Y,Z, six derivative fields, three zero fields and a
// pandas MultiIndex; here X,Y,Z are computed once per cell (as float when fd_fp32_cast,
// Q11), differenced on chip and the 2x2 Gram eigenproblem is solved in closed form.
// HBM traffic: read x_dep,y_dep once (+halo re-reads served by L2), write sigma once.
// Three kernels, one arithmetic per type:
//   sigma_march_kernel_f32   float32, sigma only, even width (the default): a wave walks down its rows with
//                            five rows of X,Y,Z in registers, x-neighbours by wavefront shuffle
//   sigma_kernel_f32         float32, sigma only, any width: a 64 x 16 tile + halo through LDS
//   sigma_kernel<T,S>        float64 and/or the 9-plane tensor output, numba's typing of the stencil (S)
#include "lcs_common.h"

namespace {

constexpr int SW = 64;  // tile width  (longitude)
constexpr int SH = 16;  // tile height (latitude)
constexpr int HALO = 2;
constexpr int LW = SW + 2 * HALO;
constexpr int LH = SH + 2 * HALO;
constexpr int SBLOCK = 256;

__device__ __forceinline__ void sincos_t(float a, float *s, float *c) { sincosf(a, s, c); }
__device__ __forceinline__ void sincos_t(double a, double *s, double *c) { sincos(a, s, c); }

template <typename T>
struct SigmaArgs {
    const T *x_dep, *y_dep, *seed_lat;
    int in_row0, n_in_rows, nx, ny_global;
    T dlat, dlon;
    int layout;
    int out_row0, n_out_rows;
    T *sigma;   // may be null
    T *tensor;  // null, or 9 planes [n_out_rows*nx] in the reference's merge order (LCS.py:220)
};

// T: arithmetic type of positions; S: type X,Y,Z are differenced in
template <typename T, typename S>
__global__ void __launch_bounds__(SBLOCK) sigma_kernel(const SigmaArgs<T> A) {
#pragma clang fp contract(off)
    __shared__ S sX[LH][LW + 1];
    __shared__ S sY[LH][LW + 1];
    __shared__ S sZ[LH][LW + 1];
    const int ntx = (A.nx + SW - 1) / SW;
    const int tyi = blockIdx.x / ntx, txi = blockIdx.x - tyi * ntx;
    const int gy0 = A.out_row0 + tyi * SH;  // global row of the tile's first output row
    const int gx0 = txi * SW;
    const T PI = T(3.141592653589793);
    const T R = T(6371000);

    // stage X,Y,Z for the tile + halo
    for (int i = threadIdx.x; i < LW * LH; i += SBLOCK) {
        const int ly = i / LW, lx = i - ly * LW;
        const int gy = gy0 - HALO + ly;            // global row
        int gx = gx0 - HALO + lx;                  // cyclic column (tools.py:225-228)
        gx %= A.nx;
        if (gx < 0) gx += A.nx;
        const int ry = gy - A.in_row0;             // row inside the input window
        S vx = S(0), vy = S(0), vz = S(0);
        if (gy >= 0 && gy < A.ny_global && ry >= 0 && ry < A.n_in_rows) {
            const size_t o = (size_t)ry * A.nx + gx;
            const T lon = (A.x_dep[o] * PI) / T(180);            // LCS.py:195
            const T lat = ((A.y_dep[o] - T(90)) * PI) / T(180);  // LCS.py:196 (colatitude - pi)
            T sl, cl, so, co;
            sincos_t(lat, &sl, &cl);
            sincos_t(lon, &so, &co);
            vx = (S)((R * sl) * co);  // LCS.py:197
            vy = (S)((R * sl) * so);  // LCS.py:198
            vz = (S)(R * cl);         // LCS.py:199
        }
        sX[ly][lx] = vx;
        sY[ly][lx] = vy;
        sZ[ly][lx] = vz;
    }
    __syncthreads();

    const T dy = ((PI / T(180)) * A.dlat) * R;  // tools.py:256
    for (int i = threadIdx.x; i < SW * SH; i += SBLOCK) {
        const int oy = i / SW, ox = i - oy * SW;
        const int gy = gy0 + oy, gx = gx0 + ox;
        if (gy >= A.out_row0 + A.n_out_rows || gx >= A.nx) continue;
        const int ly = oy + HALO, lx = ox + HALO;
        // numba typing of tools.py:204-207: S differences, double scaling, S store
        auto centred = [](S p1, S m1, S p2, S m2) -> S {
            const S d1 = p1 - m1, d2 = p2 - m2;
            return (S)((4.0 / 3.0) * (double)d1 / 2.0 - (1.0 / 3.0) * (double)d2 / 4.0);
        };
        auto ddx = [&](S(*a)[LW + 1]) -> S {
            return centred(a[ly][lx + 1], a[ly][lx - 1], a[ly][lx + 2], a[ly][lx - 2]);
        };
        auto ddy = [&](S(*a)[LW + 1]) -> S {
            if (gy < 2) return (S)((double)(a[ly + 1][lx] - a[ly][lx]) / 2.0);                  // tools.py:210-213
            if (gy >= A.ny_global - 2) return (S)((double)(a[ly][lx] - a[ly - 1][lx]) / 2.0);  // tools.py:214-217
            return centred(a[ly + 1][lx], a[ly - 1][lx], a[ly + 2][lx], a[ly - 2][lx]);
        };
        const T latr = (A.seed_lat[gy - A.in_row0] * PI) / T(180);  // tools.py:254
        const T dx = (((PI / T(180)) * A.dlon) * R) * cos(latr);   // tools.py:255
        // derivative / metric: the division is done in T (float64 / float32 as numpy would)
        const T ta = (T)ddx(sX) / dx, tb = (T)ddy(sX) / dy;  // dXdx, dXdy
        const T tc = (T)ddx(sY) / dx, td = (T)ddy(sY) / dy;  // dYdx, dYdy
        const T te = (T)ddx(sZ) / dx, tf = (T)ddy(sZ) / dy;  // dZdx, dZdy
        const size_t oidx = (size_t)(gy - A.out_row0) * A.nx + gx;
        if (A.tensor) {
            const size_t plane = (size_t)A.n_out_rows * A.nx;
            A.tensor[oidx] = ta;
            A.tensor[plane + oidx] = tb;
            A.tensor[2 * plane + oidx] = tc;
            A.tensor[3 * plane + oidx] = td;
            A.tensor[4 * plane + oidx] = te;
            A.tensor[5 * plane + oidx] = tf;
            A.tensor[6 * plane + oidx] = T(0);  // dXdr, dYdr, dZdr (LCS.py:206-208)
            A.tensor[7 * plane + oidx] = T(0);
            A.tensor[8 * plane + oidx] = T(0);
        }
        if (!A.sigma) continue;
        const double a_ = ta, b_ = tb, c_ = tc, d_ = td, e_ = te, f_ = tf;
        double p, q, r;
        if (A.layout == LC_LAYOUT_REFERENCE) {
            // M = [[a,b,c],[d,e,f],[0,0,0]] (LCS.py:153): Gram matrix of its two non-zero rows
            p = a_ * a_ + b_ * b_ + c_ * c_;
            q = d_ * d_ + e_ * e_ + f_ * f_;
            r = a_ * d_ + b_ * e_ + c_ * f_;
        } else {
            // Jacobian [[a,b],[c,d],[e,f]]: F^T F
            p = a_ * a_ + c_ * c_ + e_ * e_;
            q = b_ * b_ + d_ * d_ + f_ * f_;
            r = a_ * b_ + c_ * d_ + e_ * f_;
        }
        const double dpq = p - q;
        const double disc = sqrt(dpq * dpq + 4.0 * r * r);
        const double lam = 0.5 * ((p + q) + disc);
        A.sigma[oidx] = (T)sqrt(lam);  // NaN in -> NaN out (Q14)
    }
}

// ======================================================================================
// float fast path of K3.  The float result cannot be bit-identical to the reference
// (numpy float32 sin/cos, LAPACK sgesdd) anyway, so this instantiation spends as few VALU
// cycles per cell as it can: bounded-argument sincos (Cody-Waite by pi/2 + cephes
// minimax polynomials, ~1 ulp), float stencil with the 4th-order weights folded
// (2/3, -1/12), reciprocal metrics per row, float closed form.  Tile 64 x 16 outputs per
// 256 threads (halo redundancy (68*20)/(64*16) = 1.33).  Measured on 4096^2 cells (ms): 64x64 0.201,
// 64x32 0.105, 128x16 0.112, 64x16 0.0905, 64x12 0.096, 32x32 0.094, 32x16 0.101, 128x8 0.099, 64x8 0.108 --
// occupancy (LDS per workgroup) matters more than halo redundancy.
// ======================================================================================
#ifndef LCS_SIGMA_MROWS
#define LCS_SIGMA_MROWS 20
#endif
constexpr int FW = 64, FH = 16;
constexpr int FLW = FW + 2 * HALO, FLH = FH + 2 * HALO;

__device__ __forceinline__ void bounded_sincosf(float a, float *sn, float *cs) {  // |a| < 64
    const float n = rintf(a * 0.636619772367581343f);  // 2/pi
    float r = fmaf(n, -1.5703125f, a);
    r = fmaf(n, -4.837512969970703125e-4f, r);
    r = fmaf(n, -7.54978995489188216e-8f, r);
    const float z = r * r;
    const float sp = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f), z * r, r);
    const float cp = fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f),
                          z * z, fmaf(-0.5f, z, 1.0f));
    const int q = (int)n;
    const bool odd = q & 1;  // odd quadrant: sine and cosine swap
    const unsigned s1 = __builtin_bit_cast(unsigned, odd ? cp : sp), c1 = __builtin_bit_cast(unsigned, odd ? sp : cp);
    // signs as bit operations: sine flips in quadrants 2, 3 (bit 1 of q), cosine in quadrants 1, 2 (bit 1 of q + 1)
    *sn = __builtin_bit_cast(float, s1 ^ (((unsigned)q << 30) & 0x80000000u));
    *cs = __builtin_bit_cast(float, c1 ^ (((unsigned)(q + 1) << 30) & 0x80000000u));
}

// X, Y, Z of one departure point (LCS.py:195-199), float
__device__ __forceinline__ void sphere_xyz_f32(float xd, float yd, float &vx, float &vy, float &vz) {
#pragma clang fp contract(off)
    const float D2R = 3.141592653589793f / 180.0f;
    const float R = 6371000.0f;
    const float lon = xd * D2R;            // LCS.py:195
    const float lat = (yd - 90.0f) * D2R;  // LCS.py:196
    float sl, cl, so, co;
    if (fabsf(lat) < 64.0f && fabsf(lon) < 64.0f) {
        bounded_sincosf(lat, &sl, &cl);
        bounded_sincosf(lon, &so, &co);
    } else {  // out of the polynomial's range, or NaN: library path
        sincosf(lat, &sl, &cl);
        sincosf(lon, &so, &co);
    }
    const float rs = R * sl;
    vx = rs * co;  // LCS.py:197
    vy = rs * so;  // LCS.py:198
    vz = R * cl;   // LCS.py:199
}

// 1 / dx of a seed row (tools.py:254-255), float
__device__ __forceinline__ float inv_dx_f32(float seed_lat, float dlon) {
#pragma clang fp contract(off)
    const float latr = (seed_lat * 3.141592653589793f) / 180.0f;                           // tools.py:254
    return 1.0f / ((((3.141592653589793f / 180.0f) * dlon) * 6371000.0f) * cosf(latr));  // tools.py:255
}

// the two stencils with the 4th-order weights folded: (4/3)/2 and -(1/3)/4 of tools.py:204-207
__device__ __forceinline__ float centred_f32(float p1, float m1, float p2, float m2) {
#pragma clang fp contract(off)
    return __builtin_fmaf(2.0f / 3.0f, p1 - m1, (-1.0f / 12.0f) * (p2 - m2));
}
__device__ __forceinline__ float ddy_f32(int gy, int ny_global, float m2, float m1, float c0, float p1, float p2) {
#pragma clang fp contract(off)
    if (gy < 2) return 0.5f * (p1 - c0);              // tools.py:210-213
    if (gy >= ny_global - 2) return 0.5f * (c0 - m1);  // tools.py:214-217
    return centred_f32(p1, m1, p2, m2);
}

// largest singular value from the six derivatives (closed-form 2x2 Gram eigenvalue), float
__device__ __forceinline__ float sigma_from_derivatives_f32(int layout, float a_, float b_, float c_, float d_, float e_,
                                                            float f_) {
#pragma clang fp contract(off)
    float p, q, r;
    if (layout == LC_LAYOUT_REFERENCE) {  // LCS.py:153 (Q13)
        p = __builtin_fmaf(c_, c_, __builtin_fmaf(b_, b_, a_ * a_));
        q = __builtin_fmaf(f_, f_, __builtin_fmaf(e_, e_, d_ * d_));
        r = __builtin_fmaf(c_, f_, __builtin_fmaf(b_, e_, a_ * d_));
    } else {
        p = __builtin_fmaf(e_, e_, __builtin_fmaf(c_, c_, a_ * a_));
        q = __builtin_fmaf(f_, f_, __builtin_fmaf(d_, d_, b_ * b_));
        r = __builtin_fmaf(e_, f_, __builtin_fmaf(c_, d_, a_ * b_));
    }
    const float dpq = p - q;
    // v_sqrt_f32 (1 ulp) instead of the correctly rounded expansion: 14 fewer instructions per root
    const float disc = __builtin_amdgcn_sqrtf(__builtin_fmaf(dpq, dpq, (4.0f * r) * r));
    return __builtin_amdgcn_sqrtf(0.5f * ((p + q) + disc));
}

// general float kernel (any width, any alignment): X, Y, Z of a 64 x 16 tile + halo through LDS
__global__ void __launch_bounds__(SBLOCK) sigma_kernel_f32(const SigmaArgs<float> A) {
    __shared__ float sX[FLH][FLW + 1];
    __shared__ float sY[FLH][FLW + 1];
    __shared__ float sZ[FLH][FLW + 1];
    __shared__ float s_inv_dx[FH];
    const int ntx = (A.nx + FW - 1) / FW;
    const int tyi = blockIdx.x / ntx, txi = blockIdx.x - tyi * ntx;
    const int gy0 = A.out_row0 + tyi * FH;
    const int gx0 = txi * FW;

    if (threadIdx.x < FH) {
        const int gy = gy0 + (int)threadIdx.x;
        s_inv_dx[threadIdx.x] = gy < A.out_row0 + A.n_out_rows ? inv_dx_f32(A.seed_lat[gy - A.in_row0], A.dlon) : 0.0f;
    }
    for (int i = threadIdx.x; i < FLW * FLH; i += SBLOCK) {
        const int ly = i / FLW, lx = i - ly * FLW;
        const int gy = gy0 - HALO + ly;
        int gx = gx0 - HALO + lx;
        gx = gx < 0 ? gx + A.nx : (gx >= A.nx ? gx - A.nx : gx);
        if (gx < 0 || gx >= A.nx) {  // grids narrower than the tile: general modulo
            gx %= A.nx;
            if (gx < 0) gx += A.nx;
        }
        const int ry = gy - A.in_row0;
        float vx = 0.0f, vy = 0.0f, vz = 0.0f;
        if (gy >= 0 && gy < A.ny_global && ry >= 0 && ry < A.n_in_rows) {
            const size_t o = (size_t)ry * A.nx + gx;
            sphere_xyz_f32(A.x_dep[o], A.y_dep[o], vx, vy, vz);
        }
        sX[ly][lx] = vx;
        sY[ly][lx] = vy;
        sZ[ly][lx] = vz;
    }
    __syncthreads();

    const float inv_dy = 1.0f / (((3.141592653589793f / 180.0f) * A.dlat) * 6371000.0f);  // tools.py:256
    for (int i = threadIdx.x; i < FW * FH; i += SBLOCK) {
        const int oy = i / FW, ox = i - oy * FW;
        const int gy = gy0 + oy, gx = gx0 + ox;
        if (gy >= A.out_row0 + A.n_out_rows || gx >= A.nx) continue;
        const int ly = oy + HALO, lx = ox + HALO;
        const float inv_dx = s_inv_dx[oy];
        auto ddx = [&](float(*a)[FLW + 1]) -> float {
            return centred_f32(a[ly][lx + 1], a[ly][lx - 1], a[ly][lx + 2], a[ly][lx - 2]) * inv_dx;
        };
        auto ddy = [&](float(*a)[FLW + 1]) -> float {
            return ddy_f32(gy, A.ny_global, a[ly - 2][lx], a[ly - 1][lx], a[ly][lx], a[ly + 1][lx], a[ly + 2][lx]) * inv_dy;
        };
        const float a_ = ddx(sX), b_ = ddy(sX), c_ = ddx(sY), d_ = ddy(sY), e_ = ddx(sZ), f_ = ddy(sZ);
        A.sigma[(size_t)(gy - A.out_row0) * A.nx + gx] = sigma_from_derivatives_f32(A.layout, a_, b_, c_, d_, e_, f_);
    }
}

// ======================================================================================
// float, sigma only, even width: the marching kernel.  A WAVE owns a span of 128 columns (two per lane, one 8-byte
// load per field and row) and walks down MROWS output rows with the X, Y, Z of five rows in registers: d/dy comes
// from the registers, d/dx from the two neighbouring lanes by wavefront shuffle (`ds_bpermute_b32`: 12 per row, no
// LDS memory, no barrier).  Lanes 0 and 63 are halo lanes (124 columns written per wave), rows -2..+2 around the
// strip are loaded as halo: the sincos work per output cell is 1.03 x (1 + 4/MROWS) instead of the LDS tile's 1.33.
// Five rows of loads are in flight per wave (a register ring indexed like the window).  Same arithmetic as sigma_kernel_f32, bit for bit.
// ======================================================================================
#ifndef LCS_SIGMA_MBLOCK
#define LCS_SIGMA_MBLOCK 256
#endif
constexpr int MBLOCK = LCS_SIGMA_MBLOCK;  // threads per workgroup of the marching kernel (waves are independent)
constexpr int MCOLS = 2;                 // columns per lane
constexpr int MSPAN_OUT = 62 * MCOLS;    // columns written per wave

template <int MROWS, int LAYOUT>
__global__ void __launch_bounds__(MBLOCK) sigma_march_kernel_f32(const SigmaArgs<float> A, int nspans, int nstrips) {
    static_assert(MROWS <= 64, "row metrics live one per lane");
    const int lane = threadIdx.x & 63;
    const int w = blockIdx.x * (MBLOCK / 64) + __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);  // wave-uniform
    const int strip = w / nspans, span = w - strip * nspans;
    if (strip >= nstrips) return;
    const int gy0 = A.out_row0 + strip * MROWS;
    const int nrows = min(MROWS, A.out_row0 + A.n_out_rows - gy0);
    const int col = span * MSPAN_OUT + (lane - 1) * MCOLS;
    const bool writes = lane >= 1 && lane <= 62 && col < A.nx;
    int c = col < 0 ? col + A.nx : col;  // cyclic column of this lane's pair (tools.py:225-228); nx is even
    if (c >= A.nx) {
        c -= A.nx;
        if (c >= A.nx) c %= A.nx;
    }
    const int from_prev = ((lane + 63) & 63) * 4, from_next = ((lane + 1) & 63) * 4;  // bpermute byte addresses
    // 1/dx of this strip's rows, one per lane, broadcast by readlane in the loop
    float my_inv_dx = 0.0f;
    if (lane < nrows) my_inv_dx = inv_dx_f32(A.seed_lat[gy0 + lane - A.in_row0], A.dlon);
    const float inv_dy = 1.0f / (((3.141592653589793f / 180.0f) * A.dlat) * 6371000.0f);  // tools.py:256

    auto row_ok = [&](int gy) {  // wave-uniform
        const int ry = gy - A.in_row0;
        return gy >= 0 && gy < A.ny_global && ry >= 0 && ry < A.n_in_rows;
    };
    auto load_row = [&](int gy, float2 &xd, float2 &yd) {  // always issued (a row outside the window reads the nearest
        const int ry = min(max(gy - A.in_row0, 0), A.n_in_rows - 1);  // one and is zeroed below): no branch around loads
        const float *px = A.x_dep + (size_t)ry * A.nx, *py = A.y_dep + (size_t)ry * A.nx;
        xd = *reinterpret_cast<const float2 *>(px + c);
        yd = *reinterpret_cast<const float2 *>(py + c);
    };
    auto shuf = [&](int addr, float v) {
        return __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(addr, __builtin_bit_cast(int, v)));
    };
    float X[5][MCOLS], Y[5][MCOLS], Z[5][MCOLS];  // rows gy-2 .. gy+2; slot = (row - first row) mod 5
    float2 RX[5], RY[5];                          // the next five rows as loaded: five rows of loads in flight per wave
    const int base = gy0 - 2;
#pragma unroll
    for (int j = 0; j < 5; ++j) {
        load_row(base + j, RX[j], RY[j]);
    }
    // rows enter in slot order, so with the loop unrolled by five every register index below is static and the
    // compiler counts the loads in flight itself (vmcnt)
    for (int o = 0; o * 5 - 4 < nrows; ++o) {
#pragma unroll
        for (int S4 = 0; S4 < 5; ++S4) {         // S4: slot of the entering row
            const int i = o * 5 + S4 - 4;        // output row gy0 + i once rows up to gy0 + i + 2 are in
            if (i < nrows) {                     // wave-uniform
                const int S0 = (S4 + 1) % 5, S1 = (S4 + 2) % 5, S2 = (S4 + 3) % 5, S3 = (S4 + 4) % 5;
                const int gyn = gy0 + i + 2;     // the row entering the window
                const float2 cx = RX[S4], cy = RY[S4];
                load_row(gyn + 5, RX[S4], RY[S4]);  // its ring slot goes to the row five further down
                if (row_ok(gyn)) {
                    sphere_xyz_f32(cx.x, cy.x, X[S4][0], Y[S4][0], Z[S4][0]);
                    sphere_xyz_f32(cx.y, cy.y, X[S4][1], Y[S4][1], Z[S4][1]);
                } else {
                    X[S4][0] = X[S4][1] = Y[S4][0] = Y[S4][1] = Z[S4][0] = Z[S4][1] = 0.0f;
                }
                if (i >= 0) {
                    const int gy = gy0 + i;
                    const float inv_dx = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, my_inv_dx), i));
                    const bool edge = gy < 2 || gy >= A.ny_global - 2;  // one-sided rows (Q12): wave-uniform, rare
                    float dxs[3][MCOLS], dys[3][MCOLS];
                    auto field = [&](float (&F)[5][MCOLS], int n) {
                        const float pk0 = shuf(from_prev, F[S2][0]), pk1 = shuf(from_prev, F[S2][1]);
                        const float nk0 = shuf(from_next, F[S2][0]), nk1 = shuf(from_next, F[S2][1]);
                        dxs[n][0] = centred_f32(F[S2][1], pk1, nk0, pk0) * inv_dx;
                        dxs[n][1] = centred_f32(nk0, F[S2][0], nk1, pk1) * inv_dx;
#pragma unroll
                        for (int k = 0; k < MCOLS; ++k) {
                            float d = centred_f32(F[S3][k], F[S1][k], F[S4][k], F[S0][k]);
                            if (edge) d = ddy_f32(gy, A.ny_global, F[S0][k], F[S1][k], F[S2][k], F[S3][k], F[S4][k]);
                            dys[n][k] = d * inv_dy;
                        }
                    };
                    field(X, 0);
                    field(Y, 1);
                    field(Z, 2);
                    float sig[MCOLS];
#pragma unroll
                    for (int k = 0; k < MCOLS; ++k)
                        sig[k] = sigma_from_derivatives_f32(LAYOUT, dxs[0][k], dys[0][k], dxs[1][k], dys[1][k], dxs[2][k], dys[2][k]);
                    if (writes)
                        *reinterpret_cast<float2 *>(A.sigma + (size_t)(gy - A.out_row0) * A.nx + col) = make_float2(sig[0], sig[1]);
                }
            }
        }
    }
}

template <typename T>
int sigma_impl(lc_ctx *ctx, const void *x_dep, const void *y_dep, int in_row0, int n_in_rows, int nx, int ny_global,
               const void *seed_lat, double dlat, double dlon, int fd_fp32_cast, int layout, int out_row0,
               int n_out_rows, void *sigma_out, void *tensor_out = nullptr) {
    SigmaArgs<T> A;
    A.x_dep = (const T *)x_dep;
    A.y_dep = (const T *)y_dep;
    A.seed_lat = (const T *)seed_lat;
    A.in_row0 = in_row0;
    A.n_in_rows = n_in_rows;
    A.nx = nx;
    A.ny_global = ny_global;
    A.dlat = (T)dlat;
    A.dlon = (T)dlon;
    A.layout = layout;
    A.out_row0 = out_row0;
    A.n_out_rows = n_out_rows;
    A.sigma = (T *)sigma_out;
    A.tensor = (T *)tensor_out;
    if constexpr (sizeof(T) == 4) {
        // the marching kernel's waves walk MROWS + 4 rows one after the other: a latency floor of ~20 us whatever the
        // size, so below 2^23 cells the LDS-tile kernel (more, shorter-lived workgroups) is the faster one
        // (2048^2: 27.5 vs 29.2 us; 2896^2: 50.9 vs 43.2).  sigma_march 1 forces it (tests, A/B), 2 = by size.
        const bool march = ctx->sigma_march == 1 || (ctx->sigma_march == 2 && (long long)nx * n_out_rows >= (1ll << 23));
        if (!tensor_out && nx % MCOLS == 0 && nx >= 2 * MCOLS && march &&
            (((uintptr_t)x_dep | (uintptr_t)y_dep | (uintptr_t)sigma_out) & 7) == 0) {  // float, sigma only, even width
            constexpr int MROWS = LCS_SIGMA_MROWS;
            const int nspans = (nx + MSPAN_OUT - 1) / MSPAN_OUT, nstrips = (n_out_rows + MROWS - 1) / MROWS;
            const int waves = nspans * nstrips;
            const dim3 grid((waves + MBLOCK / 64 - 1) / (MBLOCK / 64));
            ctx->last_sigma_kernel = "sigma_march_kernel_f32";
            if (layout == LC_LAYOUT_REFERENCE)
                hipLaunchKernelGGL((sigma_march_kernel_f32<MROWS, LC_LAYOUT_REFERENCE>), grid, dim3(MBLOCK), 0, ctx->stream, A,
                                   nspans, nstrips);
            else
                hipLaunchKernelGGL((sigma_march_kernel_f32<MROWS, LC_LAYOUT_PHYSICAL>), grid, dim3(MBLOCK), 0, ctx->stream, A,
                                   nspans, nstrips);
            LC_HIP_CHECK(hipGetLastError());
            return LC_OK;
        }
        if (!tensor_out) {  // float, sigma only, any width: the LDS-tile kernel
            const int fx = (nx + FW - 1) / FW, fy = (n_out_rows + FH - 1) / FH;
            ctx->last_sigma_kernel = "sigma_kernel_f32";
            hipLaunchKernelGGL(sigma_kernel_f32, dim3(fx * fy), dim3(SBLOCK), 0, ctx->stream, A);
            LC_HIP_CHECK(hipGetLastError());
            return LC_OK;
        }
    }
    const int ntx = (nx + SW - 1) / SW, nty = (n_out_rows + SH - 1) / SH;
    ctx->last_sigma_kernel = sizeof(T) == 4 ? "sigma_kernel<float, float>"
                             : (fd_fp32_cast ? "sigma_kernel<double, float>" : "sigma_kernel<double, double>");
    if (fd_fp32_cast || sizeof(T) == 4)
        hipLaunchKernelGGL((sigma_kernel<T, float>), dim3(ntx * nty), dim3(SBLOCK), 0, ctx->stream, A);
    else
        hipLaunchKernelGGL((sigma_kernel<T, double>), dim3(ntx * nty), dim3(SBLOCK), 0, ctx->stream, A);
    LC_HIP_CHECK(hipGetLastError());
    return LC_OK;
}

// tools.fourth_order_derivative on its own (LCS/tools.py:190-245): index-space stencil on a 2-D array, numba typing
// (S differences, double scaling, S store).  dim 1: cyclic in longitude when isglobal (:220-228), else the one-sided
// difference / 2 on the two first and two last columns (:229-244), as dim 0 always does on its rows (:210-217).
template <typename S>
__global__ void index_derivative_kernel(const S *__restrict__ a, S *__restrict__ out, int ny, int nx, int dim, int isglobal) {
#pragma clang fp contract(off)
    const size_t n = (size_t)ny * nx;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        const int y = (int)(i / nx), x = (int)(i - (size_t)y * nx);
        double r;
        if (dim == 0) {
            if (y < 2)
                r = (double)(a[i + nx] - a[i]) / 2.0;
            else if (y >= ny - 2)
                r = (double)(a[i] - a[i - nx]) / 2.0;
            else
                r = (4.0 / 3.0) * (double)(a[i + nx] - a[i - nx]) / 2.0 -
                    (1.0 / 3.0) * (double)(a[i + 2 * (size_t)nx] - a[i - 2 * (size_t)nx]) / 4.0;
        } else if (isglobal) {
            const S *row = a + (size_t)y * nx;
            const int xp1 = (x + 1) % nx, xm1 = (x - 1 + nx) % nx, xp2 = (x + 2) % nx, xm2 = (x - 2 + nx) % nx;
            r = (4.0 / 3.0) * (double)(row[xp1] - row[xm1]) / 2.0 - (1.0 / 3.0) * (double)(row[xp2] - row[xm2]) / 4.0;
        } else {
            if (x < 2)
                r = (double)(a[i + 1] - a[i]) / 2.0;
            else if (x >= nx - 2)
                r = (double)(a[i] - a[i - 1]) / 2.0;
            else
                r = (4.0 / 3.0) * (double)(a[i + 1] - a[i - 1]) / 2.0 - (1.0 / 3.0) * (double)(a[i + 2] - a[i - 2]) / 4.0;
        }
        out[i] = (S)r;
    }
}

}  // namespace

extern "C" int lc_fourth_order_derivative(lc_ctx *ctx, const void *in_dev, int dtype, int ny, int nx, int dim,
                                          int isglobal, void *out_dev) {
    LC_REQUIRE(ctx, "lc_fourth_order_derivative: null context");
    LC_REQUIRE(dtype == LC_F32 || dtype == LC_F64, "lc_fourth_order_derivative: bad dtype %d", dtype);
    LC_REQUIRE(in_dev && out_dev && in_dev != out_dev, "lc_fourth_order_derivative: bad pointers");
    LC_REQUIRE(ny >= 5 && nx >= 5, "lc_fourth_order_derivative: grid %dx%d too small", ny, nx);
    LC_REQUIRE(dim == 0 || dim == 1, "Dim must be either 0 or 1.");
    LC_HIP_CHECK(hipSetDevice(ctx->device));
    const size_t n = (size_t)ny * nx;
    const int blocks = (int)((n + 255) / 256 < 8192 ? (n + 255) / 256 : 8192);
    if (dtype == LC_F32)
        hipLaunchKernelGGL(index_derivative_kernel<float>, dim3(blocks), dim3(256), 0, ctx->stream,
                           (const float *)in_dev, (float *)out_dev, ny, nx, dim, isglobal != 0);
    else
        hipLaunchKernelGGL(index_derivative_kernel<double>, dim3(blocks), dim3(256), 0, ctx->stream,
                           (const double *)in_dev, (double *)out_dev, ny, nx, dim, isglobal != 0);
    LC_HIP_CHECK(hipGetLastError());
    return LC_OK;
}

extern "C" int lc_sigma(lc_ctx *ctx, const void *x_dep, const void *y_dep, int dtype, int in_row0, int n_in_rows,
                        int nx, int ny_global, const void *seed_lat_dev, double dlat, double dlon, int fd_fp32_cast,
                        int tensor_layout, int out_row0, int n_out_rows, void *sigma_out) {
    LC_REQUIRE(ctx, "lc_sigma: null context");
    LC_REQUIRE(dtype == LC_F32 || dtype == LC_F64, "lc_sigma: bad dtype %d", dtype);
    LC_REQUIRE(x_dep && y_dep && seed_lat_dev && sigma_out, "lc_sigma: null pointer");
    LC_REQUIRE(nx >= 5 && ny_global >= 5, "lc_sigma: grid %dx%d too small for the 5-point stencil", ny_global, nx);
    LC_REQUIRE(tensor_layout == LC_LAYOUT_REFERENCE || tensor_layout == LC_LAYOUT_PHYSICAL, "lc_sigma: bad layout");
    LC_REQUIRE(n_in_rows >= 1 && in_row0 >= 0 && in_row0 + n_in_rows <= ny_global, "lc_sigma: bad input window");
    LC_REQUIRE(n_out_rows >= 1 && out_row0 >= in_row0 && out_row0 + n_out_rows <= in_row0 + n_in_rows,
               "lc_sigma: output rows outside the input window");
    // every output row needs r-2..r+2 unless the global one-sided rule covers it
    const int need_lo = out_row0 < 2 ? 0 : out_row0 - 2;
    const int last = out_row0 + n_out_rows - 1;
    const int need_hi = last >= ny_global - 2 ? ny_global - 1 : last + 2;
    LC_REQUIRE(in_row0 <= need_lo && in_row0 + n_in_rows - 1 >= need_hi,
               "lc_sigma: rows [%d,%d] need halo rows [%d,%d] but the input holds [%d,%d]", out_row0, last, need_lo,
               need_hi, in_row0, in_row0 + n_in_rows - 1);
    LC_HIP_CHECK(hipSetDevice(ctx->device));
    if (dtype == LC_F32)
        return sigma_impl<float>(ctx, x_dep, y_dep, in_row0, n_in_rows, nx, ny_global, seed_lat_dev, dlat, dlon,
                                 fd_fp32_cast, tensor_layout, out_row0, n_out_rows, sigma_out);
    return sigma_impl<double>(ctx, x_dep, y_dep, in_row0, n_in_rows, nx, ny_global, seed_lat_dev, dlat, dlon,
                              fd_fp32_cast, tensor_layout, out_row0, n_out_rows, sigma_out);
}

extern "C" int lc_flowmap_gradient(lc_ctx *ctx, const void *x_dep, const void *y_dep, int dtype, int ny, int nx,
                                   const void *seed_lat_dev, double dlat, double dlon, int fd_fp32_cast,
                                   void *def_tensor_out) {
    LC_REQUIRE(ctx, "lc_flowmap_gradient: null context");
    LC_REQUIRE(dtype == LC_F32 || dtype == LC_F64, "lc_flowmap_gradient: bad dtype %d", dtype);
    LC_REQUIRE(x_dep && y_dep && seed_lat_dev && def_tensor_out, "lc_flowmap_gradient: null pointer");
    LC_REQUIRE(nx >= 5 && ny >= 5, "lc_flowmap_gradient: grid %dx%d too small for the 5-point stencil", ny, nx);
    LC_HIP_CHECK(hipSetDevice(ctx->device));
    if (dtype == LC_F32)
        return sigma_impl<float>(ctx, x_dep, y_dep, 0, ny, nx, ny, seed_lat_dev, dlat, dlon, fd_fp32_cast,
                                 LC_LAYOUT_REFERENCE, 0, ny, nullptr, def_tensor_out);
    return sigma_impl<double>(ctx, x_dep, y_dep, 0, ny, nx, ny, seed_lat_dev, dlat, dlon, fd_fp32_cast,
                              LC_LAYOUT_REFERENCE, 0, ny, nullptr, def_tensor_out);
}
