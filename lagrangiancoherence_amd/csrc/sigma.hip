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
// pandas MultiIndex; here a workgroup computes X,Y,Z once for its tile plus a 2-cell
// halo into LDS (as float when fd_fp32_cast, Q11), differences from LDS and solves the
// 2x2 Gram eigenproblem in closed form.  HBM traffic: read x_dep,y_dep once (+halo
// re-reads served by L2), write sigma once.
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
constexpr int FW = 64, FH = 16;
constexpr int FLW = FW + 2 * HALO, FLH = FH + 2 * HALO;

__device__ __forceinline__ void fast_sincosf(float a, float *sn, float *cs) {
    if (!(fabsf(a) < 64.0f)) {  // out of the bounded range (or NaN): library path
        sincosf(a, sn, cs);
        return;
    }
    const float n = rintf(a * 0.636619772367581343f);  // 2/pi
    float r = fmaf(n, -1.5703125f, a);
    r = fmaf(n, -4.837512969970703125e-4f, r);
    r = fmaf(n, -7.54978995489188216e-8f, r);
    const float z = r * r;
    const float sp = fmaf(fmaf(fmaf(-1.9515295891e-4f, z, 8.3321608736e-3f), z, -1.6666654611e-1f), z * r, r);
    const float cp = fmaf(fmaf(fmaf(2.443315711809948e-5f, z, -1.388731625493765e-3f), z, 4.166664568298827e-2f),
                          z * z, fmaf(-0.5f, z, 1.0f));
    const int q = (int)n;
    const float s1 = (q & 1) ? cp : sp, c1 = (q & 1) ? sp : cp;
    *sn = (q & 2) ? -s1 : s1;
    *cs = ((q + 1) & 2) ? -c1 : c1;
}

__global__ void __launch_bounds__(SBLOCK) sigma_kernel_f32(const SigmaArgs<float> A) {
    __shared__ float sX[FLH][FLW + 1];
    __shared__ float sY[FLH][FLW + 1];
    __shared__ float sZ[FLH][FLW + 1];
    __shared__ float s_inv_dx[FH];
    const int ntx = (A.nx + FW - 1) / FW;
    const int tyi = blockIdx.x / ntx, txi = blockIdx.x - tyi * ntx;
    const int gy0 = A.out_row0 + tyi * FH;
    const int gx0 = txi * FW;
    const float D2R = 3.141592653589793f / 180.0f;
    const float R = 6371000.0f;

    if (threadIdx.x < FH) {
        const int gy = gy0 + (int)threadIdx.x;
        float v = 0.0f;
        if (gy < A.out_row0 + A.n_out_rows) {
            const float latr = (A.seed_lat[gy - A.in_row0] * 3.141592653589793f) / 180.0f;  // tools.py:254
            v = 1.0f / ((((3.141592653589793f / 180.0f) * A.dlon) * R) * cosf(latr));      // tools.py:255
        }
        s_inv_dx[threadIdx.x] = v;
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
            const float lon = A.x_dep[o] * D2R;             // LCS.py:195
            const float lat = (A.y_dep[o] - 90.0f) * D2R;   // LCS.py:196
            float sl, cl, so, co;
            fast_sincosf(lat, &sl, &cl);
            fast_sincosf(lon, &so, &co);
            const float rs = R * sl;
            vx = rs * co;  // LCS.py:197
            vy = rs * so;  // LCS.py:198
            vz = R * cl;   // LCS.py:199
        }
        sX[ly][lx] = vx;
        sY[ly][lx] = vy;
        sZ[ly][lx] = vz;
    }
    __syncthreads();

    const float inv_dy = 1.0f / (((3.141592653589793f / 180.0f) * A.dlat) * R);  // tools.py:256
    const float W1 = 2.0f / 3.0f, W2 = -1.0f / 12.0f;  // (4/3)/2 and -(1/3)/4 of tools.py:204-207
    for (int i = threadIdx.x; i < FW * FH; i += SBLOCK) {
        const int oy = i / FW, ox = i - oy * FW;
        const int gy = gy0 + oy, gx = gx0 + ox;
        if (gy >= A.out_row0 + A.n_out_rows || gx >= A.nx) continue;
        const int ly = oy + HALO, lx = ox + HALO;
        const float inv_dx = s_inv_dx[oy];
        auto ddx = [&](float(*a)[FLW + 1]) -> float {
            return fmaf(W1, a[ly][lx + 1] - a[ly][lx - 1], W2 * (a[ly][lx + 2] - a[ly][lx - 2])) * inv_dx;
        };
        auto ddy = [&](float(*a)[FLW + 1]) -> float {
            float d;
            if (gy < 2)
                d = 0.5f * (a[ly + 1][lx] - a[ly][lx]);  // tools.py:210-213
            else if (gy >= A.ny_global - 2)
                d = 0.5f * (a[ly][lx] - a[ly - 1][lx]);  // tools.py:214-217
            else
                d = fmaf(W1, a[ly + 1][lx] - a[ly - 1][lx], W2 * (a[ly + 2][lx] - a[ly - 2][lx]));
            return d * inv_dy;
        };
        const float a_ = ddx(sX), b_ = ddy(sX), c_ = ddx(sY), d_ = ddy(sY), e_ = ddx(sZ), f_ = ddy(sZ);
        const size_t oidx = (size_t)(gy - A.out_row0) * A.nx + gx;
        float p, q, r;
        if (A.layout == LC_LAYOUT_REFERENCE) {  // LCS.py:153 (Q13)
            p = a_ * a_ + b_ * b_ + c_ * c_;
            q = d_ * d_ + e_ * e_ + f_ * f_;
            r = a_ * d_ + b_ * e_ + c_ * f_;
        } else {
            p = a_ * a_ + c_ * c_ + e_ * e_;
            q = b_ * b_ + d_ * d_ + f_ * f_;
            r = a_ * b_ + c_ * d_ + e_ * f_;
        }
        const float dpq = p - q;
        const float disc = sqrtf(fmaf(dpq, dpq, 4.0f * r * r));
        A.sigma[oidx] = sqrtf(0.5f * ((p + q) + disc));
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
        if (!tensor_out) {  // float, sigma only: the fast kernel
            const int fx = (nx + FW - 1) / FW, fy = (n_out_rows + FH - 1) / FH;
            hipLaunchKernelGGL(sigma_kernel_f32, dim3(fx * fy), dim3(SBLOCK), 0, ctx->stream, A);
            LC_HIP_CHECK(hipGetLastError());
            return LC_OK;
        }
    }
    const int ntx = (nx + SW - 1) / SW, nty = (n_out_rows + SH - 1) / SH;
    if (fd_fp32_cast || sizeof(T) == 4)
        hipLaunchKernelGGL((sigma_kernel<T, float>), dim3(ntx * nty), dim3(SBLOCK), 0, ctx->stream, A);
    else
        hipLaunchKernelGGL((sigma_kernel<T, double>), dim3(ntx * nty), dim3(SBLOCK), 0, ctx->stream, A);
    LC_HIP_CHECK(hipGetLastError());
    return LC_OK;
}

// tools.fourth_order_derivative on its own (LCS/tools.py:190-228, isglobal branch): index-space
// stencil on a 2-D array, numba typing (S differences, double scaling, S store).
template <typename S>
__global__ void index_derivative_kernel(const S *__restrict__ a, S *__restrict__ out, int ny, int nx, int dim) {
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
        } else {
            const S *row = a + (size_t)y * nx;
            const int xp1 = (x + 1) % nx, xm1 = (x - 1 + nx) % nx, xp2 = (x + 2) % nx, xm2 = (x - 2 + nx) % nx;
            r = (4.0 / 3.0) * (double)(row[xp1] - row[xm1]) / 2.0 - (1.0 / 3.0) * (double)(row[xp2] - row[xm2]) / 4.0;
        }
        out[i] = (S)r;
    }
}

}  // namespace

extern "C" int lc_fourth_order_derivative(lc_ctx *ctx, const void *in_dev, int dtype, int ny, int nx, int dim,
                                          void *out_dev) {
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
                           (const float *)in_dev, (float *)out_dev, ny, nx, dim);
    else
        hipLaunchKernelGGL(index_derivative_kernel<double>, dim3(blocks), dim3(256), 0, ctx->stream,
                           (const double *)in_dev, (double *)out_dev, ny, nx, dim);
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
