// Global pre-processing of LCS.__call__ (isglobal=True), SURVEY.md section 8f rank 2:
//
//   lc_regrid_common_grid   LCS/LCS.py:107-114   u.interp(linear) onto the fixed 0.5 degree grid, targets outside
//                                                the source range filled from u.reindex(method='nearest')
//   lc_spectral_truncate    LCS/LCS.py:115-118   windspharm VectorWind(u, v).truncate(f, truncation=T)
//
// Regrid: the index / weight tables depend on the coordinates only and are built on the host; ONE fused kernel
// does latitude lerp -> longitude lerp -> nearest fill with scipy.interpolate.interp1d's operation order
// (slope = (y_hi - y_lo) / (x_hi - x_lo) in the field's dtype promoted by the float64 coordinates;
// y = slope * (x_new - x_lo) + y_lo; latitude pass first, as xarray's interp does for two 1-D coordinates).
//
// Truncation: a dense linear operator on SPHEREPACK's equally spaced grid theta_i = i pi / (nlat - 1):
// zonal DFT restricted to m <= T  ->  per m an (nlat x nlat) matrix P[m] = synthesis . analysis  ->  inverse
// DFT.  P[m]: exact integral of the trigonometric interpolant of the m-th zonal coefficient (cosine series for
// even m, sine series for odd m -- Swarztrauber's "Z functions") against Pbar^m_n sin(theta), n <= T, evaluated
// with Gauss-Legendre nodes in cos(theta).  Operators are built on the host in double, once per (nlat, nlon, T)
// and cached on the context; the three products are hand-written float64 kernels (plain FMA: the matrices are
// a few hundred on a side, the whole truncation of a 97-level u,v series is ~10 GFLOP).
// PARITY: restates the published algorithm; NOT pinned against pyspharm (not installable here).
#include <cmath>
#include <vector>

#include "lcs_common.h"

struct lc_trunc_cache {
    int nlat = 0, nlon = 0, T = -1, gridtype = -1;
    double *P = nullptr;   // [T+1][nlat][nlat]
    double *F = nullptr;   // [nlon][2(T+1)]   forward:  cos | sin
    double *G = nullptr;   // [2(T+1)][nlon]   inverse weights
};

void lc_trunc_cache_free(lc_trunc_cache *c) {
    if (!c) return;
    if (c->P) (void)hipFree(c->P);
    if (c->F) (void)hipFree(c->F);
    if (c->G) (void)hipFree(c->G);
    delete c;
}

namespace {

// ---------------------------------------------------------------------------------------------
// regrid
// ---------------------------------------------------------------------------------------------
struct AxisPlan {
    std::vector<int> lo, near;
    std::vector<double> t, d;       // x_new - x_lo, x_hi - x_lo
    std::vector<unsigned char> inside;
};

// np.searchsorted(src, v, side='left')
int lower_bound_idx(const double *src, int n, double v) {
    int a = 0, b = n;
    while (a < b) {
        const int m = (a + b) / 2;
        if (src[m] < v)
            a = m + 1;
        else
            b = m;
    }
    return a;
}

AxisPlan axis_plan(const double *src, int n, const double *dst, int nd) {
    AxisPlan p;
    p.lo.resize(nd);
    p.near.resize(nd);
    p.t.resize(nd);
    p.d.resize(nd);
    p.inside.resize(nd);
    for (int k = 0; k < nd; ++k) {
        const int s = lower_bound_idx(src, n, dst[k]);
        int hi = s < 1 ? 1 : (s > n - 1 ? n - 1 : s);  // scipy interp1d._call_linear: clip(searchsorted, 1, n-1)
        p.lo[k] = hi - 1;
        p.t[k] = dst[k] - src[hi - 1];
        p.d[k] = src[hi] - src[hi - 1];
        p.inside[k] = dst[k] >= src[0] && dst[k] <= src[n - 1];
        // pandas Index.get_indexer(method='nearest'), increasing index: left only if strictly closer
        const int r = s > n - 1 ? n - 1 : s, l = r - 1 < 0 ? 0 : r - 1;
        p.near[k] = std::fabs(dst[k] - src[l]) < std::fabs(src[r] - dst[k]) ? l : r;
    }
    return p;
}

struct RegridArgs {
    int nt, ny_s, nx_s, ny_d, nx_d;
    const int *jlo, *jn, *ilo, *in_;
    const double *ty, *dy, *tx, *dx;
    const unsigned char *in_y, *in_x;
};

template <typename T>
__global__ void regrid_kernel(const T *__restrict__ src, const RegridArgs A, double *__restrict__ out) {
#pragma clang fp contract(off)
    const size_t n = (size_t)A.nt * A.ny_d * A.nx_d;
    for (size_t o = blockIdx.x * (size_t)blockDim.x + threadIdx.x; o < n; o += (size_t)gridDim.x * blockDim.x) {
        const int i = (int)(o % A.nx_d);
        const size_t r = o / A.nx_d;
        const int j = (int)(r % A.ny_d), t = (int)(r / A.ny_d);
        const T *lvl = src + (size_t)t * A.ny_s * A.nx_s;
        const int jl = A.jlo[j], il = A.ilo[i];
        const double ty = A.ty[j], dy = A.dy[j];
        auto lat_lerp = [&](int ii) -> double {  // tmp[t, j, ii] of the latitude pass
            const T a = lvl[(size_t)jl * A.nx_s + ii], b = lvl[(size_t)(jl + 1) * A.nx_s + ii];
            const T diff = b - a;                 // y_hi - y_lo in the field's dtype (numpy), then promoted
            return ((double)diff / dy) * ty + (double)a;
        };
        const double x_lo = lat_lerp(il), x_hi = lat_lerp(il + 1);
        const double v = ((x_hi - x_lo) / A.dx[i]) * A.tx[i] + x_lo;
        const bool ok = A.in_y[j] && A.in_x[i] && !(v != v);
        out[o] = ok ? v : (double)lvl[(size_t)A.jn[j] * A.nx_s + A.in_[i]];  // LCS.py:109,113
    }
}

// ---------------------------------------------------------------------------------------------
// truncation operators (host, double)
// ---------------------------------------------------------------------------------------------
// Pbar^m_n(x), n = m..nmax, orthonormal on [-1, 1]; out[(n-m)*nx + q]
void legendre_normalized(int m, int nmax, const std::vector<double> &x, std::vector<double> &out) {
    const size_t nx = x.size();
    out.assign((size_t)(nmax - m + 1) * nx, 0.0);
    for (size_t q = 0; q < nx; ++q) {
        const double s = std::sqrt(std::fmax(0.0, 1.0 - x[q] * x[q]));
        double pmm = std::sqrt(0.5);
        for (int k = 1; k <= m; ++k) pmm = -std::sqrt((2 * k + 1) / (2.0 * k)) * s * pmm;
        out[q] = pmm;
        if (nmax > m) out[nx + q] = std::sqrt(2 * m + 3.0) * x[q] * pmm;
        for (int n = m + 2; n <= nmax; ++n) {
            const double a = std::sqrt((4.0 * n * n - 1.0) / ((double)n * n - (double)m * m));
            const double b = std::sqrt((((double)n - 1.0) * (n - 1.0) - (double)m * m) / (4.0 * (n - 1.0) * (n - 1.0) - 1.0));
            out[(size_t)(n - m) * nx + q] = a * (x[q] * out[(size_t)(n - m - 1) * nx + q] - b * out[(size_t)(n - m - 2) * nx + q]);
        }
    }
}

// Gauss-Legendre nodes / weights on [-1, 1] (Newton on P_n from the Chebyshev guess)
void gauss_legendre(int n, std::vector<double> &x, std::vector<double> &w) {
    x.resize(n);
    w.resize(n);
    const double pi = 3.14159265358979323846;
    for (int i = 0; i < (n + 1) / 2; ++i) {
        double z = std::cos(pi * (i + 0.75) / (n + 0.5)), pp = 1.0;
        for (int it = 0; it < 100; ++it) {
            double p1 = 1.0, p2 = 0.0;
            for (int j = 1; j <= n; ++j) {
                const double p3 = p2;
                p2 = p1;
                p1 = ((2.0 * j - 1.0) * z * p2 - (j - 1.0) * p3) / j;
            }
            pp = n * (z * p1 - p2) / (z * z - 1.0);
            const double z1 = z;
            z = z1 - p1 / pp;
            if (std::fabs(z - z1) < 1e-16) break;
        }
        x[i] = -z;
        x[n - 1 - i] = z;
        w[i] = w[n - 1 - i] = 2.0 / ((1.0 - z * z) * pp * pp);
    }
}

// Gaussian grid (windspharm gridtype 'gaussian' -> SPHEREPACK shags/shsgs): the rows sit on the Gauss-Legendre nodes
// x_i = sin(lat_i) and the analysis is that quadrature, a^m_n = sum_j w_j Pbar^m_n(x_j) g_m(x_j) -- exact for
// band-limited fields -- so P[m][i][j] = sum_{n=m..T} Pbar^m_n(x_i) Pbar^m_n(x_j) w_j.
void build_projectors_gaussian(int nlat, int T, std::vector<double> &P) {
    std::vector<double> x, w, S;
    gauss_legendre(nlat, x, w);                 // ascending in x = south -> north
    std::vector<double> xd(nlat), wd(nlat);     // row 0 = northernmost
    for (int i = 0; i < nlat; ++i) {
        xd[i] = x[nlat - 1 - i];
        wd[i] = w[nlat - 1 - i];
    }
    P.assign((size_t)(T + 1) * nlat * nlat, 0.0);
    for (int m = 0; m <= T; ++m) {
        const int nn = T - m + 1;
        legendre_normalized(m, T, xd, S);       // (nn, nlat)
        double *Pm = P.data() + (size_t)m * nlat * nlat;
        for (int i = 0; i < nlat; ++i)
            for (int a = 0; a < nn; ++a) {
                const double s = S[(size_t)a * nlat + i];
                const double *Sa = S.data() + (size_t)a * nlat;
                double *Pr = Pm + (size_t)i * nlat;
                for (int j = 0; j < nlat; ++j) Pr[j] += s * Sa[j] * wd[j];
            }
    }
}

// P[m] (row-major nlat x nlat, row 0 = north pole) for m = 0..T, SPHEREPACK's equally spaced grid
void build_projectors(int nlat, int T, std::vector<double> &P) {
    const int N = nlat - 1;
    const double pi = 3.14159265358979323846;
    std::vector<double> cth(nlat), xq, wq;
    for (int i = 0; i < nlat; ++i) cth[i] = std::cos(i * pi / N);
    gauss_legendre(2 * N, xq, wq);
    const int Q = (int)xq.size();
    std::vector<double> tq(Q);
    for (int q = 0; q < Q; ++q) tq[q] = std::acos(xq[q]);
    P.assign((size_t)(T + 1) * nlat * nlat, 0.0);
    std::vector<double> S, Pq, basis, integ, A;
    for (int m = 0; m <= T; ++m) {
        const int nn = T - m + 1;
        legendre_normalized(m, T, cth, S);   // (nn, nlat)  synthesis
        legendre_normalized(m, T, xq, Pq);   // (nn, Q)
        const bool even = m % 2 == 0;
        const int k0 = even ? 0 : 1, nk = even ? N + 1 : N - 1;
        basis.assign((size_t)nk * Q, 0.0);   // cos(k theta_q) | sin(k theta_q)
        for (int k = 0; k < nk; ++k)
            for (int q = 0; q < Q; ++q) basis[(size_t)k * Q + q] = even ? std::cos((k + k0) * tq[q]) : std::sin((k + k0) * tq[q]);
        integ.assign((size_t)nn * nk, 0.0);  // integral basis_k Pbar^m_n sin(theta) dtheta
        for (int a = 0; a < nn; ++a)
            for (int k = 0; k < nk; ++k) {
                double s = 0.0;
                for (int q = 0; q < Q; ++q) s += Pq[(size_t)a * Q + q] * wq[q] * basis[(size_t)k * Q + q];
                integ[(size_t)a * nk + k] = s;
            }
        A.assign((size_t)nn * nlat, 0.0);    // analysis: integ . B, B = the interpolant's coefficient map
        for (int k = 0; k < nk; ++k) {
            const int kk = k + k0;
            for (int i = 0; i < nlat; ++i) {
                double b = (2.0 / N) * (even ? std::cos((double)kk * i * pi / N) : std::sin((double)kk * i * pi / N));
                if (even) {
                    if (i == 0 || i == N) b *= 0.5;
                    if (kk == 0 || kk == N) b *= 0.5;
                }
                if (b == 0.0) continue;
                for (int a = 0; a < nn; ++a) A[(size_t)a * nlat + i] += integ[(size_t)a * nk + k] * b;
            }
        }
        double *Pm = P.data() + (size_t)m * nlat * nlat;
        for (int i = 0; i < nlat; ++i)
            for (int a = 0; a < nn; ++a) {
                const double s = S[(size_t)a * nlat + i];
                const double *Ar = A.data() + (size_t)a * nlat;
                double *Pr = Pm + (size_t)i * nlat;
                for (int j = 0; j < nlat; ++j) Pr[j] += s * Ar[j];
            }
    }
}

// ---------------------------------------------------------------------------------------------
// truncation kernels (double)
// ---------------------------------------------------------------------------------------------
constexpr int DFT_ROWS = 4;   // field rows per block in the forward DFT

// X[b][i][c] = sum_j g[b][i][j] F[j][c],  g = the field with latitude flipped to north -> south
template <typename T>
__global__ void __launch_bounds__(256) dft_forward_kernel(const T *__restrict__ f, int nrows_total, int nlat, int nlon, int C,
                                                           const double *__restrict__ F, double *__restrict__ X, int rows_per_block) {
    extern __shared__ double s_row[];  // rows_per_block (<= DFT_ROWS: as many as fit 64 KB of LDS) x nlon
    const int row0 = blockIdx.x * rows_per_block;
    for (int e = threadIdx.x; e < rows_per_block * nlon; e += blockDim.x) {
        const int r = e / nlon, j = e - r * nlon, row = row0 + r;
        double v = 0.0;
        if (row < nrows_total) {
            const int b = row / nlat, i = row - b * nlat;
            v = (double)f[((size_t)b * nlat + (nlat - 1 - i)) * nlon + j];
        }
        s_row[e] = v;
    }
    __syncthreads();
    const int r = threadIdx.x / 64;
    if (r >= rows_per_block || row0 + r >= nrows_total) return;
    const double *g = s_row + (size_t)r * nlon;
    for (int c = threadIdx.x % 64; c < C; c += 64) {  // any truncation: the wave walks the 2 (T + 1) columns 64 at a time
        double acc = 0.0;
        for (int j = 0; j < nlon; ++j) acc = fma(g[j], F[(size_t)j * C + c], acc);
        X[(size_t)(row0 + r) * C + c] = acc;
    }
}

// H[b][i][c] = sum_j P[m(c)][i][j] X[b][j][c]: per m one (nlat x nlat) . (nlat x 2 nb) product, 32 x 32 tiles
constexpr int PT = 32;
__global__ void __launch_bounds__(256) project_kernel(const double *__restrict__ P, const double *__restrict__ X, int nb, int nlat,
                                                      int T1, double *__restrict__ H) {
    __shared__ double sP[PT][PT + 1], sX[PT][PT + 1];
    const int m = blockIdx.z, C = 2 * T1;
    const int i0 = blockIdx.y * PT, n0 = blockIdx.x * PT;  // n indexes (b, part): column n -> b = n / 2, c = m + (n & 1) * T1
    const int tx = threadIdx.x % PT, ty = threadIdx.x / PT;  // 32 x 8 threads, 4 outputs each
    const double *Pm = P + (size_t)m * nlat * nlat;
    double acc[4] = {0, 0, 0, 0};
    for (int k0 = 0; k0 < nlat; k0 += PT) {
        for (int e = threadIdx.x; e < PT * PT; e += 256) {
            const int a = e / PT, b = e % PT;
            sP[a][b] = (i0 + a < nlat && k0 + b < nlat) ? Pm[(size_t)(i0 + a) * nlat + k0 + b] : 0.0;
            const int n = n0 + b, j = k0 + a;
            sX[a][b] = (n < 2 * nb && j < nlat) ? X[((size_t)(n >> 1) * nlat + j) * C + m + (n & 1) * T1] : 0.0;
        }
        __syncthreads();
#pragma unroll 8
        for (int k = 0; k < PT; ++k) {
            const double xv = sX[k][tx];
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = fma(sP[ty + 8 * q][k], xv, acc[q]);
        }
        __syncthreads();
    }
    const int n = n0 + tx;
    if (n >= 2 * nb) return;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int i = i0 + ty + 8 * q;
        if (i < nlat) H[((size_t)(n >> 1) * nlat + i) * C + m + (n & 1) * T1] = acc[q];
    }
}

// out[b][nlat-1-i][l] = sum_c H[b][i][c] G[c][l]   (back to latitude ascending, field dtype)
template <typename T>
__global__ void __launch_bounds__(256) dft_inverse_kernel(const double *__restrict__ H, int nlat, int nlon, int C,
                                                           const double *__restrict__ G, T *__restrict__ out) {
    extern __shared__ double sh[];  // C spectral coefficients of the row
    const int row = blockIdx.x;  // (b, i)
    for (int c = threadIdx.x; c < C; c += blockDim.x) sh[c] = H[(size_t)row * C + c];
    __syncthreads();
    const int b = row / nlat, i = row - b * nlat;
    T *dst = out + ((size_t)b * nlat + (nlat - 1 - i)) * nlon;
    for (int l = threadIdx.x; l < nlon; l += blockDim.x) {
        double acc = 0.0;
        for (int c = 0; c < C; ++c) acc = fma(sh[c], G[(size_t)c * nlon + l], acc);
        dst[l] = (T)acc;
    }
}

int ensure_operators(lc_ctx *ctx, int nlat, int nlon, int T, int gridtype) {
    lc_trunc_cache *c = ctx->trunc;
    if (c && c->nlat == nlat && c->nlon == nlon && c->T == T && c->gridtype == gridtype) return LC_OK;
    lc_trunc_cache_free(c);
    ctx->trunc = nullptr;
    c = new lc_trunc_cache;
    const int T1 = T + 1, C = 2 * T1;
    std::vector<double> P, F((size_t)nlon * C), G((size_t)C * nlon);
    if (gridtype == LC_GRID_GAUSSIAN)
        build_projectors_gaussian(nlat, T, P);
    else
        build_projectors(nlat, T, P);
    const double pi = 3.14159265358979323846;
    for (int j = 0; j < nlon; ++j)
        for (int m = 0; m < T1; ++m) {
            const double ang = 2.0 * pi * ((double)j * m) / nlon;
            const double cs = std::cos(ang), sn = std::sin(ang), scale = (m == 0 ? 1.0 : 2.0) / nlon;
            F[(size_t)j * C + m] = cs;
            F[(size_t)j * C + T1 + m] = sn;
            G[(size_t)m * nlon + j] = scale * cs;
            G[(size_t)(T1 + m) * nlon + j] = scale * sn;
        }
    hipError_t e = hipMalloc((void **)&c->P, P.size() * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void **)&c->F, F.size() * sizeof(double));
    if (e == hipSuccess) e = hipMalloc((void **)&c->G, G.size() * sizeof(double));
    if (e == hipSuccess) e = hipMemcpy(c->P, P.data(), P.size() * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(c->F, F.data(), F.size() * sizeof(double), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMemcpy(c->G, G.data(), G.size() * sizeof(double), hipMemcpyHostToDevice);
    if (e != hipSuccess) {
        lc_trunc_cache_free(c);
        lc_set_error("lc_spectral_truncate: operator upload failed: %s", hipGetErrorString(e));
        return e == hipErrorOutOfMemory ? LC_ENOMEM : LC_EHIP;
    }
    c->nlat = nlat;
    c->nlon = nlon;
    c->T = T;
    c->gridtype = gridtype;
    ctx->trunc = c;
    return LC_OK;
}

template <typename T>
int truncate_impl(lc_ctx *ctx, const T *f, int nb, int nlat, int nlon, int Tr, T *out) {
    const int T1 = Tr + 1, C = 2 * T1;
    hipStream_t st = ctx->stream;
    double *X = nullptr, *H = nullptr;
    const size_t n = (size_t)nb * nlat * C;
    LC_HIP_CHECK(hipMallocAsync((void **)&X, 2 * n * sizeof(double), st));
    H = X + n;
    const int rows = nb * nlat;
    int rpb = (int)((size_t)64 * 1024 / ((size_t)nlon * sizeof(double)));  // field rows a block stages in LDS
    rpb = rpb > DFT_ROWS ? DFT_ROWS : rpb;
    hipLaunchKernelGGL((dft_forward_kernel<T>), dim3((rows + rpb - 1) / rpb), dim3(256),
                       (size_t)rpb * nlon * sizeof(double), st, f, rows, nlat, nlon, C, ctx->trunc->F, X, rpb);
    hipLaunchKernelGGL(project_kernel, dim3((2 * nb + PT - 1) / PT, (nlat + PT - 1) / PT, T1), dim3(256), 0, st,
                       ctx->trunc->P, X, nb, nlat, T1, H);
    hipLaunchKernelGGL((dft_inverse_kernel<T>), dim3(rows), dim3(256), (size_t)C * sizeof(double), st, H, nlat, nlon, C, ctx->trunc->G, out);
    const hipError_t le = hipGetLastError();
    (void)hipFreeAsync(X, st);
    LC_HIP_CHECK(le);
    return LC_OK;
}

}  // namespace

extern "C" int lc_regrid_common_grid(lc_ctx *ctx, const void *src_dev, int dtype, int nt, int ny_s, int nx_s,
                                     const double *src_lat_host, const double *src_lon_host, const double *dst_lat_host,
                                     int ny_d, const double *dst_lon_host, int nx_d, double *out_dev) {
    LC_REQUIRE(ctx, "lc_regrid_common_grid: null context");
    LC_REQUIRE(dtype == LC_F32 || dtype == LC_F64, "lc_regrid_common_grid: bad dtype %d", dtype);
    LC_REQUIRE(src_dev && out_dev && src_lat_host && src_lon_host && dst_lat_host && dst_lon_host,
               "lc_regrid_common_grid: null pointer");
    LC_REQUIRE(nt >= 1 && ny_s >= 2 && nx_s >= 2 && ny_d >= 1 && nx_d >= 1, "lc_regrid_common_grid: bad sizes");
    for (int k = 1; k < ny_s; ++k) LC_REQUIRE(src_lat_host[k] > src_lat_host[k - 1], "lc_regrid_common_grid: latitude must ascend");
    for (int k = 1; k < nx_s; ++k) LC_REQUIRE(src_lon_host[k] > src_lon_host[k - 1], "lc_regrid_common_grid: longitude must ascend");
    LC_HIP_CHECK(hipSetDevice(ctx->device));
    const AxisPlan py = axis_plan(src_lat_host, ny_s, dst_lat_host, ny_d), px = axis_plan(src_lon_host, nx_s, dst_lon_host, nx_d);
    // one upload: ints | doubles | bytes
    const size_t ni = 2 * ((size_t)ny_d + nx_d), nd = ni, nbts = (size_t)ny_d + nx_d;
    std::vector<char> host(ni * sizeof(int) + nd * sizeof(double) + nbts);
    double *hd = (double *)host.data();  // doubles first (alignment)
    int *hi = (int *)(host.data() + nd * sizeof(double));
    unsigned char *hb = (unsigned char *)(host.data() + nd * sizeof(double) + ni * sizeof(int));
    std::copy(py.t.begin(), py.t.end(), hd);
    std::copy(py.d.begin(), py.d.end(), hd + ny_d);
    std::copy(px.t.begin(), px.t.end(), hd + 2 * ny_d);
    std::copy(px.d.begin(), px.d.end(), hd + 2 * ny_d + nx_d);
    std::copy(py.lo.begin(), py.lo.end(), hi);
    std::copy(py.near.begin(), py.near.end(), hi + ny_d);
    std::copy(px.lo.begin(), px.lo.end(), hi + 2 * ny_d);
    std::copy(px.near.begin(), px.near.end(), hi + 2 * ny_d + nx_d);
    std::copy(py.inside.begin(), py.inside.end(), hb);
    std::copy(px.inside.begin(), px.inside.end(), hb + ny_d);
    char *dev = nullptr;
    LC_HIP_CHECK(hipMallocAsync((void **)&dev, host.size(), ctx->stream));
    hipError_t e = hipMemcpyAsync(dev, host.data(), host.size(), hipMemcpyHostToDevice, ctx->stream);
    if (e == hipSuccess) e = hipStreamSynchronize(ctx->stream);  // `host` is pageable and about to go out of scope
    if (e != hipSuccess) {
        (void)hipFreeAsync(dev, ctx->stream);
        LC_HIP_CHECK(e);
    }
    RegridArgs A;
    A.nt = nt;
    A.ny_s = ny_s;
    A.nx_s = nx_s;
    A.ny_d = ny_d;
    A.nx_d = nx_d;
    const double *dd = (const double *)dev;
    const int *di = (const int *)(dev + nd * sizeof(double));
    const unsigned char *db = (const unsigned char *)(dev + nd * sizeof(double) + ni * sizeof(int));
    A.ty = dd;
    A.dy = dd + ny_d;
    A.tx = dd + 2 * ny_d;
    A.dx = dd + 2 * ny_d + nx_d;
    A.jlo = di;
    A.jn = di + ny_d;
    A.ilo = di + 2 * ny_d;
    A.in_ = di + 2 * ny_d + nx_d;
    A.in_y = db;
    A.in_x = db + ny_d;
    const size_t n = (size_t)nt * ny_d * nx_d;
    const int blocks = (int)((n + 255) / 256 < 16384 ? (n + 255) / 256 : 16384);
    if (dtype == LC_F32)
        hipLaunchKernelGGL((regrid_kernel<float>), dim3(blocks), dim3(256), 0, ctx->stream, (const float *)src_dev, A, out_dev);
    else
        hipLaunchKernelGGL((regrid_kernel<double>), dim3(blocks), dim3(256), 0, ctx->stream, (const double *)src_dev, A, out_dev);
    const hipError_t le = hipGetLastError();
    (void)hipFreeAsync(dev, ctx->stream);
    LC_HIP_CHECK(le);
    return LC_OK;
}

// windspharm's grid inspection (windspharm/tools? `inspect_gridtype`, reached from VectorWind at LCS/LCS.py:116):
// latitudes equally spaced to 5e-4 degrees must equal the global equally spaced grid of that size (poles included for
// an odd count, half a spacing away from them for an even one) -> 'regular'; otherwise they must equal the Gaussian
// latitudes of that size to 5e-4 degrees -> 'gaussian'; anything else is an error.  Host arithmetic only.
extern "C" int lc_inspect_gridtype(const double *lat_ascending, int nlat, int *gridtype_out) {
    LC_REQUIRE(lat_ascending && gridtype_out && nlat >= 3, "lc_inspect_gridtype: bad arguments");
    const double tol = 5e-4, d0 = std::fabs(lat_ascending[1] - lat_ascending[0]);
    bool equal = true;
    for (int i = 1; i < nlat; ++i) equal = equal && std::fabs(std::fabs(lat_ascending[i] - lat_ascending[i - 1]) - d0) < tol;
    if (equal) {
        const double first = nlat % 2 ? -90.0 : -90.0 + 90.0 / nlat, last = -first;
        for (int i = 0; i < nlat; ++i) {
            const double want = first + (last - first) * i / (nlat - 1);
            if (std::fabs(lat_ascending[i] - want) > tol) {
                lc_set_error("equally-spaced latitudes are invalid (they may be non-global): row %d is %g, a global grid of %d rows has %g",
                             i, lat_ascending[i], nlat, want);
                return LC_EINVAL;
            }
        }
        *gridtype_out = LC_GRID_REGULAR;
        return LC_OK;
    }
    std::vector<double> x, w;
    gauss_legendre(nlat, x, w);  // ascending
    for (int i = 0; i < nlat; ++i) {
        const double want = std::asin(x[i]) * 180.0 / 3.14159265358979323846;
        if (std::fabs(lat_ascending[i] - want) > tol) {
            lc_set_error("latitudes are neither equally-spaced or Gaussian (row %d is %g, the Gaussian grid of %d rows has %g)", i,
                         lat_ascending[i], nlat, want);
            return LC_EINVAL;
        }
    }
    *gridtype_out = LC_GRID_GAUSSIAN;
    return LC_OK;
}

extern "C" int lc_spectral_truncate(lc_ctx *ctx, const void *f_dev, int dtype, int nbatch, int nlat, int nlon, int truncation,
                                    int gridtype, void *out_dev) {
    LC_REQUIRE(ctx, "lc_spectral_truncate: null context");
    LC_REQUIRE(dtype == LC_F32 || dtype == LC_F64, "lc_spectral_truncate: bad dtype %d", dtype);
    LC_REQUIRE(f_dev && out_dev, "lc_spectral_truncate: null pointer");
    LC_REQUIRE(nbatch >= 1 && nlat >= 3 && nlon >= 4, "lc_spectral_truncate: bad sizes");
    LC_REQUIRE(truncation >= 0, "lc_spectral_truncate: truncation must be >= 0");
    LC_REQUIRE(gridtype == LC_GRID_REGULAR || gridtype == LC_GRID_GAUSSIAN, "lc_spectral_truncate: bad gridtype %d", gridtype);
    LC_REQUIRE(truncation <= nlat - 1 && truncation <= (nlon - 1) / 2, "lc_spectral_truncate: truncation %d too high for a %dx%d grid",
               truncation, nlat, nlon);
    if ((size_t)2 * (truncation + 1) * sizeof(double) > 48 * 1024) {  // a row's coefficients sit in LDS in the inverse DFT
        lc_set_error("lc_spectral_truncate: truncation %d exceeds this build's limit of %d", truncation, 48 * 1024 / 16 - 1);
        return LC_EUNSUPPORTED;
    }
    if ((size_t)nlon * sizeof(double) > 64 * 1024) {  // the forward DFT stages whole field rows in LDS (4 at a time where they fit)
        lc_set_error("lc_spectral_truncate: %d longitudes exceed this build's limit of %d", nlon, 64 * 1024 / 8);
        return LC_EUNSUPPORTED;
    }
    LC_HIP_CHECK(hipSetDevice(ctx->device));
    const int s = ensure_operators(ctx, nlat, nlon, truncation, gridtype);
    if (s != LC_OK) return s;
    if (dtype == LC_F32) return truncate_impl<float>(ctx, (const float *)f_dev, nbatch, nlat, nlon, truncation, (float *)out_dev);
    return truncate_impl<double>(ctx, (const double *)f_dev, nbatch, nlat, nlon, truncation, (double *)out_dev);
}
