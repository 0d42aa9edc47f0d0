// Ridge classification of tools.find_ridges_spherical_hessian (LCS/tools.py:99-138): per grid point
// the eigen-decomposition numpy.linalg.eig (LAPACK dgeev) returns for the symmetric 2x2 Hessian, the
// reference's row-indexed "eigenvector" (tools.py:107), its dot product with the gradient, the
// eigenvalue of largest magnitude and the ridge mask.  The reference loops over points in Python.
//
// dgeev on a symmetric 2x2 is closed form (no balancing, trivial Hessenberg reduction, dlahqr on one
// 2x2 block): the Ahues-Tisseur deflation test, else one dlanv2 standardisation (two branches), then
// dtrevc on the resulting triangle and dgeev's unit 2-norm scaling.  Restated line by line below and
// pinned against numpy.linalg.eig itself (tests/test_ridges.py: eigenvalues bit-exact, vectors 3e-16).
#include "lcs_common.h"

namespace {

struct Eig2 {
    double w0, w1;
    double v00, v01, v10, v11;  // V[row][col]; columns are the eigenvectors
};

__device__ __forceinline__ double dlapy2(double x, double y) {
    const double xa = fabs(x), ya = fabs(y);
    const double w = fmax(xa, ya), z = fmin(xa, ya);
    if (z == 0.0) return w;
    const double q = z / w;
    return w * sqrt(1.0 + q * q);
}

// unit 2-norm after dtrevc's max-norm scaling (dgeev)
__device__ __forceinline__ void normalise(double &x, double &y) {
    const double e = fmax(fabs(x), fabs(y));
    x /= e;
    y /= e;
    const double n = hypot(x, y);
    x /= n;
    y /= n;
}

__device__ Eig2 dgeev_sym2(double a, double b, double d) {
#pragma clang fp contract(off)
    const double ulp = 2.220446049250313e-16;   // dlamch('P')
    const double safmin = 2.2250738585072014e-308;
    const double smlnum = safmin * (2.0 / ulp);  // dlahqr: safmin*(nh/ulp)
    Eig2 r;
    const double ab = fabs(b);
    bool deflate = ab <= smlnum;
    if (!deflate && ab <= ulp * (fabs(a) + fabs(d))) {
        const double aa = fmax(fabs(d), fabs(a - d)), bb = fmin(fabs(d), fabs(a - d));
        const double s = aa + ab;
        deflate = ab * (ab / s) <= fmax(smlnum, ulp * (bb * (aa / s)));
    }
    if (deflate) {  // T = [[a,b],[0,d]]: first vector e1, second solves (a-d) x = -b
        const double smin = fmax(ulp * fabs(d), smlnum);
        double csr = a - d;
        if (fabs(csr) < smin) csr = smin;
        double x1 = (-b) / csr, x2 = 1.0;
        normalise(x1, x2);
        r.w0 = a;
        r.w1 = d;
        r.v00 = 1.0;
        r.v10 = 0.0;
        r.v01 = x1;
        r.v11 = x2;
        return r;
    }
    const double p = 0.5 * (a - d);
    const double scale = fmax(fabs(p), ab);
    const double z0 = (p / scale) * p + (ab / scale) * ab;
    if (z0 >= 4.0 * ulp) {  // dlanv2: real eigenvalues, T comes out diagonal
        const double z = p + copysign(sqrt(scale) * sqrt(z0), p);
        r.w0 = d + z;
        r.w1 = d - (ab / z) * ab;
        const double tau = dlapy2(b, z);
        const double cs = z / tau, sn = b / tau;
        r.v00 = cs;
        r.v01 = -sn;
        r.v10 = sn;
        r.v11 = cs;
        return r;
    }
    // dlanv2: "complex eigenvalues, or real (almost) equal eigenvalues" with c == b
    const double sigma = b + b, temp = a - d;
    const double tau = dlapy2(sigma, temp);
    double cs = sqrt(0.5 * (1.0 + fabs(sigma) / tau));
    double sn = -((0.5 * temp) / (tau * cs)) * copysign(1.0, sigma);
    const double AA = a * cs + b * sn, BB = -a * sn + b * cs, CC = b * cs + d * sn, DD = -b * sn + d * cs;
    const double A2 = AA * cs + CC * sn, B2 = BB * cs + DD * sn, C2 = -AA * sn + CC * cs, D2 = -BB * sn + DD * cs;
    const double mid = 0.5 * (A2 + D2);
    double Af = mid, Df = mid, Bf = B2;
    if (C2 != 0.0) {
        if (B2 != 0.0) {
            if (copysign(1.0, B2) == copysign(1.0, C2)) {  // real eigenvalues: reduce to upper triangular form
                const double sab = sqrt(fabs(B2)), sac = sqrt(fabs(C2));
                const double p3 = copysign(sab * sac, C2);
                const double t3 = 1.0 / sqrt(fabs(B2 + C2));
                Af = mid + p3;
                Df = mid - p3;
                Bf = B2 - C2;
                const double cs1 = sab * t3, sn1 = sac * t3;
                const double t = cs * cs1 - sn * sn1;
                sn = cs * sn1 + sn * cs1;
                cs = t;
            }
        } else {  // copy C to B
            Bf = -C2;
            const double t = cs;
            cs = -sn;
            sn = t;
        }
    }
    const double smin = fmax(ulp * fabs(Df), smlnum);
    double csr = Af - Df;
    if (fabs(csr) < smin) csr = smin;
    const double y1 = (-Bf) / csr;
    double c00 = cs, c10 = sn;                         // Z[:,0]
    double c01 = cs * y1 - sn, c11 = sn * y1 + cs;      // Z[:,0]*y1 + Z[:,1]
    normalise(c00, c10);
    normalise(c01, c11);
    r.w0 = Af;
    r.w1 = Df;
    r.v00 = c00;
    r.v10 = c10;
    r.v01 = c01;
    r.v11 = c11;
    return r;
}

__global__ void ridge_kernel(const double *__restrict__ hxx, const double *__restrict__ hxy,
                             const double *__restrict__ hyy, const double *__restrict__ gx,
                             const double *__restrict__ gy, size_t n, double tol, double *__restrict__ mask,
                             double *__restrict__ eigmin, double *__restrict__ dt_out,
                             double *__restrict__ eigvec_out) {
#pragma clang fp contract(off)
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        auto clean = [](double h) { return (fabs(h) != INFINITY && h == h) ? h : 0.0; };  // tools.py:92-93
        const Eig2 e = dgeev_sym2(clean(hxx[i]), clean(hxy[i]), clean(hyy[i]));
        const bool second = e.w1 < e.w0;                          // np.argmin: first index on ties
        const double r0 = second ? e.v10 : e.v00, r1 = second ? e.v11 : e.v01;  // a ROW of V (tools.py:107)
        const double dt = r0 * gx[i] + r1 * gy[i];                // tools.py:115
        const double em = fabs(e.w1) > fabs(e.w0) ? e.w1 : e.w0;  // eigenvalue of largest magnitude (tools.py:118)
        double m = (fabs(dt) <= tol) ? dt : 0.0;                  // tools.py:136
        m = (fabs(dt) > tol) ? m : 1.0;                           // tools.py:137 (a NaN dt ends up 1)
        m = (em < 0.0) ? m : 0.0;                                 // tools.py:138: sign(eigmin) == -1
        mask[i] = m;
        eigmin[i] = em;
        if (dt_out) dt_out[i] = dt;
        if (eigvec_out) {
            eigvec_out[i] = r0;
            eigvec_out[n + i] = r1;
        }
    }
}

}  // namespace

extern "C" int lc_ridge_classify(lc_ctx *ctx, const void *hxx, const void *hxy, const void *hyy, const void *gx,
                                 const void *gy, size_t n, double tolerance, void *mask_out, void *eigmin_out,
                                 void *dt_out, void *eigvec_out) {
    LC_REQUIRE(ctx, "lc_ridge_classify: null context");
    LC_REQUIRE(hxx && hxy && hyy && gx && gy && mask_out && eigmin_out, "lc_ridge_classify: null pointer");
    LC_HIP_CHECK(hipSetDevice(ctx->device));
    if (n == 0) return LC_OK;
    const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
    hipLaunchKernelGGL(ridge_kernel, dim3(blocks), dim3(256), 0, ctx->stream, (const double *)hxx, (const double *)hxy,
                       (const double *)hyy, (const double *)gx, (const double *)gy, n, tolerance, (double *)mask_out,
                       (double *)eigmin_out, (double *)dt_out, (double *)eigvec_out);
    LC_HIP_CHECK(hipGetLastError());
    return LC_OK;
}
