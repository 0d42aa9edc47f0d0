/* Pure-C client of liblcs_hip.so: proves the boundary is a C ABI (no Python, no torch, no C++ types).
 *
 *   gcc -std=c99 -I include tests/c/abi_smoke.c -o abi_smoke -L lagrangiancoherence_amd -llcs_hip -lm
 *
 * KAT-2 of SURVEY.md section 8c through lc_lcs_host: uniform zonal wind u0, v = 0, K SETTLS iterations.
 * Every interior seed must move by (1+K)*dt*u0*180/(pi*R*|cos lat|) per step (reference quirks Q4, Q5) and
 * keep its latitude; sigma must be finite.  Also checks the error path (bad interp_order -> LC_EUNSUPPORTED
 * with a message).  Exit code 0 = pass. */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>

#include "lcs_hip.h"

#define NT 4
#define NY 21
#define NX 36

/* The device-pointer route from plain C: lc_malloc / lc_memcpy_*, lc_field_pack, then the time-level loop in two pieces --
 * lc_advect over levels [0, 2) and lc_advect_from over [2, 3) from its result -- against one lc_advect over [0, 3): the
 * loop of LCS/trajectory.py:80-126 carries only positions, so the two must agree bit for bit.  A sheared wind this time
 * (u grows with latitude, v with longitude) so that every seed samples between nodes. */
static int continuation_check(lc_ctx *ctx, const double *u_uniform, const double *v0, const double *lat, const double *lon) {
    static double u[NT * NY * NX], v[NT * NY * NX], xa[NY * NX], ya[NY * NX], xb[NY * NX], yb[NY * NX];
    (void)u_uniform;
    (void)v0;
    for (int t = 0; t < NT; ++t)
        for (int j = 0; j < NY; ++j)
            for (int i = 0; i < NX; ++i) {
                u[(t * NY + j) * NX + i] = 5.0 + 0.3 * j + 0.7 * t;
                v[(t * NY + j) * NX + i] = 0.2 * (i - NX / 2) - 0.1 * t;
            }
    const size_t fb = sizeof u, pe = lc_packed_elems(NT, NY, NX), ee = lc_packed_elems(NT - 1, NY, NX), sb = sizeof xa;
    void *du, *dv, *lin, *ext, *dlat, *dlon, *x1, *y1, *x2, *y2;
    int st = 0;
    st |= lc_malloc(ctx, fb, &du) | lc_malloc(ctx, fb, &dv) | lc_malloc(ctx, pe * 8, &lin) | lc_malloc(ctx, ee * 8, &ext);
    st |= lc_malloc(ctx, NY * 8, &dlat) | lc_malloc(ctx, NX * 8, &dlon);
    st |= lc_malloc(ctx, sb, &x1) | lc_malloc(ctx, sb, &y1) | lc_malloc(ctx, sb, &x2) | lc_malloc(ctx, sb, &y2);
    st |= lc_memcpy_h2d(ctx, du, u, fb) | lc_memcpy_h2d(ctx, dv, v, fb) | lc_memcpy_h2d(ctx, dlat, lat, NY * 8) |
          lc_memcpy_h2d(ctx, dlon, lon, NX * 8);
    st |= lc_field_pack(ctx, du, dv, LC_F64, NT, NY, NX, 1, lin, ext);
#define ADV_ARGS lin, NULL, ext, LC_F64, NT, NY, NX, lat[0], lat[NY - 1], lon[0], lon[NX - 1], dlat, NY, dlon, NX, 0, NY
    st |= lc_advect(ctx, ADV_ARGS, 900.0, 2, 1, LC_X_CYCLIC, 0, NT - 1, x1, y1, NULL, NULL);               /* levels [0, 3) */
    st |= lc_advect(ctx, ADV_ARGS, 900.0, 2, 1, LC_X_CYCLIC, 0, 2, x2, y2, NULL, NULL);                    /* levels [0, 2) */
    st |= lc_advect_from(ctx, ADV_ARGS, x2, y2, 900.0, 2, 1, LC_X_CYCLIC, 2, 1, x2, y2, NULL, NULL);       /* [2, 3), in place */
    st |= lc_sync(ctx);
    st |= lc_memcpy_d2h(ctx, xa, x1, sb) | lc_memcpy_d2h(ctx, ya, y1, sb) | lc_memcpy_d2h(ctx, xb, x2, sb) | lc_memcpy_d2h(ctx, yb, y2, sb);
    if (st != LC_OK) {
        fprintf(stderr, "device-pointer route: %d %s\n", st, lc_last_error());
        return 1;
    }
    int diff = 0, moved = 0;
    for (int k = 0; k < NY * NX; ++k) {
        diff += xa[k] != xb[k] || ya[k] != yb[k];
        moved += xa[k] != lon[k % NX];
    }
    printf("lc_build_id = %s; lc_advect_from continuation: %d differences, %d of %d seeds moved\n", lc_build_id(), diff, moved, NY * NX);
    /* lc_advect_ex from C: the argument structure as this compiler lays it out, with the RAW planes as the order-1 source and
     * NO packed_lin (float64: Euler sample and pole rows straight from u, v) -- the bits of the packed_lin call above */
    {
        lc_advect_args a = {0};
        a.struct_size = sizeof a;
        a.packed_ext = ext;
        a.u_raw = du;
        a.v_raw = dv;
        a.dtype = LC_F64;
        a.nt = NT;
        a.ny_f = NY;
        a.nx_f = NX;
        a.lat_min = lat[0];
        a.lat_max = lat[NY - 1];
        a.lon_min = lon[0];
        a.lon_max = lon[NX - 1];
        a.seed_lat_dev = dlat;
        a.ny = NY;
        a.seed_lon_dev = dlon;
        a.nx = NX;
        a.row0 = 0;
        a.ny_global = NY;
        a.timestep = 900.0;
        a.settls_order = 2;
        a.interp_order = 1;
        a.cyclic_x = LC_X_CYCLIC;
        a.t0 = 0;
        a.nsteps = NT - 1;
        a.n_members = 1;
        a.x_out = x2;
        a.y_out = y2;
        int s2 = lc_advect_ex(ctx, &a) | lc_sync(ctx) | lc_memcpy_d2h(ctx, xb, x2, sb) | lc_memcpy_d2h(ctx, yb, y2, sb);
        int rawdiff = 0;
        for (int k = 0; k < NY * NX; ++k) rawdiff += xa[k] != xb[k] || ya[k] != yb[k];
        a.struct_size = sizeof a - 8;                      /* a client compiled against another layout is refused */
        const int refused = lc_advect_ex(ctx, &a) == LC_EINVAL;
        printf("lc_advect_ex with raw planes: status %d, %d differences from the packed_lin form, wrong struct_size refused: %d\n",
               s2, rawdiff, refused);
        diff += (s2 != LC_OK) + rawdiff + !refused;
    }
    void *all[] = {du, dv, lin, ext, dlat, dlon, x1, y1, x2, y2};
    for (unsigned k = 0; k < sizeof all / sizeof all[0]; ++k) lc_free(ctx, all[k]);
    return diff != 0 || moved < NY * NX / 2;
}

int main(void) {
    static double u[NT * NY * NX], v[NT * NY * NX], lat[NY], lon[NX];
    static double sigma[NY * NX], x[NY * NX], y[NY * NX];
    const double u0 = 7.0, dt = 600.0, R = 6371000.0, PI = 3.141592653589793;
    const int K = 4;
    for (int j = 0; j < NY; ++j) lat[j] = -80.0 + 8.0 * j;
    for (int i = 0; i < NX; ++i) lon[i] = -180.0 + 10.0 * i;
    for (int k = 0; k < NT * NY * NX; ++k) {
        u[k] = u0;
        v[k] = 0.0;
    }
    printf("lc_version = %d\n", lc_version());
    if (lc_version() != LC_VERSION) { /* argument lists changed between versions: header and library must match */
        fprintf(stderr, "liblcs_hip.so is ABI %d, this client was compiled against %d\n", lc_version(), LC_VERSION);
        return 4;
    }
    lc_ctx *ctx = NULL;
    if (lc_ctx_create(0, &ctx) != LC_OK) {
        fprintf(stderr, "lc_ctx_create: %s\n", lc_last_error());
        return 2;
    }
    int st = lc_lcs_host(ctx, u, v, LC_F64, NT, NY, NX, lat, lon, lat, NY, lon, NX, dt, K, /*interp_order*/ 1,
                         /*cyclic_x*/ 1, /*t0*/ 0, /*nsteps*/ NT - 1, /*gauss_sigma*/ 0.0, /*fd_fp32_cast*/ 1,
                         LC_LAYOUT_REFERENCE, sigma, x, y, NULL, NULL);
    if (st != LC_OK) {
        fprintf(stderr, "lc_lcs_host: %d %s\n", st, lc_last_error());
        return 3;
    }
    int bad = 0;
    for (int j = 1; j < NY - 1; ++j) {               /* rows 0 and NY-1 are 'constant'-mode pole rows (Q3) */
        const double dl = (NT - 1) * (1 + K) * dt * u0 * 180.0 / (PI * R * fabs(cos(lat[j] * PI / 180.0)));
        for (int i = 1; i < NX; ++i) {               /* column 0 (lon == -180) is rewritten to 0 by Q7 */
            const double moved = x[j * NX + i] - lon[i];
            if (fabs(moved - dl) > 1e-11 * dl || y[j * NX + i] != lat[j] || !isfinite(sigma[j * NX + i])) ++bad;
        }
    }
    printf("uniform-wind known answer: %d mismatches\n", bad);
    st = lc_lcs_host(ctx, u, v, LC_F64, NT, NY, NX, lat, lon, lat, NY, lon, NX, dt, K, /*interp_order*/ 6, 1, 0,
                     NT - 1, 0.0, 1, LC_LAYOUT_REFERENCE, sigma, x, y, NULL, NULL);
    printf("interp_order=6 -> status %d (%s)\n", st, lc_last_error());
    if (st != LC_EUNSUPPORTED) ++bad;
    bad += continuation_check(ctx, u, v, lat, lon);
    lc_ctx_destroy(ctx);
    return bad ? 1 : 0;
}
