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
    lc_ctx_destroy(ctx);
    return bad ? 1 : 0;
}
