"""Helpers for the parity tests at BASELINE.json's full sizes.

The oracle cannot run millions of seeds, but (i) advection is independent per seed, so the engine's answer on a
SUBSET of a huge seed grid must equal the oracle run on exactly those seeds, and (ii) sigma couples a seed only to
its +-2 neighbours (LCS/tools.py:202-228), so the engine's sigma on a contiguous WINDOW must equal the oracle run
on that window plus a 2-seed halo.  In both cases the rows handed to the oracle are framed by the global grid's
first / last ``order`` rows, because the reference classifies pole rows by seed-row INDEX (LCS/tools.py:24-33).
"""
import numpy as np


def subset(n, k, edge, must=()):
    """About k indices in [0, n): the first and last `edge`, every index in `must`, the rest evenly spread."""
    inner = np.round(np.linspace(edge, n - 1 - edge, max(k - 2 * edge, 2))).astype(int)
    return np.unique(np.concatenate([np.arange(edge), inner, np.asarray(must, dtype=int), np.arange(n - edge, n)]))


def lon_err(a, b):
    """|a - b| on the circle: one ulp can flip the +-180 rewrite (Q7)."""
    d = np.abs(np.asarray(a, dtype=np.float64) - np.asarray(b, dtype=np.float64))
    return np.minimum(d, np.abs(d - 360.0))


def oracle_subset(O, u, v, lat, lon, slat, slon, rows, cols, dtype, **kw):
    """Oracle departure points at seeds (rows x cols) of the global seed grid, in `dtype` arithmetic.
    `rows` must contain the global first / last `interp_order` rows (subset() does that)."""
    c = lambda a: np.asarray(a).astype(dtype, copy=False)
    return O.parcel_propagation(c(u), c(v), c(lat), c(lon), seed_lat=c(slat)[rows], seed_lon=c(slon)[cols], **kw)


def oracle_window(O, u, v, lat, lon, slat, slon, r0, r1, c0, c1, dtype, interp_order, **kw):
    """(x_dep, y_dep, sigma) of the oracle on seed window rows [r0, r1) x cols [c0, c1) of the global grid.

    The oracle advects window + 2-seed halo (framed by the global pole rows), then runs the reference's
    flowmap_gradient + matrix 2-norm on that block; the halo ring, where the block's own one-sided / cyclic
    edge rules differ from the global grid's, is dropped.  A window starting at global row 0 keeps the genuine
    one-sided rows (Q12)."""
    ny, nx = len(slat), len(slon)
    assert c0 >= 2 and c1 <= nx - 2, "window columns must leave room for the halo"
    ra, rb = max(r0 - 2, 0), min(r1 + 2, ny)
    assert ra == 0 or ra >= interp_order, "window must start at row 0 or clear the pole rows"
    assert rb <= ny - interp_order - 0 or rb == ny
    head = np.arange(0, interp_order) if ra > 0 else np.arange(0, 0)
    tail = np.arange(ny - interp_order, ny) if rb < ny else np.arange(0, 0)
    rows = np.concatenate([head, np.arange(ra, rb), tail])
    cols = np.arange(c0 - 2, c1 + 2)
    c = lambda a: np.asarray(a).astype(dtype, copy=False)
    x, y = O.parcel_propagation(c(u), c(v), c(lat), c(lon), seed_lat=c(slat)[rows], seed_lon=c(slon)[cols],
                                interp_order=interp_order, **kw)
    xb, yb = x[head.size:head.size + rb - ra], y[head.size:head.size + rb - ra]
    # the grid spacing is the GLOBAL grid's coord[1] - coord[0] (LCS/tools.py:255-256), not the window's
    gl, go = c(slat), c(slon)
    tens = O.flowmap_gradient(xb, yb, gl[ra:rb], go[cols], dlat=gl[1] - gl[0], dlon=go[1] - go[0])
    sig = O.sigma_max(tens)
    i0 = r0 - ra
    return xb[i0:i0 + r1 - r0, 2:-2], yb[i0:i0 + r1 - r0, 2:-2], sig[i0:i0 + r1 - r0, 2:-2]


def band(err_gpu, err_oracle, label, med_floor, p99_floor, max_floor, med_k=2.0, p99_k=3.0, max_k=4.0):
    """The float32 engine is judged against the float64 answer and must sit in the band of the float32
    ORACLE's own error: median, 99th percentile and maximum."""
    eg, eo = np.asarray(err_gpu).ravel(), np.asarray(err_oracle).ravel()
    st = lambda e: (float(np.median(e)), float(np.percentile(e, 99)), float(e.max()))
    (gm, gp, gx), (om, op, ox) = st(eg), st(eo)
    print(f"{label}: gpu32 median {gm:.3e} p99 {gp:.3e} max {gx:.3e} | oracle32 median {om:.3e} p99 {op:.3e} max {ox:.3e}")
    assert gm <= max(med_k * om, med_floor), f"{label}: median {gm:.3e} vs oracle32 {om:.3e}"
    assert gp <= max(p99_k * op, p99_floor), f"{label}: p99 {gp:.3e} vs oracle32 {op:.3e}"
    assert gx <= max(max_k * ox, max_floor), f"{label}: max {gx:.3e} vs oracle32 {ox:.3e}"


def q7_teleports(eng, field, slat, slon, seeds, t0=0, nsteps=None, interp_order=1, K=4, timestep=-900.0):
    """Which of `seeds` [(row, col), ...] the engine sent through the reference's longitude seam defect (Q7):
    `x % 180` maps a parcel that lands EXACTLY on -180.0 to longitude 0 (LCS/trajectory.py:93-94,119-120).  In
    float64 that is a measure-zero event; in float32 (ulp 1.5e-5 at 180) it hits about one parcel in a million
    per seam crossing, and WHICH parcel depends on the last bit, so engine and float32 oracle teleport
    different ones.  Signature: the trajectory contains x == 0.0 exactly, straight after a step near -180."""
    out = []
    ny = len(slat)
    for r, c in seeds:
        res = eng.advect(field, np.asarray(slat)[r:r + 1], np.asarray(slon)[c:c + 1], timestep, K, interp_order, True,
                         t0=t0, nsteps=nsteps, row0=int(r), ny_global=ny, return_traj=True)
        tx = res[2].cpu().numpy()[:, 0, 0]
        hit = np.nonzero(tx[1:] == 0.0)[0]
        out.append(bool(hit.size) and bool(np.any(tx[hit] < -170.0)))
    return out


def split_teleports(err_gpu, err_oracle, limit=0.5):
    """Indices where the engine is off by more than `limit` degrees while the float32 oracle is not (candidates
    for Q7 teleports; the caller verifies each with q7_teleports) and the mask of everything else."""
    cand = np.argwhere((np.asarray(err_gpu) > limit) & (np.asarray(err_oracle) <= limit))
    keep = np.ones(np.shape(err_gpu), dtype=bool)
    for i, j in cand:
        keep[i, j] = False
    return [tuple(int(k) for k in ij) for ij in cand], keep


def dilate(mask_bad, r=2):
    """Cells within +-r (rows and columns) of a bad seed: the sigma stencil's reach (Q12)."""
    m = np.asarray(mask_bad, dtype=bool)
    out = m.copy()
    for dy in range(-r, r + 1):
        for dx in range(-r, r + 1):
            sh = np.zeros_like(m)
            ys = slice(max(dy, 0), m.shape[0] + min(dy, 0)); yd = slice(max(-dy, 0), m.shape[0] + min(-dy, 0))
            xs = slice(max(dx, 0), m.shape[1] + min(dx, 0)); xd = slice(max(-dx, 0), m.shape[1] + min(-dx, 0))
            sh[yd, xd] = m[ys, xs]
            out |= sh
    return out


def positions_check(eng, field, slat, slon, rows, cols, xg, yg, o32, o64, label, floors, interp_order=1, **tele_kw):
    """Engine departure points at seeds (rows x cols) inside the float32 oracle's band around the float64
    answer.  A seed the engine is off by > 0.5 degrees on (and the oracle is not) must be a verified Q7
    teleport; those (and the oracle's own) are left out of the statistics.  Returns the keep mask."""
    (x32, y32), (x64, y64) = o32, o64
    eg = np.maximum(lon_err(xg, x64), np.abs(np.asarray(yg, dtype=np.float64) - y64))
    eo = np.maximum(lon_err(x32, x64), np.abs(np.asarray(y32, dtype=np.float64) - y64))
    cand, keep = split_teleports(eg, eo)
    assert len(cand) <= 3, f"{label}: {len(cand)} seeds off by > 0.5 degrees"
    if cand:
        seeds = [(int(rows[i]), int(cols[j])) for i, j in cand]
        ok = q7_teleports(eng, field, slat, slon, seeds, interp_order=interp_order, **tele_kw)
        print(f"{label}: {len(cand)} Q7 teleport(s) at {seeds} verified {ok}")
        assert all(ok), f"{label}: large position error that is NOT a Q7 teleport"
    keep &= split_teleports(eo, eg)[1]
    band(eg[keep], eo[keep], f"{label} positions", *floors)
    return keep
