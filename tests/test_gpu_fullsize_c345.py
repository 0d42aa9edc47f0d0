"""Parity at full size for BASELINE configs[2] sigma, configs[3] (C4) and configs[4] (C5), on ONE GPU.

C4 (8192^2 seeds x 384 steps, 8-GPU row-sharded) runs as its 8 row shards one after the other
(`row_partition`, `row0` / `ny_global`), C5 (64 start times x 2048^2 x 200 steps) as members {0, 31, 63}
through `ensemble_lcs`.  Positions: subset of the seeds vs the oracle on exactly those seeds.  sigma: contiguous
64x64 windows vs the oracle on window + 2-seed halo, float32 judged against the float64 answer inside the
band of the float32 oracle's own error (median, p99, max).  Helpers in tests/_fullsize.py.
"""
import numpy as np
import pytest

from lagrangiancoherence_amd import flows, sharded
from tests._fullsize import band, dilate, oracle_subset, oracle_window, positions_check, subset

pytestmark = pytest.mark.gpu
KW = dict(timestep=-900.0, SETTLS_order=4, cyclic_xboundary=True)

# The kernel each BASELINE configuration is dispatched to.  Every test below asserts the name of the kernel that ran
# (lc_ctx_last_advect_kernel), so a change of the launcher's size thresholds cannot silently move a configuration onto a
# kernel variant that has no oracle anchor at that size.
EXPECT = {
    ("c3", 1): "advect_lds2_kernel<4, true, 0>",          # 4096^2 seeds: two seeds per lane, tall patches
    ("c3", 3): "advect_lds2_o3_kernel<4, true, 0>",
    ("c4 shard", 1): "advect_lds2_kernel<4, true, 0>",    # 1024 x 8192 seeds = 2^23 per call
    ("c5 member", 1): "advect_lds_kernel<1, 4, true>",    # one member alone: 2^22 seeds, one seed per lane
    ("c5 rank", 1): "advect_lds2_kernel<4, true, 3>",     # a rank's 8 members through lc_advect_batch: two MEMBERS per lane
}


@pytest.fixture(scope="module")
def eng():
    from lagrangiancoherence_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


@pytest.fixture(scope="module")
def O():
    from oracle import lcs_oracle
    return lcs_oracle


def _np(t):
    return t.cpu().numpy()


# ------------------------------------------------------------------------------------------ C3 sigma
@pytest.fixture(scope="module")
def c3(eng):
    u, v, lat, lon = flows.era5_like(nt=97)
    slat, slon = flows.seed_grid(4096, 4096, lat, lon)
    res = {}
    for order, nt in ((1, 97), (3, 25)):       # the order-3 oracle re-runs scipy's prefilter 18x per step: 24 steps
        f = eng.prepare_field(u[:nt], v[:nt], lat, lon, order)
        r = eng.lcs(f, slat, slon, -900.0, SETTLS_order=4, interp_order=order, cyclic_xboundary=True)
        # the kernels BASELINE configs[2] dispatches to (2^24 seeds per call: two seeds per lane)
        assert eng.last_advect_kernel() == EXPECT[("c3", order)], eng.last_advect_kernel()
        assert eng.last_sigma_kernel() == "sigma_march_kernel_f32", eng.last_sigma_kernel()
        res[order] = {k: _np(r[k]) for k in ("sigma", "x_dep", "y_dep")}
        res[order]["field"] = f
        del r
    return u, v, lat, lon, slat, slon, res


def _window_check(eng, g, slat, slon, win, o32, o64, label, order, floors_s, floors_p, **tele_kw):
    """sigma and positions of one window inside the float32 oracle's band; seeds sent through the Q7 seam
    defect are verified one by one and left out, with their stencil neighbours."""
    r0, r1, c0, c1 = win
    keep = positions_check(eng, g["field"], slat, slon, np.arange(r0, r1), np.arange(c0, c1), g["x_dep"][r0:r1, c0:c1],
                           g["y_dep"][r0:r1, c0:c1], o32[:2], o64[:2], label, floors_p, interp_order=order, **tele_kw)
    sg = g["sigma"][r0:r1, c0:c1].astype(np.float64)
    assert np.isfinite(sg).all()
    ks = ~dilate(~keep)
    band(np.abs(sg / o64[2] - 1)[ks], np.abs(o32[2] / o64[2] - 1)[ks], f"{label} sigma", *floors_s)


# 64x64 windows of the 4096^2 grid (0.044 degree spacing, where the float32 cancellation in the X, Y, Z
# differences of Q11 is at its worst): equator, 60 N, and the south pole edge with its one-sided rows (Q12)
WINDOWS = {"equator": (2016, 2080, 1000, 1064), "60N": (3380, 3444, 3000, 3064), "pole_edge": (0, 64, 2000, 2064)}


@pytest.mark.parametrize("name", list(WINDOWS))
def test_config3_sigma_windows_vs_oracle(eng, c3, O, name):
    u, v, lat, lon, slat, slon, res = c3
    win = WINDOWS[name]
    o32 = oracle_window(O, u, v, lat, lon, slat, slon, *win, np.float32, 1, **KW)
    o64 = oracle_window(O, u, v, lat, lon, slat, slon, *win, np.float64, 1, **KW)
    # floors (what the engine may always have, whatever the float32 oracle's own error): by window since round 5 -- the measured
    # values are in profiles/r05/parity_stats.txt; a floor sat an order of magnitude above them at 60N (p99 2.4e-5 against 5e-4)
    floors_s, floors_p = {"equator": ((1e-4, 5e-4, 1e-3), (1e-4, 5e-4, 2e-3)), "60N": ((4e-4, 1.5e-3, 2.5e-3), (2e-5, 5e-5, 1e-4)),
                          "pole_edge": ((1e-4, 1e-3, 1e-2), (3e-4, 5e-4, 2e-3))}[name]
    _window_check(eng, res[1], slat, slon, win, o32, o64, f"C3 {name} (order 1, 96 steps)", 1, floors_s, floors_p)


def test_config3_sigma_window_order3(eng, c3, O):
    u, v, lat, lon, slat, slon, res = c3
    win = WINDOWS["60N"]
    kw = dict(KW, interp_order=3)
    o32 = oracle_window(O, u[:25], v[:25], lat, lon, slat, slon, *win, np.float32, **kw)
    o64 = oracle_window(O, u[:25], v[:25], lat, lon, slat, slon, *win, np.float64, **kw)
    _window_check(eng, res[3], slat, slon, win, o32, o64, "C3 60N (order 3, 24 steps)", 3,
                  (1.5e-4, 6e-4, 1e-3), (3e-5, 1.5e-4, 2.5e-4))   # (measured 1.2e-4 / 4.6e-4 / 6.0e-4 and 2.5e-5 / 1.0e-4 / 1.6e-4)


# ------------------------------------------------------------------------------------------ C4
def test_config4_row_shards_full_size(eng, O):
    """configs[3]: 8192^2 seeds x 384 steps on the 720x1440 series, as the 8 row shards of the 8-GPU layout."""
    import torch
    NY = NX = 8192
    WORLD = 8
    u, v, lat, lon = flows.era5_like(nt=385)
    slat, slon = flows.seed_grid(NY, NX, lat, lon)
    f = eng.prepare_field(u, v, lat, lon, 1)
    x = torch.empty((NY, NX), dtype=torch.float32, device="cuda")
    y = torch.empty_like(x)
    bounds = [sharded.row_partition(NY, WORLD, r) for r in range(WORLD)]
    for lo, hi in bounds:                                          # advection needs no communication
        xs, ys = eng.advect(f, slat[lo:hi], slon, -900.0, 4, 1, True, row0=lo, ny_global=NY)
        assert eng.last_advect_kernel() == EXPECT[("c4 shard", 1)], eng.last_advect_kernel()
        x[lo:hi], y[lo:hi] = xs, ys
    # (a) positions: a subset containing every shard's first and last row, vs the oracle on those seeds
    must = [r for lo, hi in bounds for r in (lo, lo + 1, hi - 2, hi - 1)]
    rows, cols = subset(NY, 24, 1, must), subset(NX, 40, 0)
    xg, yg = _np(x[rows][:, cols]), _np(y[rows][:, cols])
    kw = dict(KW, interp_order=1)
    o32 = oracle_subset(O, u, v, lat, lon, slat, slon, rows, cols, np.float32, **kw)
    o64 = oracle_subset(O, u, v, lat, lon, slat, slon, rows, cols, np.float64, **kw)
    positions_check(eng, f, slat, slon, rows, cols, xg, yg, o32, o64, "C4 (384 steps)", (4e-4, 5e-3, 2e-2))
    # (b) sigma of every shard from its halo window == sigma of the whole grid, bit for bit
    dlat, dlon = float(slat[1] - slat[0]), float(slon[1] - slon[0])
    s_full = eng.sigma(x, y, slat, dlat, dlon)
    for lo, hi in bounds:
        n_lo, n_hi = sharded.halo_rows(NY, lo, hi)
        a, b = lo - n_lo, hi + n_hi
        s = eng.sigma(x[a:b], y[a:b], slat[a:b], dlat, dlon, ny_global=NY, in_row0=a, out_row0=lo, n_out_rows=hi - lo)
        assert torch.equal(s, s_full[lo:hi]), f"shard rows {lo}:{hi}"
    assert bool(torch.isfinite(s_full).all())
    # (c) sigma on a window straddling the boundary between shards 3 and 4, vs the oracle
    r0, r1, c0, c1 = 4064, 4128, 5000, 5064
    o32 = oracle_window(O, u, v, lat, lon, slat, slon, r0, r1, c0, c1, np.float32, 1, **KW)
    o64 = oracle_window(O, u, v, lat, lon, slat, slon, r0, r1, c0, c1, np.float64, 1, **KW)
    g = {"x_dep": _np(x), "y_dep": _np(y), "sigma": _np(s_full), "field": f}
    _window_check(eng, g, slat, slon, (r0, r1, c0, c1), o32, o64, "C4 window across the shard 3|4 boundary", 1,
                  (1e-3, 1e-2, 1e-1), (4e-4, 5e-3, 2e-2))


# ------------------------------------------------------------------------------------------ C5
def test_config5_ensemble_members_full_size(eng, O):
    """configs[4]: members 0, 31, 63 of 64 start times x 2048^2 seeds x 200 steps on the 264-level series."""
    NE, NS, N = 64, 200, 2048
    u, v, lat, lon = flows.era5_like(nt=NE + NS)
    slat, slon = flows.seed_grid(N, N, lat, lon)
    f = eng.prepare_field(u, v, lat, lon, 1)
    rows, cols = subset(N, 64, 1), subset(N, 64, 0)
    r0, r1, c0, c1 = 1400, 1464, 300, 364
    for e in (0, 31, 63):
        mine, sig, xd, yd = sharded.ensemble_lcs(eng, f, slat, slon, -900.0, NE, NS, rank=e, world=NE, SETTLS_order=4,
                                                 interp_order=1, return_dpts=True)
        assert mine == [e]
        assert eng.last_advect_kernel() == EXPECT[("c5 member", 1)], eng.last_advect_kernel()
        kw = dict(KW, interp_order=1, t0=e, nsteps=NS)
        o32 = oracle_subset(O, u, v, lat, lon, slat, slon, rows, cols, np.float32, **kw)
        o64 = oracle_subset(O, u, v, lat, lon, slat, slon, rows, cols, np.float64, **kw)
        xg, yg = _np(xd[0][rows][:, cols]), _np(yd[0][rows][:, cols])
        positions_check(eng, f, slat, slon, rows, cols, xg, yg, o32, o64, f"C5 member {e} (200 steps)",
                        (2e-4, 2e-3, 1e-2), t0=e, nsteps=NS)
        kw = dict(KW, t0=e, nsteps=NS)
        w32 = oracle_window(O, u, v, lat, lon, slat, slon, r0, r1, c0, c1, np.float32, 1, **kw)
        w64 = oracle_window(O, u, v, lat, lon, slat, slon, r0, r1, c0, c1, np.float64, 1, **kw)
        assert np.isfinite(_np(sig[0])).all()
        g = {"x_dep": _np(xd[0]), "y_dep": _np(yd[0]), "sigma": _np(sig[0]), "field": f}
        _window_check(eng, g, slat, slon, (r0, r1, c0, c1), w32, w64, f"C5 member {e} window", 1,
                      (5e-4, 5e-3, 5e-2), (2e-4, 2e-3, 1e-2), t0=e, nsteps=NS)


def test_config5_one_rank_of_eight_as_dispatched(eng, O):
    """configs[4] as the 8-GPU layout runs it: rank 7's block of 8 consecutive start times (members 56..63) x 2048^2
    seeds x 200 steps in ONE lc_advect_batch call -- level-major chunks, two members per lane of the two-seed kernel
    (`advect_lds2_kernel<4, true, 3>`, the kernel the C5 throughput is quoted on).  The reference runs its start-time
    loop one series at a time (LCS/trajectory.py:80-126 per start time); the oracle does exactly that for the first
    member of the block (first of a lane pair), a middle one (second of a pair) and the last (second of a pair, whose
    last step reads the series' last level): positions on a seed subset and sigma on a 64x64 window."""
    NE, NS, N, WORLD, RANK = 64, 200, 2048, 8, 7
    u, v, lat, lon = flows.era5_like(nt=NE + NS)
    slat, slon = flows.seed_grid(N, N, lat, lon)
    f = eng.prepare_field(u, v, lat, lon, 1)
    mine, sig, xd, yd = sharded.ensemble_lcs(eng, f, slat, slon, -900.0, NE, NS, rank=RANK, world=WORLD, SETTLS_order=4,
                                             interp_order=1, return_dpts=True)
    assert mine == list(range(56, 64))
    assert eng.last_advect_kernel() == EXPECT[("c5 rank", 1)], eng.last_advect_kernel()
    # the pair path walks the window [0, nsteps + 1) in chunks of sharded.ENSEMBLE_CHUNK levels
    assert eng.last_advect_launches() == -(-(NS + 1) // sharded.ENSEMBLE_CHUNK)
    rows, cols = subset(N, 48, 1), subset(N, 48, 0)
    r0, r1, c0, c1 = 1400, 1464, 300, 364
    for e in (56, 59, 63):
        i = e - mine[0]
        kw = dict(KW, interp_order=1, t0=e, nsteps=NS)
        o32 = oracle_subset(O, u, v, lat, lon, slat, slon, rows, cols, np.float32, **kw)
        o64 = oracle_subset(O, u, v, lat, lon, slat, slon, rows, cols, np.float64, **kw)
        xg, yg = _np(xd[i][rows][:, cols]), _np(yd[i][rows][:, cols])
        positions_check(eng, f, slat, slon, rows, cols, xg, yg, o32, o64, f"C5 rank 7 member {e} (200 steps, batch)",
                        (2e-4, 2e-3, 1e-2), t0=e, nsteps=NS)
        kw = dict(KW, t0=e, nsteps=NS)
        w32 = oracle_window(O, u, v, lat, lon, slat, slon, r0, r1, c0, c1, np.float32, 1, **kw)
        w64 = oracle_window(O, u, v, lat, lon, slat, slon, r0, r1, c0, c1, np.float64, 1, **kw)
        assert np.isfinite(_np(sig[i])).all()
        g = {"x_dep": _np(xd[i]), "y_dep": _np(yd[i]), "sigma": _np(sig[i]), "field": f}
        _window_check(eng, g, slat, slon, (r0, r1, c0, c1), w32, w64, f"C5 rank 7 member {e} window", 1,
                      (5e-4, 5e-3, 5e-2), (2e-4, 2e-3, 1e-2), t0=e, nsteps=NS)
