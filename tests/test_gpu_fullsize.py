"""Parity at BASELINE.json's full sizes (configs[1] and configs[2]).

The oracle cannot run 4096^2 seeds in seconds, but advection is independent per seed, so the
GPU's answer at a SUBSET of the seeds must equal the oracle run on exactly those seeds.  The
subset keeps the first/last `order` rows so the oracle's pole-row rule (by seed row index,
LCS/tools.py:24-33) selects the same rows as in the full grid.  Plus size-independent
properties: the uniform-wind known answer and sharded == unsharded at full size.
"""
import os

import numpy as np
import pytest

from lagrangiancoherence_amd import flows

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from lagrangiancoherence_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


def _o3_f64_pack(planes):
    """The prefilter stage lc_field_pack takes for float64 coefficients at order 3 on a grid of 64 x 64 nodes or more: one pass
    (a suite run with LCS_FUSED_PREFILTER=0 forces the two streaming sweeps it replaced)."""
    import os
    if os.environ.get("LCS_FUSED_PREFILTER", "1") == "0":
        return "prefilter_cols_stream_kernel + prefilter_rows_stream_kernel"
    return f"prefilter_fused_stream_kernel<{planes}>"


def _subset(n, k, edge):
    """k indices in [0, n) containing the first and last `edge` rows."""
    inner = np.unique(np.round(np.linspace(edge, n - 1 - edge, k - 2 * edge)).astype(int))
    return np.concatenate([np.arange(edge), inner, np.arange(n - edge, n)])


@pytest.fixture(scope="module")
def c3():
    u, v, lat, lon = flows.era5_like(nt=97)                    # 720 x 1440, float32
    slat, slon = flows.seed_grid(4096, 4096, lat, lon)
    return u, v, lat, lon, slat, slon


@pytest.mark.parametrize("order", [1, 3])
def test_config3_full_size_subset_vs_oracle(eng, c3, order):
    from oracle import lcs_oracle as O
    u, v, lat, lon, slat, slon = c3
    f = eng.prepare_field(u, v, lat, lon, order)
    assert eng.last_pack_kernel() == {1: "pack_fused_kernel", 3: "prefilter_fir_kernel"}[order], eng.last_pack_kernel()
    x, y = eng.advect(f, slat, slon, -900.0, SETTLS_order=4, interp_order=order, cyclic_xboundary=True)
    # the kernel BASELINE configs[2] is dispatched to (asserted so a change of the launcher's size thresholds cannot
    # silently move the configuration onto a kernel variant without an oracle anchor)
    assert eng.last_advect_kernel() == {1: "advect_lds2_kernel<4, true, 0>", 3: "advect_lds2_o3_kernel<4, true, 0>"}[order]
    rows, cols = _subset(4096, 40, order), _subset(4096, 40, 0)
    xg = x[rows][:, cols].cpu().numpy().astype(np.float64)
    yg = y[rows][:, cols].cpu().numpy().astype(np.float64)
    kw = dict(timestep=-900.0, SETTLS_order=4, interp_order=order, cyclic_xboundary=True)
    x32, y32 = O.parcel_propagation(u, v, lat, lon, seed_lat=slat[rows], seed_lon=slon[cols], **kw)
    x64, y64 = O.parcel_propagation(u.astype(np.float64), v.astype(np.float64), lat.astype(np.float64),
                                    lon.astype(np.float64), seed_lat=slat[rows].astype(np.float64),
                                    seed_lon=slon[cols].astype(np.float64), **kw)

    from tests._fullsize import positions_check
    # 96 steps x 5 position updates in float32 at |x| ~ 180 (ulp 1.5e-5): a few 1e-4 degrees, amplified where the
    # flow stretches.  The GPU must sit in the band of the float32 oracle's own error: median, p99 and maximum.
    positions_check(eng, f, slat, slon, rows, cols, xg, yg, (x32, y32), (x64, y64), f"C3 order {order}",
                    (1e-4, 5e-4, 2e-3), interp_order=order)


def test_config3_full_size_settls_order_0_subset_vs_oracle(eng, c3):
    """SETTLS_order = 0 is the LIBRARY default (LCS/trajectory.py:14, LCS/LCS.py:26): one Euler sample per level and nothing to
    stage a tile for -- the two-seed kernel compiled for K = 0, no tile and no iteration blocks (round 6; below 2^23
    seeds the one-seed direct-gather kernel): another code path than the K = 4 anchors above (include/lcs_hip.h).  The same
    subset-vs-oracle anchor at 4096^2 seeds x 96 steps, the dispatched kernel asserted by name; and K = 1, 2 (their own instances)."""
    from oracle import lcs_oracle as O
    from tests._fullsize import positions_check
    u, v, lat, lon, slat, slon = c3
    f = eng.prepare_field(u, v, lat, lon, 1)
    rows, cols = _subset(4096, 40, 1), _subset(4096, 40, 0)
    for K, kernel in ((0, "advect_lds2_kernel<0, true, 0>"), (1, "advect_lds2_kernel<1, true, 0>"), (2, "advect_lds2_kernel<2, true, 0>")):
        x, y = eng.advect(f, slat, slon, -900.0, SETTLS_order=K, interp_order=1, cyclic_xboundary=True)
        assert eng.last_advect_kernel() == kernel, (K, eng.last_advect_kernel())
        xg = x[rows][:, cols].cpu().numpy().astype(np.float64)
        yg = y[rows][:, cols].cpu().numpy().astype(np.float64)
        kw = dict(timestep=-900.0, SETTLS_order=K, interp_order=1, cyclic_xboundary=True)
        x32, y32 = O.parcel_propagation(u, v, lat, lon, seed_lat=slat[rows], seed_lon=slon[cols], **kw)
        x64, y64 = O.parcel_propagation(u.astype(np.float64), v.astype(np.float64), lat.astype(np.float64),
                                        lon.astype(np.float64), seed_lat=slat[rows].astype(np.float64),
                                        seed_lon=slon[cols].astype(np.float64), **kw)
        # (1 + K) position updates per step instead of 5: the floors scale with the updates made
        positions_check(eng, f, slat, slon, rows, cols, xg, yg, (x32, y32), (x64, y64), f"C3 K={K}",
                        (1e-4, 5e-4, 2e-3), interp_order=1, **({"K": K} if K != 4 else {}))


def test_settls_order_0_two_seed_kernel_equals_the_direct_kernel_bit_for_bit(eng, c3):
    """Round 6: from 2^23 seeds per call SETTLS_order = 0 runs on the two-seed kernel (compiled for K = 0 when cyclic, the
    run-time-K instance otherwise) instead of the one-seed direct-gather kernel -- the same Euler sample, update and clamps, so
    the same bits: 2048 x 4096 seeds x 24 levels, cyclic and not, with the trajectory."""
    u, v, lat, lon, slat, slon = c3
    f = eng.prepare_field(u[:25], v[:25], lat, lon, 1)
    rows = slat[1024:3072]
    for cyclic, name in ((True, "advect_lds2_kernel<0, true, 0>"), (False, "advect_lds2_kernel<0, false, 0>")):
        kw = dict(SETTLS_order=0, interp_order=1, cyclic_xboundary=cyclic, noncyclic_clamp="pointwise", row0=1024, ny_global=4096)
        a = eng.advect(f, rows, slon, -900.0, **kw)
        assert eng.last_advect_kernel() == name, eng.last_advect_kernel()
        try:
            eng.set_lds_tiles(0)
            b = eng.advect(f, rows, slon, -900.0, **kw)
            assert eng.last_advect_kernel() == "advect_kernel_f32<1>", eng.last_advect_kernel()
        finally:
            eng.set_lds_tiles(-1)
        for p_, q_ in zip(a, b):
            assert bool((p_ == q_).all()), (cyclic, name)
    # ... SETTLS_order 1, 2, 3: instances of their own too; the one-seed LDS kernel (what a row shard below 2^23 seeds runs) gives the same bits
    for K in (1, 2, 3):
        kw = dict(SETTLS_order=K, interp_order=1, cyclic_xboundary=True, row0=1024, ny_global=4096)
        a = eng.advect(f, rows, slon, -900.0, **kw)
        assert eng.last_advect_kernel() == f"advect_lds2_kernel<{K}, true, 0>", eng.last_advect_kernel()
        try:
            eng.set_lds_tiles(2)
            b = eng.advect(f, rows, slon, -900.0, **kw)
            assert eng.last_advect_kernel().startswith("advect_lds_kernel<1"), eng.last_advect_kernel()
        finally:
            eng.set_lds_tiles(-1)
        assert bool((a[0] == b[0]).all() and (a[1] == b[1]).all()), K
    # ... interp_order = 3 with SETTLS_order = 0, the reference's DEFAULT arguments (LCS/trajectory.py:14-16): its own instance too
    f3 = eng.prepare_field(u[:25], v[:25], lat, lon, 3)
    kw = dict(SETTLS_order=0, interp_order=3, cyclic_xboundary=True, row0=1024, ny_global=4096)
    for cyclic, name in ((True, "advect_lds2_o3_kernel<0, true, 0>"), (False, "advect_lds2_o3_kernel<0, false, 0>")):
        kw.update(cyclic_xboundary=cyclic, noncyclic_clamp="pointwise")
        a3 = eng.advect(f3, rows, slon, -900.0, **kw)
        assert eng.last_advect_kernel() == name, eng.last_advect_kernel()
        try:    # (the one-seed LDS kernel, what a row shard below 2^23 seeds runs: bit for bit; the order-3 direct-gather kernel agrees to rounding only)
            eng.set_lds_tiles(2)
            b3 = eng.advect(f3, rows, slon, -900.0, **kw)
            assert eng.last_advect_kernel().startswith("advect_lds_kernel<3"), eng.last_advect_kernel()
        finally:
            eng.set_lds_tiles(-1)
        assert bool((a3[0] == b3[0]).all() and (a3[1] == b3[1]).all()), eng.last_advect_kernel()
    del f3, a3, b3
    # ... and with the trajectory (whole-line stores: the run-time-K instance of the line-store patch mode)
    t = eng.advect(f, rows, slon, -900.0, SETTLS_order=0, interp_order=1, cyclic_xboundary=True, return_traj=True, row0=1024, ny_global=4096)
    assert eng.last_advect_kernel() == "advect_lds2_kernel<-1, true, 2>", eng.last_advect_kernel()
    c = eng.advect(f, rows, slon, -900.0, SETTLS_order=0, interp_order=1, cyclic_xboundary=True, row0=1024, ny_global=4096)
    assert bool((t[0] == c[0]).all() and (t[1] == c[1]).all() and (t[2][-1] == c[0]).all() and (t[3][-1] == c[1]).all())


def test_config3_sharded_equals_unsharded_full_size(eng, c3):
    u, v, lat, lon, slat, slon = c3
    f = eng.prepare_field(u[:9], v[:9], lat, lon, 1)
    x, y = eng.advect(f, slat, slon, -900.0, SETTLS_order=4, interp_order=1)
    xb, yb = eng.advect(f, slat[1024:2048], slon, -900.0, SETTLS_order=4, interp_order=1, row0=1024, ny_global=4096)
    assert bool((xb == x[1024:2048]).all()) and bool((yb == y[1024:2048]).all())
    dlat, dlon = float(slat[1] - slat[0]), float(slon[1] - slon[0])
    s = eng.sigma(x, y, slat, dlat, dlon)
    sb = eng.sigma(x[1022:2050], y[1022:2050], slat[1022:2050], dlat, dlon, ny_global=4096, in_row0=1022,
                   out_row0=1024, n_out_rows=1024)
    assert bool((sb == s[1024:2048]).all()) and bool(np.isfinite(s.cpu().numpy()).all())


def test_uniform_wind_known_answer_full_size(eng):
    # KAT-2 at 4096^2 seeds, float64: every interior seed moves (1+K)*dt*u0*conversion_x(lat_seed) per step (Q4, Q5)
    lat = np.linspace(-80.0, 80.0, 161)
    lon = -180.0 + 0.5 * np.arange(720)
    nt, u0, dt, K = 5, 11.0, 600.0, 4
    U = np.full((nt, lat.size, lon.size), u0)
    f = eng.prepare_field(U, np.zeros_like(U), lat, lon, 1)
    slat = np.linspace(-80.0, 80.0, 4096)
    slon = np.linspace(-180.0, 179.5, 4096)
    x, y = eng.advect(f, slat, slon, dt, SETTLS_order=K, interp_order=1)
    x, y = x.cpu().numpy(), y.cpu().numpy()
    assert np.array_equal(y, np.broadcast_to(slat[:, None], y.shape))
    dl = (nt - 1) * (1 + K) * dt * u0 * 180 / (np.pi * 6371000 * np.abs(np.cos(np.deg2rad(slat))))
    # columns whose whole path stays west of the index-scale seam (Q2) and off the pole rows (Q3)
    cols = slon + dl.max() < 179.0
    got = (x - slon[None, :])[1:-1][:, cols]
    np.testing.assert_allclose(got[:, 1:], np.broadcast_to(dl[1:-1, None], got[:, 1:].shape), rtol=1e-11)


def test_config2_full_size_subset_vs_oracle(eng):
    """configs[1]: 1024^2 nodes, moving vortex, 200 steps, float64, seeds = field nodes."""
    from oracle import lcs_oracle as O
    u, v, lat, lon = flows.config2()
    f = eng.prepare_field(u, v, lat, lon, 1)
    r = eng.lcs(f, lat, lon, -900.0, SETTLS_order=4, interp_order=1, cyclic_xboundary=True)
    family = "advect_wg64_kernel" if os.environ.get("LCS_F64_WG_TILE") == "1" else "advect_lds64_kernel"
    assert eng.last_advect_kernel() == family + "<4, true, 1>", eng.last_advect_kernel()   # (order-1 source: the raw planes)
    rows, cols = _subset(1024, 24, 1), _subset(1024, 24, 0)
    xr_, yr_ = O.parcel_propagation(u, v, lat, lon, timestep=-900.0, SETTLS_order=4, interp_order=1,
                                    cyclic_xboundary=True, seed_lat=lat[rows], seed_lon=lon[cols])
    xg = r["x_dep"][rows][:, cols].cpu().numpy()
    yg = r["y_dep"][rows][:, cols].cpu().numpy()
    print(f"C2: max |dx| {np.abs(xg - xr_).max():.3e} |dy| {np.abs(yg - yr_).max():.3e} deg")
    np.testing.assert_allclose(xg, xr_, rtol=0, atol=1e-9)
    np.testing.assert_allclose(yg, yr_, rtol=0, atol=1e-9)
    s = r["sigma"].cpu().numpy()
    assert np.isfinite(s).all() and s.max() > 1.5      # the vortex does stretch the flow map


def test_config2_order3_full_size_subset_vs_oracle(eng):
    """configs[1] as the reference itself would run it: float64, seeds = field nodes (LCS/trajectory.py:68-70) and the
    default interp_order=3 (:16), on the float64 order-3 LDS-tile kernel.  12 steps: the order-3 oracle re-runs scipy's
    whole-field prefilter 18 times per step."""
    from oracle import lcs_oracle as O
    u, v, lat, lon = flows.config2(nt=13)
    f = eng.prepare_field(u, v, lat, lon, 3)
    assert eng.last_pack_kernel() == _o3_f64_pack("double"), eng.last_pack_kernel()   # both prefilter sweeps in one pass (round 5)
    x, y = eng.advect(f, lat, lon, -900.0, SETTLS_order=4, interp_order=3, cyclic_xboundary=True)
    assert eng.last_advect_kernel() == "advect_lds64_o3_kernel<4, true>", eng.last_advect_kernel()
    rows, cols = _subset(1024, 24, 3), _subset(1024, 24, 0)
    xr_, yr_ = O.parcel_propagation(u, v, lat, lon, timestep=-900.0, SETTLS_order=4, interp_order=3,
                                    cyclic_xboundary=True, seed_lat=lat[rows], seed_lon=lon[cols])
    xg, yg = x[rows][:, cols].cpu().numpy(), y[rows][:, cols].cpu().numpy()
    print(f"C2 order 3: max |dx| {np.abs(xg - xr_).max():.3e} |dy| {np.abs(yg - yr_).max():.3e} deg")
    np.testing.assert_allclose(xg, xr_, rtol=0, atol=1e-9)
    np.testing.assert_allclose(yg, yr_, rtol=0, atol=1e-9)


def test_config3_return_traj_whole_line_stores_full_size(eng, c3):
    """return_traj (LCS/trajectory.py:125-139) in float32 at configs[2]'s seed count: the two-seed kernel with the
    trajectory slab stores (`advect_lds2_kernel<4, true, 2>`: positions through LDS, whole 128-byte lines, non-temporal).
    Every stored level of a seed subset against the oracle's trajectory list, inside the float32 oracle's band."""
    import torch
    from oracle import lcs_oracle as O
    from tests._fullsize import positions_check
    u, v, lat, lon, slat, slon = c3
    nt = 25
    f = eng.prepare_field(u[:nt], v[:nt], lat, lon, 1)
    x, y, tx, ty = eng.advect(f, slat, slon, -900.0, SETTLS_order=4, interp_order=1, cyclic_xboundary=True, return_traj=True)
    assert eng.last_advect_kernel() == "advect_lds2_kernel<4, true, 2>", eng.last_advect_kernel()
    assert tuple(tx.shape) == (nt, 4096, 4096)
    # entry 0 is the seed grid (trajectory.py:73-74), the last entry the departure points
    assert torch.equal(tx[0], torch.from_numpy(slon).cuda()[None, :].expand(4096, 4096))
    assert torch.equal(ty[0], torch.from_numpy(slat).cuda()[:, None].expand(4096, 4096))
    assert torch.equal(tx[-1], x) and torch.equal(ty[-1], y)
    rows, cols = _subset(4096, 40, 1), _subset(4096, 40, 0)
    ri, ci = torch.from_numpy(rows).cuda(), torch.from_numpy(cols).cuda()
    xg = tx[:, ri][:, :, ci].cpu().numpy().astype(np.float64)
    yg = ty[:, ri][:, :, ci].cpu().numpy().astype(np.float64)
    kw = dict(timestep=-900.0, SETTLS_order=4, interp_order=1, cyclic_xboundary=True, return_traj=True)
    c = lambda a, dt: np.asarray(a).astype(dt, copy=False)
    o = {}
    for dt in (np.float32, np.float64):
        o[dt] = O.parcel_propagation(c(u[:nt], dt), c(v[:nt], dt), c(lat, dt), c(lon, dt), seed_lat=c(slat, dt)[rows],
                                     seed_lon=c(slon, dt)[cols], **kw)
    for lev in range(1, nt):
        # floors grow with the level as the error does (level 1 measures 3.9e-6 / 2.4e-5 / 4.1e-5, level 24 2.3e-5 / 8.6e-4 / 2.1e-3:
        # profiles/r05/parity_stats.txt); until round 5 every level had level 24's
        g = 0.1 + 0.9 * (lev - 1) / 23.0
        positions_check(eng, f, slat, slon, rows, cols, xg[lev], yg[lev], (o[np.float32][0][lev], o[np.float32][1][lev]),
                        (o[np.float64][0][lev], o[np.float64][1][lev]), f"C3 traj level {lev}", (1e-4 * g, 5e-4 * g, 2e-3 * g), nsteps=lev)


@pytest.mark.parametrize("order", [1, 3])
def test_config2_float32_wind_full_size_subset_vs_oracle(eng, order):
    """configs[1]'s shape as reanalysis data reach the reference: a float32 wind on float64 coordinates, seeds = field nodes
    (LCS/trajectory.py:68-70, 86-87, 110-112: numpy promotes).  The kernels that keep the wind float32
    (`advect_lds64w_kernel<4, true>`; at order 3 `advect_lds64w_o3_kernel<4, true>` on float64 coefficients that the one-pass
    prefilter forms straight from the float32 planes) at full size, a seed subset against the oracle run on the same float32
    arrays -- which reproduces numpy's promotion by being numpy."""
    from oracle import lcs_oracle as O
    u, v, lat, lon = flows.config2(nt=13)
    u32, v32 = u.astype(np.float32), v.astype(np.float32)
    f = eng.prepare_field(u32, v32, lat, lon, order)
    assert f.wind_f32 and f.dtype == np.float64
    assert eng.last_pack_kernel() == {1: "pack_fused_kernel", 3: _o3_f64_pack("float")}[order], eng.last_pack_kernel()
    x, y = eng.advect(f, lat, lon, -900.0, SETTLS_order=4, interp_order=order, cyclic_xboundary=True)
    assert eng.last_advect_kernel() == {1: "advect_lds64w_kernel<4, true>", 3: "advect_lds64w_o3_kernel<4, true>"}[order], eng.last_advect_kernel()
    rows, cols = _subset(1024, 24, order), _subset(1024, 24, 0)
    xr_, yr_ = O.parcel_propagation(u32, v32, lat, lon, timestep=-900.0, SETTLS_order=4, interp_order=order,
                                    cyclic_xboundary=True, seed_lat=lat[rows], seed_lon=lon[cols])
    assert xr_.dtype == np.float64
    xg, yg = x[rows][:, cols].cpu().numpy(), y[rows][:, cols].cpu().numpy()
    print(f"C2 float32 wind order {order}: max |dx| {np.abs(xg - xr_).max():.3e} |dy| {np.abs(yg - yr_).max():.3e} deg")
    np.testing.assert_allclose(xg, xr_, rtol=0, atol=1e-9)
    np.testing.assert_allclose(yg, yr_, rtol=0, atol=1e-9)
