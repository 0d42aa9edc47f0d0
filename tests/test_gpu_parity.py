"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle on the
same seeded inputs, and against the committed golden fixtures.

Tolerances (stated here once):
  float64  positions |dx|,|dy| <= 1e-9 degrees, sigma relative 1e-7 (the float32 cast
           of X,Y,Z before differencing (Q11) amplifies 1e-13-degree position
           differences: ulp(6.4e6f)=0.5 m over dx ~ 1e5 m), sigma w/o the cast 1e-9.
  float32  judged against the float64 oracle on the same (float32-valued) inputs:
           the GPU's error must be of the size of the float32 *oracle's* own error.
"""
import os

import numpy as np
import pytest

from lagrangiancoherence_amd import flows

pytestmark = pytest.mark.gpu

GOLD = os.path.join(os.path.dirname(__file__), "golden")
POS_ATOL64 = 1e-9
SIG_RTOL64 = 1e-7


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
    return t.detach().cpu().numpy()


def _rand_field(seed, nt=4, ny=23, nx=31, dtype=np.float64, scale=20.0):
    rng = np.random.default_rng(seed)
    lat = np.linspace(-80, 80, ny).astype(dtype)
    lon = (-180 + 360.0 / nx * np.arange(nx)).astype(dtype)
    # smooth-ish random wind so trajectories stay sane
    u = (scale * rng.standard_normal((nt, ny, nx))).astype(dtype)
    v = (0.5 * scale * rng.standard_normal((nt, ny, nx))).astype(dtype)
    return u, v, lat, lon


# ------------------------------------------------------------------ field image
def test_pack_order1_image(eng):
    u, v, lat, lon = _rand_field(1)
    f = eng.prepare_field(u, v, lat, lon, 1, lin_image=True)
    nt, ny, nx = u.shape
    img = _np(f.lin).reshape(nt, ny + 3, nx + 3, 2)
    assert np.array_equal(img[:, 1:ny + 1, 1:nx + 1, 0], u)
    assert np.array_equal(img[:, 1:ny + 1, 1:nx + 1, 1], v)

    def mir(i, n):
        i = abs(i)
        return 2 * (n - 1) - i if i > n - 1 else i
    for py in (0, ny + 1, ny + 2):
        for px in (0, 5, nx + 1, nx + 2):
            assert img[2, py, px, 0] == u[2, mir(py - 1, ny), mir(px - 1, nx)]
    assert img[1, 4, 0, 1] == v[1, 3, 1] and img[1, 4, nx + 2, 1] == v[1, 3, nx - 3]


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("nt", [1, 2, 7, 8, 9, 17])
def test_pack_walks_levels_in_chunks_and_can_build_the_fused_level_image_alone(eng, dtype, nt):
    """The order-1 pack carries level t+1 forward over chunks of 8 levels (each raw level read once): every level of lin and
    of ext = 2 F[t] - F[t+1] for series lengths around the chunk size; lc_field_pack(order 1, packed_dev = NULL) writes the
    fused-level image alone (what a float64 caller with the raw planes as the order-1 source needs), bit-identical."""
    import ctypes as C
    from lagrangiancoherence_amd import _capi
    u, v, lat, lon = _rand_field(100 + nt, nt=nt, ny=9, nx=300, dtype=dtype)      # 303 padded nodes per row: two blocks in x
    ny, nx = u.shape[1:]
    ud, vd = eng.to_device(u, dtype), eng.to_device(v, dtype)
    n = eng.lib.lc_packed_elems(nt, ny, nx)
    lin = eng._empty((n,), dtype)
    ext = eng._empty((max(eng.lib.lc_packed_elems(nt - 1, ny, nx), 1),), dtype)
    ext2 = eng._empty(tuple(ext.shape), dtype).fill_(float("nan"))
    lc = _capi.LC_F32 if dtype == np.float32 else _capi.LC_F64
    eng._use_current_stream()
    _capi.check(eng.lib.lc_field_pack(eng.ctx, eng._ptr(ud), eng._ptr(vd), lc, nt, ny, nx, 1, eng._ptr(lin),
                                      eng._ptr(ext) if nt >= 2 else None), eng.lib)

    def mir(i, m):
        i = np.abs(i)
        return np.where(i > m - 1, 2 * (m - 1) - i, i)
    iy, ix = mir(np.arange(-1, ny + 2), ny), mir(np.arange(-1, nx + 2), nx)
    want = np.stack([u[:, iy][:, :, ix], v[:, iy][:, :, ix]], axis=-1)             # (nt, ny+3, nx+3, 2)
    assert np.array_equal(_np(lin).reshape(want.shape), want)
    if nt >= 2:
        two = dtype(2)
        assert np.array_equal(_np(ext).reshape(nt - 1, ny + 3, nx + 3, 2), two * want[:-1] - want[1:])
        _capi.check(eng.lib.lc_field_pack(eng.ctx, eng._ptr(ud), eng._ptr(vd), lc, nt, ny, nx, 1, None, eng._ptr(ext2)), eng.lib)
        assert np.array_equal(_np(ext2), _np(ext))
    with pytest.raises(ValueError, match="packed_dev may only be NULL"):
        _capi.check(eng.lib.lc_field_pack(eng.ctx, eng._ptr(ud), eng._ptr(vd), lc, nt, ny, nx, 1, None, None), eng.lib)
    with pytest.raises(ValueError, match="packed_dev may only be NULL"):
        _capi.check(eng.lib.lc_field_pack(eng.ctx, eng._ptr(ud), eng._ptr(vd), lc, nt, ny, nx, 3, None, eng._ptr(ext2)), eng.lib)


@pytest.mark.parametrize("dtype,order,fuse", [(np.float64, 1, True), (np.float64, 1, False), (np.float64, 3, True), (np.float64, 3, False),
                                              (np.float64, 2, False), (np.float32, 3, True), (np.float32, 2, False)])
@pytest.mark.parametrize("K", [0, 4])
def test_raw_planes_as_the_order1_source_equal_the_lin_image_bit_for_bit(eng, dtype, order, fuse, K):
    """lc_advect_ex with the raw wind planes instead of the order-1 image (the default of every field except float32 at
    order 1): the pole seed rows' order-1 / 'constant' samples (LCS/tools.py:31-39) and, in float64 at order 1, the Euler
    sample (LCS/trajectory.py:82-84) read the same node values through the same arithmetic -- departure points, trajectories,
    standalone samples and the non-cyclic clamp's sub-step path are bit-identical to the legacy packed_lin form, in every
    kernel family (LDS tiles, direct gathers, exact order)."""
    u, v, lat, lon = flows.era5_like(nt=6, ny=40, nx=72)
    u, v, lat, lon = (a.astype(dtype) for a in (u, v, lat, lon))
    if order == 1:                                        # (a spline prefilter would smear them over whole lines)
        u[2, 38:, 70:] = np.inf                           # non-finite nodes next to the last row / column (mirror pads)
    # seeds = field nodes plus the exact last node row / column of the index map (c = n - 1: the window whose neighbour is
    # the mirrored pad) and a denser block
    slat = np.unique(np.concatenate([lat, np.linspace(lat[0], lat[-1], 57).astype(dtype)]))
    slon = np.unique(np.concatenate([lon, np.linspace(lon[0], lon[-1], 101).astype(dtype)]))
    f_raw = eng.prepare_field(u, v, lat, lon, order, fuse_levels=fuse)
    f_lin = eng.prepare_field(u, v, lat, lon, order, fuse_levels=fuse, lin_image=True)
    assert f_raw.lin is None and f_raw.u is not None and f_lin.lin is not None and f_lin.u is None
    for lds in ((-1, 0) if dtype == np.float64 or order == 3 else (-1,)):
        eng.set_lds_tiles(lds)
        try:
            for cyc in (True, False):
                a = eng.advect(f_raw, slat, slon, -1800.0, K, order, cyc, return_traj=True)
                ka = eng.last_advect_kernel()
                b = eng.advect(f_lin, slat, slon, -1800.0, K, order, cyc, return_traj=True)
                kb = eng.last_advect_kernel()
                for p, q in zip(a, b):
                    assert np.array_equal(_np(p), _np(q), equal_nan=True), (ka, kb, cyc, lds)
                if dtype == np.float64 and order == 1 and K > 0 and cyc:
                    assert ka.endswith(", 1>") and kb.endswith(", 0>"), (ka, kb)     # the raw / lin kernel variants
        finally:
            eng.set_lds_tiles(-1)
    # a row block of a sharded grid (the pole rule by global row index) and the standalone sample
    a = eng.advect(f_raw, slat[:9], slon, -1800.0, K, order, True, row0=0, ny_global=slat.size)
    b = eng.advect(f_lin, slat[:9], slon, -1800.0, K, order, True, row0=0, ny_global=slat.size)
    assert all(np.array_equal(_np(p), _np(q), equal_nan=True) for p, q in zip(a, b))
    px, py = np.meshgrid(slon, slat)
    sa = eng.sample(f_raw, px, py, level=2, interp_order=order)
    sb = eng.sample(f_lin, px, py, level=2, interp_order=order)
    assert all(np.array_equal(_np(p), _np(q), equal_nan=True) for p, q in zip(sa, sb))
    if order != 1:   # "order 1 is always available" on a field prepared for another order
        f_fresh = eng.prepare_field(u, v, lat, lon, order, fuse_levels=fuse)   # sample() FIRST: float32 builds its order-1 image on demand there too
        sa = eng.sample(f_fresh, px, py, level=2, interp_order=1)
        sb = eng.sample(f_lin, px, py, level=2, interp_order=1)
        assert all(np.array_equal(_np(p), _np(q), equal_nan=True) for p, q in zip(sa, sb))
        assert (f_fresh.lin is not None) == (dtype == np.float32)
        a = eng.advect(f_raw, slat, slon, -1800.0, K, 1, True)
        b = eng.advect(f_lin, slat, slon, -1800.0, K, 1, True)
        assert all(np.array_equal(_np(p), _np(q), equal_nan=True) for p, q in zip(a, b))
    # the raw planes are BORROWED from the caller when they were device tensors: an in-place write afterwards is refused
    ud, vd = eng.to_device(u, dtype), eng.to_device(v, dtype)
    f_b = eng.prepare_field(ud, vd, lat, lon, order, fuse_levels=fuse)
    if f_b.u is not None:
        ud.add_(1.0)
        with pytest.raises(RuntimeError, match="modified in place"):
            eng.advect(f_b, slat[:9], slon, -1800.0, K, order, True, row0=0, ny_global=slat.size)


@pytest.mark.parametrize("K,cyclic", [(4, True), (2, True), (4, False), (1, False)])
def test_float64_fused_levels_without_any_packed_image_equal_the_ext_image_bit_for_bit(eng, K, cyclic):
    """lc_advect_args.fuse_levels_raw: float64 at order 1 with the fused-level value 2 F[t] - F[t+1] formed from the raw
    planes inside the kernels (tile staging, out-of-tile gathers, the direct kernel) -- the expression lc_field_pack
    evaluates, one rounding -- so a field prepared with ext_image=False holds NO packed image and still gives the bits of
    the ext-image form: LDS-tile and direct kernels, trajectories, a row block with a continuation, pole rows, seeds on
    the last node row / column (mirrored neighbour), a seed grid sparser and one denser than the field."""
    u, v, lat, lon = flows.era5_like(nt=9, ny=72, nx=144)
    u, v, lat, lon = (a.astype(np.float64) for a in (u * 2.0, v, lat, lon))
    f_img = eng.prepare_field(u, v, lat, lon, 1, ext_image=True)
    f_raw = eng.prepare_field(u, v, lat, lon, 1, ext_image=False)
    assert f_raw.lin is None and f_raw.ext is None and f_raw.cub is None and f_raw.fuse_raw and f_img.ext is not None
    for sny, snx in ((150, 200), (40, 60), (300, 512), (72, 144)):
        slat, slon = (a.astype(np.float64) for a in flows.seed_grid(sny, snx, lat, lon))
        try:
            for mode in (-1, 0):
                eng.set_lds_tiles(mode)
                out = []
                for f in (f_img, f_raw):
                    r = eng.advect(f, slat, slon, -1800.0, SETTLS_order=K, interp_order=1, cyclic_xboundary=cyclic,
                                   noncyclic_clamp="pointwise", return_traj=True)
                    name = eng.last_advect_kernel()
                    lo, hi = 0, sny // 2
                    rb = eng.advect(f, slat[lo:hi], slon, -1800.0, SETTLS_order=K, interp_order=1, cyclic_xboundary=cyclic,
                                    noncyclic_clamp="pointwise", row0=lo, ny_global=sny, t0=3, nsteps=5, start=(r[2][3][lo:hi], r[3][3][lo:hi]))
                    out.append(([_np(t) for t in r] + [_np(t) for t in rb], name))
                (a, na), (b, nb) = out
                assert na.endswith(", 1>") and nb.endswith(", 2>") and ("lds64" in nb) == (mode == -1), (na, nb)
                for p, q in zip(a, b):
                    assert np.array_equal(p, q), (sny, snx, mode, na, nb)
        finally:
            eng.set_lds_tiles(-1)
    # the reference's outer-product clamp falls back to the two-sample sub-step path from the same field
    rng = np.random.default_rng(3)
    la, lo_ = np.linspace(-40, 40, 41), np.linspace(-60, 50, 56)
    uu, vv = 30 + 25 * rng.standard_normal((5, 41, 56)), 8 * rng.standard_normal((5, 41, 56))
    xa, ya = eng.advect(eng.prepare_field(uu, vv, la, lo_, 1, ext_image=True), la, lo_, 7200.0, 2, 1, False)
    ka = eng.last_advect_kernel()
    xb, yb = eng.advect(eng.prepare_field(uu, vv, la, lo_, 1, ext_image=False), la, lo_, 7200.0, 2, 1, False)
    assert ka == eng.last_advect_kernel() == "outer_substep_kernel" and np.array_equal(_np(xa), _np(xb)) and np.array_equal(_np(ya), _np(yb))


@pytest.mark.parametrize("dtype,order", [(np.float64, 1), (np.float64, 3), (np.float32, 1), (np.float32, 3)])
def test_settls_order_0_one_call_form_builds_no_fused_level_image_and_gives_the_same_bits(eng, dtype, order):
    """Engine.pack_and_advect(SETTLS_order=0) -- what the drop-in calls for the library's default SETTLS_order -- does not build
    the fused-level image nothing would read (float64 at order 1: no packed image at all; order 3: the coefficients only;
    float32: the order-1 / coefficient image only) and returns the bits of prepare_field() + advect(): cyclic and not, with
    the trajectory, dense and sparse seeds.  An explicit ext_image / fuse_levels is respected."""
    u, v, lat, lon = flows.era5_like(nt=9, ny=72, nx=144)
    u, v, lat, lon = (a.astype(dtype) for a in (u * 2.0, v, lat, lon))
    full = eng.prepare_field(u, v, lat, lon, order)
    assert full.ext is not None
    for sny, snx in ((72, 144), (150, 200), (40, 60)):
        slat, slon = (a.astype(dtype) for a in flows.seed_grid(sny, snx, lat, lon))
        for cyclic in (True, False):
            kw = dict(cyclic_xboundary=cyclic, noncyclic_clamp="pointwise", return_traj=True)
            a = eng.advect(full, slat, slon, -1800.0, SETTLS_order=0, interp_order=order, **kw)
            f, *b = eng.pack_and_advect(u, v, lat, lon, slat, slon, -1800.0, 0, order, **kw)
            assert f.ext is None and (f.lin is None) == (dtype == np.float64 or order == 3)
            if dtype == np.float64 and order == 1:
                assert f.cub is None and f.fuse_raw
            for p_, q_ in zip(a, b):
                assert np.array_equal(_np(p_), _np(q_)), (sny, snx, cyclic, eng.last_advect_kernel())
    f, *_ = eng.pack_and_advect(u, v, lat, lon, slat, slon, -1800.0, 0, order, ext_image=True)
    assert f.ext is not None


@pytest.mark.parametrize("K,cyclic", [(4, True), (2, True), (4, False), (1, False)])
def test_float64_workgroup_shared_tile_equals_the_per_wave_tiles_bit_for_bit(eng, monkeypatch, K, cyclic):
    """advect_wg64_kernel (LCS_F64_WG_TILE=1 at context creation; round 6's structural attempt at config 2, off by default):
    four waves 2 x 2 around ONE 24 x 20-node tile, the anchor predicted a level ahead, one barrier per level.  Same locate /
    lerp / clamp functions and the same out-of-tile fallback, so the bits are advect_lds64_kernel's: every source form (lin +
    ext images, raw planes + ext image, raw planes only), trajectories, a row block continued from level 3, pole rows, seed
    grids that are the nodes, sparser, denser, and not a multiple of the 16 x 16 patch."""
    from lagrangiancoherence_amd.engine import Engine
    monkeypatch.setenv("LCS_F64_WG_TILE", "1")
    wg = Engine(0)
    monkeypatch.delenv("LCS_F64_WG_TILE")
    try:
        u, v, lat, lon = flows.era5_like(nt=9, ny=72, nx=144)
        u, v, lat, lon = (a.astype(np.float64) for a in (u * 2.0, v, lat, lon))
        for ext_image, raw_tag in ((True, ", 1>"), (False, ", 2>")):
            fa, fb = (e.prepare_field(u, v, lat, lon, 1, ext_image=ext_image) for e in (eng, wg))
            for sny, snx in ((72, 144), (150, 200), (40, 60), (131, 77)):
                slat, slon = (a.astype(np.float64) for a in flows.seed_grid(sny, snx, lat, lon))
                out = []
                for e, f in ((eng, fa), (wg, fb)):
                    r = e.advect(f, slat, slon, -1800.0, SETTLS_order=K, interp_order=1, cyclic_xboundary=cyclic,
                                 noncyclic_clamp="pointwise", return_traj=True)
                    name = e.last_advect_kernel()
                    lo, hi = 0, sny // 2
                    rb = e.advect(f, slat[lo:hi], slon, -1800.0, SETTLS_order=K, interp_order=1, cyclic_xboundary=cyclic,
                                  noncyclic_clamp="pointwise", row0=lo, ny_global=sny, t0=3, nsteps=5, start=(r[2][3][lo:hi], r[3][3][lo:hi]))
                    out.append(([_np(t) for t in r] + [_np(t) for t in rb], name))
                (a, na), (b, nb) = out
                assert na.startswith("advect_lds64_kernel") and nb.startswith("advect_wg64_kernel") and na.endswith(raw_tag) and nb.endswith(raw_tag), (na, nb)
                for p, q in zip(a, b):
                    assert np.array_equal(p, q), (sny, snx, ext_image, na, nb)
    finally:
        wg.close()


@pytest.mark.parametrize("K,cyclic", [(4, True), (2, True), (4, False), (0, True)])
def test_float64_order3_fused_levels_without_the_ext_image_equal_it_bit_for_bit(eng, K, cyclic):
    """lc_advect_args.fuse_levels_raw at order 3: float64 with the fused-level COEFFICIENTS 2 c[t] - c[t+1] formed from the
    coefficient image inside the kernels (the iteration tile while it is staged, the out-of-tile windows from both levels,
    the direct kernel) -- lc_field_pack's own expression, one rounding -- so a field prepared with ext_image=False holds the
    coefficient image alone and gives the bits of the ext-image form (the default: measured faster): LDS-tile and direct
    kernels, trajectories, a row block with a continuation, pole rows, sparse and dense seed grids, seeds = nodes."""
    u, v, lat, lon = flows.era5_like(nt=9, ny=72, nx=144)
    u, v, lat, lon = (a.astype(np.float64) for a in (u * 2.0, v, lat, lon))
    f_img = eng.prepare_field(u, v, lat, lon, 3)
    f_cub = eng.prepare_field(u, v, lat, lon, 3, ext_image=False)
    assert f_cub.ext is None and f_cub.cub is not None and f_cub.fuse_raw and f_img.ext is not None and not f_img.fuse_raw
    assert np.array_equal(_np(f_cub.cub), _np(f_img.cub))          # pads included (pads_only_kernel == pads_ext_kernel's)
    for sny, snx in ((150, 200), (40, 60), (300, 512), (72, 144)):
        slat, slon = (a.astype(np.float64) for a in flows.seed_grid(sny, snx, lat, lon))
        try:
            for mode in (-1, 0):
                eng.set_lds_tiles(mode)
                out = []
                for f in (f_img, f_cub):
                    r = eng.advect(f, slat, slon, -1800.0, SETTLS_order=K, interp_order=3, cyclic_xboundary=cyclic,
                                   noncyclic_clamp="pointwise", return_traj=True)
                    name = eng.last_advect_kernel()
                    lo, hi = 0, sny // 2
                    rb = eng.advect(f, slat[lo:hi], slon, -1800.0, SETTLS_order=K, interp_order=3, cyclic_xboundary=cyclic,
                                    noncyclic_clamp="pointwise", row0=lo, ny_global=sny, t0=3, nsteps=5, start=(r[2][3][lo:hi], r[3][3][lo:hi]))
                    out.append(([_np(t) for t in r] + [_np(t) for t in rb], name))
                (a, na), (b, nb) = out
                if mode == -1:
                    assert "lds64_o3" in na and "lds64_o3" in nb and nb.endswith("cub>") and not na.endswith("cub>"), (na, nb)
                else:
                    assert na == nb == "advect_kernel<double, 3, true, 0>", (na, nb)
                for p, q in zip(a, b):
                    assert np.array_equal(p, q), (sny, snx, mode, na, nb)
        finally:
            eng.set_lds_tiles(-1)
    # order 1 on such a field ("order 1 is always available": the raw planes, two-sample form) is untouched by the flag
    slat, slon = (a.astype(np.float64) for a in flows.seed_grid(60, 90, lat, lon))
    a = eng.advect(f_img, slat, slon, -1800.0, K, 1, cyclic, noncyclic_clamp="pointwise")
    b = eng.advect(f_cub, slat, slon, -1800.0, K, 1, cyclic, noncyclic_clamp="pointwise")
    assert all(np.array_equal(_np(p), _np(q)) for p, q in zip(a, b))


@pytest.mark.parametrize("dtype,order", [(np.float64, 3), (np.float64, 1), (np.float32, 3)])
@pytest.mark.parametrize("chunk", [1, 3, 4, 7])
def test_pipelined_pack_and_advect_equals_the_serial_form_bit_for_bit(eng, O, dtype, order, chunk):
    """Engine.pack_and_advect(pipeline=True): the images of level chunk k+1 packed on a side stream while chunk k is advected,
    each advect continuing in place (lc_advect_from) -- the images and the departure points of prepare_field + advect, bit
    for bit, for chunk sizes that divide the series, do not, and leave a last chunk of one level; the field it returns is
    complete (a second advect from it gives the same answer).  The default (pipeline=None) is the serial form everywhere since
    round 5 (with the one-pass float64 prefilter the pipelined form no longer pays); what cannot be pipelined falls back."""
    u, v, lat, lon = flows.era5_like(nt=9, ny=72, nx=144)
    u, v, lat, lon = (a.astype(dtype) for a in (u, v, lat, lon))
    slat, slon = (a.astype(dtype) for a in flows.seed_grid(150, 200, lat, lon))
    f0 = eng.prepare_field(u, v, lat, lon, order)
    x0, y0 = eng.advect(f0, slat, slon, -1800.0, 4, order, True)
    f1, x1, y1 = eng.pack_and_advect(u, v, lat, lon, slat, slon, -1800.0, 4, order, True, pipeline=True, chunk=chunk)
    assert np.array_equal(_np(x0), _np(x1)) and np.array_equal(_np(y0), _np(y1)), (dtype, order, chunk)
    assert f1.lin is None and np.array_equal(_np(f1.ext), _np(f0.ext))
    if dtype == np.float64 and order == 3:       # ... and without the ext image (ext_image=False: the kernels form it from cub)
        f2, x3, y3 = eng.pack_and_advect(u, v, lat, lon, slat, slon, -1800.0, 4, order, True, pipeline=True, chunk=chunk, ext_image=False)
        assert f2.ext is None and f2.fuse_raw and np.array_equal(_np(f2.cub), _np(f0.cub)) and np.array_equal(_np(x3), _np(x0)) and np.array_equal(_np(y3), _np(y0))
        assert eng.last_advect_kernel().endswith("cub>")
    if order == 3:
        assert np.array_equal(_np(f1.cub), _np(f0.cub))
    x2, y2 = eng.advect(f1, slat, slon, -1800.0, 4, order, True)
    assert np.array_equal(_np(x2), _np(x0)) and np.array_equal(_np(y2), _np(y0))
    if dtype == np.float64:
        xo, yo = O.parcel_propagation(u, v, lat, lon, timestep=-1800.0, SETTLS_order=4, interp_order=order, cyclic_xboundary=True,
                                      seed_lat=slat, seed_lon=slon)
        d = np.abs(_np(x1) - xo)
        assert np.minimum(d, np.abs(d - 360)).max() < POS_ATOL64 and np.abs(_np(y1) - yo).max() < POS_ATOL64
    if chunk == 3:
        # the default form on this small grid is the serial one; trajectories, the non-cyclic clamp and the exact order fall back
        assert not eng.pipeline_pays(dtype, order, True, 8, 150 * 200, True) and not eng.pipeline_pays(np.float64, 3, True, 200, 1 << 20, True)
        assert not eng.pipeline_pays(np.float64, 1, True, 200, 1 << 20, True) and not eng.pipeline_pays(np.float32, 3, True, 200, 1 << 24, True)
        r = eng.pack_and_advect(u, v, lat, lon, slat, slon, -1800.0, 4, order, True, pipeline=True, chunk=chunk, return_traj=True)
        assert len(r) == 5 and np.array_equal(_np(r[1]), _np(x0)) and np.array_equal(_np(r[3][-1]), _np(x0))
        r = eng.pack_and_advect(u, v, lat, lon, slat, slon, -1800.0, 4, order, False, pipeline=True, chunk=chunk, noncyclic_clamp="pointwise")
        xa, _ = eng.advect(f0, slat, slon, -1800.0, 4, order, False, noncyclic_clamp="pointwise")
        assert np.array_equal(_np(r[1]), _np(xa))
        w = eng.lcs_wind(u, v, lat, lon, slat, slon, -1800.0, 4, order, True, pipeline=True)
        l = eng.lcs(f0, slat, slon, -1800.0, 4, order, True)
        assert np.array_equal(_np(w["sigma"]), _np(l["sigma"])) and np.array_equal(_np(w["x_dep"]), _np(x0))


@pytest.mark.parametrize("dtype,tol", [(np.float64, 2e-13), (np.float32, 2e-5)])
def test_prefilter_matches_scipy(eng, O, dtype, tol):
    u, v, lat, lon = _rand_field(2, nt=2, ny=19, nx=37, dtype=dtype, scale=1.0)
    f = eng.prepare_field(u, v, lat, lon, 3)
    nt, ny, nx = u.shape
    img = _np(f.cub).reshape(nt, ny + 3, nx + 3, 2).astype(np.float64)
    for t in range(nt):
        np.testing.assert_allclose(img[t, 1:ny + 1, 1:nx + 1, 0], O.spline_prefilter_mirror(u[t]), atol=tol)
        np.testing.assert_allclose(img[t, 1:ny + 1, 1:nx + 1, 1], O.spline_prefilter_mirror(v[t]), atol=tol)
    # pads mirror the coefficients
    assert np.array_equal(img[0, 0, 1:nx + 1, 0], img[0, 2, 1:nx + 1, 0])
    assert np.array_equal(img[0, ny + 2, 1:nx + 1, 1], img[0, ny - 2, 1:nx + 1, 1])


# ------------------------------------------------------------------ orders 2, 4, 5 (generic direct kernels)
@pytest.mark.parametrize("order", [2, 4, 5])
def test_general_spline_orders_vs_scipy_and_oracle(eng, O, order):
    """LCS/trajectory.py:16 + LCS/tools.py:26-30 hand any interp_order to scipy.  Prefilter vs
    scipy.ndimage.spline_filter itself, one interpolation pass vs scipy.ndimage.map_coordinates through the
    oracle's xr_map_coordinates (index scale, pole rows, wrap), then whole advections vs the oracle."""
    from scipy.ndimage import spline_filter
    tol = 1e-12 if order == 2 else 5e-11       # orders 4, 5: scipy's pole constants differ in the last bit
    u, v, lat, lon = _rand_field(40 + order, nt=4, ny=21, nx=37)
    f = eng.prepare_field(u, v, lat, lon, order)
    nt, ny, nx = u.shape
    img = _np(f.cub).reshape(nt, ny + 3, nx + 3, 2)
    np.testing.assert_allclose(img[1, 1:ny + 1, 1:nx + 1, 0], spline_filter(u[1], order=order, mode="mirror"), atol=tol)
    rng = np.random.default_rng(order)
    py = rng.uniform(lat[0] - 5, lat[-1] + 5, (ny, nx))
    px = rng.uniform(lon[0] - 30, lon[-1] + 30, (ny, nx))
    su, sv = eng.sample(f, px, py, level=2, interp_order=order)
    np.testing.assert_allclose(_np(su), O.xr_map_coordinates(u[2], lat, lon, px, py, order=order), rtol=0, atol=20 * tol)
    np.testing.assert_allclose(_np(sv), O.xr_map_coordinates(v[2], lat, lon, px, py, order=order), rtol=0, atol=20 * tol)
    for K, cyc in ((0, True), (2, True), (1, False)):
        kw = dict(timestep=-3600.0, SETTLS_order=K, interp_order=order, cyclic_xboundary=cyc)
        x, y = eng.advect(f, lat, lon, -3600.0, K, order, cyc)
        xr_, yr_ = O.parcel_propagation(u, v, lat, lon, **kw)
        np.testing.assert_allclose(_np(x), xr_, rtol=0, atol=POS_ATOL64)
        np.testing.assert_allclose(_np(y), yr_, rtol=0, atol=POS_ATOL64)
    # float32 fields: scipy evaluates in double and rounds to the field dtype; so does the generic kernel
    u32, v32, la32, lo32 = (a.astype(np.float32) for a in (u, v, lat, lon))
    f32 = eng.prepare_field(u32, v32, la32, lo32, order)
    x, y = eng.advect(f32, la32, lo32, -3600.0, 2, order, True)
    x32, y32 = O.parcel_propagation(u32, v32, la32, lo32, timestep=-3600.0, SETTLS_order=2, interp_order=order, cyclic_xboundary=True)
    d = np.abs(_np(x).astype(np.float64) - x32)
    d = np.maximum(np.minimum(d, np.abs(d - 360)), np.abs(_np(y) - y32))
    # (the float32 coefficient image and float32 positions meet a spatially uncorrelated random wind)
    assert np.percentile(d, 99) < 2e-4 and d.max() < 5e-3
    with pytest.raises(ValueError):
        eng.advect(f, lat, lon, -3600.0, 1, 3, True)           # prepared for another order
    with pytest.raises(ValueError):
        eng.prepare_field(u, v, lat, lon, 6)


# ------------------------------------------------------------------ advection
@pytest.mark.parametrize("order", [1, 3])
@pytest.mark.parametrize("K", [0, 1, 4])
@pytest.mark.parametrize("dt", [-3600.0, 5400.0])
def test_advect_f64_random_field(eng, O, order, K, dt):
    u, v, lat, lon = _rand_field(3 + K, nt=5)
    f = eng.prepare_field(u, v, lat, lon, order)
    x, y = eng.advect(f, lat, lon, dt, SETTLS_order=K, interp_order=order, cyclic_xboundary=True)
    xr_, yr_ = O.parcel_propagation(u, v, lat, lon, timestep=dt, SETTLS_order=K, interp_order=order,
                                    cyclic_xboundary=True)
    np.testing.assert_allclose(_np(y), yr_, rtol=0, atol=POS_ATOL64)
    np.testing.assert_allclose(_np(x), xr_, rtol=0, atol=POS_ATOL64)


def test_advect_noncyclic_clamp(eng, O):
    u, v, lat, lon = _rand_field(11, nt=4, scale=60.0)
    f = eng.prepare_field(u, v, lat, lon, 1)
    x, y = eng.advect(f, lat, lon, 7200.0, SETTLS_order=2, interp_order=1, cyclic_xboundary=False,
                      noncyclic_clamp="pointwise")
    xr_, yr_ = O.parcel_propagation(u, v, lat, lon, timestep=7200.0, SETTLS_order=2, interp_order=1,
                                    cyclic_xboundary=False, noncyclic_clamp="pointwise")
    np.testing.assert_allclose(_np(x), xr_, rtol=0, atol=POS_ATOL64)
    np.testing.assert_allclose(_np(y), yr_, rtol=0, atol=POS_ATOL64)
    assert _np(x).min() >= lon.min() and _np(x).max() <= lon.max()


@pytest.mark.parametrize("order", [1, 3])
@pytest.mark.parametrize("K", [0, 2])
def test_advect_noncyclic_reference_outer_clamp(eng, O, order, K):
    """cyclic_xboundary=False as the reference computes it (Q9): `positions_x[np.where(x < x_min)] = x_min` on a
    DataArray sets the whole cross product of offending rows x columns (LCS/trajectory.py:96-97, 122-123)."""
    u, v, lat, lon = _rand_field(11 + K, nt=4, scale=60.0)
    f = eng.prepare_field(u, v, lat, lon, order)
    kw = dict(timestep=7200.0, SETTLS_order=K, interp_order=order, cyclic_xboundary=False)
    x, y, tx, ty = eng.advect(f, lat, lon, 7200.0, K, order, False, return_traj=True)   # default = reference_outer
    assert eng.last_advect_kernel() == "outer_substep_kernel"
    xr_, yr_ = O.parcel_propagation(u, v, lat, lon, noncyclic_clamp="reference_outer", return_traj=True, **kw)
    xp_, _ = O.parcel_propagation(u, v, lat, lon, noncyclic_clamp="pointwise", **kw)
    assert np.abs(xr_[-1] - xp_).max() > 1.0          # the two rules really differ on this input
    np.testing.assert_allclose(_np(tx), xr_, rtol=0, atol=POS_ATOL64)
    np.testing.assert_allclose(_np(ty), yr_, rtol=0, atol=POS_ATOL64)
    assert np.array_equal(_np(x), _np(tx)[-1]) and np.array_equal(_np(y), _np(ty)[-1])


@pytest.mark.parametrize("fuse", [False, True])
def test_advect_noncyclic_reference_outer_restarts_at_the_chunk_where_a_parcel_first_leaves(eng, O, fuse):
    """A long regional series whose parcels stay inside the box for the first 21 levels: the fused kernel runs in chunks
    of 16 levels with the clamp flag read back after each, and the sub-step path (the reference's outer-product rule)
    restarts from the positions saved before the SECOND chunk instead of from the seed grid -- same answer as the oracle
    running the rule from the start, trajectories included."""
    u, v, lat, lon = _rand_field(9, nt=40, ny=19, nx=27, scale=3.0)
    u[:21] *= 0.02                      # nearly at rest, then a strong zonal flow pushes parcels out of both edges
    u[21:] = u[21:] * 10 + 40.0
    f = eng.prepare_field(u, v, lat, lon, 1, fuse_levels=fuse)
    kw = dict(timestep=3600.0, SETTLS_order=1, interp_order=1, cyclic_xboundary=False)
    x, y, tx, ty = eng.advect(f, lat, lon, 3600.0, 1, 1, False, return_traj=True)
    assert eng.last_advect_kernel() == "outer_substep_kernel"
    xr_, yr_ = O.parcel_propagation(u, v, lat, lon, noncyclic_clamp="reference_outer", return_traj=True, **kw)
    xp_, _ = O.parcel_propagation(u, v, lat, lon, noncyclic_clamp="pointwise", **kw)
    assert np.abs(xr_[-1] - xp_).max() > 1.0        # the two clamps really differ on this flow
    np.testing.assert_allclose(_np(tx), xr_, rtol=0, atol=POS_ATOL64)
    np.testing.assert_allclose(_np(ty), yr_, rtol=0, atol=POS_ATOL64)
    assert np.array_equal(_np(x), _np(tx)[-1]) and np.array_equal(_np(y), _np(ty)[-1])
    # nothing had left the box by the end of the first chunk (level 16): that is where the restart positions come from
    assert (np.abs(xr_[16]) <= np.abs(lon).max()).all()


def test_advect_noncyclic_reference_outer_float32_and_fast_path(eng, O):
    # float32: same band as the float32 oracle against float64
    u, v, lat, lon = _rand_field(5, nt=4, scale=60.0, dtype=np.float32)
    f = eng.prepare_field(u, v, lat, lon, 1)
    x, y = eng.advect(f, lat, lon, 7200.0, 2, 1, False)
    kw = dict(timestep=7200.0, SETTLS_order=2, interp_order=1, cyclic_xboundary=False, noncyclic_clamp="reference_outer")
    x32, _ = O.parcel_propagation(u, v, lat, lon, **kw)
    x64, _ = O.parcel_propagation(*(a.astype(np.float64) for a in (u, v, lat, lon)), **kw)
    # a 1-ulp difference can move a parcel across the bound and switch a whole row x column cross product:
    # compare where float32 oracle and float64 oracle agree on the clamp pattern
    same = np.abs(x32 - x64) < 1e-2
    eg = np.abs(_np(x).astype(np.float64) - x64)[same]
    eo = np.abs(x32 - x64)[same]
    print(f"outer clamp float32: gpu median {np.median(eg):.2e} p99 {np.percentile(eg, 99):.2e} max {eg.max():.2e}; "
          f"oracle32 median {np.median(eo):.2e} p99 {np.percentile(eo, 99):.2e}")
    assert same.mean() > 0.9 and (eg < 1e-2).mean() > 0.97      # the same rows x columns were clamped
    assert np.median(eg) <= max(4 * np.median(eo), 1e-5) and np.percentile(eg, 90) < 1e-3
    # purely meridional wind: no longitude changes, nothing leaves the box (any zonal wind pushes the last seed
    # column out, whose index wraps to node 1 under Q2): the fused kernel's answer stands, no sub-step path
    u2, v2 = np.zeros_like(u), v * np.float32(0.1)
    f2 = eng.prepare_field(u2, v2, lat, lon, 1)
    xa, ya = eng.advect(f2, lat, lon, 7200.0, 2, 1, False)
    assert eng.last_advect_kernel() != "outer_substep_kernel"
    xb, yb = eng.advect(f2, lat, lon, 7200.0, 2, 1, False, noncyclic_clamp="pointwise")
    assert np.array_equal(_np(xa), _np(xb)) and np.array_equal(_np(ya), _np(yb))
    # the rule couples every row through the offending columns: a call on a row block needs the flag all-reduce of the
    # sharded driver (Engine.set_flag_allreduce) and is refused without it -- no silent per-point fallback
    for clamp in (None, "reference_outer"):
        with pytest.raises(ValueError, match="lc_ctx_set_flag_allreduce"):
            eng.advect(f2, lat[2:9], lon, 7200.0, 2, 1, False, row0=2, ny_global=lat.size, noncyclic_clamp=clamp)
    xs, _ = eng.advect(f2, lat[2:9], lon, 7200.0, 2, 1, False, row0=2, ny_global=lat.size, noncyclic_clamp="pointwise")
    assert np.array_equal(_np(xs), _np(xb)[2:9])
    # with a reducer in place (here a stand-in for one rank holding every row block in turn: the flags of the whole grid
    # come from the unsharded run) the row-block call runs the same sub-step path
    import ctypes as C
    from lagrangiancoherence_amd import _capi
    calls = []
    cb = _capi.FLAG_ALLREDUCE_FN(lambda user, ptr, n: (calls.append(int(n)), 0)[1])
    _capi.check(eng.lib.lc_ctx_set_flag_allreduce(eng.ctx, C.cast(cb, C.c_void_p), None), eng.lib)
    try:
        xw, yw = eng.advect(f, lat, lon, 7200.0, 2, 1, False)          # whole grid through the hook: 1 + 2 per sub-step calls
        assert eng.last_advect_kernel() == "outer_substep_kernel"
        nsub = (u.shape[0] - 1) * 3
        assert calls == [1] + [lon.size] * (2 * nsub)
    finally:
        eng.set_flag_allreduce(enable=False)
    xr0, yr0 = eng.advect(f, lat, lon, 7200.0, 2, 1, False)
    assert np.array_equal(_np(xw), _np(xr0)) and np.array_equal(_np(yw), _np(yr0))


def test_kat_zero_wind_and_uniform_wind(eng):
    # KAT-1 / KAT-2 straight on the GPU
    lat = np.linspace(-80, 80, 21)
    lon = -180 + 10.0 * np.arange(36)
    U0 = np.zeros((4, 21, 36))
    f = eng.prepare_field(U0, U0, lat, lon, 1)
    x, y = eng.advect(f, lat, lon, -3600.0, SETTLS_order=2, interp_order=1)
    X, Y = np.meshgrid(lon, lat)
    exp = X.copy()
    exp[X == -180] = 0.0                       # Q7
    assert np.array_equal(_np(x), exp) and np.array_equal(_np(y), Y)
    u0, dt, K = 7.0, 600.0, 4
    f = eng.prepare_field(np.full_like(U0, u0), U0, lat, lon, 1)
    x, y = eng.advect(f, lat, lon, dt, SETTLS_order=K, interp_order=1)
    dlon = (1 + K) * dt * u0 * 180 / (np.pi * 6371000 * np.abs(np.cos(np.deg2rad(lat))))   # Q4
    moved = _np(x) - X
    np.testing.assert_allclose(moved[1:-1, 1:], np.broadcast_to((3 * dlon)[1:-1, None], (19, 35)), rtol=1e-12)
    assert np.array_equal(moved[-1, 1:], np.zeros(35))      # Q2+Q3: last row sees cval=0
    assert np.array_equal(_np(y), Y)


def test_seed_grid_and_t0_window(eng, O):
    u, v, lat, lon = _rand_field(21, nt=7, ny=25, nx=40)
    slat = np.linspace(lat[0], lat[-1], 37)
    slon = np.linspace(lon[0], lon[-1], 53)
    f = eng.prepare_field(u, v, lat, lon, 3)
    for order in (1, 3):
        x, y = eng.advect(f, slat, slon, -1800.0, SETTLS_order=2, interp_order=order, t0=2, nsteps=3)
        xr_, yr_ = O.parcel_propagation(u, v, lat, lon, timestep=-1800.0, SETTLS_order=2, interp_order=order,
                                        cyclic_xboundary=True, seed_lat=slat, seed_lon=slon, t0=2, nsteps=3)
        np.testing.assert_allclose(_np(x), xr_, rtol=0, atol=POS_ATOL64)
        np.testing.assert_allclose(_np(y), yr_, rtol=0, atol=POS_ATOL64)


def test_return_traj(eng, O):
    u, v, lat, lon = _rand_field(31, nt=5)
    f = eng.prepare_field(u, v, lat, lon, 1)
    x, y, tx, ty = eng.advect(f, lat, lon, 3600.0, SETTLS_order=1, interp_order=1, return_traj=True)
    txr, tyr = O.parcel_propagation(u, v, lat, lon, timestep=3600.0, SETTLS_order=1, interp_order=1,
                                    cyclic_xboundary=True, return_traj=True)
    assert tuple(tx.shape) == txr.shape
    np.testing.assert_allclose(_np(tx), txr, rtol=0, atol=POS_ATOL64)
    np.testing.assert_allclose(_np(ty), tyr, rtol=0, atol=POS_ATOL64)
    assert np.array_equal(_np(tx[-1]), _np(x)) and np.array_equal(_np(ty[0]), np.meshgrid(lon, lat)[1])


def test_row_sharded_advect_is_bit_identical(eng):
    # rows advected as three blocks (with global row offsets) == rows advected at once
    u, v, lat, lon = _rand_field(41, nt=4, ny=29, nx=33)
    f = eng.prepare_field(u, v, lat, lon, 3)
    x, y = eng.advect(f, lat, lon, -3600.0, SETTLS_order=2, interp_order=3)
    for lo, hi in ((0, 9), (9, 20), (20, 29)):
        xb, yb = eng.advect(f, lat[lo:hi], lon, -3600.0, SETTLS_order=2, interp_order=3, row0=lo, ny_global=29)
        assert np.array_equal(_np(xb), _np(x)[lo:hi]) and np.array_equal(_np(yb), _np(y)[lo:hi])


# ------------------------------------------------------------------ sigma
@pytest.mark.parametrize("layout", ["reference", "physical"])
def test_sigma_f64_vs_oracle(eng, O, layout):
    rng = np.random.default_rng(5)
    lat = np.linspace(-70, 70, 45)
    lon = -180 + 4.0 * np.arange(90)
    X, Y = np.meshgrid(lon, lat)
    xd = X + rng.uniform(-3, 3, X.shape)
    yd = np.clip(Y + rng.uniform(-3, 3, X.shape), -70, 70)
    dt_ = O.flowmap_gradient(xd, yd, lat, lon)
    ref = O.sigma_max(dt_, layout)
    sig = _np(eng.sigma(xd, yd, lat, lat[1] - lat[0], lon[1] - lon[0], tensor_layout=layout))
    np.testing.assert_allclose(sig, ref, rtol=SIG_RTOL64)
    # without the Q11 cast the stencil is clean float64
    ref2 = O.sigma_max(O.flowmap_gradient(xd, yd, lat, lon, fd_fp32_cast=False), layout)
    sig2 = _np(eng.sigma(xd, yd, lat, lat[1] - lat[0], lon[1] - lon[0], tensor_layout=layout, fd_fp32_cast=False))
    np.testing.assert_allclose(sig2, ref2, rtol=1e-9)


def test_sigma_nan_in_nan_out(eng):
    lat = np.linspace(-40, 40, 21)
    lon = -180 + 10.0 * np.arange(36)
    X, Y = np.meshgrid(lon, lat)
    xd = X.copy()
    xd[10, 7] = np.nan
    sig = _np(eng.sigma(xd, Y, lat, lat[1] - lat[0], lon[1] - lon[0]))
    bad = np.argwhere(np.isnan(sig))
    # the NaN poisons exactly the stencil footprint: +-2 along each axis through (10,7)
    exp = {(10 + d, 7) for d in (-2, -1, 1, 2)} | {(10, 7 + d) for d in (-2, -1, 1, 2)}
    assert {tuple(b) for b in bad} == exp


def test_sigma_row_window_with_halo(eng):
    rng = np.random.default_rng(6)
    lat = np.linspace(-60, 60, 41)
    lon = -180 + 5.0 * np.arange(72)
    X, Y = np.meshgrid(lon, lat)
    xd = X + rng.uniform(-2, 2, X.shape)
    yd = Y + rng.uniform(-2, 2, X.shape)
    dlat, dlon = lat[1] - lat[0], lon[1] - lon[0]
    full = _np(eng.sigma(xd, yd, lat, dlat, dlon))
    # block rows [15,30) computed from rows [13,32)
    blk = _np(eng.sigma(xd[13:32], yd[13:32], lat[13:32], dlat, dlon, ny_global=41, in_row0=13, out_row0=15,
                        n_out_rows=15))
    assert np.array_equal(blk, full[15:30])
    # first block [0,15) needs rows [0,17); last block [30,41) needs [28,41)
    b0 = _np(eng.sigma(xd[:17], yd[:17], lat[:17], dlat, dlon, ny_global=41, in_row0=0, out_row0=0, n_out_rows=15))
    b2 = _np(eng.sigma(xd[28:], yd[28:], lat[28:], dlat, dlon, ny_global=41, in_row0=28, out_row0=30, n_out_rows=11))
    assert np.array_equal(b0, full[:15]) and np.array_equal(b2, full[30:])
    with pytest.raises(ValueError):   # missing halo is an error, not a silent wrong answer
        eng.sigma(xd[15:30], yd[15:30], lat[15:30], dlat, dlon, ny_global=41, in_row0=15, out_row0=15, n_out_rows=15)


@pytest.mark.parametrize("ny,nx", [(5, 8), (41, 72), (37, 124), (70, 126), (33, 250), (64, 1440), (131, 2050)])
@pytest.mark.parametrize("layout", ["reference", "physical"])
def test_sigma_marching_and_lds_tile_kernels_agree_bitwise(eng, O, ny, nx, layout):
    """float32 sigma: the marching kernel (registers + wavefront shuffles, the default on even widths) and the LDS-tile
    kernel are one arithmetic -- every width class (one span, span edge at 124/126/250, many spans), pole rows,
    NaNs, row windows with halo; and both sit in the float32 oracle's band."""
    rng = np.random.default_rng(ny * 10007 + nx)
    lat = np.linspace(-89.5, 89.5, ny)
    lon = -180 + (360.0 / nx) * np.arange(nx)
    X, Y = np.meshgrid(lon, lat)
    xd = (X + rng.uniform(-1, 1, X.shape) * (360.0 / nx)).astype(np.float32)
    yd = np.clip(Y + rng.uniform(-1, 1, X.shape) * (179.0 / ny), -90, 90).astype(np.float32)
    if ny > 8:
        xd[ny // 2, nx // 3] = np.nan
    dlat, dlon = float(lat[1] - lat[0]), float(lon[1] - lon[0])
    lat32 = lat.astype(np.float32)
    try:
        eng.set_sigma_march(1)      # forced: by default grids below 2^23 cells take the LDS-tile kernel
        a = _np(eng.sigma(xd, yd, lat32, dlat, dlon, tensor_layout=layout))
        assert eng.last_sigma_kernel() == "sigma_march_kernel_f32"
        eng.set_sigma_march(0)
        b = _np(eng.sigma(xd, yd, lat32, dlat, dlon, tensor_layout=layout))
        assert eng.last_sigma_kernel() == "sigma_kernel_f32"
        assert np.array_equal(a, b, equal_nan=True)
        if ny >= 41:   # a row window with halo, through both kernels
            kw = dict(ny_global=ny, in_row0=9, out_row0=11, n_out_rows=17)
            eng.set_sigma_march(1)
            wa = _np(eng.sigma(xd[9:30], yd[9:30], lat32[9:30], dlat, dlon, tensor_layout=layout, **kw))
            eng.set_sigma_march(0)
            wb = _np(eng.sigma(xd[9:30], yd[9:30], lat32[9:30], dlat, dlon, tensor_layout=layout, **kw))
            assert np.array_equal(wa, wb, equal_nan=True) and np.array_equal(wa, a[11:28], equal_nan=True)
    finally:
        eng.set_sigma_march(-1)
    ok = np.isfinite(a)
    assert (~ok).sum() == (8 if ny > 8 else 0)
    # against the float64 oracle on the same float32 inputs: float32 X, Y, Z differences carry the Q11 noise
    ref = O.sigma_max(O.flowmap_gradient(xd.astype(np.float64), yd.astype(np.float64), lat, lon), layout)
    rel = np.abs(a[ok] - ref[ok]) / np.maximum(ref[ok], 1e-30)
    assert np.median(rel) < 2e-3


def test_gaussian_filter_vs_scipy(eng):
    from scipy.ndimage import gaussian_filter
    rng = np.random.default_rng(8)
    a = rng.standard_normal((37, 53))
    for s in (0.8, 2.0):
        np.testing.assert_allclose(_np(eng.gaussian_filter(a, s)), gaussian_filter(a, sigma=s), rtol=0, atol=1e-14)
    a32 = a.astype(np.float32)
    np.testing.assert_allclose(_np(eng.gaussian_filter(a32, 1.5)), gaussian_filter(a32, sigma=1.5), rtol=0, atol=1e-6)


# ------------------------------------------------------------------ golden fixtures (whole path)
def _load(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


@pytest.mark.parametrize("tag,dt,K", [("bwd_k4", -21600, 4), ("fwd_k2", 21600, 2), ("fwd_k4", 21600, 4)])
@pytest.mark.parametrize("order", [3, 1])
def test_golden_config1(eng, tag, dt, K, order):
    g = _load(f"g1_{tag}_o{order}")
    u, v, lat, lon = flows.config1()
    f = eng.prepare_field(u, v, lat, lon, order)
    r = eng.lcs(f, lat, lon, dt, SETTLS_order=K, interp_order=order, cyclic_xboundary=True)
    np.testing.assert_allclose(_np(r["x_dep"]), g["x_dep"], rtol=0, atol=POS_ATOL64)
    np.testing.assert_allclose(_np(r["y_dep"]), g["y_dep"], rtol=0, atol=POS_ATOL64)
    np.testing.assert_allclose(_np(r["sigma"]), g["sigma"], rtol=SIG_RTOL64)


def test_golden_config1_trajectories(eng):
    g = _load("g1_traj_bwd_k4_o3")
    u, v, lat, lon = flows.config1()
    f = eng.prepare_field(u, v, lat, lon, 3)
    x, y, tx, ty = eng.advect(f, lat, lon, -21600, SETTLS_order=4, interp_order=3, return_traj=True)
    np.testing.assert_allclose(_np(tx), g["traj_x"], rtol=0, atol=POS_ATOL64)
    np.testing.assert_allclose(_np(ty), g["traj_y"], rtol=0, atol=POS_ATOL64)


def test_golden_config2_downsampled(eng):
    g = _load("g2_c2_128_k4_o1")
    u, v, lat, lon = flows.config2(n=128, nt=21)
    f = eng.prepare_field(u, v, lat, lon, 1)
    r = eng.lcs(f, lat, lon, -900, SETTLS_order=4, interp_order=1, cyclic_xboundary=True)
    np.testing.assert_allclose(_np(r["x_dep"]), g["x_dep"], rtol=0, atol=POS_ATOL64)
    np.testing.assert_allclose(_np(r["y_dep"]), g["y_dep"], rtol=0, atol=POS_ATOL64)
    np.testing.assert_allclose(_np(r["sigma"]), g["sigma"], rtol=SIG_RTOL64)


@pytest.mark.parametrize("order", [1, 3])
def test_golden_config3_mini_f32(eng, order):
    """float32 path: error vs the float64 answer must be of the size of the float32 oracle's own error."""
    g = _load(f"g3_c3mini_k4_o{order}")
    u, v, lat, lon = flows.era5_like(nt=13, ny=72, nx=144)
    slat, slon = flows.seed_grid(96, 160, lat, lon)
    f = eng.prepare_field(u, v, lat, lon, order)
    assert f.dtype == np.float32
    r = eng.lcs(f, slat, slon, -900, SETTLS_order=4, interp_order=order, cyclic_xboundary=True)
    x, y, s = (_np(r[k]).astype(np.float64) for k in ("x_dep", "y_dep", "sigma"))
    ex_o = np.abs(g["x_dep"] - g["x_dep64"])
    ey_o = np.abs(g["y_dep"] - g["y_dep64"])
    ex_g = np.abs(x - g["x_dep64"])
    ey_g = np.abs(y - g["y_dep64"])
    print(f"order {order}: oracle32 err x {ex_o.max():.3e} y {ey_o.max():.3e}; gpu32 err x {ex_g.max():.3e} y {ey_g.max():.3e}")
    # the east-most seed column sits on the +-180 seam where 1 ulp flips the cyclic rewrite: compare mod 360 there
    ex_g = np.minimum(ex_g, np.abs(ex_g - 360))
    assert ex_g.max() <= max(4 * ex_o.max(), 2e-4) and ey_g.max() <= max(4 * ey_o.max(), 1e-4)
    es_o = np.abs(g["sigma"] - g["sigma64"]) / g["sigma64"]
    es_g = np.abs(s - g["sigma64"]) / g["sigma64"]
    print(f"          sigma rel err oracle32 median {np.median(es_o):.3e} max {es_o.max():.3e}; "
          f"gpu32 median {np.median(es_g):.3e} max {es_g.max():.3e}")
    assert np.median(es_g) <= max(4 * np.median(es_o), 1e-4)
    assert np.percentile(es_g, 99) <= max(4 * np.percentile(es_o, 99), 1e-3)


# ------------------------------------------------------------------ one-call host route + errors
def test_lcs_host_route_matches_engine(eng):
    from lagrangiancoherence_amd.engine import lcs_host
    u, v, lat, lon = flows.config1()
    out = lcs_host(u, v, lat, lon, -21600, SETTLS_order=4, interp_order=3, cyclic_xboundary=True, return_traj=True)
    # float64 at the example's size: the host route keeps numpy / scipy's operation order (LC_F64_AUTO)
    f = eng.prepare_field(u, v, lat, lon, 3, fuse_levels=False)
    r = eng.lcs(f, lat, lon, -21600, SETTLS_order=4, interp_order=3, cyclic_xboundary=True)
    assert eng.last_advect_kernel() == "advect_kernel<double, 3, false, 0>"
    assert np.array_equal(out["x_dep"], _np(r["x_dep"])) and np.array_equal(out["sigma"], _np(r["sigma"]))
    assert out["traj_x"].shape == (8, 89, 180) and np.array_equal(out["traj_x"][-1], out["x_dep"])
    g = np.load(os.path.join(GOLD, "g1_bwd_k4_o3.npz"))
    assert np.abs(out["x_dep"] - g["x_dep"]).max() < 1e-12 and np.abs(out["y_dep"] - g["y_dep"]).max() < 1e-12
    # 'fast' = the engine's default fused-level form, bit for bit; inside the float64 tolerance of the reference
    fast = lcs_host(u, v, lat, lon, -21600, SETTLS_order=4, interp_order=3, cyclic_xboundary=True, float64_fidelity="fast")
    rf = eng.lcs(eng.prepare_field(u, v, lat, lon, 3), lat, lon, -21600, SETTLS_order=4, interp_order=3, cyclic_xboundary=True)
    assert np.array_equal(fast["x_dep"], _np(rf["x_dep"])) and np.abs(fast["x_dep"] - g["x_dep"]).max() < POS_ATOL64


def test_gauss_sigma_path(eng, O):
    u, v, lat, lon = flows.config1()
    f = eng.prepare_field(u, v, lat, lon, 1)
    r = eng.lcs(f, lat, lon, -21600, SETTLS_order=2, interp_order=1, gauss_sigma=1.5)
    s, _, _ = O.lcs(u, v, lat, lon, timestep=-21600, SETTLS_order=2, interp_order=1, cyclic_xboundary=True,
                    gauss_sigma=1.5)
    np.testing.assert_allclose(_np(r["sigma"]), s, rtol=SIG_RTOL64)


def test_errors(eng):
    u, v, lat, lon = _rand_field(51)
    with pytest.raises(ValueError):
        eng.prepare_field(u, v, lat, lon, 0)           # reference: slice(0,-0) empty -> reshape error
    with pytest.raises(ValueError):
        eng.prepare_field(u, v, lat[::-1], lon, 1)     # must be ascending
    f = eng.prepare_field(u, v, lat, lon, 1)
    with pytest.raises(ValueError):
        eng.advect(f, lat, lon, 3600.0, interp_order=3)    # no coefficient image
    with pytest.raises(ValueError):
        eng.advect(f, lat, lon, 3600.0, interp_order=1, t0=2, nsteps=5)   # runs past the last level


@pytest.mark.parametrize("order", [1, 3])
def test_lds_tile_and_direct_gather_kernels_agree(eng, order, monkeypatch):
    """Both float kernels (per-wave LDS tiles / direct gather) on a flow with jets, a seam crossing and
    polar rows, so the tile fallback paths run; they share the arithmetic, so they must agree to rounding."""
    u, v, lat, lon = flows.era5_like(nt=9, ny=72, nx=144)
    u = (u * 2.5).astype(np.float32)                  # up to ~170 m/s: windows leave the tiles
    slat, slon = flows.seed_grid(150, 200, lat, lon)
    f = eng.prepare_field(u, v, lat, lon, order)
    out = {}
    try:
        for flag in ("0", "1"):
            eng.set_lds_tiles(int(flag))     # (the environment variable is read once, at context creation)
            x, y = eng.advect(f, slat, slon, -1800.0, SETTLS_order=4, interp_order=order)
            assert ("lds" in eng.last_advect_kernel()) == (flag == "1")
            out[flag] = (_np(x).astype(np.float64), _np(y).astype(np.float64))
    finally:
        eng.set_lds_tiles(-1)
    dx = np.abs(out["0"][0] - out["1"][0])
    dx = np.minimum(dx, np.abs(dx - 360))
    dy = np.abs(out["0"][1] - out["1"][1])
    print(f"order {order}: LDS vs direct max |dx| {dx.max():.2e} |dy| {dy.max():.2e}, identical {np.mean(dx == 0):.3f}")
    assert np.percentile(dx, 99) < 1e-4 and np.percentile(dy, 99) < 1e-4
    assert dx.max() < 5e-2 and dy.max() < 5e-2


@pytest.mark.parametrize("dtype,tol", [(np.float64, 5e-13), (np.float32, 4e-5)])
def test_prefilter_wide_rows_lds_kernel(eng, O, dtype, tol):
    # nx >= 64 takes the LDS-chunked longitude sweep (3 chunks here, ragged last chunk, ragged row block)
    u, v, lat, lon = _rand_field(61, nt=2, ny=37, nx=150, dtype=dtype, scale=1.0)
    f = eng.prepare_field(u, v, lat, lon, 3)
    nt, ny, nx = u.shape
    img = _np(f.cub).reshape(nt, ny + 3, nx + 3, 2).astype(np.float64)
    for t in range(nt):
        np.testing.assert_allclose(img[t, 1:ny + 1, 1:nx + 1, 0], O.spline_prefilter_mirror(u[t]), atol=tol)
        np.testing.assert_allclose(img[t, 1:ny + 1, 1:nx + 1, 1], O.spline_prefilter_mirror(v[t]), atol=tol)


def test_one_pass_prefilter_matches_the_recursive_sweeps(eng, O, monkeypatch):
    """float32, order 3: the truncated-convolution prefilter (one pass over the raw field, the default) against the
    recursive sweeps (LCS_FIR_PREFILTER=0 at context creation) and against scipy's recursion in float64: ragged
    tiles, pads, the fused-level image; a grid too small for the 14-node halo takes the sweeps either way."""
    from lagrangiancoherence_amd.engine import Engine
    u, v, lat, lon = _rand_field(77, nt=3, ny=97, nx=150, dtype=np.float32, scale=20.0)
    monkeypatch.setenv("LCS_FIR_PREFILTER", "0")
    eng0 = Engine(0)
    monkeypatch.delenv("LCS_FIR_PREFILTER")
    a = eng.prepare_field(u, v, lat, lon, 3)
    assert eng.last_pack_kernel() == "prefilter_fir_kernel", eng.last_pack_kernel()
    b = eng0.prepare_field(u, v, lat, lon, 3)
    assert eng0.last_pack_kernel() == "prefilter_cols_kernel + prefilter_rows_kernel", eng0.last_pack_kernel()
    scale = float(np.abs(u).max())
    for name in ("cub", "ext"):
        x, y = _np(getattr(a, name)).astype(np.float64), _np(getattr(b, name)).astype(np.float64)
        nlev = 3 if name == "cub" else 2          # ext has no last level
        x, y = x.reshape(-1, 100, 153, 2)[:nlev], y.reshape(-1, 100, 153, 2)[:nlev]
        assert x.shape[0] == nlev
        assert np.abs(x - y).max() < 3e-6 * scale, name      # a few float32 ulps of the field's scale
    img = _np(a.cub).astype(np.float64).reshape(3, 100, 153, 2)
    for t in range(3):
        np.testing.assert_allclose(img[t, 1:98, 1:151, 0], O.spline_prefilter_mirror(u[t].astype(np.float64)), atol=3e-6 * scale)
    # pads mirror the coefficients exactly
    assert np.array_equal(img[:, 0, 1:151], img[:, 2, 1:151]) and np.array_equal(img[:, 98, 1:151], img[:, 96, 1:151])
    assert np.array_equal(img[:, 99, 1:151], img[:, 95, 1:151]) and np.array_equal(img[:, :, 0], img[:, :, 2])
    assert np.array_equal(img[:, :, 151], img[:, :, 149]) and np.array_equal(img[:, :, 152], img[:, :, 148])
    # small grid: both contexts run the same sweeps
    u2, v2, lat2, lon2 = _rand_field(78, nt=2, ny=12, nx=40, dtype=np.float32, scale=5.0)
    assert np.array_equal(_np(eng.prepare_field(u2, v2, lat2, lon2, 3).cub), _np(eng0.prepare_field(u2, v2, lat2, lon2, 3).cub))


@pytest.mark.parametrize("ny,nx", [(97, 150), (64, 64), (130, 65), (65, 97), (37, 150), (97, 40), (81, 96), (200, 257),
                                   (300, 150), (513, 70), (260, 700), (1030, 333)])
def test_float64_streaming_prefilter_matches_the_two_march_sweeps_and_scipy(eng, O, monkeypatch, ny, nx):
    """float64, order 3: the streaming prefilter (default; lines of 64 nodes or more, anticausal walks started 32 nodes
    ahead; both axes of 64 nodes or more: ONE pass, the longitude recursion across the lanes of the workgroup --
    prefilter_fused_stream_kernel; otherwise one streaming sweep per axis that is long enough) against the two-march
    kernels (LCS_FIR_PREFILTER=0) and against scipy's recursion: line lengths at the limit (64), one node over a chunk, a
    ragged last chunk, a ragged row block, one axis too short for the streaming form, columns cut into row pieces (300
    rows and more on an idle chip: lcplan::fused_prefilter_split), lines of several workgroups (700 columns), a partly
    filled last wave; pads and the fused-level image follow."""
    from lagrangiancoherence_amd.engine import Engine
    u, v, lat, lon = _rand_field(1000 + ny + nx, nt=3, ny=ny, nx=nx, dtype=np.float64, scale=20.0)
    monkeypatch.setenv("LCS_FIR_PREFILTER", "0")
    eng0 = Engine(0)
    monkeypatch.delenv("LCS_FIR_PREFILTER")
    a = eng.prepare_field(u, v, lat, lon, 3)
    cols, rows = ("prefilter_cols_stream_kernel" if ny >= 64 else "prefilter_cols_kernel"), ("prefilter_rows_stream_kernel" if nx >= 64 else "prefilter_rows_kernel")
    if os.environ.get("LCS_FUSED_PREFILTER", "1") != "0":
        assert eng.last_pack_kernel() == ("prefilter_fused_stream_kernel<double>" if ny >= 64 and nx >= 64 else f"{cols} + {rows}"), eng.last_pack_kernel()
    b = eng0.prepare_field(u, v, lat, lon, 3)
    assert eng0.last_pack_kernel() == "prefilter_cols_kernel + prefilter_rows_kernel", eng0.last_pack_kernel()
    scale = float(np.abs(u).max())
    for name in ("cub", "ext"):
        x, y = _np(getattr(a, name)), _np(getattr(b, name))
        nlev = 3 if name == "cub" else 2
        x, y = x.reshape(-1, ny + 3, nx + 3, 2)[:nlev], y.reshape(-1, ny + 3, nx + 3, 2)[:nlev]
        assert np.abs(x - y).max() <= 3e-15 * scale * (3 if name == "ext" else 1), name   # a few last bits of the line's scale (fma grouping)
    img = _np(a.cub).reshape(3, ny + 3, nx + 3, 2)
    for t in range(3):
        np.testing.assert_allclose(img[t, 1:ny + 1, 1:nx + 1, 0], O.spline_prefilter_mirror(u[t]), rtol=0, atol=2e-14 * scale)
        np.testing.assert_allclose(img[t, 1:ny + 1, 1:nx + 1, 1], O.spline_prefilter_mirror(v[t]), rtol=0, atol=2e-14 * scale)
    assert np.array_equal(img[:, 0, 1:nx + 1], img[:, 2, 1:nx + 1]) and np.array_equal(img[:, :, nx + 2], img[:, :, nx - 2])
    if ny >= 64 and nx >= 64:
        # the one-pass kernel against the two streaming sweeps it replaces (LCS_FUSED_PREFILTER=0): other summation order, last bits
        fused_on = os.environ.get("LCS_FUSED_PREFILTER", "1") != "0"      # (a suite run with the fallback forced compares it with itself)
        monkeypatch.setenv("LCS_FUSED_PREFILTER", "0")
        eng2 = Engine(0)
        monkeypatch.undo()
        c2 = _np(eng2.prepare_field(u, v, lat, lon, 3).cub)
        c1 = _np(a.cub)
        assert np.array_equal(c1, c2) != fused_on and np.abs(c1 - c2).max() <= 3e-15 * scale
        # ... and its float32-wind instance (LC_F64_WIND_F32: float32 planes in, float64 coefficients out) = the float64 one on the same values
        u32, v32 = u.astype(np.float32), v.astype(np.float32)
        fw = eng.prepare_field(u32, v32, lat, lon, 3)
        fd = eng.prepare_field(u32.astype(np.float64), v32.astype(np.float64), lat, lon, 3)
        assert np.array_equal(_np(fw.cub), _np(fd.cub))
        eng2.close()


def test_float64_order3_coefficients_do_not_depend_on_how_the_launch_cut_the_level(eng):
    """Round 5's advisor finding: prefilter_fused_stream_kernel cuts the workgroups of its last round into row pieces
    (lcplan::fused_prefilter_split: which levels, and into how many pieces, follows from nt and the CU count), and a piece used
    to restart the latitude march from a 64-term sum where a whole march carries its recursion on -- the same level came out
    in different bits as part of another series length.  Now EVERY march restarts at every multiple of FUSED_PIECE_ALIGN
    (256) rows by the same function, and pieces begin there only.  Packed here: the same levels of a 600-row field (room for
    a cut at rows 256 and 512) as a series of 2, 3, 5 and 9 levels -- items % CUs and the cut differ from one to the next --
    and as sub-ranges; every level's coefficient image must be identical in all of them, bit for bit."""
    ny, nx, nt = 600, 130, 9
    u, v, lat, lon = _rand_field(4242, nt=nt, ny=ny, nx=nx, dtype=np.float64, scale=20.0)
    full = _np(eng.prepare_field(u, v, lat, lon, 3).cub).reshape(-1, ny + 3, nx + 3, 2)[:nt].copy()
    if os.environ.get("LCS_FUSED_PREFILTER", "1") != "0":      # (a suite run with the two-sweep fallback forced checks that one: it never cut)
        assert eng.last_pack_kernel() == "prefilter_fused_stream_kernel<double>", eng.last_pack_kernel()
    assert np.isfinite(full).all()
    for n in (2, 3, 5):
        part = _np(eng.prepare_field(u[:n], v[:n], lat, lon, 3).cub).reshape(-1, ny + 3, nx + 3, 2)[:n]
        assert np.array_equal(part, full[:n]), f"levels 0..{n - 1} packed as a series of {n} differ from the same levels of a series of {nt}"
    tail = _np(eng.prepare_field(u[4:], v[4:], lat, lon, 3).cub).reshape(-1, ny + 3, nx + 3, 2)[:nt - 4]
    assert np.array_equal(tail, full[4:]), "levels 4..8 packed on their own differ from the same levels of the whole series"
    # and the restart rows are what scipy's recursion gives, to the same few last bits as everywhere else
    from oracle import lcs_oracle as O
    scale = float(np.abs(u).max())
    np.testing.assert_allclose(full[1, 1:ny + 1, 1:nx + 1, 0], O.spline_prefilter_mirror(u[1]), rtol=0, atol=2e-14 * scale)


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("order", [1, 3])
def test_launch_geometry_does_not_change_results(dtype, order, monkeypatch):
    """The global pole rows run in leading workgroups of their own and the tile rows are dealt out from the poles
    (LCS_POLE_BLOCKS, LCS_TILE_ORDER, LCS_XCD_CHUNK_ROWS, read at context creation): none of it may change a bit.
    Shapes where every row is a pole row, row windows holding only the lower / only the upper pole rows / none."""
    from lagrangiancoherence_amd.engine import Engine
    for k in ("LCS_POLE_BLOCKS", "LCS_TILE_ORDER", "LCS_XCD_CHUNK_ROWS"):
        monkeypatch.delenv(k, raising=False)
    eng = Engine(0)            # a fresh context with the default geometry (the shared one may carry kernel choices)
    for k, v in (("LCS_POLE_BLOCKS", "0"), ("LCS_TILE_ORDER", "0"), ("LCS_XCD_CHUNK_ROWS", "0")):
        monkeypatch.setenv(k, v)
    plain = Engine(0)
    u, v, lat, lon = flows.era5_like(nt=6, ny=40, nx=90)
    u, v, lat, lon = (a.astype(dtype) for a in (u, v, lat, lon))
    try:
        fa, fb = eng.prepare_field(u, v, lat, lon, order), plain.prepare_field(u, v, lat, lon, order)
        for sny, snx in ((2 * order, 70), (2 * order + 1, 9), (75, 133), (33, 64), (70, 96)):
            slat, slon = (a.astype(dtype) for a in flows.seed_grid(sny, snx, lat, lon))
            windows = [(0, sny)] + ([(0, sny // 2), (sny // 2, sny), (order, sny - order)] if sny > 4 * order else [])
            for lo, hi in windows:
                kw = dict(SETTLS_order=2, interp_order=order, cyclic_xboundary=True, row0=lo, ny_global=sny)
                xa, ya = eng.advect(fa, slat[lo:hi], slon, -3600.0, **kw)
                xb, yb = plain.advect(fb, slat[lo:hi], slon, -3600.0, **kw)
                assert bool((xa == xb).all()) and bool((ya == yb).all()), (sny, snx, lo, hi)
            if dtype == np.float32 and snx % 4 == 0:
                # trajectories too, with the two-seed kernels forced: whole-line slab stores beside pole rows that are
                # advected by leading workgroups (default) or inside the tiles (LCS_POLE_BLOCKS=0)
                for e_ in (eng, plain):
                    e_.set_lds_tiles(1)
                ta = eng.advect(fa, slat, slon, -3600.0, SETTLS_order=4, interp_order=order, return_traj=True)
                tb = plain.advect(fb, slat, slon, -3600.0, SETTLS_order=4, interp_order=order, return_traj=True)
                for e_ in (eng, plain):
                    e_.set_lds_tiles(-1)
                assert all(bool((a == b).all()) for a, b in zip(ta, tb)), (sny, snx, "traj")
    finally:
        plain.close()
        eng.close()


def test_ensemble_members_are_t0_windows(eng, O):
    # BASELINE config 5 in miniature: member e = the same seeds started at time level e
    from lagrangiancoherence_amd import sharded
    u, v, lat, lon = _rand_field(71, nt=9, ny=25, nx=40)
    slat = np.linspace(lat[0], lat[-1], 30)
    slon = np.linspace(lon[0], lon[-1], 44)
    f = eng.prepare_field(u, v, lat, lon, 1)
    got = {}
    for rank in range(2):          # two "ranks" in one process: the partition is what is under test
        mine, sig = sharded.ensemble_lcs(eng, f, slat, slon, -1800.0, n_members=5, nsteps=4, rank=rank, world=2,
                                         SETTLS_order=2, interp_order=1)
        for e, s in zip(mine, sig):
            got[e] = _np(s)
    assert sorted(got) == [0, 1, 2, 3, 4]
    for e in (0, 3, 4):
        ref, _, _ = O.lcs(u, v, lat, lon, timestep=-1800.0, SETTLS_order=2, interp_order=1, cyclic_xboundary=True,
                          seed_lat=slat, seed_lon=slon, t0=e, nsteps=4)
        np.testing.assert_allclose(got[e], ref, rtol=SIG_RTOL64)
    with pytest.raises(ValueError):
        sharded.ensemble_lcs(eng, f, slat, slon, -1800.0, n_members=6, nsteps=4)
    # members alternate between HIP streams (default 2): same bits as one stream, departure points included
    a = sharded.ensemble_lcs(eng, f, slat, slon, -1800.0, n_members=5, nsteps=4, SETTLS_order=2, return_dpts=True, streams=1)
    for ns in (2, 3):
        b = sharded.ensemble_lcs(eng, f, slat, slon, -1800.0, n_members=5, nsteps=4, SETTLS_order=2, return_dpts=True,
                                 streams=ns)
        assert a[0] == b[0] and all(bool((x == y).all()) for x, y in zip(a[1:], b[1:]))
    # lc_advect_batch: all members through one launch per level chunk == member by member, in place continuation too
    x1, y1 = eng.advect_batch(f, slat, slon, -1800.0, 5, 4, SETTLS_order=2, interp_order=1)
    assert bool((x1 == a[2]).all()) and bool((y1 == a[3]).all())
    xa, ya = eng.advect_batch(f, slat, slon, -1800.0, 5, 1, SETTLS_order=2, interp_order=1)
    eng.advect_batch(f, slat, slon, -1800.0, 5, 3, SETTLS_order=2, interp_order=1, t0=1, start=(xa, ya), out=(xa, ya))
    assert bool((xa == a[2]).all()) and bool((ya == a[3]).all())
    with pytest.raises(ValueError):
        eng.advect_batch(f, slat, slon, -1800.0, 6, 4, SETTLS_order=2)        # member 5 would need level 9 of 9
    # members that are not consecutive start times take the per-member launches on side streams
    got = sharded.ensemble_advect(eng, f, slat, slon, -1800.0, [0, 2, 4], 4, SETTLS_order=2, level_chunk=3)
    for (gx, gy), e in zip(got, (0, 2, 4)):
        assert bool((gx == a[2][e]).all()) and bool((gy == a[3][e]).all())
    # level-major order (every member's first chunk of levels, then every member's next: ensemble_advect) == member by
    # member (level_chunk=0), bit for bit, whatever the chunk and the number of streams
    for chunk, ns in ((0, 1), (1, 2), (3, 1), (3, 3), (None, 2)):
        b = sharded.ensemble_lcs(eng, f, slat, slon, -1800.0, n_members=5, nsteps=4, SETTLS_order=2, return_dpts=True,
                                 streams=ns, level_chunk=chunk)
        assert a[0] == b[0] and all(bool((x == y).all()) for x, y in zip(a[1:], b[1:])), (chunk, ns)


def test_ensemble_with_the_reference_noncyclic_clamp_on_a_long_series(eng, O):
    """cyclic_xboundary=False through ensemble_lcs with more steps than ENSEMBLE_CHUNK (round-3 advisor finding: the
    level-major path continued each member in place with start positions, which LC_X_CLAMP_REFERENCE_OUTER refuses).  The
    reference's clamp is decided per member over its whole series: member-major, each member == its own lc_advect."""
    from lagrangiancoherence_amd import sharded
    nsteps = sharded.ENSEMBLE_CHUNK + 5
    u, v, lat, lon = _rand_field(23, nt=nsteps + 3, ny=19, nx=27, scale=3.0)
    u[:] = u * 4 + 25.0                          # a zonal flow that pushes parcels out of the regional box
    f = eng.prepare_field(u, v, lat, lon, 1)
    mine, sig, xd, yd = sharded.ensemble_lcs(eng, f, lat, lon, 3600.0, n_members=3, nsteps=nsteps, SETTLS_order=1,
                                             interp_order=1, cyclic_xboundary=False, return_dpts=True)
    assert mine == [0, 1, 2]
    for e in mine:
        r = eng.lcs(f, lat, lon, 3600.0, SETTLS_order=1, interp_order=1, cyclic_xboundary=False, t0=e, nsteps=nsteps)
        assert eng.last_advect_kernel() == "outer_substep_kernel"       # parcels did leave: the reference's rule ran
        assert bool((xd[e] == r["x_dep"]).all()) and bool((yd[e] == r["y_dep"]).all()) and bool((sig[e] == r["sigma"]).all())
        xr_, yr_ = O.parcel_propagation(u, v, lat, lon, timestep=3600.0, SETTLS_order=1, interp_order=1,
                                        cyclic_xboundary=False, noncyclic_clamp="reference_outer", t0=e, nsteps=nsteps)
        np.testing.assert_allclose(_np(xd[e]), xr_, rtol=0, atol=POS_ATOL64)
        np.testing.assert_allclose(_np(yd[e]), yr_, rtol=0, atol=POS_ATOL64)


@pytest.mark.parametrize("order", [1, 3])
def test_float32_wind_on_float64_coordinates_follows_numpy_promotion(eng, O, order):
    """The reference with float32 winds and float64 lat/lon: samples come back float32 (Q10), latitude
    increments are formed in float32, longitude increments in float64.  lc_advect's LC_F64_WIND_F32 mode
    must reproduce that to float64 rounding -- plain float64 arithmetic on the same values does not."""
    u, v, lat, lon = _rand_field(81, nt=6, ny=27, nx=40)
    u32, v32 = u.astype(np.float32), v.astype(np.float32)
    f = eng.prepare_field(u32, v32, lat, lon, order)
    assert f.dtype == np.float64 and f.wind_f32 and f.ext is None
    x, y = eng.advect(f, lat, lon, -3600.0, SETTLS_order=3, interp_order=order)
    # the wind stays float32 (LC_F64_WIND_F32_LIN32): per-wave LDS tiles of both levels -- of the float32 image at order 1, of the
    # float64 coefficients packed straight from the float32 planes at order 3
    assert eng.last_advect_kernel() == ("advect_lds64w_kernel<-1, true>" if order == 1 else "advect_lds64w_o3_kernel<-1, true>")
    assert f.u is None and f.u32 is not None and (f.lin32 is not None) == (order == 1) and (f.cub is not None) == (order == 3)
    xr_, yr_ = O.parcel_propagation(u32, v32, lat, lon, timestep=-3600.0, SETTLS_order=3, interp_order=order,
                                    cyclic_xboundary=True)
    assert xr_.dtype == np.float64
    np.testing.assert_allclose(_np(x), xr_, rtol=0, atol=POS_ATOL64)
    np.testing.assert_allclose(_np(y), yr_, rtol=0, atol=POS_ATOL64)
    # the same float32-valued wind pushed through pure float64 arithmetic is a different (1e-7-ish) answer
    f64 = eng.prepare_field(u32.astype(np.float64), v32.astype(np.float64), lat, lon, order)
    x64, y64 = eng.advect(f64, lat, lon, -3600.0, SETTLS_order=3, interp_order=order)
    assert np.abs(_np(y64) - yr_).max() > 100 * POS_ATOL64
    # the float32 planes are BORROWED when they were device tensors (pole rows and Euler samples read them live next to
    # images packed from their old values): an in-place write afterwards is refused on these paths too (round-5 advisor)
    ud, vd = eng.to_device(u32, np.float32), eng.to_device(v32, np.float32)
    fb = eng.prepare_field(ud, vd, lat, lon, order)
    assert fb.u32 is ud and fb.planes32_version is not None
    eng.advect(fb, lat, lon, -3600.0, SETTLS_order=3, interp_order=order)
    vd.mul_(2.0)
    with pytest.raises(RuntimeError, match="modified in place"):
        eng.advect(fb, lat, lon, -3600.0, SETTLS_order=3, interp_order=order)


@pytest.mark.parametrize("K,cyclic", [(4, True), (3, True), (4, False), (0, True)])
def test_float32_wind_kept_float32_equals_the_float64_images_bit_for_bit(eng, O, K, cyclic):
    """LC_F64_WIND_F32_LIN32 (the default for float32 winds on float64 coordinates at order 1): the order-1 image stays
    float32 and a node is widened as it is read -- the same doubles into the same sums as LC_F64_WIND_F32 on float64 images
    of the wind (numpy's promotion: LCS/trajectory.py:86-87,110-112), so departure points, trajectories, row
    blocks with a continuation and pole rows are bit-identical: LDS-tile kernel, direct kernel (SETTLS_order 0 or
    lc_ctx_set_lds_tiles(0)), sparse and dense seed grids.  The reference's outer-product clamp, another interpolation
    order and sample() fall back to float64 planes made on demand."""
    u, v, lat, lon = flows.era5_like(nt=9, ny=72, nx=144)
    u, v, lat, lon = u * np.float32(2.0), v, lat.astype(np.float64), lon.astype(np.float64)
    f_new = eng.prepare_field(u, v, lat, lon, 1)
    f_old = eng.prepare_field(u, v, lat, lon, 1, lin_image=False)             # float64 planes as the order-1 source (round 4)
    assert f_new.wind_f32 and f_new.lin32 is not None and f_new.u is None and f_old.lin32 is None and f_old.u is not None
    for sny, snx in ((150, 200), (40, 60), (300, 512), (72, 144)):
        slat, slon = flows.seed_grid(sny, snx, lat, lon)
        try:
            for mode in (-1, 0):
                eng.set_lds_tiles(mode)
                out = []
                for f in (f_old, f_new):
                    r = eng.advect(f, slat, slon, -1800.0, SETTLS_order=K, interp_order=1, cyclic_xboundary=cyclic,
                                   noncyclic_clamp="pointwise", return_traj=True)
                    name = eng.last_advect_kernel()
                    lo, hi = 0, sny // 2
                    rb = eng.advect(f, slat[lo:hi], slon, -1800.0, SETTLS_order=K, interp_order=1, cyclic_xboundary=cyclic,
                                    noncyclic_clamp="pointwise", row0=lo, ny_global=sny, t0=3, nsteps=5, start=(r[2][3][lo:hi], r[3][3][lo:hi]))
                    out.append(([_np(t) for t in r] + [_np(t) for t in rb], name))
                (a, na), (b, nb) = out
                assert na == "advect_kernel<double, 1, false, 1>", na
                assert nb == ("advect_lds64w_kernel<%s, %s>" % (4 if K == 4 else -1, "true" if cyclic else "false") if (mode == -1 and K > 0) else "advect_w32_kernel"), nb
                for p_, q_ in zip(a, b):
                    assert p_.dtype == np.float64 and np.array_equal(p_, q_), (sny, snx, mode, na, nb)
        finally:
            eng.set_lds_tiles(-1)
    # what the float32-kept form does not serve takes float64 planes made on demand: the reference's outer-product clamp ...
    rng = np.random.default_rng(3)
    la, lo_ = np.linspace(-40, 40, 41), np.linspace(-60, 50, 56)
    uu, vv = (30 + 25 * rng.standard_normal((5, 41, 56))).astype(np.float32), (8 * rng.standard_normal((5, 41, 56))).astype(np.float32)
    g = eng.prepare_field(uu, vv, la, lo_, 1)
    xa, ya = eng.advect(g, la, lo_, 7200.0, 2, 1, False)
    assert eng.last_advect_kernel() == "outer_substep_kernel" and g.u is not None
    xo, yo = O.parcel_propagation(uu, vv, la, lo_, timestep=7200.0, SETTLS_order=2, interp_order=1, cyclic_xboundary=False,
                                  noncyclic_clamp="reference_outer")
    assert np.abs(_np(xa) - xo).max() < POS_ATOL64 and np.abs(_np(ya) - yo).max() < POS_ATOL64
    # ... and sample()
    px, py = np.meshgrid(lon[::7], lat[::5])
    sa, sb = eng.sample(f_new, px, py, level=2), eng.sample(f_old, px, py, level=2)
    assert all(np.array_equal(_np(p_), _np(q_)) for p_, q_ in zip(sa, sb))
    # the C ABI refuses what the dtype does not cover
    a = eng._advect_args(f_new, 1, eng.to_device(lat, np.float64), lat.size, eng.to_device(lon, np.float64), lon.size, 0, lat.size, None, None,
                         -1800.0, 2, 1, 0, 8, 1, 0, *(eng._empty((lat.size, lon.size), np.float64) for _ in range(2)), None, None)
    a.interp_order = 3
    from lagrangiancoherence_amd import _capi
    import ctypes
    with pytest.raises(ValueError, match="LC_F64_WIND_F32_LIN32"):
        _capi.check(eng.lib.lc_advect_ex(eng.ctx, ctypes.byref(a)), eng.lib)


@pytest.mark.parametrize("K,cyclic", [(4, True), (2, True), (4, False), (0, True)])
def test_float32_wind_kept_float32_at_order3_equals_the_float64_planes_form_bit_for_bit(eng, K, cyclic):
    """LC_F64_WIND_F32_LIN32 at interp_order 3 (the reference's default order on float32 reanalysis winds): the float64 spline
    coefficients packed straight from the float32 planes (lc_field_pack(LC_F64_WIND_F32)) equal those of a float64 copy of
    the wind, and the per-wave-tile kernel -- scipy's weights and tap order, numpy's index map, two samples per iteration
    rounded to float32 -- equals the generic kernel on float64 planes bit for bit: departure points, trajectories, a row
    block with a continuation, pole rows (order 1 from the float32 planes), LDS and direct kernels, sparse and dense seeds."""
    u, v, lat, lon = flows.era5_like(nt=9, ny=72, nx=144)
    u, v, lat, lon = u * np.float32(2.0), v, lat.astype(np.float64), lon.astype(np.float64)
    f_new = eng.prepare_field(u, v, lat, lon, 3)
    f_old = eng.prepare_field(u, v, lat, lon, 3, lin_image=False)            # float64 planes (round 4)
    assert f_new.u32 is not None and f_new.u is None and f_old.u is not None and f_old.u32 is None
    assert np.array_equal(_np(f_new.cub), _np(f_old.cub))
    for sny, snx in ((150, 200), (40, 60), (300, 512), (72, 144)):
        slat, slon = flows.seed_grid(sny, snx, lat, lon)
        try:
            for mode in (-1, 0):
                eng.set_lds_tiles(mode)
                out = []
                for f in (f_old, f_new):
                    r = eng.advect(f, slat, slon, -1800.0, SETTLS_order=K, interp_order=3, cyclic_xboundary=cyclic,
                                   noncyclic_clamp="pointwise", return_traj=True)
                    name = eng.last_advect_kernel()
                    lo, hi = 0, sny // 2
                    rb = eng.advect(f, slat[lo:hi], slon, -1800.0, SETTLS_order=K, interp_order=3, cyclic_xboundary=cyclic,
                                    noncyclic_clamp="pointwise", row0=lo, ny_global=sny, t0=3, nsteps=5, start=(r[2][3][lo:hi], r[3][3][lo:hi]))
                    out.append(([_np(t) for t in r] + [_np(t) for t in rb], name))
                (a, na), (b, nb) = out
                assert na == "advect_kernel<double, 3, false, 0>", na
                assert nb == ("advect_lds64w_o3_kernel<%s, %s>" % (4 if K == 4 else -1, "true" if cyclic else "false") if mode == -1 else "advect_kernel<double, 3, false, 0>"), nb
                for p_, q_ in zip(a, b):
                    assert p_.dtype == np.float64 and np.array_equal(p_, q_), (sny, snx, mode, na, nb)
        finally:
            eng.set_lds_tiles(-1)
    # order 1 on such a field ("order 1 is always available") takes float64 planes made on demand
    slat, slon = flows.seed_grid(60, 90, lat, lon)
    a = eng.advect(f_new, slat, slon, -1800.0, K, 1, cyclic, noncyclic_clamp="pointwise")
    b = eng.advect(f_old, slat, slon, -1800.0, K, 1, cyclic, noncyclic_clamp="pointwise")
    assert f_new.u is not None and all(np.array_equal(_np(p_), _np(q_)) for p_, q_ in zip(a, b))


@pytest.mark.parametrize("order", [1, 3])
def test_float_path_options_and_tiny_field(eng, O, order):
    """float32 kernels beyond the headline settings: a field smaller than an LDS tile (direct-gather kernel is
    launched instead), non-cyclic clamp, trajectories, K=0 and K=1, forward time."""
    # (a) tiny field: 9 x 11 nodes < 16 x 8 / 32 x 16 tile
    u, v, lat, lon = _rand_field(91, nt=5, ny=9, nx=11, dtype=np.float32, scale=8.0)
    f = eng.prepare_field(u, v, lat, lon, order)
    for K in (0, 1, 3):
        x, y = eng.advect(f, lat, lon, 1800.0, SETTLS_order=K, interp_order=order)
        x64, y64 = O.parcel_propagation(u.astype(np.float64), v.astype(np.float64), lat.astype(np.float64),
                                        lon.astype(np.float64), timestep=1800.0, SETTLS_order=K, interp_order=order,
                                        cyclic_xboundary=True)
        dx = np.abs(_np(x) - x64)
        assert np.minimum(dx, np.abs(dx - 360)).max() < 2e-4 and np.abs(_np(y) - y64).max() < 2e-4
    # (b) larger field through the LDS kernel: non-cyclic clamp + trajectories
    u, v, lat, lon = _rand_field(92, nt=6, ny=40, nx=64, dtype=np.float32, scale=30.0)
    f = eng.prepare_field(u, v, lat, lon, order)
    x, y, tx, ty = eng.advect(f, lat, lon, -3600.0, SETTLS_order=2, interp_order=order, cyclic_xboundary=False,
                              return_traj=True, noncyclic_clamp="pointwise")   # the fused kernel's own clamp
    tx64, ty64 = O.parcel_propagation(u.astype(np.float64), v.astype(np.float64), lat.astype(np.float64),
                                      lon.astype(np.float64), timestep=-3600.0, SETTLS_order=2, interp_order=order,
                                      cyclic_xboundary=False, return_traj=True, noncyclic_clamp="pointwise")
    ex, ey = np.abs(_np(tx) - tx64), np.abs(_np(ty) - ty64)
    # random (spatially uncorrelated) wind: neighbouring nodes differ by tens of m/s, so float32 position
    # rounding is amplified quickly; the bulk must still agree closely and nothing may run away
    assert np.percentile(ex, 99) < 5e-3 and np.percentile(ey, 99) < 5e-3 and ex.max() < 5.0 and ey.max() < 5.0
    assert np.array_equal(_np(tx[-1]), _np(x)) and _np(x).min() >= lon.min() and _np(x).max() <= lon.max()
    assert np.array_equal(_np(tx[0]), np.meshgrid(lon, lat)[0])


def test_lcs_host_float32_route(eng):
    """The torch-free host entry point with float32 arrays takes the same kernels (fused levels, LDS tiles)
    as the engine route: identical results."""
    from lagrangiancoherence_amd.engine import lcs_host
    u, v, lat, lon = flows.era5_like(nt=7, ny=72, nx=144)
    slat, slon = flows.seed_grid(90, 130, lat, lon)
    for order in (1, 3):
        out = lcs_host(u, v, lat, lon, -900.0, SETTLS_order=4, interp_order=order, cyclic_xboundary=True,
                       seed_lat=slat, seed_lon=slon)
        f = eng.prepare_field(u, v, lat, lon, order)
        r = eng.lcs(f, slat, slon, -900.0, SETTLS_order=4, interp_order=order)
        assert out["x_dep"].dtype == np.float32
        assert np.array_equal(out["x_dep"], _np(r["x_dep"])) and np.array_equal(out["y_dep"], _np(r["y_dep"]))
        assert np.array_equal(out["sigma"], _np(r["sigma"]))


@pytest.mark.parametrize("dtype,order", [(np.float32, 1), (np.float32, 3), (np.float64, 1), (np.float64, 3)])
def test_lcs_host_pipelined_route_is_bit_identical_to_the_serial_route(eng, dtype, order):
    """lc_ctx_set_host_pipeline (the default of the one-call host route since round 6): the wind travels through the pinned
    staging ring in level chunks and the pack + advect kernels of chunk c run while chunk c + 1 is on the bus (44 levels here:
    three chunks); a sub-range of the series moves only its own levels.  Bit-identical to the serial route (plain copies of the
    whole series, one pack, one advect) and to the engine's, departure points and sigma; a series too short to pipeline, the
    per-point clamp and trajectories (serial forms over the staged copies) too."""
    from lagrangiancoherence_amd.engine import lcs_host
    u, v, lat, lon = flows.era5_like(nt=44, ny=72, nx=144)
    slat, slon = flows.seed_grid(90, 130, lat, lon)
    u, v, lat, lon, slat, slon = (a.astype(dtype) for a in (u, v, lat, lon, slat, slon))
    kw = dict(SETTLS_order=4, interp_order=order, cyclic_xboundary=True, seed_lat=slat, seed_lon=slon, float64_fidelity="fast")
    a = lcs_host(u, v, lat, lon, -900.0, **kw)
    b = lcs_host(u, v, lat, lon, -900.0, pipeline=False, **kw)
    f = eng.prepare_field(u, v, lat, lon, order)
    r = eng.lcs(f, slat, slon, -900.0, SETTLS_order=4, interp_order=order)
    for k in ("x_dep", "y_dep", "sigma"):
        assert a[k].dtype == dtype and np.array_equal(a[k], b[k]) and np.array_equal(a[k], _np(r[k])), k
    assert np.isfinite(a["sigma"]).all()
    # levels 5 .. 40 of the series (35 steps: pipelined, two chunks and a short one) and 7 .. 12 (serial, those levels only)
    for t0, n in ((5, 35), (7, 5)):
        a = lcs_host(u, v, lat, lon, -900.0, t0=t0, nsteps=n, **kw)
        b = lcs_host(u, v, lat, lon, -900.0, t0=t0, nsteps=n, pipeline=False, **kw)
        r = eng.lcs(f, slat, slon, -900.0, SETTLS_order=4, interp_order=order, t0=t0, nsteps=n)
        for k in ("x_dep", "y_dep", "sigma"):
            assert np.array_equal(a[k], b[k]) and np.array_equal(a[k], _np(r[k])), (t0, n, k)
    # trajectories and the per-point clamp: serial kernels, staged copies
    kw2 = dict(kw, cyclic_xboundary=False, noncyclic_clamp="pointwise", return_traj=True)
    a, b = lcs_host(u, v, lat, lon, -900.0, **kw2), lcs_host(u, v, lat, lon, -900.0, pipeline=False, **kw2)
    for k in ("x_dep", "sigma", "traj_x", "traj_y"):
        assert np.array_equal(a[k], b[k], equal_nan=True), k


def test_large_host_arrays_travel_through_the_staging_ring(eng):
    """Engine.to_device / to_host move arrays of 8 MB or more through the context's pinned staging ring (lc_copy_to_device /
    lc_copy_to_host: ordered with the current stream, the source reusable at once) -- the same bytes as torch's own copies, for
    sizes across the ring's 32 MB pieces and longer than the ring, in both dtypes, with the ring switched off too."""
    import torch
    rng = np.random.default_rng(5)
    for n, dtype in ((3 * 1024 * 1024 + 17, np.float32), (40 * 1024 * 1024 // 8 + 3, np.float64), (150 * 1024 * 1024 // 4 + 1, np.float32)):
        a = rng.standard_normal(n).astype(dtype)
        for on in (1, 0):
            _capi_check = eng.lib.lc_ctx_set_host_pipeline(eng.ctx, on)
            assert _capi_check == 0
            t = eng.to_device(a, dtype)
            a_keep = a.copy()
            a[:] = 0                                       # the source may be reused as soon as to_device returns
            assert torch.equal(t, torch.from_numpy(a_keep).to(t.device))
            t2 = t * 2                                     # work enqueued on the stream after the copy sees the data
            back = eng.to_host(t2)
            assert back.dtype == dtype and np.array_equal(back, a_keep * 2)
            a[:] = a_keep
    eng.lib.lc_ctx_set_host_pipeline(eng.ctx, 1)
    # a transposed view of a contiguous array (the example's (latitude, longitude, time) winds asked for as (time, latitude,
    # longitude)): the buffer travels as it lies, the permutation runs on the device -- same values, contiguous result
    base = rng.standard_normal((90, 180, 200)).astype(np.float32)                       # (latitude, longitude, time), 13 MB
    for view in (base.transpose(2, 0, 1), base.transpose(2, 1, 0), base.transpose(1, 0, 2), base[:, :, ::2].transpose(2, 0, 1)):
        t = eng.to_device(view, np.float32)
        assert t.is_contiguous() and tuple(t.shape) == view.shape and np.array_equal(t.cpu().numpy(), view)
    t64 = eng.to_device(base.transpose(2, 0, 1), np.float64)                             # a dtype change: the ordinary path
    assert t64.dtype == torch.float64 and np.array_equal(t64.cpu().numpy(), base.transpose(2, 0, 1).astype(np.float64))


@pytest.mark.parametrize("order", [1, 3])
def test_float64_fused_levels_option(eng, O, order):
    """float64 default (fuse_levels=True): one sample of 2F[t]-F[t+1] per SETTLS iteration, index map by multiplication,
    fused lerps -- results move by rounding only.  fuse_levels=False keeps numpy / scipy's operation order (two samples,
    true divisions, scipy's tap sum) and lands another three orders closer to the oracle."""
    u, v, lat, lon = flows.config2(n=96, nt=13)
    f_exact = eng.prepare_field(u, v, lat, lon, order, fuse_levels=False)
    f_fused = eng.prepare_field(u, v, lat, lon, order)
    assert f_exact.ext is None and not f_exact.fuse_raw and f_fused.ext is not None and not f_fused.fuse_raw
    xe, ye = eng.advect(f_exact, lat, lon, -900.0, SETTLS_order=4, interp_order=order)
    assert eng.last_advect_kernel() == {1: "advect_kernel<double, 1, false, 1>", 3: "advect_kernel<double, 3, false, 0>"}[order]
    xf, yf = eng.advect(f_fused, lat, lon, -900.0, SETTLS_order=4, interp_order=order)
    assert eng.last_advect_kernel() == ("advect_lds64_kernel<4, true, 1>" if order == 1 else "advect_lds64_o3_kernel<4, true>")
    xo, yo = O.parcel_propagation(u, v, lat, lon, timestep=-900.0, SETTLS_order=4, interp_order=order,
                                  cyclic_xboundary=True)
    for got, ref in ((xf, xo), (yf, yo)):
        d = np.abs(_np(got) - ref)
        assert np.minimum(d, np.abs(d - 360)).max() < 1e-10            # rounding-level (fp64 tolerance, degrees)
    assert np.abs(_np(xe) - xo).max() < 1e-12 and np.abs(_np(ye) - yo).max() < 1e-12


@pytest.mark.parametrize("order", [1, 3])
@pytest.mark.parametrize("K", [0, 4])
def test_float64_fast_form_on_the_last_node_row_decides_the_wrap_as_numpy_does(eng, O, order, K):
    """scipy's 'wrap' map is discontinuous at the last node (tools.py:21-22 scales by n / span, so c = n - 1 is an ordinary
    coordinate and anything above lands next to node 0).  A seed row that sits on the last node row of a coarser field has
    c = n - 1 to rounding: the float64 fast form (index map by multiplication) and numpy's (n (x - min)) / span fell on
    different sides there (found by the randomised run of tests/test_gpu_hypothesis.py: 0.34 degrees on a whole seed row).
    Within 1e-12 of n - 1 the fast form now takes numpy's expression: same side, same sample."""
    rng = np.random.default_rng(6739)
    lat = np.linspace(-60.0, 60.0, 18)
    lon = -180 + 360.0 / 23 * np.arange(23)
    u = 5.0 * rng.standard_normal((3, 18, 23))
    v = 2.5 * rng.standard_normal((3, 18, 23))
    slat = np.linspace(lat[0], lat[-1], 19)            # row 17: (53.33.. + 60) * 18 / 120 = 17 = n - 1 to rounding
    slon = np.linspace(lon[0], lon[-1], 24)            # column 22: c = 22 = n - 1 to rounding
    cy = 18 * (slat - lat[0]) / (lat[-1] - lat[0])
    cx = 23 * (slon - lon[0]) / (lon[-1] - lon[0])
    assert np.abs(cy - 17).min() < 1e-13 and np.abs(cx - 22).min() < 1e-13
    f = eng.prepare_field(u, v, lat, lon, order)
    for mode in (-1, 0):
        try:
            eng.set_lds_tiles(mode)
            x, y = eng.advect(f, slat, slon, -5400.0, SETTLS_order=K, interp_order=order)
        finally:
            eng.set_lds_tiles(-1)
        xo, yo = O.parcel_propagation(u, v, lat, lon, timestep=-5400.0, SETTLS_order=K, interp_order=order, cyclic_xboundary=True,
                                      seed_lat=slat, seed_lon=slon)
        d = np.abs(_np(x) - xo)
        assert np.minimum(d, np.abs(d - 360)).max() < POS_ATOL64 and np.abs(_np(y) - yo).max() < POS_ATOL64


@pytest.mark.parametrize("dtype", [np.float32, np.float64])
def test_non_finite_and_huge_wind_values_stay_contained(eng, dtype):
    """NaN, +-inf and 1e30 m/s at a few wind nodes: every kernel variant finishes (the rare-case branches
    of the fast float kernel see NaN/garbage coordinates), latitudes stay inside the clamp bounds (Q8: a NaN
    latitude becomes y_min), and at order 1 only parcels that touched a poisoned cell are affected."""
    u, v, lat, lon = flows.era5_like(nt=9, ny=90, nx=180)
    u, v, lat, lon = (a.astype(dtype) for a in (u, v, lat, lon))
    u[3, 40, 60] = np.nan
    v[5, 20, 100] = np.inf
    u[2, 70, 10] = -np.inf
    v[1, 5, 5] = 1e30
    u[6, 80, 170] = -1e30
    slat, slon = flows.seed_grid(300, 400, lat, lon)
    for order in (1, 3):
        f = eng.prepare_field(u, v, lat, lon, order)
        for cyclic in (True, False):
            x, y = eng.advect(f, slat, slon, -1800.0, SETTLS_order=4, interp_order=order, cyclic_xboundary=cyclic)
            xn, yn = _np(x), _np(y)
            assert np.nanmin(yn) >= lat.min() and np.nanmax(yn) <= lat.max()
            assert not np.isnan(yn).any()                                      # Q8
            if order == 1:
                assert (~np.isfinite(xn)).mean() < 0.01
            sig = _np(eng.sigma(x, y, slat, slat[1] - slat[0], slon[1] - slon[0]))
            assert sig.shape == xn.shape


# ------------------------------------------------------------------ continuation, level chunks, wide patches
@pytest.mark.parametrize("dtype", [np.float32, np.float64])
@pytest.mark.parametrize("order,lds", [(1, -1), (1, 1), (1, 2), (1, 0), (3, -1)])
def test_advect_from_continues_bit_for_bit(eng, dtype, order, lds):
    """lc_advect_from: levels [t0, t0+a) then [t0+a, t0+a+b) from the first call's positions == one call over a+b
    levels, bit for bit, in every kernel family (two-seed / one-seed LDS tiles, direct gathers, float64), with
    trajectories, in place, and on a row block with pole rows (LCS/trajectory.py:80-126 carries only positions)."""
    u, v, lat, lon = flows.era5_like(nt=10, ny=60, nx=120)
    u, v, lat, lon = (a.astype(dtype) for a in (u * 1.5, v, lat, lon))
    slat, slon = (a.astype(dtype) for a in flows.seed_grid(141, 202, lat, lon))
    f = eng.prepare_field(u, v, lat, lon, order)
    kw = dict(SETTLS_order=4, interp_order=order, cyclic_xboundary=True)
    try:
        eng.set_lds_tiles(lds)
        x, y, tx, ty = eng.advect(f, slat, slon, -1800.0, t0=1, nsteps=8, return_traj=True, **kw)
        xa, ya, txa, tya = eng.advect(f, slat, slon, -1800.0, t0=1, nsteps=3, return_traj=True, **kw)
        xb, yb, txb, tyb = eng.advect(f, slat, slon, -1800.0, t0=4, nsteps=5, return_traj=True, start=(xa, ya), **kw)
        assert bool((xb == x).all()) and bool((yb == y).all())
        assert bool((txa == tx[:4]).all()) and bool((txb == tx[3:]).all()) and bool((tyb == ty[3:]).all())
        # a row block (global row offset: the lower pole rows are in it), without trajectories
        lo, hi = 0, 77
        xr, yr = eng.advect(f, slat[lo:hi], slon, -1800.0, t0=4, nsteps=5, row0=lo, ny_global=141,
                            start=(xa[lo:hi], ya[lo:hi]), **kw)
        assert bool((xr == x[lo:hi]).all()) and bool((yr == y[lo:hi]).all())
    finally:
        eng.set_lds_tiles(-1)


@pytest.mark.parametrize("order,lds,traj", [(1, 1, False), (1, 1, True), (1, 2, True), (3, -1, False), (1, 0, True)])
def test_level_chunks_do_not_change_results(eng, order, lds, traj):
    """lc_ctx_set_level_chunk: the series as consecutive launches of 1 / 3 / 4 levels == one launch, bit for bit."""
    u, v, lat, lon = flows.era5_like(nt=12, ny=60, nx=120)
    slat, slon = flows.seed_grid(130, 210, lat, lon)
    f = eng.prepare_field(u, v, lat, lon, order)
    kw = dict(SETTLS_order=4, interp_order=order, cyclic_xboundary=True, return_traj=traj)
    try:
        eng.set_lds_tiles(lds)
        eng.set_level_chunk(0)
        ref = eng.advect(f, slat, slon, -1800.0, **kw)
        for chunk in (1, 3, 4, 64):
            eng.set_level_chunk(chunk)
            got = eng.advect(f, slat, slon, -1800.0, **kw)
            for a, b in zip(ref, got):
                assert bool((a == b).all()), (chunk, tuple(a.shape))
    finally:
        eng.set_level_chunk(-1)
        eng.set_lds_tiles(-1)
    with pytest.raises(ValueError):
        eng.set_level_chunk(-2)


@pytest.mark.parametrize("order", [1, 3])
@pytest.mark.parametrize("sny,snx", [(130, 212), (131, 211), (64, 32), (40, 100), (23, 18), (200, 512)])
def test_patch_modes_of_the_two_seed_kernel_agree_bitwise(sny, snx, order, monkeypatch):
    """Which seeds a wave of the two-seed kernel holds and how trajectories are stored (LCS_PATCH_MODE: 0 tall patches,
    per-lane stores; 1 wide patches, paired stores; 2 whole-line stores through an LDS slab and one workgroup barrier per
    level, the default with trajectories) only decides which lane holds which seed: positions and trajectories are
    bit-identical -- widths that are no multiple of 4 or 32 (per-lane fallback), grid edges, pole rows and level chunks
    included."""
    from lagrangiancoherence_amd.engine import Engine
    u, v, lat, lon = flows.era5_like(nt=7, ny=60, nx=120)
    slat, slon = flows.seed_grid(sny, snx, lat, lon)
    out = {}
    for flag in ("0", "1", "2", ""):
        if flag:
            monkeypatch.setenv("LCS_PATCH_MODE", flag)
        else:
            monkeypatch.delenv("LCS_PATCH_MODE")
        e = Engine(0)
        try:
            e.set_lds_tiles(1)
            f = e.prepare_field(u, v, lat, lon, order)
            out[flag] = [_np(t) for t in e.advect(f, slat, slon, -1800.0, SETTLS_order=4, interp_order=order, return_traj=True)]
            want = flag or ("2" if snx % 4 == 0 and snx >= 32 else "0")      # the default: lines where rows are 16-byte aligned
            assert e.last_advect_kernel() == ("advect_lds2_kernel<4, true, %s>" if order == 1 else "advect_lds2_o3_kernel<4, true, %s>") % want
            out[flag + "n"] = [_np(t) for t in e.advect(f, slat, slon, -1800.0, SETTLS_order=4, interp_order=order)]
            e.set_level_chunk(4)
            out[flag + "c"] = [_np(t) for t in e.advect(f, slat, slon, -1800.0, SETTLS_order=2, interp_order=order, return_traj=True)]
            if flag == "0":     # the one-seed-per-lane kernel of the same order: the same bits
                e.set_level_chunk(0)
                e.set_lds_tiles(2)
                one = [_np(t) for t in e.advect(f, slat, slon, -1800.0, SETTLS_order=4, interp_order=order, return_traj=True)]
                assert "advect_lds_kernel" in e.last_advect_kernel()
                for a, b in zip(out["0"], one):
                    assert np.array_equal(a, b)
        finally:
            e.close()
    for flag in ("1", "2", ""):
        for suffix in ("", "n", "c"):
            for a, b in zip(out["0" + suffix], out[flag + suffix]):
                assert np.array_equal(a, b), (flag, suffix)
    assert np.array_equal(out["0"][0], out["0n"][0])


@pytest.mark.parametrize("order,lds", [(1, 1), (1, 2), (1, 0), (3, 1), (3, 2)])
def test_advect_batch_equals_member_by_member_in_every_float_kernel(eng, order, lds):
    """lc_advect_batch (member = blockIdx.y, one launch per level chunk over all members) against one lc_advect per
    member: bit for bit, two-seed / one-seed LDS-tile kernels and direct gathers, orders 1 and 3, with level chunks."""
    u, v, lat, lon = flows.era5_like(nt=12, ny=60, nx=120)
    slat, slon = flows.seed_grid(101, 144, lat, lon)
    f = eng.prepare_field(u, v, lat, lon, order)
    try:
        eng.set_lds_tiles(lds)
        eng.set_level_chunk(3)
        xb, yb = eng.advect_batch(f, slat, slon, -1800.0, 4, 7, SETTLS_order=4, interp_order=order, t0=1)
        eng.set_level_chunk(0)
        for m in range(4):
            x, y = eng.advect(f, slat, slon, -1800.0, SETTLS_order=4, interp_order=order, t0=1 + m, nsteps=7)
            assert bool((xb[m] == x).all()) and bool((yb[m] == y).all()), m
    finally:
        eng.set_level_chunk(-1)
        eng.set_lds_tiles(-1)


@pytest.mark.parametrize("members,stride,nsteps,chunk,K", [(4, 1, 7, 3, 4), (5, 1, 6, 0, 4), (3, 2, 5, 2, 2), (2, 0, 4, 0, 1),
                                                         (7, 3, 5, 4, 4), (2, 1, 2, 1, 4), (6, 1, 9, -1, 3), (9, 2, 8, 5, 4),
                                                         (4, 0, 3, 2, 4)])
def test_member_pairs_in_the_two_seed_kernel_equal_member_by_member(eng, members, stride, nsteps, chunk, K):
    """lc_advect_batch through the two-seed order-1 kernel holds two consecutive MEMBERS per lane (PATCH_PAIR: launches
    walk the pair's level window, the second member t0_stride levels behind the first): bit for bit one lc_advect per
    member -- odd member counts (a pair without a second member), strides 0..3, level chunks that cut the window before /
    at / after the second member's start, pole rows in leading workgroups and inside the tiles, a continuation in
    place; and the tall-patch form of the same kernel (LCS_PATCH_MODE=0) for the same call."""
    from lagrangiancoherence_amd.engine import Engine
    u, v, lat, lon = flows.era5_like(nt=30, ny=60, nx=120)
    slat, slon = flows.seed_grid(101, 144, lat, lon)
    slat = slat.copy(); slat[0], slat[-1] = -90.0, 90.0          # global pole rows (generic per-seed path, Q3)
    f = eng.prepare_field(u, v, lat, lon, 1)
    ref = []
    eng.set_level_chunk(0)
    for m in range(members):
        ref.append(eng.advect(f, slat, slon, -1800.0, SETTLS_order=K, interp_order=1, t0=2 + m * stride, nsteps=nsteps))
    try:
        eng.set_lds_tiles(1)
        eng.set_level_chunk(chunk)
        xb, yb = eng.advect_batch(f, slat, slon, -1800.0, members, nsteps, SETTLS_order=K, interp_order=1, t0=2, t0_stride=stride)
        want = ", 3>" if nsteps > stride else ", 0>"
        assert eng.last_advect_kernel().endswith(want), eng.last_advect_kernel()
        for m in range(members):
            assert bool((xb[m] == ref[m][0]).all()) and bool((yb[m] == ref[m][1]).all()), m
        # continuation in place: the first steps, then the rest from where they stopped
        n1 = max(1, nsteps // 2)
        xa, ya = eng.advect_batch(f, slat, slon, -1800.0, members, n1, SETTLS_order=K, interp_order=1, t0=2, t0_stride=stride)
        if nsteps > n1:
            eng.advect_batch(f, slat, slon, -1800.0, members, nsteps - n1, SETTLS_order=K, interp_order=1, t0=2 + n1,
                             t0_stride=stride, start=(xa, ya), out=(xa, ya))
        for m in range(members):
            assert bool((xa[m] == ref[m][0]).all()) and bool((ya[m] == ref[m][1]).all()), ("continued", m)
    finally:
        eng.set_level_chunk(-1)
        eng.set_lds_tiles(-1)
    # pole rows inside the tiles (LCS_POLE_BLOCKS=0) and the tall patches (LCS_PATCH_MODE=0): other contexts
    import os
    for env, val, suffix in (("LCS_POLE_BLOCKS", "0", want), ("LCS_PATCH_MODE", "0", ", 0>"), ("LCS_PATCH_MODE", "3", ", 3>")):
        os.environ[env] = val
        try:
            e2 = Engine(0)
        finally:
            del os.environ[env]
        f2_ = e2.prepare_field(u, v, lat, lon, 1)
        e2.set_lds_tiles(1)
        e2.set_level_chunk(chunk)
        xb, yb = e2.advect_batch(f2_, slat, slon, -1800.0, members, nsteps, SETTLS_order=K, interp_order=1, t0=2, t0_stride=stride)
        if nsteps > stride:
            assert e2.last_advect_kernel().endswith(suffix), e2.last_advect_kernel()
        for m in range(members):
            assert bool((xb[m] == ref[m][0]).all()) and bool((yb[m] == ref[m][1]).all()), (env, val, m)


@pytest.mark.parametrize("order", [1, 3])
@pytest.mark.parametrize("K,cyclic", [(4, True), (2, True), (4, False), (1, False), (0, True)])
def test_float64_lds_tile_and_direct_kernels_agree_bitwise(eng, O, K, cyclic, order):
    """float64, fused levels: the per-wave LDS-tile kernels (default; order 3 also at K = 0) and the direct-gather kernels
    (lc_ctx_set_lds_tiles(0)) share locate_fast64 / lerp_fast64 / cubic_taps_fast64 -- bit-identical on a flow with jets,
    a seam crossing, pole rows, a seed grid sparser and one denser than the field, trajectories, a row block and a
    continuation."""
    u, v, lat, lon = flows.era5_like(nt=9, ny=72, nx=144)
    u, v, lat, lon = (a.astype(np.float64) for a in (u * 2.0, v, lat, lon))
    f = eng.prepare_field(u, v, lat, lon, order)
    for sny, snx in ((150, 200), (40, 60), (300, 512)):
        slat, slon = (a.astype(np.float64) for a in flows.seed_grid(sny, snx, lat, lon))
        out = {}
        try:
            for mode in (-1, 0):
                eng.set_lds_tiles(mode)
                r = eng.advect(f, slat, slon, -1800.0, SETTLS_order=K, interp_order=order, cyclic_xboundary=cyclic,
                               noncyclic_clamp="pointwise", return_traj=True)
                assert ("lds64" in eng.last_advect_kernel()) == (mode == -1 and (K > 0 or order == 3))
                lo, hi = 0, sny // 2
                rb = eng.advect(f, slat[lo:hi], slon, -1800.0, SETTLS_order=K, interp_order=order, cyclic_xboundary=cyclic,
                                noncyclic_clamp="pointwise", row0=lo, ny_global=sny, t0=3, nsteps=5, start=(r[2][3][lo:hi], r[3][3][lo:hi]))
                out[mode] = [_np(t) for t in r] + [_np(t) for t in rb]
        finally:
            eng.set_lds_tiles(-1)
        for a, b in zip(out[-1], out[0]):
            assert np.array_equal(a, b), (sny, snx)
        assert np.array_equal(out[-1][4], out[-1][0][:sny // 2])       # the continued row block lands on the whole run's result
    xo, yo = O.parcel_propagation(u, v, lat, lon, timestep=-1800.0, SETTLS_order=K, interp_order=order, cyclic_xboundary=cyclic,
                                  seed_lat=slat, seed_lon=slon, noncyclic_clamp="pointwise")
    d = np.abs(out[-1][0] - xo)
    assert np.minimum(d, np.abs(d - 360)).max() < POS_ATOL64 and np.abs(out[-1][1] - yo).max() < POS_ATOL64
