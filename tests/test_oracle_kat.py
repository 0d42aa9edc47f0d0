"""Known-answer tests pinning the CPU oracle.

The reference has no tests or fixtures, so these KATs are what stands between
the oracle and "it merely agrees with itself": analytic answers, and direct
comparisons with the third-party kernels (scipy.ndimage / numpy.linalg) the
reference calls.
"""
import numpy as np
import pytest
from scipy.ndimage import map_coordinates, spline_filter

from oracle import lcs_oracle as O

R = 6371000


def _grid(ny=21, nx=36):
    lat = np.linspace(-80, 80, ny)
    lon = -180 + 360.0 / nx * np.arange(nx)
    return lat, lon


# ---------------------------------------------------------------- KAT-1
@pytest.mark.parametrize("order", [1, 3])
@pytest.mark.parametrize("K", [0, 2])
def test_kat1_zero_wind(order, K):
    lat, lon = _grid()
    U = np.zeros((4, lat.size, lon.size))
    x, y = O.parcel_propagation(U, U, lat, lon, timestep=-3600, SETTLS_order=K,
                                interp_order=order, cyclic_xboundary=True)
    X, Y = np.meshgrid(lon, lat)
    assert np.array_equal(y, Y)
    # Q7: lon == -180 exactly is not > -180, so it becomes (-180 % 180) == 0
    expect = X.copy()
    expect[X == -180] = 0.0
    assert np.array_equal(x, expect)


# ---------------------------------------------------------------- KAT-2
@pytest.mark.parametrize("K", [0, 1, 2, 4])
def test_kat2_uniform_zonal_wind(K):
    lat, lon = _grid(ny=21, nx=36)
    u0, dt, nt = 7.0, 600.0, 4
    U = np.full((nt, lat.size, lon.size), u0)
    V = np.zeros_like(U)
    x, y = O.parcel_propagation(U, V, lat, lon, timestep=dt, SETTLS_order=K,
                                interp_order=1, cyclic_xboundary=True)
    X, Y = np.meshgrid(lon, lat)
    assert np.array_equal(y, Y)
    # Q4: every SETTLS iteration ADDS 0.5*dt*c*(ua + 2u - u) = dt*c*u0.
    dlon = (1 + K) * dt * u0 * 180 / (np.pi * R * np.abs(np.cos(np.deg2rad(lat))))
    steps = nt - 1
    # rows 0 and ny-1 are pole rows (order 1): 'constant' mode.  The last row
    # maps to index ny > ny-1 and sees zero wind (Q2 + Q3); row 0 maps to index 0
    # and is inside.  Column nx-1 (index nx*(nx-1)/(nx-1)=nx > nx-1) sees zero
    # wind on pole rows only.
    moved = x - X
    inner = slice(1, -1)
    cols = slice(1, None)   # skip the lon==-180 column (Q7 rewrite to 0)
    np.testing.assert_allclose(moved[inner, cols], np.broadcast_to(
        (steps * dlon)[inner, None], moved[inner, cols].shape), rtol=1e-12)
    assert np.array_equal(moved[-1, cols], np.zeros_like(moved[-1, cols]))
    # row 0 ('constant'): inside except near its east edge, where the index
    # nx*(x-lon_min)/(lon_max-lon_min) exceeds nx-1 as soon as the seed has moved
    np.testing.assert_allclose(moved[0, 1:-3], steps * dlon[0], rtol=1e-12)
    assert moved[0, -1] == 0.0


# ---------------------------------------------------------------- KAT-3
@pytest.mark.parametrize("order,mode", [(1, "wrap"), (3, "wrap"), (1, "constant"), (2, "wrap"), (4, "wrap"), (5, "wrap")])
def test_kat3_interp_restated_vs_scipy(order, mode):
    rng = np.random.default_rng(3)
    f = rng.standard_normal((17, 23))
    ny, nx = f.shape
    cy = np.concatenate([rng.uniform(-3, ny + 3, 4000), rng.uniform(-3 * ny, 3 * ny, 2000),
                         [0, ny - 1, ny, -0.0, 1.0, ny - 2.0, 2 * (ny - 1), -(ny - 1.0)]])
    cx = np.concatenate([rng.uniform(-3, nx + 3, 4000), rng.uniform(-3 * nx, 3 * nx, 2000),
                         [0, nx - 1, nx, 5.0, -0.0, nx - 1.0, 0.0, 2 * (nx - 1.0)]])
    ref = map_coordinates(f, np.array([cy, cx]), order=order, mode=mode)
    got = O.interp_restated(f, cy, cx, order, mode)
    # orders 4 and 5: scipy's prefilter pole constants differ from sqrt-expression ones in the last bit and
    # the filter gain (~400) carries that into the coefficients
    np.testing.assert_allclose(got, ref, rtol=0, atol=2e-14 if order <= 3 else 2e-12)


# ---------------------------------------------------------------- KAT-4
@pytest.mark.parametrize("order", [2, 3, 4, 5])
def test_kat4_prefilter_vs_scipy(order):
    rng = np.random.default_rng(4)
    f = rng.standard_normal((19, 31))
    np.testing.assert_allclose(O.spline_prefilter_mirror(f, order),
                               spline_filter(f, order=order, mode="mirror"),
                               rtol=0, atol=5e-14 if order <= 3 else 5e-12)


# ---------------------------------------------------------------- KAT-5
def test_kat5_sigma_closed_form_vs_svd():
    rng = np.random.default_rng(5)
    dt = rng.standard_normal((9, 40, 50)) * np.array([1, 5, .1, 2, 1, 3, 0, 0, 0])[:, None, None]
    dt[6:] = 0
    ref = O.sigma_max(dt, "reference")
    np.testing.assert_allclose(O.sigma_max_closed_form(dt, "reference"), ref, rtol=1e-13)
    # Q13: reference layout is M = [[dXdx,dXdy,dYdx],[dYdy,dZdx,dZdy],[0,0,0]]
    M = np.zeros((3, 3))
    M[0], M[1] = dt[0:3, 7, 9], dt[3:6, 7, 9]
    assert np.isclose(ref[7, 9], np.linalg.svd(M, compute_uv=False)[0], rtol=1e-13)
    # physical layout = Jacobian d(X,Y,Z)/d(x,y): largest eigenvalue of F^T F
    phys = O.sigma_max(dt, "physical")
    F = np.array([[dt[0, 7, 9], dt[1, 7, 9]], [dt[2, 7, 9], dt[3, 7, 9]], [dt[4, 7, 9], dt[5, 7, 9]]])
    assert np.isclose(phys[7, 9], np.sqrt(np.linalg.eigvalsh(F.T @ F)[-1]), rtol=1e-13)
    np.testing.assert_allclose(O.sigma_max_closed_form(dt, "physical"), phys, rtol=1e-13)


def test_kat5_nan_in_nan_out():
    rng = np.random.default_rng(6)
    dt = rng.standard_normal((9, 6, 7))
    dt[6:] = 0
    dt[2, 3, 4] = np.nan
    s = O.sigma_max(dt)
    assert np.isnan(s[3, 4]) and np.isnan(s).sum() == 1      # Q14


# ---------------------------------------------------------------- KAT-6
def test_kat6_identity_flow_map():
    # x_dep = lon, y_dep = lat.  X = R sin(LAT) cos(LON) with LAT = lat-90 deg.
    lat = np.arange(-60.0, 60.5, 0.5)
    lon = -180 + 0.5 * np.arange(720)
    X, Y = np.meshgrid(lon, lat)
    clean = O.flowmap_gradient(X, Y, lat, lon, fd_fp32_cast=False)
    ref = O.flowmap_gradient(X, Y, lat, lon)
    lam, colat = np.deg2rad(X), np.deg2rad(Y - 90)
    coslat = np.cos(np.deg2rad(Y))
    # analytic derivatives wrt metric distance: d/dx = 1/(R cos(lat)) d/dlam, d/dy = 1/R d/dphi
    ana = np.stack([
        -np.sin(colat) * np.sin(lam) / coslat,      # dXdx
        np.cos(colat) * np.cos(lam),                # dXdy
        np.sin(colat) * np.cos(lam) / coslat,       # dYdx
        np.cos(colat) * np.sin(lam),                # dYdy
        np.zeros_like(X),                           # dZdx
        -np.sin(colat)])                            # dZdy
    inner = (slice(None), slice(2, -2), slice(None))
    # 4th-order stencil at 0.5 degree: truncation ~ h^4/30 ~ 2e-10
    np.testing.assert_allclose(clean[:6][inner], ana[inner], atol=5e-9)
    # Q11: fp32 rounding of |X| <= 6.4e6 (ulp 0.5 m) over dx = 55 km*cos(lat): ~1e-5 noise
    assert np.abs(ref[:6][inner] - ana[inner]).max() < 5e-5
    assert np.abs(ref[:6][inner] - ana[inner]).max() > 1e-8   # the cast is really there
    assert np.array_equal(ref[6:], np.zeros_like(ref[6:]))
    # sigma of the identity map with the reference's scrambled layout is finite, O(1)
    s = O.sigma_max(ref)
    assert np.isfinite(s).all() and 0.9 < s[2:-2].min() and s[2:-2].max() < 2.1


# ---------------------------------------------------------------- KAT-7
def test_kat7_stencil_polynomial():
    ny, nx = 12, 16
    i = np.arange(ny, dtype=np.float64)[:, None]
    j = np.arange(nx, dtype=np.float64)[None, :]
    a = (i ** 3 + j ** 2) * np.ones((ny, nx))
    a32 = a.astype(np.float32)                     # exactly representable
    d0 = O.fourth_order_derivative(a32, dim=0)
    # 4th-order stencil is exact for cubics: d/di i^3 = 3 i^2 on interior rows
    np.testing.assert_array_equal(d0[2:-2], np.broadcast_to(3 * i[2:-2] ** 2, (ny - 4, nx)).astype(np.float32))
    # Q12: two pole rows each side use a one-sided difference divided by 2
    for r in (0, 1):
        np.testing.assert_array_equal(d0[r], (a32[r + 1] - a32[r]) / 2)
    for r in (ny - 1, ny - 2):
        np.testing.assert_array_equal(d0[r], (a32[r] - a32[r - 1]) / 2)
    d1 = O.fourth_order_derivative(a32, dim=1)
    # interior columns: exact 2j; cyclic wrap columns see the jump (nx-1)^2 -> 0
    np.testing.assert_array_equal(d1[:, 2:-2], np.broadcast_to(2 * j[:, 2:-2], (ny, nx - 4)).astype(np.float32))
    jj = np.arange(nx)
    b = (jj.astype(np.float64) ** 2)
    exp = (4 / 3) * (b[(jj + 1) % nx] - b[(jj - 1) % nx]) / 2 - (1 / 3) * (b[(jj + 2) % nx] - b[(jj - 2) % nx]) / 4
    np.testing.assert_allclose(d1[3], exp.astype(np.float32), rtol=1e-7)
    assert d0.dtype == np.float32 and d1.dtype == np.float32
    # isglobal=False (LCS/tools.py:229-244): no wrap -- interior columns as before, the two first / last columns the
    # one-sided difference divided by 2, exactly as dim 0 treats its rows; dim 0 itself ignores the flag
    d1r = O.fourth_order_derivative(a32, dim=1, isglobal=False)
    np.testing.assert_array_equal(d1r[:, 2:-2], d1[:, 2:-2])
    for c in (0, 1):
        np.testing.assert_array_equal(d1r[:, c], (a32[:, c + 1] - a32[:, c]) / 2)
    for c in (nx - 1, nx - 2):
        np.testing.assert_array_equal(d1r[:, c], (a32[:, c] - a32[:, c - 1]) / 2)
    np.testing.assert_array_equal(O.fourth_order_derivative(a32, dim=0, isglobal=False), d0)


def test_q7_cyclic_wrap_values():
    x = np.array([[-365.0, -180.0, 180.0, 190.0, 365.0, -179.0, 179.0]])
    y = np.zeros_like(x)
    xo, _ = O._clamp(x, y, -180, 179, -90, 90, True, "pointwise")
    np.testing.assert_array_equal(xo, [[175.0, 0.0, -180.0, -170.0, -175.0, -179.0, 179.0]])


def test_q8_nan_latitude_becomes_ymin():
    x = np.array([[0.0]])
    y = np.array([[np.nan]])
    _, yo = O._clamp(x, y, -180, 179, -88, 88, True, "pointwise")
    assert yo[0, 0] == -88


def test_q9_noncyclic_clamp_variants():
    x = np.array([[0.0, 5.0, 0.0], [0.0, 0.0, 0.0], [-9.0, 0.0, 0.0]])
    y = np.zeros_like(x)
    xp, _ = O._clamp(x, y, -4, 4, -90, 90, False, "pointwise")
    np.testing.assert_array_equal(xp, [[0, 4, 0], [0, 0, 0], [-4, 0, 0]])
    xr_, _ = O._clamp(x, y, -4, 4, -90, 90, False, "reference_outer")
    # rows {2} x cols {0} for the min clamp, rows {0} x cols {1} for the max clamp
    np.testing.assert_array_equal(xr_, [[0, 4, 0], [0, 0, 0], [-4, 0, 0]])
    x2 = np.array([[9.0, 0.0], [0.0, 9.0]])
    xo2, _ = O._clamp(x2, np.zeros_like(x2), -4, 4, -90, 90, False, "reference_outer")
    np.testing.assert_array_equal(xo2, [[4, 4], [4, 4]])   # the outer-product defect


def test_q6_backward_uses_stored_order():
    lat, lon = _grid()
    rng = np.random.default_rng(7)
    U = rng.standard_normal((3, lat.size, lon.size))
    V = rng.standard_normal((3, lat.size, lon.size))
    xb, yb = O.parcel_propagation(U, V, lat, lon, timestep=-600, interp_order=1, cyclic_xboundary=True)
    # negative dt with the SAME field order; reversing the data gives something else
    xr_, yr_ = O.parcel_propagation(U[::-1], V[::-1], lat, lon, timestep=-600, interp_order=1, cyclic_xboundary=True)
    assert not np.allclose(xb, xr_)
    # one Euler step by hand on an interior seed
    va = map_coordinates(V[0], [[5 * lat.size / (lat.size - 1)], [7 * lon.size / (lon.size - 1)]], order=1, mode="wrap")[0]
    x1, y1 = O.parcel_propagation(U[:2], V[:2], lat, lon, timestep=-600, interp_order=1, cyclic_xboundary=True)
    assert np.isclose(y1[5, 7], lat[5] + -600 * (180 / (R * np.pi)) * va, rtol=1e-14)


def test_dtype_follows_inputs_q10():
    lat, lon = _grid()
    rng = np.random.default_rng(8)
    U = rng.standard_normal((3, lat.size, lon.size)).astype(np.float32)
    x, y = O.parcel_propagation(U, U, lat.astype(np.float32), lon.astype(np.float32), timestep=-600,
                                SETTLS_order=1, interp_order=1, cyclic_xboundary=True)
    assert x.dtype == np.float32 and y.dtype == np.float32
    x, y = O.parcel_propagation(U, U, lat, lon, timestep=-600, SETTLS_order=1, interp_order=1,
                                cyclic_xboundary=True)
    assert x.dtype == np.float64           # fp64 coords keep fp64 positions
    out = O.xr_map_coordinates(U[0], lat, lon, *np.meshgrid(lon, lat), order=1)
    assert out.dtype == np.float32


def test_return_traj_shape_and_first_entry():
    lat, lon = _grid()
    rng = np.random.default_rng(9)
    U = rng.standard_normal((4, lat.size, lon.size))
    tx, ty = O.parcel_propagation(U, U, lat, lon, timestep=600, interp_order=1, cyclic_xboundary=True,
                                  return_traj=True)
    assert tx.shape == (4, lat.size, lon.size)
    X, Y = np.meshgrid(lon, lat)
    assert np.array_equal(tx[0], X) and np.array_equal(ty[0], Y)
    x, y = O.parcel_propagation(U, U, lat, lon, timestep=600, interp_order=1, cyclic_xboundary=True)
    assert np.array_equal(tx[-1], x) and np.array_equal(ty[-1], y)


def test_order0_raises_like_reference():
    lat, lon = _grid()
    U = np.zeros((2, lat.size, lon.size))
    with pytest.raises(ValueError):
        O.parcel_propagation(U, U, lat, lon, interp_order=0)
