"""Ridge extraction (LCS/tools.py:52-155).

CPU: the closed form of LAPACK dgeev on a symmetric 2x2 (what the HIP kernel implements) against
numpy.linalg.eig itself -- every branch: deflation, real eigenvalues, almost-equal eigenvalues.
GPU: the device implementation against the oracle, which calls numpy.linalg.eig per point as the
reference does."""
import numpy as np
import pytest

from oracle import ridges_oracle as RO


def _matrices(n=120000, seed=0):
    rng = np.random.default_rng(seed)
    a = rng.standard_normal(n) * 10.0 ** rng.integers(-20, 3, n)
    b = rng.standard_normal(n) * 10.0 ** rng.integers(-22, 3, n)
    d = rng.standard_normal(n) * 10.0 ** rng.integers(-20, 3, n)
    a[:100] = d[:100]
    b[100:200] = 0
    a[200:300] = 0
    d[300:400] = -a[300:400]
    d[600:700] = 0
    a[700:800] = 0
    d[700:800] = 0
    a[2000:3000] = d[2000:3000] * (1 + 1e-16 * rng.integers(-5, 5, 1000))
    a[5000:60000] *= 1e-10      # realistic Hessian magnitudes (FTLE per metre^2)
    b[5000:60000] *= 1e-10
    d[5000:60000] *= 1e-10
    return a, b, d


def test_closed_form_equals_numpy_linalg_eig():
    a, b, d = _matrices()
    H = np.stack([np.stack([a, b], -1), np.stack([b, d], -1)], -2)
    w, V = np.linalg.eig(H)
    w0, w1, Vc = RO.dlanv2_sym(a, b, d)
    assert np.array_equal(w[:, 0], w0) and np.array_equal(w[:, 1], w1)          # eigenvalues bit for bit
    assert np.abs(V - Vc).max() < 1e-15
    # all three branches were exercised
    ulp = np.finfo(float).eps
    p, ab = 0.5 * (a - d), np.abs(b)
    with np.errstate(all="ignore"):
        z0 = (p / np.maximum(np.abs(p), ab)) * p + (ab / np.maximum(np.abs(p), ab)) * ab
    assert ((z0 < 4 * ulp) & (b != 0)).sum() > 100 and (ab <= ulp * (np.abs(a) + np.abs(d))).sum() > 100


def test_oracle_quirks():
    lat = np.linspace(-60, 60, 41)
    lon = -180 + 4.5 * np.arange(80)
    LON, LAT = np.meshgrid(lon, lat)
    f = np.exp(-((LAT - 5) / 10.0) ** 2) * (1 + 0.2 * np.cos(np.deg2rad(3 * LON)))   # a zonal ridge at 5N
    mask, eigmin, dt = RO.find_ridges_spherical_hessian(f, lat, lon, sigma=0.5)
    assert set(np.unique(mask)) <= {0.0, 1.0} and mask.sum() > 0
    assert (eigmin[mask == 1] < 0).all()                                       # R4 + tools.py:138
    # on the crest (5N, a grid row) the gradient across it vanishes and the curvature is negative
    assert mask[np.abs(lat - 5).argmin()].mean() > 0.9
    # on the flanks the gradient (1e-6 per metre) is far above the tolerance (5e-7): not a ridge
    assert mask[np.abs(lat - 14).argmin()].sum() == 0
    # R3: a NaN gradient flags the point (if eigmin < 0): both where() conditions are False for NaN
    m2 = np.where(np.abs(np.nan) <= 1.0, np.nan, 0)
    m2 = np.where(np.abs(np.nan) > 1.0, m2, 1)
    assert m2 == 1


@pytest.mark.gpu
def test_ridge_classify_kernel_vs_numpy_eig():
    from lagrangiancoherence_amd.engine import Engine
    eng = Engine(0)
    a, b, d = _matrices(60000, seed=3)
    rng = np.random.default_rng(4)
    gx, gy = rng.standard_normal(a.size), rng.standard_normal(a.size)
    a[10], b[11], d[12] = np.inf, np.nan, -np.inf                              # cleaned to 0 (tools.py:92-93)
    gx[13] = np.nan                                                             # R3
    tol = 0.3
    mask, eigmin, dt, vec = (t.cpu().numpy() for t in eng.ridge_classify(a, b, d, gx, gy, tol, return_eigvec=True))
    ac, bc, dc = (np.where(np.isfinite(v), v, 0.0) for v in (a, b, d))
    w, V = np.linalg.eig(np.stack([np.stack([ac, bc], -1), np.stack([bc, dc], -1)], -2))
    n = np.arange(a.size)
    row = V[n, np.argmin(w, axis=1), :]
    dt_ref = row[:, 0] * gx + row[:, 1] * gy
    em_ref = w[n, np.argmax(np.abs(w), axis=1)]
    # the device's double sqrt/divide can differ from the host's in the last bit
    np.testing.assert_allclose(eigmin, em_ref, rtol=2e-15, atol=0)
    np.testing.assert_allclose(vec, row.T, rtol=0, atol=5e-15)                  # tools.py:107, the ROW of V
    np.testing.assert_allclose(dt, dt_ref, rtol=0, atol=5e-15 * max(1.0, np.nanmax(np.abs(dt_ref))), equal_nan=True)
    m = np.where(np.abs(dt_ref) <= tol, dt_ref, 0)
    m = np.where(np.abs(dt_ref) > tol, m, 1)
    m = np.where(np.sign(em_ref) == -1, m, 0)
    borderline = np.abs(np.abs(dt_ref) - tol) < 1e-12
    assert np.array_equal(mask[~borderline], m[~borderline])
    assert mask[13] == (1.0 if em_ref[13] < 0 else 0.0)
    eng.close()


@pytest.mark.gpu
def test_find_ridges_drop_in_vs_oracle():
    from LagrangianCoherence.LCS.tools import find_ridges_spherical_hessian
    from lagrangiancoherence_amd import flows
    from tests import labelled
    from lagrangiancoherence_amd.dropin import get_engine
    u, v, lat, lon = flows.config1()
    eng = get_engine()
    f = eng.prepare_field(u, v, lat, lon, 1)
    r = eng.lcs(f, lat, lon, -21600, SETTLS_order=4, interp_order=1)
    ftle = np.log(r["sigma"].cpu().numpy()) / 2                               # examples/ideal_vortex.py:288
    da = labelled.DataArray(ftle.T, ["longitude", "latitude"], {"latitude": lat, "longitude": lon}, name="ftle")
    for sigma, tol in ((0.5, 0.0005e-3), (1.2, 2e-6)):
        ridges, eigmin = find_ridges_spherical_hessian(da, sigma=sigma, tolerance_threshold=tol)
        assert ridges.dims == ("longitude", "latitude") and ridges.shape == (180, 89)   # original dim order
        m_ref, e_ref, dt_ref = RO.find_ridges_spherical_hessian(ftle, lat, lon, sigma=sigma, tolerance_threshold=tol)
        np.testing.assert_allclose(eigmin.values.T, e_ref, rtol=1e-12, atol=1e-25)
        borderline = np.abs(np.abs(dt_ref) - tol) < 1e-9 * tol
        assert np.array_equal(ridges.values.T[~borderline], m_ref[~borderline])
        assert 0 < ridges.values.sum() < ridges.values.size
        # return_eigvectors=True: the reference's six-tuple (tools.py:140-147)
        six = find_ridges_spherical_hessian(da, sigma=sigma, tolerance_threshold=tol, return_eigvectors=True)
        ref6 = RO.find_ridges_spherical_hessian(ftle, lat, lon, sigma=sigma, tolerance_threshold=tol,
                                                return_eigvectors=True)
        assert len(six) == 6
        assert np.array_equal(six[0].values, ridges.values) and np.array_equal(six[1].values, eigmin.values)
        assert six[3].dims == ("eigvectors", "longitude", "latitude") and six[4].dims == ("elements", "longitude", "latitude")
        assert list(six[3]["eigvectors"].values) == ["d2dadxdy", "d2dadydx"]
        assert list(six[4]["elements"].values) == ["ddadx", "ddady"]
        assert six[2].dims == six[5].dims == ("longitude", "latitude")
        scale = np.nanmax(np.abs(ref6[2]))
        np.testing.assert_allclose(six[2].values.T, ref6[2], rtol=0, atol=1e-12 * scale)           # raw product
        np.testing.assert_allclose(six[3].values.transpose(0, 2, 1), ref6[3], rtol=0, atol=1e-12)  # unit vectors
        np.testing.assert_allclose(six[4].values.transpose(0, 2, 1), ref6[4], rtol=1e-12, atol=0)  # gradient
        # arctan(e0/e1) is ill-conditioned where e1 ~ 0 (the angle jumps between -90 and +90): compare elsewhere
        ok = np.abs(ref6[3][1]) > 1e-6
        a_ref = np.where(np.isnan(ref6[5]), 0, ref6[5])
        a_got = np.where(np.isnan(six[5].values.T), 0, six[5].values.T)
        np.testing.assert_allclose(a_got[ok], a_ref[ok], rtol=0, atol=1e-6)


@pytest.mark.gpu
def test_find_ridges_regional_branch_vs_oracle():
    """find_ridges_spherical_hessian(isglobal=False) (LCS/tools.py:52-55,77-81): every derivative takes the regional
    longitude stencil (one-sided on the two first / last columns, tools.py:229-244) instead of the cyclic one."""
    from LagrangianCoherence.LCS.tools import find_ridges_spherical_hessian
    from tests import labelled
    lat = np.linspace(-30, 30, 61)
    lon = np.linspace(-80, -20, 91)                                              # a regional box: nothing to wrap
    LON, LAT = np.meshgrid(lon, lat)
    f = np.exp(-((LAT - 0.3 * (LON + 50)) / 6.0) ** 2) + 0.1 * np.sin(np.deg2rad(9 * LON)) * np.cos(np.deg2rad(7 * LAT))
    da = labelled.DataArray(f, ["latitude", "longitude"], {"latitude": lat, "longitude": lon}, name="ftle")
    tol = 2e-7
    out = {}
    for g in (False, True):
        ridges, eigmin = find_ridges_spherical_hessian(da, sigma=0.8, tolerance_threshold=tol, isglobal=g)
        m_ref, e_ref, dt_ref = RO.find_ridges_spherical_hessian(f, lat, lon, sigma=0.8, tolerance_threshold=tol, isglobal=g)
        np.testing.assert_allclose(eigmin.values, e_ref, rtol=1e-12, atol=1e-25)
        borderline = np.abs(np.abs(dt_ref) - tol) < 1e-9 * tol
        assert np.array_equal(ridges.values[~borderline], m_ref[~borderline])
        out[g] = eigmin.values
    # the two branches differ on the box's edge columns (second derivatives: 4 columns deep) and nowhere else
    assert np.array_equal(out[False][:, 4:-4], out[True][:, 4:-4]) and not np.array_equal(out[False][:, :4], out[True][:, :4])
