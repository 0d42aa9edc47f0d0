"""Global pre-processing kernels behind the C ABI (lc_regrid_common_grid, lc_spectral_truncate; LCS.py:105-118)
against the CPU oracle (scipy interp1d + pandas nearest; the restated SPHEREPACK truncation)."""
import numpy as np
import pytest
import torch

from lagrangiancoherence_amd import preprocess as PP
from oracle import preprocess_oracle as PO

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng():
    from lagrangiancoherence_amd.engine import Engine
    e = Engine(0)
    yield e
    e.close()


def _field(nt=2, ny=37, nx=72, lat0=-90.0, lat1=90.0):
    rng = np.random.default_rng(0)
    lat = np.linspace(lat0, lat1, ny)
    lon = -180 + 360.0 / nx * np.arange(nx)
    return rng.standard_normal((nt, ny, nx)), lat, lon


def test_regrid_kernel_matches_oracle(eng):
    for kw in ({}, dict(lat0=-80.0, lat1=80.0), dict(ny=19, nx=40), dict(nt=5, ny=181, nx=360)):
        u, lat, lon = _field(**kw)
        ref, lats, lons = PO.regrid_common_grid(u, lat, lon)
        got, glats, glons = PP.regrid_common_grid(eng, u, lat, lon)
        assert np.array_equal(glats, lats) and np.array_equal(glons, lons)
        np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=0, atol=1e-14)
    # float32 input comes back float64, as xarray's interp does, with scipy's promotion: (y_hi - y_lo) is formed
    # in float32, the slope and the result in float64
    u32 = u.astype(np.float32)
    got, _, _ = PP.regrid_common_grid(eng, u32, lat, lon)
    assert got.dtype == torch.float64
    np.testing.assert_allclose(got.cpu().numpy(), PO.regrid_common_grid(u32, lat, lon)[0], rtol=0, atol=1e-14)
    # NaN in the source: interp gives NaN there, the nearest source node fills it (LCS.py:113) -- unless that is NaN too
    u2, lat2, lon2 = _field()
    u2[0, 10, 20] = np.nan
    ref = PO.regrid_common_grid(u2, lat2, lon2)[0]
    got = PP.regrid_common_grid(eng, u2, lat2, lon2)[0].cpu().numpy()
    assert np.array_equal(np.isnan(got), np.isnan(ref)) and np.isnan(ref).any()
    np.testing.assert_allclose(got[~np.isnan(ref)], ref[~np.isnan(ref)], rtol=0, atol=1e-14)


def test_nearest_tie_goes_to_the_larger_index_like_pandas(eng):
    src = np.array([0.0, 1.0, 2.0, 3.0])
    dst = np.array([-1.0, 3.5, 4.0])                        # all outside: nearest fill decides
    u = np.arange(16.0).reshape(1, 4, 4)
    got = eng.regrid(u, src, src, dst, dst).cpu().numpy()
    ref = PO.regrid_common_grid(u, src, src, dst, dst)[0]
    assert np.array_equal(got, ref) and got[0, 0, 0] == 0.0 and got[0, 2, 2] == 15.0
    # ties inside the range never reach the nearest path; a tie outside cannot occur -- check pandas' rule on a
    # 2-node axis through the oracle's own indexer instead
    import pandas as pd
    assert list(pd.Index(src).get_indexer([0.5, 1.5], method="nearest")) == [1, 2]


def _harmonic(m, n, nlat, nlon, phase=0.3):
    theta = np.arange(nlat) * np.pi / (nlat - 1)
    lam = 2 * np.pi * np.arange(nlon) / nlon
    P = PO.legendre_normalized(m, n, np.cos(theta))[n - m]
    return (P[:, None] * np.cos(m * lam + phase)[None, :])[::-1]       # ascending latitude


def test_truncation_kernels_are_the_exact_projector_on_band_limited_fields(eng):
    nlat, nlon, T = 60, 121, 8
    trunc = lambda f: PP.spectral_truncate(eng, f, T).cpu().numpy()
    for m, n in [(0, 0), (0, 5), (1, 1), (1, 8), (2, 8), (8, 8), (3, 7)]:
        f = _harmonic(m, n, nlat, nlon)
        np.testing.assert_allclose(trunc(f), f, atol=2e-12)            # degree <= T: unchanged
    for m, n in [(0, 9), (0, 30), (1, 30), (2, 31), (0, 59), (1, 58), (8, 40)]:
        np.testing.assert_allclose(trunc(_harmonic(m, n, nlat, nlon)), 0.0, atol=2e-12)   # T < n <= nlat-1: removed
    np.testing.assert_allclose(trunc(_harmonic(9, 12, nlat, nlon)), 0.0, atol=2e-12)       # zonal wavenumber above T
    g = np.random.default_rng(1).standard_normal((3, nlat, nlon))      # arbitrary data: a projector is idempotent
    once = trunc(g)
    np.testing.assert_allclose(trunc(once), once, atol=1e-11)
    assert np.abs(once).max() < np.abs(g).max()


def test_truncation_kernels_match_oracle_on_the_reference_grid(eng):
    rng = np.random.default_rng(2)
    f = rng.standard_normal((2, 3, 360, 721))                          # leading dims are batch
    ref = PO.spectral_truncate(f, 20)
    got = PP.spectral_truncate(eng, f, 20)
    assert tuple(got.shape) == f.shape
    np.testing.assert_allclose(got.cpu().numpy(), ref, rtol=0, atol=5e-13)
    got32 = PP.spectral_truncate(eng, f.astype(np.float32), 20)
    assert got32.dtype == torch.float32
    np.testing.assert_allclose(got32.cpu().numpy(), ref, atol=5e-6)
    # another (nlat, nlon, T): the context's operator cache is rebuilt, then reused
    f2 = rng.standard_normal((4, 91, 180))
    for _ in range(2):
        np.testing.assert_allclose(PP.spectral_truncate(eng, f2, 12).cpu().numpy(), PO.spectral_truncate(f2, 12), atol=5e-13)
    np.testing.assert_allclose(PP.spectral_truncate(eng, f[0], 20).cpu().numpy(), ref[0], rtol=0, atol=5e-13)
    with pytest.raises(ValueError):
        eng.spectral_truncate(np.zeros((40, 90)), 45)                  # T > (nlon - 1) / 2
    # truncations beyond 31 (T42, T63 ... what windspharm accepts): the kernels walk the spectral columns 64 at a time
    f3 = rng.standard_normal((2, 96, 192))
    for T in (42, 63):
        np.testing.assert_allclose(PP.spectral_truncate(eng, f3, T).cpu().numpy(), PO.spectral_truncate(f3, T), atol=2e-12)
    # more than 2048 longitudes (a 0.125 degree grid): the forward DFT stages fewer rows per block
    f4 = rng.standard_normal((1, 33, 2880))
    np.testing.assert_allclose(PP.spectral_truncate(eng, f4, 10).cpu().numpy(), PO.spectral_truncate(f4, 10), atol=5e-12)


def test_truncation_on_gaussian_latitudes(eng):
    """windspharm gridtype 'gaussian' (LCS.py:115-117 when interp_to_common_grid=False hands it a Gaussian grid)."""
    rng = np.random.default_rng(7)
    nlat, nlon = 64, 128
    glat, _ = PO.gaussian_latitudes(nlat)
    assert PP.inspect_gridtype(glat) == "gaussian"
    f = rng.standard_normal((3, nlat, nlon))
    for T in (21, 42):
        got = PP.spectral_truncate(eng, f, T, "gaussian").cpu().numpy()
        np.testing.assert_allclose(got, PO.spectral_truncate(f, T, "gaussian"), atol=2e-12)
        np.testing.assert_allclose(PP.spectral_truncate(eng, got, T, "gaussian").cpu().numpy(), got, atol=1e-11)   # idempotent
    # the same rows read as SPHEREPACK's equally spaced grid give another field: the grid type matters
    assert np.abs(PP.spectral_truncate(eng, f, 21, "regular").cpu().numpy() - PO.spectral_truncate(f, 21, "gaussian")).max() > 1e-3


def test_global_host_route_equals_the_dropin_and_the_oracle():
    """lc_lcs_global_host (torch-free: host arrays in, host arrays out) = the drop-in's LCS(...)(ds, isglobal=True):
    regrid, T20, cubic interpolation, cyclic (LCS.py:105-157); and both against the oracle's composition."""
    from lagrangiancoherence_amd import flows
    from tests import labelled
    from lagrangiancoherence_amd.engine import lcs_global_host
    from LagrangianCoherence.LCS.LCS import LCS
    from oracle import lcs_oracle as O
    u, v, lat, lon = flows.config1()
    out = lcs_global_host(u, v, lat, lon, -21600.0, SETTLS_order=4)
    assert out["sigma"].shape == (360, 721) and out["sigma"].dtype == np.float64
    assert np.array_equal(out["latitude"], PP.COMMON_LATS) and np.array_equal(out["longitude"], PP.COMMON_LONS)
    times = np.arange(u.shape[0]).astype("datetime64[h]")
    mk = lambda a: labelled.DataArray(a, ("time", "latitude", "longitude"), {"time": times, "latitude": lat, "longitude": lon})
    eig, xd, yd = LCS(timestep=-21600, SETTLS_order=4, return_dpts=True)(u=mk(u), v=mk(v), isglobal=True, verbose=False)
    assert np.array_equal(eig.values[0], out["sigma"]) and np.array_equal(xd.values, out["x_dep"])
    ur, lats, lons = PO.regrid_common_grid(u, lat, lon)
    vr, _, _ = PO.regrid_common_grid(v, lat, lon)
    s, x, y = O.lcs(PO.spectral_truncate(ur, 20), PO.spectral_truncate(vr, 20), lats, lons, timestep=-21600.0,
                    SETTLS_order=4, interp_order=3, cyclic_xboundary=True)
    np.testing.assert_allclose(out["x_dep"], x, rtol=0, atol=1e-8)
    np.testing.assert_allclose(out["y_dep"], y, rtol=0, atol=1e-8)
    np.testing.assert_allclose(out["sigma"][20:-20], s[20:-20], rtol=1e-7)
    # no regrid, no truncation: the example's own 89 x 180 grid, float32 in -> float32 out
    o2 = lcs_global_host(u.astype(np.float32), v.astype(np.float32), lat, lon, -21600.0, SETTLS_order=4, interp_order=1,
                         interp_to_common_grid=False, truncation=None)
    assert o2["sigma"].shape == (89, 180) and o2["sigma"].dtype == np.float32 and np.isfinite(o2["sigma"]).all()
    with pytest.raises(ValueError, match="non-global"):                 # windspharm refuses the 89-row grid
        lcs_global_host(u, v, lat, lon, -21600.0, interp_to_common_grid=False, truncation=20)
