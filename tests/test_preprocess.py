"""Global pre-processing (LCS.py:105-118), CPU side: the oracle's known answers and the
host logic of the product (windspharm's grid inspection).  The product's kernels (lc_regrid_common_grid,
lc_spectral_truncate) are compared with this oracle on the GPU in tests/test_preprocess_gpu.py."""
import numpy as np
import pytest

from lagrangiancoherence_amd import preprocess as PP
from oracle import preprocess_oracle as PO


# ------------------------------------------------------------------ regrid
def _field(nt=2, ny=37, nx=72, lat0=-90.0, lat1=90.0):
    rng = np.random.default_rng(0)
    lat = np.linspace(lat0, lat1, ny)
    lon = -180 + 360.0 / nx * np.arange(nx)
    return rng.standard_normal((nt, ny, nx)), lat, lon


def test_regrid_oracle_known_answers():
    u, lat, lon = _field()
    out, lats, lons = PO.regrid_common_grid(u, lat, lon)
    assert out.shape == (2, 360, 721) and lats[0] == -89.75 and lons[-1] == 179.5 and not np.isnan(out).any()
    # a field linear in lat and lon is reproduced exactly inside the source range
    LON, LAT = np.meshgrid(lon, lat)
    lin = (2.0 * LAT - 0.5 * LON + 3.0)[None]
    o, _, _ = PO.regrid_common_grid(lin, lat, lon)
    LONt, LATt = np.meshgrid(lons, lats)
    inside = LONt <= lon[-1]
    np.testing.assert_allclose(o[0][inside], (2.0 * LATt - 0.5 * LONt + 3.0)[inside], atol=1e-11)
    # east of the last source longitude (175 here) the nearest source column is copied (LCS.py:109,113)
    j = np.abs(lat - lats[100]).argmin()
    assert o[0][100, -1] != (2.0 * LATt - 0.5 * LONt + 3.0)[100, -1]
    col = o[0][:, lons > lon[-1]]
    assert np.allclose(col, col[:, :1])                       # constant along lon there: nearest fill
    # a source grid that does not reach the poles: rows beyond it are nearest-filled too
    u2, lat2, lon2 = _field(lat0=-80.0, lat1=80.0)
    o2, _, _ = PO.regrid_common_grid(u2, lat2, lon2)
    assert np.array_equal(o2[:, 0], o2[:, 5]) and not np.isnan(o2).any()


# ------------------------------------------------------------------ spectral truncation
def _harmonic(m, n, nlat, nlon, phase=0.3):
    theta = np.arange(nlat) * np.pi / (nlat - 1)
    lam = 2 * np.pi * np.arange(nlon) / nlon
    P = PO.legendre_normalized(m, n, np.cos(theta))[n - m]
    return (P[:, None] * np.cos(m * lam + phase)[None, :])[::-1]       # ascending latitude


def test_truncation_is_the_exact_projector_on_band_limited_fields():
    nlat, nlon, T = 60, 121, 8
    trunc = lambda f: PO.spectral_truncate(f, T)
    for m, n in [(0, 0), (0, 5), (1, 1), (1, 8), (2, 8), (8, 8), (3, 7)]:
        f = _harmonic(m, n, nlat, nlon)
        np.testing.assert_allclose(trunc(f), f, atol=2e-12)            # degree <= T: unchanged
    for m, n in [(0, 9), (0, 30), (1, 30), (2, 31), (0, 59), (1, 58), (8, 40)]:
        np.testing.assert_allclose(trunc(_harmonic(m, n, nlat, nlon)), 0.0, atol=2e-12)   # T < n <= nlat-1: removed
    f = _harmonic(9, 12, nlat, nlon)                                   # zonal wavenumber above T
    np.testing.assert_allclose(trunc(f), 0.0, atol=2e-12)
    g = np.random.default_rng(1).standard_normal((3, nlat, nlon))      # arbitrary data: a projector is idempotent
    once = trunc(g)
    np.testing.assert_allclose(trunc(once), once, atol=1e-11)
    assert np.abs(once).max() < np.abs(g).max()


def test_grid_inspection_like_windspharm():
    PP.check_regular_global_lat(PP.COMMON_LATS)                        # 360 rows at +-(90 - 0.25): accepted
    PP.check_regular_global_lat(np.linspace(-90, 90, 181))             # odd count: poles + equator
    with pytest.raises(ValueError, match="non-global"):
        PP.check_regular_global_lat(np.arange(-88.0, 89.0, 2.0))       # the example's own 89-row grid
    with pytest.raises(ValueError, match="equally-spaced"):
        PP.check_regular_global_lat(np.array([-60.0, -10.0, 0.0, 70.0]))
    with pytest.raises(ValueError, match="too high"):
        PP.spectral_truncate(None, np.zeros((10, 30)), 20)             # refused before any engine call
