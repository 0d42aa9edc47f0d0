"""CPU oracle for the global pre-processing of LCS.__call__ (LCS/LCS.py:105-118) -- TEST INFRASTRUCTURE ONLY.

Two steps, both "next" rows of SURVEY.md section 8f (rank 2):

* ``regrid_common_grid``  LCS.py:107-114: ``u.interp(linear)`` onto lats linspace(-89.75, 89.75, 360),
  lons linspace(-180, 179.5, 721), holes (targets outside the source range) filled from
  ``u.reindex(method='nearest')``.  xarray is absent here; this calls the kernels xarray delegates to:
  ``scipy.interpolate.interp1d(kind='linear', bounds_error=False)`` along latitude then longitude, and
  ``pandas.Index.get_indexer(method='nearest')``.

* ``spectral_truncate``  LCS.py:115-118: ``windspharm.xarray.VectorWind(u, v).truncate(f, truncation=T)``
  = ``Spharmt.grdtospec(f, ntrunc=T)`` followed by ``Spharmt.spectogrd`` of pyspharm / SPHEREPACK
  (third-party, not vendored, not installed, unpinned in requirements.txt).  PARITY UNPINNED: this is a
  restatement of the *published* algorithm, not checked against pyspharm:
    - windspharm accepts the 360-row grid (+-89.75) as 'regular' and hands it to SPHEREPACK, whose
      equally spaced grid is theta_i = i*pi/(nlat-1), poles included; the rows are transformed as if
      they sat there (a quarter-degree mis-registration at the poles that the reference inherits);
    - SPHEREPACK's equally-spaced analysis (Swarztrauber 1979, "Z functions") integrates the
      trigonometric interpolant of each zonal Fourier coefficient exactly against Pbar^m_n sin(theta):
      cosine series for even m, sine series for odd m;
    - triangular truncation n <= T, synthesis on the same grid, longitudes equally spaced and periodic.
  SPHEREPACK computes in single precision; this oracle is float64 (agreement to ~1e-6 is the most one
  could ask of the real thing).  What IS pinned (tests/test_preprocess.py): the operator is the exact
  L2 projector for band-limited fields -- harmonics of degree <= T come back unchanged, degrees
  T < n <= nlat-1 vanish -- and the quadrature is cross-checked with an independent rule.
"""
from __future__ import annotations

import numpy as np
import pandas as pd
from scipy.interpolate import interp1d

__all__ = ["COMMON_LATS", "COMMON_LONS", "regrid_common_grid", "legendre_normalized", "truncation_operators",
           "spectral_truncate"]

COMMON_LATS = np.linspace(-89.75, 89.75, 180 * 2)        # LCS.py:107
COMMON_LONS = np.linspace(-180, 179.5, 360 * 2 + 1)      # LCS.py:108


def regrid_common_grid(u, lat, lon, lats=COMMON_LATS, lons=COMMON_LONS):
    """u: (nt, nlat, nlon), lat/lon ascending.  Returns (u_new, lats, lons).  LCS.py:107-114."""
    u = np.asarray(u)
    f_lat = interp1d(lat, u, kind="linear", axis=1, bounds_error=False, fill_value=np.nan, assume_sorted=True)
    tmp = f_lat(lats)                                                       # latitude first
    f_lon = interp1d(lon, tmp, kind="linear", axis=2, bounds_error=False, fill_value=np.nan, assume_sorted=True)
    u_interp = f_lon(lons)                                                  # LCS.py:111
    jn = pd.Index(lat).get_indexer(lats, method="nearest")
    i_n = pd.Index(lon).get_indexer(lons, method="nearest")
    u_reindex = u[:, jn][:, :, i_n]                                         # LCS.py:109
    return np.where(~np.isnan(u_interp), u_interp, u_reindex), lats, lons  # LCS.py:113


# ------------------------------------------------------------------------------------------------
def legendre_normalized(m, nmax, x):
    """Pbar^m_n(x), n = m..nmax, orthonormal on [-1,1]: integral Pbar^2 dx = 1.  Shape (nmax-m+1, len(x))."""
    x = np.asarray(x, dtype=np.float64)
    s = np.sqrt(np.maximum(0.0, 1.0 - x * x))
    pmm = np.full_like(x, np.sqrt(0.5))                      # Pbar^0_0
    for k in range(1, m + 1):
        pmm = -np.sqrt((2 * k + 1) / (2.0 * k)) * s * pmm     # Condon-Shortley phase (irrelevant to a projector)
    out = [pmm]
    if nmax > m:
        out.append(np.sqrt(2 * m + 3.0) * x * pmm)
    for n in range(m + 2, nmax + 1):
        a = np.sqrt((4.0 * n * n - 1.0) / (n * n - m * m))
        b = np.sqrt(((n - 1.0) ** 2 - m * m) / (4.0 * (n - 1.0) ** 2 - 1.0))
        out.append(a * (x * out[-1] - b * out[-2]))
    return np.stack(out)


def truncation_operators(nlat, T):
    """Per zonal wavenumber m = 0..T the (nlat x nlat) matrix synthesis . analysis on SPHEREPACK's
    equally spaced grid theta_i = i*pi/(nlat-1) (row 0 = north pole)."""
    N = nlat - 1
    theta = np.arange(nlat) * np.pi / N
    i = np.arange(nlat)
    ops = []
    # Gauss-Legendre in x = cos(theta): basis_k(theta) * Pbar^m_n is a polynomial in x of degree
    # <= N + T + 1 for the parity-matched basis (cos k theta = T_k(x) with even m; sin k theta * Pbar with
    # odd m = (1-x^2)^((m+1)/2) U_{k-1}(x) q(x)), so the rule is exact with Q > (N + T + 2) / 2 nodes
    Q = N + T + 8
    xq, wq = np.polynomial.legendre.leggauss(Q)
    tq = np.arccos(xq)
    for m in range(T + 1):
        P_grid = legendre_normalized(m, T, np.cos(theta))          # (n, i)   synthesis
        P_q = legendre_normalized(m, T, xq)                        # (n, q)
        if m % 2 == 0:
            k = np.arange(0, N + 1)
            B = (2.0 / N) * np.cos(np.outer(k, i) * np.pi / N)      # b_k = (2/N) sum'' g_i cos(k i pi/N)
            B[:, [0, -1]] *= 0.5
            B[[0, -1], :] *= 0.5                                    # sum'' over k in the interpolant too
            basis_q = np.cos(np.outer(k, tq))
        else:
            k = np.arange(1, N)
            B = (2.0 / N) * np.sin(np.outer(k, i) * np.pi / N)
            basis_q = np.sin(np.outer(k, tq))
        integ = (P_q * wq[None, :]) @ basis_q.T                     # (n, k): integral basis_k Pbar sin(theta) dtheta
        A = integ @ B                                               # (n, i)   analysis ("Z functions")
        ops.append(P_grid.T @ A)                                    # (i, i')
    return ops


def spectral_truncate(f, T=20):
    """f: (..., nlat, nlon) with latitude ASCENDING (south -> north), as everywhere in this repo.
    Returns the T-truncated field on the same grid.  windspharm reorders to north -> south internally
    and gives the result back on the field's own coordinates."""
    f = np.asarray(f, dtype=np.float64)
    nlat, nlon = f.shape[-2:]
    g = f[..., ::-1, :]                                             # north -> south
    F = np.fft.rfft(g, axis=-1)                                     # zonal Fourier coefficients
    ops = truncation_operators(nlat, T)
    out = np.zeros_like(F)
    for m in range(min(T, F.shape[-1] - 1) + 1):
        out[..., :, m] = np.einsum("ij,...j->...i", ops[m], F[..., :, m])
    return np.fft.irfft(out, n=nlon, axis=-1)[..., ::-1, :]
