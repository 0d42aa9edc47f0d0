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
           "truncation_operators_gaussian", "gaussian_latitudes", "spectral_truncate", "spectral_truncate_quadrature",
           "spectral_truncate_lstsq"]

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


def gaussian_latitudes(nlat):
    """Gaussian latitudes (degrees, ASCENDING) and weights of an nlat-row grid: asin of the Gauss-Legendre nodes."""
    x, w = np.polynomial.legendre.leggauss(nlat)
    return np.degrees(np.arcsin(x)), w


def truncation_operators_gaussian(nlat, T):
    """windspharm gridtype 'gaussian' (SPHEREPACK shags / shsgs): analysis by Gauss-Legendre quadrature on the grid's
    own nodes, a^m_n = sum_j w_j Pbar^m_n(x_j) g_m(x_j), synthesis on the same nodes.  Rows north -> south."""
    x, w = np.polynomial.legendre.leggauss(nlat)
    x, w = x[::-1], w[::-1]
    ops = []
    for m in range(T + 1):
        P = legendre_normalized(m, T, x)                            # (n, i)
        ops.append(P.T @ (P * w[None, :]))
    return ops


def spectral_truncate(f, T=20, gridtype="regular"):
    """f: (..., nlat, nlon) with latitude ASCENDING (south -> north), as everywhere in this repo.
    Returns the T-truncated field on the same grid.  windspharm reorders to north -> south internally
    and gives the result back on the field's own coordinates.  ``gridtype``: 'regular' (SPHEREPACK's equally
    spaced grid) or 'gaussian' -- what windspharm's inspection of the latitudes decides."""
    f = np.asarray(f, dtype=np.float64)
    nlat, nlon = f.shape[-2:]
    g = f[..., ::-1, :]                                             # north -> south
    F = np.fft.rfft(g, axis=-1)                                     # zonal Fourier coefficients
    ops = truncation_operators_gaussian(nlat, T) if gridtype == "gaussian" else truncation_operators(nlat, T)
    out = np.zeros_like(F)
    for m in range(min(T, F.shape[-1] - 1) + 1):
        out[..., :, m] = np.einsum("ij,...j->...i", ops[m], F[..., :, m])
    return np.fft.irfft(out, n=nlon, axis=-1)[..., ::-1, :]


# ------------------------------------------------------------------------------------------------
# Two INDEPENDENT formulations of the same truncation, written from other sources than the operator above.  They do
# not pin it against pyspharm (nothing can, here) -- they bound how far a correct equally-spaced analysis can be from
# it: all three agree to rounding on band-limited fields (degree <= nlat - 1 - T), and differ on anything else only by
# how each aliases the content the grid cannot represent.  tests/test_preprocess.py reports that difference on the
# regridded config-1 wind and on a non-band-limited field as the aliasing uncertainty of row f2.
# ------------------------------------------------------------------------------------------------
def _clenshaw_curtis_weights(nlat):
    """Weights w_i with sum_i w_i f(theta_i) = integral_0^pi f(theta) sin(theta) dtheta exactly for every cosine
    polynomial f of degree <= N = nlat - 1 on theta_i = i pi / N (Clenshaw & Curtis 1960; the quadrature behind the
    Driscoll-Healy / SHTns regular-grid transforms)."""
    N = nlat - 1
    i = np.arange(nlat)
    w = np.zeros(nlat)
    for k in range(0, N + 1, 2):                                    # integral cos(k theta) sin(theta) = 2 / (1 - k^2), k even
        c = (2.0 / N) * np.cos(k * i * np.pi / N)
        c[[0, -1]] *= 0.5
        w += (0.5 if k in (0, N) else 1.0) * c * 2.0 / (1.0 - k * k)
    return w


def spectral_truncate_quadrature(f, T=20):
    """Formulation 2 (discrete quadrature): a^m_n = sum_i w_i Pbar^m_n(cos theta_i) g_m(theta_i) with the
    Clenshaw-Curtis weights of the equally spaced grid -- exact while g_m Pbar^m_n is a cosine polynomial of degree
    <= nlat - 1, i.e. for fields band-limited to degree nlat - 1 - T; beyond that it aliases differently from the
    exact integral of the interpolant that SPHEREPACK's Z functions (and ``spectral_truncate``) evaluate."""
    f = np.asarray(f, dtype=np.float64)
    nlat, nlon = f.shape[-2:]
    theta = np.arange(nlat) * np.pi / (nlat - 1)
    w = _clenshaw_curtis_weights(nlat)
    F = np.fft.rfft(f[..., ::-1, :], axis=-1)
    out = np.zeros_like(F)
    for m in range(min(T, F.shape[-1] - 1) + 1):
        P = legendre_normalized(m, T, np.cos(theta))               # (n, i)
        a = np.einsum("ni,...i->...n", P * w[None, :], F[..., :, m])
        out[..., :, m] = np.einsum("ni,...n->...i", P, a)
    return np.fft.irfft(out, n=nlon, axis=-1)[..., ::-1, :]


def spectral_truncate_lstsq(f, T=20):
    """Formulation 3 (least squares): per zonal wavenumber m the coefficients a^m_n, n = m..T, that minimise the
    area-weighted misfit sum_i sin(theta_i) |g_m(theta_i) - sum_n a_n Pbar^m_n(cos theta_i)|^2 on the grid rows (the
    poles carry no area and no weight) -- the explicit fit of Y^m_n, n <= T, solved with numpy.linalg.lstsq."""
    f = np.asarray(f, dtype=np.float64)
    nlat, nlon = f.shape[-2:]
    theta = np.arange(nlat) * np.pi / (nlat - 1)
    sw = np.sqrt(np.sin(theta))
    F = np.fft.rfft(f[..., ::-1, :], axis=-1)
    out = np.zeros_like(F)
    lead = F.shape[:-2]
    for m in range(min(T, F.shape[-1] - 1) + 1):
        P = legendre_normalized(m, T, np.cos(theta))               # (n, i)
        A = (P * sw[None, :]).T                                    # (i, n)
        rhs = (F[..., :, m] * sw).reshape(-1, nlat).T              # (i, batch)
        coef = np.linalg.lstsq(A, rhs, rcond=None)[0]              # (n, batch)
        out[..., :, m] = (P.T @ coef).T.reshape(lead + (nlat,))
    return np.fft.irfft(out, n=nlon, axis=-1)[..., ::-1, :]
