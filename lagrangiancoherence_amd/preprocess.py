"""Global pre-processing of ``LCS.__call__`` (``isglobal=True``, LCS/LCS.py:105-118) on the device.

SURVEY.md section 8f rank 2 -- the callers' side of the hot path, not the hot path itself:

* :func:`regrid_common_grid` -- LCS.py:107-114: linear interpolation onto the fixed 0.5 degree grid
  (lats ``linspace(-89.75, 89.75, 360)``, lons ``linspace(-180, 179.5, 721)``), targets outside the
  source range filled with the nearest source node.  Index/weight tables are built on the host
  (they depend on coordinates only), the gathers and lerps run on the device with the same operation
  order as ``scipy.interpolate.interp1d`` (``slope * (x_new - x_lo) + y_lo``, latitude then longitude).

* :func:`spectral_truncate` -- LCS.py:115-118: ``windspharm VectorWind(u, v).truncate(f, truncation=T)``,
  i.e. spherical-harmonic analysis, triangular truncation at T, synthesis, on SPHEREPACK's equally
  spaced grid.  A dense linear operator: zonal DFT restricted to m <= T, one (nlat x nlat) projection
  per m, inverse DFT -- three plain GEMMs, which go to rocBLAS through ``torch.matmul`` (MFMA-shaped
  library work; no custom kernel).  The operators are built once per (nlat, T) on the host in float64.
  PARITY UNPINNED: pyspharm/SPHEREPACK are not installed anywhere this repo can reach; this restates
  the published algorithm (see oracle/preprocess_oracle.py for the statement and what pins it).
  SPHEREPACK works in float32; this computes in float64 and returns the field dtype.
"""
from __future__ import annotations

import functools

import numpy as np

__all__ = ["COMMON_LATS", "COMMON_LONS", "regrid_common_grid", "spectral_truncate", "check_regular_global_lat"]

COMMON_LATS = np.linspace(-89.75, 89.75, 180 * 2)        # LCS.py:107
COMMON_LONS = np.linspace(-180, 179.5, 360 * 2 + 1)      # LCS.py:108


# ---------------------------------------------------------------------------------------------
# regrid
# ---------------------------------------------------------------------------------------------
def _axis_plan(src, dst):
    """interp1d(kind='linear', bounds_error=False) bookkeeping + reindex(method='nearest') indices."""
    src = np.asarray(src, dtype=np.float64)
    dst = np.asarray(dst, dtype=np.float64)
    idx = np.clip(np.searchsorted(src, dst, side="left"), 1, src.size - 1)   # scipy _call_linear
    lo = idx - 1
    inside = (dst >= src[0]) & (dst <= src[-1])
    # nearest with pandas' tie rule for increasing indexes: left only if strictly closer
    r = np.clip(np.searchsorted(src, dst, side="left"), 0, src.size - 1)
    l = np.clip(r - 1, 0, src.size - 1)
    near = np.where(np.abs(dst - src[l]) < np.abs(src[r] - dst), l, r)
    return lo, dst - src[lo], src[lo + 1] - src[lo], inside, near


def regrid_common_grid(engine, u, lat, lon, lats=COMMON_LATS, lons=COMMON_LONS):
    """u: (nt, nlat, nlon) array or device tensor, lat/lon ascending.  Returns (device tensor, lats, lons);
    float64 like xarray's interp result."""
    torch = engine.torch
    dev = engine.device
    ud = engine.to_device(u, np.float64)
    jlo, ty, dy, in_y, jn = _axis_plan(lat, lats)
    ilo, tx, dx, in_x, i_n = _axis_plan(lon, lons)

    def t(a, dt=torch.float64):
        return torch.as_tensor(np.ascontiguousarray(a), device=dev).to(dt)
    jlo_t, ilo_t = t(jlo, torch.long), t(ilo, torch.long)
    y_lo, y_hi = ud.index_select(1, jlo_t), ud.index_select(1, jlo_t + 1)
    tmp = ((y_hi - y_lo) / t(dy)[None, :, None]) * t(ty)[None, :, None] + y_lo          # latitude first
    x_lo, x_hi = tmp.index_select(2, ilo_t), tmp.index_select(2, ilo_t + 1)
    interp = ((x_hi - x_lo) / t(dx)[None, None, :]) * t(tx)[None, None, :] + x_lo
    near = ud.index_select(1, t(jn, torch.long)).index_select(2, t(i_n, torch.long))    # LCS.py:109
    ok = t(in_y, torch.bool)[None, :, None] & t(in_x, torch.bool)[None, None, :] & ~torch.isnan(interp)
    return torch.where(ok, interp, near), np.asarray(lats, dtype=np.float64), np.asarray(lons, dtype=np.float64)


# ---------------------------------------------------------------------------------------------
# spectral truncation
# ---------------------------------------------------------------------------------------------
def check_regular_global_lat(lat):
    """windspharm's grid inspection for equally spaced latitudes (``windspharm.tools``): an even count
    must sit at +-(90 - delta/2) ..., an odd count at the poles and equator."""
    lat = np.asarray(lat, dtype=np.float64)
    n = lat.size
    d = np.abs(np.diff(lat))
    if not (np.abs(d - d[0]) < 5e-4).all():
        raise ValueError("latitudes are neither equally-spaced or Gaussian (Gaussian grids are not supported here)")
    ref = np.linspace(90, -90, n) if n % 2 else np.linspace(90 - 90.0 / n, -90 + 90.0 / n, n)
    if not np.allclose(np.sort(lat)[::-1], ref, atol=5e-4):
        raise ValueError("Invalid equally-spaced latitudes (they may be non-global)")


def _legendre_normalized(m, nmax, x):
    s = np.sqrt(np.maximum(0.0, 1.0 - x * x))
    pmm = np.full_like(x, np.sqrt(0.5))
    for k in range(1, m + 1):
        pmm = -np.sqrt((2 * k + 1) / (2.0 * k)) * s * pmm
    out = [pmm]
    if nmax > m:
        out.append(np.sqrt(2 * m + 3.0) * x * pmm)
    for n in range(m + 2, nmax + 1):
        a = np.sqrt((4.0 * n * n - 1.0) / (n * n - m * m))
        b = np.sqrt(((n - 1.0) ** 2 - m * m) / (4.0 * (n - 1.0) ** 2 - 1.0))
        out.append(a * (x * out[-1] - b * out[-2]))
    return np.stack(out)


@functools.lru_cache(maxsize=4)
def _operators(nlat: int, nlon: int, T: int):
    """(P[T+1, nlat, nlat], Fc[nlon, T+1], Fs[nlon, T+1], Gc[T+1, nlon], Gs[T+1, nlon]) in float64.

    P[m] = synthesis . analysis for zonal wavenumber m on theta_i = i*pi/(nlat-1), row 0 = north pole.
    Analysis = exact integral of the trigonometric interpolant of the m-th zonal coefficient (cosine
    series for even m, sine series for odd m) against Pbar^m_n sin(theta) (Swarztrauber's Z functions),
    evaluated with Gauss-Legendre nodes in cos(theta) (the integrands are polynomials there)."""
    N = nlat - 1
    theta = np.arange(nlat) * np.pi / N
    i = np.arange(nlat)
    xq, wq = np.polynomial.legendre.leggauss(2 * N)
    tq = np.arccos(xq)
    P = np.zeros((T + 1, nlat, nlat))
    for m in range(T + 1):
        S = _legendre_normalized(m, T, np.cos(theta))
        Pq = _legendre_normalized(m, T, xq)
        if m % 2 == 0:
            k = np.arange(0, N + 1)
            B = (2.0 / N) * np.cos(np.outer(k, i) * np.pi / N)
            B[:, [0, -1]] *= 0.5
            B[[0, -1], :] *= 0.5
            basis = np.cos(np.outer(k, tq))
        else:
            k = np.arange(1, N)
            B = (2.0 / N) * np.sin(np.outer(k, i) * np.pi / N)
            basis = np.sin(np.outer(k, tq))
        P[m] = S.T @ (((Pq * wq[None, :]) @ basis.T) @ B)
    j = np.arange(nlon)
    mm = np.arange(T + 1)
    ang = 2 * np.pi * np.outer(j, mm) / nlon
    Fc, Fs = np.cos(ang), np.sin(ang)                         # forward: X_m = Gc - i Gs
    scale = np.where(mm == 0, 1.0, 2.0)[:, None] / nlon        # inverse real DFT weights
    return P, Fc, Fs, scale * Fc.T, scale * Fs.T


def spectral_truncate(engine, f, T=20):
    """f: (..., nlat, nlon) array or device tensor, latitude ASCENDING.  Returns a device tensor of the
    same shape and dtype: the triangular-T truncation on the same grid."""
    torch = engine.torch
    if not isinstance(f, torch.Tensor):
        f = engine.to_device(f, np.asarray(f).dtype if np.asarray(f).dtype in (np.float32, np.float64) else np.float64)
    nlat, nlon = int(f.shape[-2]), int(f.shape[-1])
    if T > nlat - 1 or T > (nlon - 1) // 2:
        raise ValueError(f"truncation {T} too high for a {nlat}x{nlon} grid")
    P, Fc, Fs, Gc, Gs = (torch.as_tensor(a, device=engine.device) for a in _operators(nlat, nlon, int(T)))
    g = f.to(torch.float64).flip(-2)                          # north -> south, as windspharm orders it
    Xc, Xs = g @ Fc, g @ Fs                                    # (..., nlat, T+1)
    Hc = torch.einsum("mij,...jm->...im", P, Xc)
    Hs = torch.einsum("mij,...jm->...im", P, Xs)
    out = Hc @ Gc + Hs @ Gs
    return out.flip(-2).to(f.dtype)
