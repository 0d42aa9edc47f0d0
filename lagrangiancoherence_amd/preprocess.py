"""Global pre-processing of ``LCS.__call__`` (``isglobal=True``, LCS/LCS.py:105-118): thin host wrappers over
the two C-ABI entry points that do the work on the device (``csrc/preprocess.hip``).

SURVEY.md section 8f rank 2 -- the callers' side of the hot path, not the hot path itself:

* :func:`regrid_common_grid` -- LCS.py:107-114: linear interpolation onto the fixed 0.5 degree grid
  (lats ``linspace(-89.75, 89.75, 360)``, lons ``linspace(-180, 179.5, 721)``), targets outside the
  source range filled with the nearest source node: ``lc_regrid_common_grid`` (tables built on the host from
  the coordinates, one fused lerp-lerp-fill kernel, ``scipy.interpolate.interp1d``'s operation order).

* :func:`spectral_truncate` -- LCS.py:115-118: ``windspharm VectorWind(u, v).truncate(f, truncation=T)``,
  i.e. spherical-harmonic analysis, triangular truncation at T, synthesis, on SPHEREPACK's equally
  spaced grid: ``lc_spectral_truncate`` (zonal DFT for m <= T, one (nlat x nlat) projection per m, inverse
  DFT; float64 kernels, operators built once per (nlat, nlon, T) in C++ and cached on the context).
  PARITY UNPINNED: pyspharm/SPHEREPACK are not installed anywhere this repo can reach; this restates
  the published algorithm (see oracle/preprocess_oracle.py for the statement and what pins it).
  SPHEREPACK works in float32; this computes in float64 and returns the field dtype.
"""
from __future__ import annotations

import numpy as np

__all__ = ["COMMON_LATS", "COMMON_LONS", "regrid_common_grid", "spectral_truncate", "inspect_gridtype",
           "check_regular_global_lat"]

COMMON_LATS = np.linspace(-89.75, 89.75, 180 * 2)        # LCS.py:107
COMMON_LONS = np.linspace(-180, 179.5, 360 * 2 + 1)      # LCS.py:108


def regrid_common_grid(engine, u, lat, lon, lats=COMMON_LATS, lons=COMMON_LONS):
    """u: (nt, nlat, nlon) array or device tensor, lat/lon ascending.  Returns (device tensor, lats, lons);
    float64 like xarray's interp result."""
    lats = np.asarray(lats, dtype=np.float64)
    lons = np.asarray(lons, dtype=np.float64)
    return engine.regrid(u, lat, lon, lats, lons), lats, lons


def inspect_gridtype(lat):
    """windspharm's grid inspection (``VectorWind`` at LCS/LCS.py:116): ``'regular'`` for equally spaced global
    latitudes (an even count half a spacing away from the poles, an odd count on them), ``'gaussian'`` for Gaussian
    latitudes, ``ValueError`` with windspharm's message otherwise.  Any order; ``lc_inspect_gridtype`` does the work."""
    import ctypes as C
    from . import _capi
    lib = _capi.load()
    a = np.ascontiguousarray(np.sort(np.asarray(lat, dtype=np.float64)))
    gt = C.c_int()
    _capi.check(lib.lc_inspect_gridtype(a.ctypes.data_as(C.c_void_p), int(a.size), C.byref(gt)), lib)
    return "gaussian" if gt.value == _capi.LC_GRID_GAUSSIAN else "regular"


def check_regular_global_lat(lat):
    """Kept for callers of the earlier name: raises unless :func:`inspect_gridtype` accepts the latitudes."""
    inspect_gridtype(lat)


def spectral_truncate(engine, f, T=20, gridtype="regular"):
    """f: (..., nlat, nlon) array or device tensor, latitude ASCENDING.  Returns a device tensor of the
    same shape and dtype: the triangular-T truncation on the same grid (``gridtype``: what
    :func:`inspect_gridtype` says of the latitudes)."""
    nlat, nlon = int(f.shape[-2]), int(f.shape[-1])
    if T > nlat - 1 or T > (nlon - 1) // 2:
        raise ValueError(f"truncation {T} too high for a {nlat}x{nlon} grid")
    return engine.spectral_truncate(f, T, gridtype)
