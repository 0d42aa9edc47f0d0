"""Hot-path helpers of the reference's ``LCS/tools.py`` on the HIP engine.

* ``xr_map_coordinates``           LCS/tools.py:11-48   (one interpolation pass)
* ``derivative_spherical_coords``  LCS/tools.py:248-267
* ``fourth_order_derivative``      LCS/tools.py:190-245 (plain arrays in the reference)

The other functions of that module (ridge extraction, IDW regridding, latlonsel) are
not on the advect -> gradient -> sigma path.
"""
from __future__ import annotations

import numpy as np

from .dropin import _coord, _make, _to_np, get_engine
from .engine import common_dtype

__all__ = ["xr_map_coordinates", "fourth_order_derivative", "derivative_spherical_coords"]


def xr_map_coordinates(da, new_x, new_y, isglobal=True, order=1):
    """Interpolate a 2-D ``(latitude, longitude)`` field at positions ``new_x, new_y``
    (degrees, shaped like the field).  Signature of LCS/tools.py:11.

    ``isglobal=False`` is unreachable from the hot path and broken in the reference
    (undefined name at tools.py:47); it raises here.
    """
    if not isglobal:
        raise NotImplementedError("isglobal=False references an undefined name in the reference (LCS/tools.py:42-47)")
    f = np.asarray(da.transpose("latitude", "longitude").values)
    lat, lon = _coord(da, "latitude"), _coord(da, "longitude")
    px = np.asarray(getattr(new_x, "values", new_x))
    py = np.asarray(getattr(new_y, "values", new_y))
    eng = get_engine()
    field = eng.prepare_field(f[None], f[None], lat, lon, order, dtype=common_dtype(f, lat, lon, px, py))
    out, _ = eng.sample(field, px, py, level=0, interp_order=order)
    return _make(da, _to_np(out).astype(f.dtype, copy=False), ("latitude", "longitude"),
                 {"latitude": lat, "longitude": lon}, getattr(da, "name", None))


def fourth_order_derivative(arr, dim=0, isglobal=True):
    """4th-order 5-point difference in index space on a 2-D array, result in ``arr.dtype``.
    Signature of LCS/tools.py:191 (numba in the reference)."""
    if not isglobal and dim == 1:
        raise NotImplementedError("only the isglobal=True (cyclic longitude) branch is on the hot path")
    eng = get_engine()
    a = np.ascontiguousarray(arr)
    return _to_np(eng.index_derivative(a, dim))


def derivative_spherical_coords(da, dim=0, isglobal=True):
    """``fourth_order_derivative(values.astype('float32'))`` divided by the metric dx / dy.
    Signature of LCS/tools.py:248."""
    if dim not in (0, 1):
        raise ValueError('Dim must be either 0 or 1.')
    lat, lon = _coord(da, "latitude"), _coord(da, "longitude")
    ilat, ilon = np.argsort(lat, kind="stable"), np.argsort(lon, kind="stable")
    vals = np.asarray(da.transpose("latitude", "longitude").values)[ilat][:, ilon]
    lat, lon = lat[ilat], lon[ilon]
    EARTH_RADIUS = 6371000
    y = lat * np.pi / 180
    dx = (np.pi / 180) * (lon[1] - lon[0]) * EARTH_RADIUS * np.cos(y)
    dy = (np.pi / 180) * (lat[1] - lat[0]) * EARTH_RADIUS
    deriv = fourth_order_derivative(vals.astype('float32'), dim=dim, isglobal=isglobal)
    deriv = deriv / dy if dim == 0 else deriv / dx[:, None]
    return _make(da, deriv, ("latitude", "longitude"), {"latitude": lat, "longitude": lon}, getattr(da, "name", None))
