"""Hot-path helpers of the reference's ``LCS/tools.py`` on the HIP engine.

* ``xr_map_coordinates``           LCS/tools.py:11-48   (one interpolation pass)
* ``derivative_spherical_coords``  LCS/tools.py:248-267
* ``fourth_order_derivative``      LCS/tools.py:190-245 (plain arrays in the reference)

* ``find_ridges_spherical_hessian`` LCS/tools.py:52-155 (the consumer of the FTLE field)

The remaining functions of that module (IDW regridding, harvesine, latlonsel) have no caller on the
path.
"""
from __future__ import annotations

import numpy as np

from .dropin import _coord, _make, _to_np, get_engine
from .engine import common_dtype

__all__ = ["xr_map_coordinates", "fourth_order_derivative", "derivative_spherical_coords",
           "find_ridges_spherical_hessian"]


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
    eng = get_engine()
    a = np.ascontiguousarray(arr)
    return _to_np(eng.index_derivative(a, dim, isglobal))


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


def find_ridges_spherical_hessian(da, sigma=.5, scheme='first_order', tolerance_threshold=0.0005e-3,
                                  return_eigvectors=False, isglobal=True):
    """Hessian ridge filter in spherical coordinates.  Signature of LCS/tools.py:52-54.

    Returns ``(ridges, eigmin)`` with the input's dimension order: ``ridges`` is 1 where the reference's
    gradient/eigenvector product is within ``tolerance_threshold`` and the Hessian eigenvalue of largest
    magnitude is negative, else 0.  Gaussian smoothing, the five float32-cast 4th-order derivatives and
    the per-point ``numpy.linalg.eig`` (a Python loop in the reference) all run on the device.
    ``isglobal=False``: the longitude stencil is one-sided on the two first / last columns instead of cyclic
    (tools.py:229-244); latitude is the same either way.

    ``return_eigvectors=True`` returns the reference's six-tuple (tools.py:140-147):
    ``(ridges, eigmin, raw product, eigvectors, gradient, angle)`` -- ``eigvectors`` has a leading
    ``eigvectors`` dimension labelled ``['d2dadxdy', 'd2dadydx']`` (the reference builds it from those two
    Hessian planes, tools.py:123-124) and is zeroed where ``eigmin >= 0`` (tools.py:133); ``gradient`` has a
    leading ``elements`` dimension ``['ddadx', 'ddady']``; ``angle`` is ``180/pi * arctan(e0/e1)`` of the
    unmasked vector (tools.py:125).
    """
    dims = tuple(da.dims)
    lat, lon = _coord(da, "latitude"), _coord(da, "longitude")
    ilat, ilon = np.argsort(lat, kind="stable"), np.argsort(lon, kind="stable")       # tools.py:70-71
    vals = np.asarray(da.transpose("latitude", "longitude").values, dtype=np.float64)[ilat][:, ilon]
    lat, lon = lat[ilat], lon[ilon]
    eng = get_engine()
    torch = eng.torch
    a = eng.to_device(vals, np.float64)
    if isinstance(sigma, (float, int)) and sigma > 1e-15:                              # tools.py:74-75
        a = eng.gaussian_filter(a, sigma)
    y = lat * np.pi / 180
    dx = eng.to_device((np.pi / 180) * (lon[1] - lon[0]) * 6371000 * np.cos(y), np.float64)[:, None]   # tools.py:255
    dy = (np.pi / 180) * (lat[1] - lat[0]) * 6371000                                   # tools.py:256

    def D(f, dim):   # derivative_spherical_coords: float32 cast, index stencil, metric (tools.py:258-264)
        d = eng.index_derivative(f.to(torch.float32), dim, isglobal).to(torch.float64)   # isglobal: tools.py:77-81
        return d / dy if dim == 0 else d / dx
    ddadx, ddady = D(a, 1), D(a, 0)                                                    # tools.py:77-78
    d2x2, d2y2, dxdy = D(ddadx, 1), D(ddady, 0), D(ddadx, 0)                           # tools.py:79-81
    mask, eigmin, dt, vec = eng.ridge_classify(d2x2, dxdy, d2y2, ddadx, ddady, tolerance_threshold,
                                               return_eigvec=True)

    def out(t, lead=None, labels=None):
        arr = _to_np(t)
        coords = {"latitude": lat, "longitude": lon}
        if lead is None:
            o = _make(da, arr, ("latitude", "longitude"), coords, getattr(da, "name", None))
            return o.transpose(*dims)                                                  # tools.py:150
        coords[lead] = np.asarray(labels)
        o = _make(da, arr, (lead, "latitude", "longitude"), coords, getattr(da, "name", None))
        return o.transpose(lead, *dims)                                                # tools.py:141-147
    if not return_eigvectors:
        return out(mask), out(eigmin)
    angle = (180 / np.pi) * torch.atan(vec[0] / vec[1])                                # tools.py:125
    vec_masked = torch.where((eigmin < 0)[None], vec, torch.zeros_like(vec))           # tools.py:133
    return (out(mask), out(eigmin), out(dt), out(vec_masked, "eigvectors", ["d2dadxdy", "d2dadydx"]),
            out(torch.stack([ddadx, ddady]), "elements", ["ddadx", "ddady"]), out(angle))
