"""TEST SUPPORT -- a very small labelled-array container for machines without xarray.

The drop-in surface (`lagrangiancoherence_amd/dropin.py`) is xarray-in / xarray-out like the reference.  This image
has no xarray, so the tests exercise the same adapter code with these stand-ins: they carry exactly what the adapter
reads (``dims``, ``values``, per-dimension coordinates, scalar coordinates, ``name``) and nothing else.  The adapter
knows nothing of this module: it returns results in the class of what it was given (``type(input)(data, dims, coords,
name)``), or real ``xarray.DataArray`` objects for xarray inputs.  Not part of the product package.
"""
from __future__ import annotations

import numpy as np

__all__ = ["DataArray", "Dataset"]


class DataArray:
    def __init__(self, data, dims, coords=None, name=None):
        self.values = np.asarray(data)
        self.dims = tuple(dims)
        if self.values.ndim != len(self.dims):
            raise ValueError(f"{self.values.ndim}-d data with dims {self.dims}")
        self.coords = {}
        for k, v in (coords or {}).items():
            v = np.asarray(v)
            if k in self.dims and v.shape != (self.values.shape[self.dims.index(k)],):
                raise ValueError(f"coordinate {k!r} has shape {v.shape}")
            self.coords[k] = v
        self.name = name

    # -- what the adapter and user code touch ---------------------------------
    @property
    def shape(self):
        return self.values.shape

    @property
    def dtype(self):
        return self.values.dtype

    def __getitem__(self, key):
        if isinstance(key, str):
            c = self.coords[key]
            return DataArray(c, (key,) if c.ndim else (), {key: c} if c.ndim else {}, name=key)
        raise TypeError("only coordinate lookup by name is supported")

    def __getattr__(self, key):
        coords = self.__dict__.get("coords", {})
        if key in coords:
            return self[key]
        raise AttributeError(key)

    def copy(self, data=None):
        return DataArray(self.values.copy() if data is None else np.asarray(data), self.dims,
                         {k: v.copy() for k, v in self.coords.items()}, self.name)

    def transpose(self, *dims):
        order = [self.dims.index(d) for d in dims]
        return DataArray(self.values.transpose(order), dims, self.coords, self.name)

    def sortby(self, dim):
        idx = np.argsort(self.coords[dim], kind="stable")
        coords = dict(self.coords)
        coords[dim] = coords[dim][idx]
        return DataArray(np.take(self.values, idx, axis=self.dims.index(dim)), self.dims, coords, self.name)

    def isel(self, indexers=None, **kw):
        indexers = {**(indexers or {}), **kw}
        data, dims, coords = self.values, list(self.dims), dict(self.coords)
        for d, i in indexers.items():
            ax = dims.index(d)
            scalar = not isinstance(i, slice) and np.ndim(i) == 0
            data = data[(slice(None),) * ax + (i,)]
            coords[d] = coords[d][i]
            if scalar:
                dims.pop(ax)
        return DataArray(data, dims, coords, self.name)

    def __array__(self, dtype=None, copy=None):
        return np.asarray(self.values, dtype=dtype)

    def __repr__(self):
        return f"<labelled.DataArray {self.name!r} dims={self.dims} shape={self.shape} dtype={self.dtype}>"


def resample_linear(da, dim, freq):
    """``da.resample({dim: freq}).interpolate('linear')`` for a stand-in: the drop-in's own helper (dropin._resample_linear)."""
    from lagrangiancoherence_amd.dropin import _resample_linear
    return _resample_linear(da, dim, freq)


class Dataset:
    """Just enough for ``ds.u`` / ``ds.v`` / ``ds.copy()`` (LCS/LCS.py:81-83)."""

    def __init__(self, data_vars):
        self.data_vars = dict(data_vars)

    def __getattr__(self, key):
        dv = self.__dict__.get("data_vars", {})
        if key in dv:
            return dv[key]
        raise AttributeError(key)

    def __getitem__(self, key):
        return self.data_vars[key]

    def copy(self):
        return Dataset({k: v.copy() for k, v in self.data_vars.items()})
