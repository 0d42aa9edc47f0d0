"""Drop-in surface: the reference's labelled-array API over the HIP engine.

Same names, positional order, defaults and return conventions as

    LagrangianCoherence.LCS.LCS.LCS                       LCS/LCS.py:19-168
    LagrangianCoherence.LCS.LCS.flowmap_gradient          LCS/LCS.py:171-225
    LagrangianCoherence.LCS.trajectory.parcel_propagation LCS/trajectory.py:8-144

xarray in, xarray out.  The adapter touches only ``dims``, ``values``, coordinates and ``name`` and returns
results in the class of its inputs, so where xarray is not installed (this image) the same code runs on any
duck-typed labelled array (the tests use the stand-in of tests/labelled.py; nothing here imports it).  All arithmetic happens in the
HIP library; this file is argument handling only.

Deliberate differences from the reference (all outside the arithmetic):
  * the unconditional ``print('!'*100)`` and ``print('using s = ...')`` (LCS.py:74,126)
    are not reproduced; ``verbose`` prints the same progress lines;
  * ``isglobal=True`` runs the 0.5 degree regrid and the T20 spectral truncation on the
    device (``preprocess.py``); the truncation restates windspharm/SPHEREPACK's published
    algorithm in float64 and is NOT pinned against pyspharm (not installable here);
  * ``cyclic_xboundary=False`` (the default of both call forms) reproduces the reference's clamp as it is
    written -- outer-product assignment over offending rows x columns (LCS/trajectory.py:96-97):
    the fused kernel runs first and, only if a parcel really left the longitude range, the engine re-runs
    sub-step by sub-step with that rule (``lc_advect`` mode ``LC_X_CLAMP_REFERENCE_OUTER``);
  * mixed float32/float64 inputs are computed in float64 (engine.common_dtype);
  * float64 calls of up to 2^18 seeds (the example's 89 x 180 grid, the 360 x 721 common grid of ``isglobal=True``)
    follow numpy / scipy's operation order (LCS/trajectory.py:86-87,110-112): ~1e-13 degrees from the reference.
    Larger float64 calls take the fused-level form, which differs by rounding only (<= 1e-9 degrees, or the flow's own
    response to a 1e-12 degree seed shift where that is larger); ``get_engine().set_f64_fidelity('exact' | 'fast' |
    'auto')`` chooses explicitly (INTEGRATION.md "Behavioural notes").
"""
from __future__ import annotations

import numpy as np

from .engine import Engine, common_dtype

__all__ = ["LCS", "parcel_propagation", "flowmap_gradient", "get_engine"]

_ENGINE = None


def get_engine() -> Engine:
    """Process-wide engine on the current torch device (created on first use)."""
    global _ENGINE
    if _ENGINE is None:
        _ENGINE = Engine()
    return _ENGINE


# ---------------------------------------------------------------------------
# labelled-array adapter
# ---------------------------------------------------------------------------
def _is_xarray(obj) -> bool:
    return type(obj).__module__.split(".")[0] == "xarray"


def _make(like, data, dims, coords, name=None):
    if _is_xarray(like):
        import xarray as xr
        return xr.DataArray(data, dims=dims, coords=coords, name=name)
    # any other labelled-array class: results come back in the class they came in (constructor (data, dims, coords,
    # name); the tests' xarray-free stand-in, tests/labelled.py, is one)
    return type(like)(data, dims, coords, name)


def _resample_linear(da, dim, freq):
    """``da.resample({dim: freq}).interpolate('linear')`` (LCS/LCS.py:89-90) for a labelled array that is not an
    xarray object: the new time axis is pandas' resampling index of the old one (what xarray's grouper calls
    ``full_index``), the values go through ``scipy.interpolate.interp1d(kind='linear', bounds_error=False)`` on the
    times as float64 nanoseconds since the first one -- the two kernels xarray itself delegates to."""
    import pandas as pd
    from scipy.interpolate import interp1d
    t = pd.DatetimeIndex(np.asarray(da.coords[dim]))
    full = pd.Series(np.arange(t.size), index=t).resample(freq).asfreq().index
    t0 = t.values.astype("datetime64[ns]").min()
    x = (t.values.astype("datetime64[ns]") - t0).astype("int64").astype(np.float64)
    xn = (full.values.astype("datetime64[ns]") - t0).astype("int64").astype(np.float64)
    ax = list(da.dims).index(dim)
    vals = interp1d(x, np.asarray(da.values), kind="linear", axis=ax, bounds_error=False, assume_sorted=True)(xn)
    coords = dict(da.coords)
    coords[dim] = full.values
    return type(da)(vals, da.dims, coords, getattr(da, "name", None))


def _coord(da, dim):
    return np.asarray(da[dim].values)


def _sorted_tll(da, timedim):
    """values as (time, latitude, longitude) with lat/lon ascending, plus the coordinates.

    The reference sorts with ``sortby`` (trajectory.py:49-52, LCS.py:101-104)."""
    vals = np.asarray(da.transpose(timedim, "latitude", "longitude").values)
    lat, lon, time = _coord(da, "latitude"), _coord(da, "longitude"), _coord(da, timedim)
    ilat, ilon = np.argsort(lat, kind="stable"), np.argsort(lon, kind="stable")
    if not np.array_equal(ilat, np.arange(lat.size)):
        vals, lat = vals[:, ilat, :], lat[ilat]
    if not np.array_equal(ilon, np.arange(lon.size)):
        vals, lon = vals[:, :, ilon], lon[ilon]
    return vals, time, lat, lon


def _to_np(t):
    return get_engine().to_host(t)


# ---------------------------------------------------------------------------
# trajectory.parcel_propagation
# ---------------------------------------------------------------------------
def parcel_propagation(U, V, timestep=1, propdim="time", verbose=True, return_traj=False, SETTLS_order=0,
                       copy=False, interp_order=3, cyclic_xboundary=False):
    """Lagrangian 2-time-level advection.  Signature of LCS/trajectory.py:8-18.

    Returns ``(positions_x, positions_y)``: 2-D ``(latitude, longitude)`` arrays with a
    scalar ``propdim`` coordinate equal to the last entry of the (possibly reversed)
    time list (trajectory.py:141-142), or with ``return_traj`` 3-D
    ``(propdim, latitude, longitude)`` arrays whose entry 0 is the seed grid
    (trajectory.py:73-74,138-139).
    """
    verboseprint = print if verbose else (lambda *a, **k: None)
    u, time, lat, lon = _sorted_tll(U, propdim)
    v, _, _, _ = _sorted_tll(V, propdim)
    times = time.tolist()                                   # trajectory.py:58
    if timestep < 0:
        times.reverse()                                     # labels only (Q6)
    eng = get_engine()
    verboseprint(f"Propagating {len(times) - 1} time levels on {eng.device}")
    # pack + advect in one call (large float64 order-3 series: the pack of chunk k+1 overlaps the advect of chunk k)
    res = eng.pack_and_advect(u, v, lat, lon, lat, lon, timestep, SETTLS_order=SETTLS_order, interp_order=interp_order,
                              cyclic_xboundary=cyclic_xboundary, return_traj=return_traj,
                              fuse_levels=eng.f64_fuse_levels(common_dtype(u, v, lat, lon), lat.size * lon.size))[1:]
    coords2d = {"latitude": lat, "longitude": lon}
    if return_traj:
        assert type(times[0]).__name__ != "Datetime360Day", \
            'Cannot return trajectories with time cooridnates cftime.Datetime360Day.'   # trajectory.py:129-130
        import pandas as pd
        tindex = pd.Index(pd.to_datetime(times), name=propdim)                          # trajectory.py:138
        tcoord = tindex if _is_xarray(U) else np.asarray(tindex.values)
        dims = (propdim, "latitude", "longitude")
        px = _make(U, _to_np(res[2]), dims, {propdim: tcoord, **coords2d}, getattr(U, "name", None))
        py = _make(U, _to_np(res[3]), dims, {propdim: tcoord, **coords2d}, getattr(U, "name", None))
        return px, py
    dims = ("latitude", "longitude")
    px = _make(U, _to_np(res[0]), dims, {**coords2d, propdim: times[-1]}, getattr(U, "name", None))
    py = _make(U, _to_np(res[1]), dims, {**coords2d, propdim: times[-1]}, getattr(U, "name", None))
    return px, py


# ---------------------------------------------------------------------------
# LCS.flowmap_gradient
# ---------------------------------------------------------------------------
def flowmap_gradient(x_departure, y_departure, sigma=None):
    """The 9-component deformation tensor, dims ``(derivatives, latitude, longitude)``.
    Signature of LCS/LCS.py:171."""
    lat, lon = _coord(x_departure, "latitude"), _coord(x_departure, "longitude")
    x = np.asarray(x_departure.transpose("latitude", "longitude").values)
    y = np.asarray(y_departure.transpose("latitude", "longitude").values)
    ilat, ilon = np.argsort(lat, kind="stable"), np.argsort(lon, kind="stable")   # tools.py:250-251
    x, y, lat, lon = x[ilat][:, ilon], y[ilat][:, ilon], lat[ilat], lon[ilon]
    dtype = common_dtype(x, y, lat, lon)
    lat_t, lon_t = lat.astype(dtype), lon.astype(dtype)
    eng = get_engine()
    xd, yd = eng.to_device(x, dtype), eng.to_device(y, dtype)
    if isinstance(sigma, (float, int)) and sigma > 1e-15:                          # LCS.py:187-190
        xd, yd = eng.gaussian_filter(xd, sigma), eng.gaussian_filter(yd, sigma)
    dt = eng.flowmap_gradient(xd, yd, lat_t, float(lat_t[1] - lat_t[0]), float(lon_t[1] - lon_t[0]))
    names = ["dxdx", "dxdy", "dydx", "dydy", "dzdx", "dzdy", "dxdr", "dydr", "dzdr"]   # LCS.py:210-220
    return _make(x_departure, _to_np(dt), ("derivatives", "latitude", "longitude"),
                 {"derivatives": np.array(names), "latitude": lat, "longitude": lon})


# ---------------------------------------------------------------------------
# LCS.LCS
# ---------------------------------------------------------------------------
def _crop_strict(lat, lon, sub):
    """Row/column masks of ``latlonsel`` with strict inequalities (LCS/tools.py:158-187)."""
    def bounds(s):
        if isinstance(s, slice):
            return s.start, s.stop
        return s[0], s[-1]
    la = bounds(sub.get("latitude", sub.get("lat")))
    lo = bounds(sub.get("longitude", sub.get("lon")))
    mlat = np.ones(lat.shape, bool) if la[0] is None else (lat > la[0])
    mlat &= np.ones(lat.shape, bool) if la[1] is None else (lat < la[1])
    mlon = np.ones(lon.shape, bool) if lo[0] is None else (lon > lo[0])
    mlon &= np.ones(lon.shape, bool) if lo[1] is None else (lon < lo[1])
    return mlat, mlon


class LCS:
    """API to compute the Finite-time Lyapunov exponent in 2D wind fields (LCS/LCS.py:19-46).

    Returns sigma_max, the largest singular value of the reference's deformation
    tensor; FTLE is the caller's ``log(sigma)/2`` (examples/ideal_vortex.py:282,288).
    """
    earth_r = 6371000  # metres

    def __init__(self, timestep: float = 1, timedim='time', SETTLS_order=0, subdomain=None, return_dpts=False,
                 gauss_sigma=None):
        self.timestep = timestep
        self.SETTLS_order = SETTLS_order
        self.timedim = timedim
        self.subdomain = subdomain
        self.gauss_sigma = gauss_sigma
        self.return_dpts = return_dpts

    def __call__(self, ds=None, u=None, v=None, verbose=True, s=None, resample=None, s_is_error=False,
                 isglobal=False, return_traj=False, interp_to_common_grid=True, traj_interp_order=3, truncation=20):
        verboseprint = print if verbose else (lambda *a, **k: None)
        timestep = self.timestep
        timedim = self.timedim
        self.verbose = verbose

        if isinstance(ds, str):                                            # LCS.py:84-87
            import xarray as xr
            ds = xr.open_dataset(ds)
        if ds is not None and not isinstance(ds, str):                     # LCS.py:81-83
            u = ds.u.copy()
            v = ds.v.copy()
        if isinstance(resample, str):                                      # LCS.py:88-91
            if _is_xarray(u):
                u = u.resample({timedim: resample}).interpolate('linear')
                v = v.resample({timedim: resample}).interpolate('linear')
            else:
                u = _resample_linear(u, timedim, resample)
                v = _resample_linear(v, timedim, resample)
            timestep = np.sign(timestep) * (u[timedim].values[1] - u[timedim].values[0]) \
                .astype('timedelta64[s]').astype('float')
        assert set(u.dims) == set(v.dims), "u and v dims are different"                     # LCS.py:95
        assert set(u.dims) == {'latitude', 'longitude', timedim}, \
            'array dims should be latitude and longitude only'                             # LCS.py:96

        uu, time, lat, lon = _sorted_tll(u, timedim)                       # LCS.py:101-104
        vv, _, _, _ = _sorted_tll(v, timedim)
        eng = get_engine()
        if isglobal:
            from . import preprocess
            if interp_to_common_grid:                                      # LCS.py:106-114
                uu, lat_new, lon_new = preprocess.regrid_common_grid(eng, uu, lat, lon)
                vv, _, _ = preprocess.regrid_common_grid(eng, vv, lat, lon)
                lat, lon = lat_new, lon_new
            if truncation is not None:                                     # LCS.py:115-118
                gridtype = preprocess.inspect_gridtype(lat)              # windspharm's: equally spaced global, or Gaussian
                uu = preprocess.spectral_truncate(eng, uu, truncation, gridtype)
                vv = preprocess.spectral_truncate(eng, vv, truncation, gridtype)
            cyclic_xboundary = True                                        # LCS.py:119-120
            self.subdomain = None
        else:
            cyclic_xboundary = False

        verboseprint("*---- Parcel propagation ----*")
        dtype = common_dtype(uu, vv, lat, lon)
        lat_t, lon_t = lat.astype(dtype), lon.astype(dtype)
        res = eng.lcs_wind(uu, vv, lat, lon, lat_t, lon_t, timestep, SETTLS_order=self.SETTLS_order,
                           interp_order=traj_interp_order, cyclic_xboundary=cyclic_xboundary,
                           fuse_levels=eng.f64_fuse_levels(dtype, lat.size * lon.size),
                           gauss_sigma=self.gauss_sigma, return_traj=return_traj)          # LCS.py:129-154
        verboseprint("*---- Done eigenvalues ----*")

        sig = _to_np(res["sigma"])
        slat, slon = lat, lon
        if isinstance(self.subdomain, dict):                               # LCS.py:143-144
            mlat, mlon = _crop_strict(lat, lon, self.subdomain)
            sig, slat, slon = sig[mlat][:, mlon], lat[mlat], lon[mlon]
        timestamp = time[-1] if np.sign(timestep) == 1 else time[0]        # LCS.py:158
        eigenvalues = _make(u, sig[None], (timedim, "latitude", "longitude"),
                            {timedim: np.asarray([timestamp]), "latitude": slat, "longitude": slon},
                            getattr(u, "name", None))                      # LCS.py:159-160

        times = time.tolist()
        if timestep < 0:
            times.reverse()
        c2 = {"latitude": lat, "longitude": lon}

        def dep(k):
            return _make(u, _to_np(res[k]), ("latitude", "longitude"), {**c2, timedim: times[-1]})

        def traj(k):
            import pandas as pd
            tindex = pd.Index(pd.to_datetime(times), name=timedim)
            tcoord = tindex if _is_xarray(u) else np.asarray(tindex.values)
            return _make(u, _to_np(res[k]), (timedim, "latitude", "longitude"), {timedim: tcoord, **c2})

        if self.return_dpts and return_traj:                               # LCS.py:161-168
            return eigenvalues, dep("x_dep"), dep("y_dep"), traj("traj_x"), traj("traj_y")
        elif self.return_dpts:
            return eigenvalues, dep("x_dep"), dep("y_dep")
        elif return_traj:
            return eigenvalues, traj("traj_x"), traj("traj_y")
        return eigenvalues
