"""The drop-in surface on REAL xarray objects (examples/ideal_vortex.py:262-288).  xarray is not installed in the
build image nor on the GPU boxes of this pool, so everywhere else the adapter is exercised with
`tests/labelled.py` stand-ins; this file runs wherever `import xarray` works and is skipped
otherwise.  Every call is made twice -- xarray objects and labelled stand-ins over the same numbers -- and must
return the same values with xarray's own types, dims and coordinates."""
import numpy as np
import pytest

xr = pytest.importorskip("xarray")
pd = pytest.importorskip("pandas")

from lagrangiancoherence_amd import flows  # noqa: E402
from tests import labelled  # noqa: E402

pytestmark = pytest.mark.gpu


def _datasets():
    u, v, lat, lon = flows.config1()
    times = pd.date_range('2000-01-01', periods=u.shape[0], freq='6h')
    dims = ['latitude', 'longitude', 'time']                      # the example's order (ideal_vortex.py:124,203)
    coords = {'latitude': lat, 'longitude': lon, 'time': times}
    ds = xr.Dataset({'u': xr.DataArray(u.transpose(1, 2, 0), dims=dims, coords=coords, name='u'),
                     'v': xr.DataArray(v.transpose(1, 2, 0), dims=dims, coords=coords, name='v')})
    lc = {k: (c.values if hasattr(c, "values") else c) for k, c in coords.items()}
    lds = labelled.Dataset({'u': labelled.DataArray(u.transpose(1, 2, 0), dims, lc, name='u'),
                            'v': labelled.DataArray(v.transpose(1, 2, 0), dims, lc, name='v')})
    return ds, lds, times, lat, lon


def test_example_call_forms_on_xarray():
    from LagrangianCoherence.LCS import trajectory
    from LagrangianCoherence.LCS.LCS import LCS
    ds, lds, times, lat, lon = _datasets()
    kw = dict(timestep=-6 * 3600, propdim='time', SETTLS_order=4, copy=True, return_traj=True, cyclic_xboundary=True,
              verbose=False)
    x, y = trajectory.parcel_propagation(ds.u, ds.v, **kw)                         # ideal_vortex.py:262-270
    xl, yl = trajectory.parcel_propagation(lds.u, lds.v, **kw)
    assert isinstance(x, xr.DataArray) and x.dims == ('time', 'latitude', 'longitude')
    assert isinstance(x.indexes['time'], pd.DatetimeIndex) and x['time'].values[0] == times.values[-1]
    assert np.array_equal(x.values, xl.values) and np.array_equal(y.values, yl.values)
    assert np.array_equal(x.isel(time=0).values, np.meshgrid(lon, lat)[0])
    x2, y2 = trajectory.parcel_propagation(ds.u, ds.v, timestep=6 * 3600, SETTLS_order=2, cyclic_xboundary=True,
                                           verbose=False)                          # ideal_vortex.py:272-279
    assert x2.dims == ('latitude', 'longitude') and x2['time'].ndim == 0            # scalar time coordinate
    acs = LCS(timestep=-6 * 3600, timedim='time', SETTLS_order=4)                  # ideal_vortex.py:280-288
    kw = dict(isglobal=True, interp_to_common_grid=False, truncation=None, verbose=False)
    e = acs(ds.copy(), **kw)
    el = acs(lds.copy(), **kw)
    assert isinstance(e, xr.DataArray) and e.dims == ('time', 'latitude', 'longitude') and e.shape == (1, 89, 180)
    assert np.array_equal(e.values, el.values) and e['time'].values[0] == times.values[0]
    ftle = np.log(e) / 2                                                           # the caller's own step
    assert isinstance(ftle, xr.DataArray) and np.isfinite(ftle.values).all()


def test_return_variants_sorting_and_resample_on_xarray():
    from LagrangianCoherence.LCS.LCS import LCS
    ds, lds, times, lat, lon = _datasets()
    kw = dict(isglobal=True, interp_to_common_grid=False, truncation=None, verbose=False, traj_interp_order=1)
    out = LCS(timestep=6 * 3600, SETTLS_order=4, return_dpts=True)(ds, return_traj=True, **kw)
    outl = LCS(timestep=6 * 3600, SETTLS_order=4, return_dpts=True)(lds, return_traj=True, **kw)
    assert len(out) == 5 and all(isinstance(o, xr.DataArray) for o in out)
    for a, b in zip(out, outl):
        assert a.dims == b.dims and np.array_equal(a.values, b.values)
    assert out[0]['time'].values[0] == times.values[-1]                             # forward -> last time
    # descending latitude in, ascending out (LCS.py:101-104)
    rev = ds.isel(latitude=slice(None, None, -1))
    e = LCS(timestep=6 * 3600, SETTLS_order=1)(rev, **kw)
    assert np.array_equal(e['latitude'].values, lat)
    assert np.array_equal(e.values, LCS(timestep=6 * 3600, SETTLS_order=1)(ds, **kw).values)
    # resample through xarray's own resample().interpolate() == the stand-in's pandas + scipy composition
    r = LCS(timestep=-1.0, SETTLS_order=2)(ds, resample='3h', **kw)
    rl = LCS(timestep=-1.0, SETTLS_order=2)(lds, resample='3h', **kw)
    np.testing.assert_allclose(r.values, rl.values, rtol=1e-12)
    # the default global form with regrid + T20 (LCS.py:105-118)
    g = LCS(timestep=-6 * 3600, SETTLS_order=4)(ds, isglobal=True, verbose=False)
    gl = LCS(timestep=-6 * 3600, SETTLS_order=4)(lds, isglobal=True, verbose=False)
    assert g.shape == (1, 360, 721) and np.array_equal(g.values, gl.values)
    assert float(g['latitude'][0]) == -89.75 and float(g['longitude'][-1]) == 179.5
