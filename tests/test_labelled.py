"""The xarray stand-in used where xarray is not installed (adapter plumbing only)."""
import numpy as np
import pytest

from tests import labelled


def _da():
    return labelled.DataArray(np.arange(24.0).reshape(2, 3, 4), ['time', 'latitude', 'longitude'],
                              {'time': np.array([10, 20]), 'latitude': np.array([3.0, 1.0, 2.0]),
                               'longitude': np.arange(4.0)}, name='u')


def test_transpose_sortby_isel_copy():
    da = _da()
    t = da.transpose('latitude', 'longitude', 'time')
    assert t.shape == (3, 4, 2) and t.values[1, 2, 0] == da.values[0, 1, 2]
    s = da.sortby('latitude')
    assert list(s['latitude'].values) == [1.0, 2.0, 3.0] and np.array_equal(s.values[:, 0], da.values[:, 1])
    assert da.isel(time=1).dims == ('latitude', 'longitude') and da.isel(time=1).coords['time'] == 20
    r = da.isel(latitude=slice(None, None, -1))
    assert list(r['latitude'].values) == [2.0, 1.0, 3.0] and np.array_equal(r.values[:, 0], da.values[:, 2])
    c = da.copy()
    c.values[0, 0, 0] = -1
    assert da.values[0, 0, 0] == 0 and c.name == 'u'
    assert da.latitude.values[0] == 3.0 and np.asarray(da).shape == (2, 3, 4)


def test_dataset_and_errors():
    da = _da()
    ds = labelled.Dataset({'u': da, 'v': da.copy()})
    assert ds.u is da and ds['v'].shape == (2, 3, 4) and ds.copy().u is not da
    with pytest.raises(ValueError):
        labelled.DataArray(np.zeros((2, 2)), ['a'])
    with pytest.raises(ValueError):
        labelled.DataArray(np.zeros((2, 2)), ['a', 'b'], {'a': np.zeros(3)})
    with pytest.raises(AttributeError):
        ds.w


def test_resample_linear_is_pandas_index_plus_scipy_interp1d():
    """LCS/LCS.py:89-90 `u.resample(time='2h').interpolate('linear')` without xarray: 6-hourly -> 2-hourly."""
    import pandas as pd
    times = pd.date_range('2000-01-01', periods=4, freq='6h').values
    vals = np.random.default_rng(0).standard_normal((3, 4, 5))                 # (latitude, time, longitude)
    da = labelled.DataArray(vals, ['latitude', 'time', 'longitude'],
                            {'latitude': np.arange(3.0), 'time': times, 'longitude': np.arange(5.0)}, name='u')
    r = labelled.resample_linear(da, 'time', '2h')
    assert r.dims == da.dims and r.shape == (3, 10, 5) and r.name == 'u'
    assert np.array_equal(r['time'].values, pd.date_range('2000-01-01', periods=10, freq='2h').values)
    np.testing.assert_allclose(r.values[:, ::3], vals, rtol=1e-14, atol=1e-15)   # original instants (interp1d rounds: slope*dx + y_lo)
    np.testing.assert_allclose(r.values[:, 1], vals[:, 0] + (vals[:, 1] - vals[:, 0]) / 3, rtol=1e-14)
    np.testing.assert_allclose(r.values[:, 5], vals[:, 1] + 2 * (vals[:, 2] - vals[:, 1]) / 3, rtol=1e-14)
    # the timestep the reference derives from the new axis (LCS.py:91)
    dt = (r['time'].values[1] - r['time'].values[0]).astype('timedelta64[s]').astype('float')
    assert dt == 7200.0
