"""The adapter layer of the drop-in surface (argument intake, sorting, time labels, return variants,
assertion messages: LCS/LCS.py:72-134,158-168, LCS/trajectory.py:41-60,125-142) on the CPU.

The arithmetic is NOT under test here -- the GPU suite does that through the C ABI.  A stand-in engine
answers the adapter's calls with the CPU oracle (test infrastructure; nothing under
``lagrangiancoherence_amd/`` can reach it), so that the host logic is covered where there is no GPU."""
import os
from types import SimpleNamespace

import numpy as np
import pandas as pd
import pytest
import torch

from lagrangiancoherence_amd import dropin, flows
from tests import labelled
from lagrangiancoherence_amd.engine import common_dtype
from oracle import lcs_oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")


class OracleEngine:
    """Answers the calls ``dropin`` makes on ``engine.Engine`` with oracle results as CPU tensors."""
    torch = torch
    device = "cpu (oracle stand-in)"

    def to_device(self, a, dtype):
        return torch.as_tensor(np.ascontiguousarray(np.asarray(a, dtype=dtype)))

    def to_host(self, t):
        return t.detach().cpu().numpy()

    fuse_asked = []           # what the adapter asked of prepare_field (float64 fidelity rule of the drop-in surface)

    def f64_fuse_levels(self, dtype, n_seeds):
        return np.dtype(dtype) != np.dtype(np.float64) or n_seeds > (1 << 18)

    def prepare_field(self, u, v, lat, lon, interp_order=1, dtype=None, fuse_levels=None):
        self.fuse_asked.append(fuse_levels)
        if interp_order not in (1, 3):
            raise ValueError(f"interp_order {interp_order} unsupported")
        dt = np.dtype(dtype or common_dtype(u, v, lat, lon))
        return SimpleNamespace(u=np.asarray(u, dt), v=np.asarray(v, dt), lat=np.asarray(lat, dt),
                               lon=np.asarray(lon, dt), dtype=dt, nt=u.shape[0])

    def advect(self, f, slat, slon, timestep, SETTLS_order=0, interp_order=1, cyclic_xboundary=True, t0=0,
               nsteps=None, return_traj=False, **_):
        r = O.parcel_propagation(f.u, f.v, f.lat, f.lon, timestep=timestep, SETTLS_order=SETTLS_order,
                                 interp_order=interp_order, cyclic_xboundary=cyclic_xboundary,
                                 return_traj=return_traj, seed_lat=np.asarray(slat), seed_lon=np.asarray(slon),
                                 t0=t0, nsteps=nsteps)
        if return_traj:
            tx, ty = r
            return tuple(torch.as_tensor(a) for a in (tx[-1], ty[-1], tx, ty))
        return tuple(torch.as_tensor(a) for a in r)

    # the calls dropin makes since round 4: pack + advect (+ sigma) from the raw wind in one call
    def pack_and_advect(self, u, v, lat, lon, slat, slon, timestep, SETTLS_order=0, interp_order=1, cyclic_xboundary=True,
                        fuse_levels=None, pipeline=None, chunk=None, return_traj=False, noncyclic_clamp=None):
        f = self.prepare_field(u, v, lat, lon, interp_order, fuse_levels=fuse_levels)
        return (f, *self.advect(f, slat, slon, timestep, SETTLS_order, interp_order, cyclic_xboundary, return_traj=return_traj))

    def lcs_wind(self, u, v, lat, lon, slat, slon, timestep, SETTLS_order=0, interp_order=1, cyclic_xboundary=True,
                 fuse_levels=None, gauss_sigma=None, fd_fp32_cast=True, tensor_layout="reference", return_traj=False,
                 noncyclic_clamp=None, pipeline=None):
        f = self.prepare_field(u, v, lat, lon, interp_order, fuse_levels=fuse_levels)
        return self.lcs(f, slat, slon, timestep, SETTLS_order, interp_order, cyclic_xboundary, gauss_sigma=gauss_sigma,
                        fd_fp32_cast=fd_fp32_cast, tensor_layout=tensor_layout, return_traj=return_traj)

    def lcs(self, f, slat, slon, timestep, SETTLS_order=0, interp_order=1, cyclic_xboundary=True, t0=0, nsteps=None,
            gauss_sigma=None, fd_fp32_cast=True, tensor_layout="reference", return_traj=False):
        res = self.advect(f, slat, slon, timestep, SETTLS_order, interp_order, cyclic_xboundary, t0, nsteps, return_traj)
        x, y = res[0].numpy(), res[1].numpy()
        dtens = O.flowmap_gradient(x, y, np.asarray(slat), np.asarray(slon), sigma=gauss_sigma)
        out = {"sigma": torch.as_tensor(O.sigma_max(dtens)), "x_dep": res[0], "y_dep": res[1]}
        if return_traj:
            out["traj_x"], out["traj_y"] = res[2], res[3]
        return out

    def gaussian_filter(self, a, sigma):
        from scipy.ndimage import gaussian_filter
        return torch.as_tensor(gaussian_filter(a.numpy(), sigma=sigma))

    def flowmap_gradient(self, xd, yd, lat, dlat, dlon, **_):
        lon = np.arange(xd.shape[1]) * dlon          # only the spacing enters (tools.py:255-256)
        return torch.as_tensor(O.flowmap_gradient(xd.numpy(), yd.numpy(), np.asarray(lat), lon))


@pytest.fixture(autouse=True)
def oracle_engine(monkeypatch):
    monkeypatch.setattr(dropin, "_ENGINE", OracleEngine())


def _dataset():
    u, v, lat, lon = flows.config1()
    times = pd.date_range('2000-01-01', periods=u.shape[0], freq='6h').values
    coords = {'latitude': lat, 'longitude': lon, 'time': times}
    U = labelled.DataArray(u.transpose(1, 2, 0), ['latitude', 'longitude', 'time'], coords, name='u')
    V = labelled.DataArray(v.transpose(1, 2, 0), ['latitude', 'longitude', 'time'], coords, name='v')
    return labelled.Dataset({'u': U, 'v': V}), times, lat, lon


def test_parcel_propagation_call_forms_shapes_and_time_labels():
    from LagrangianCoherence.LCS import trajectory
    ds, times, lat, lon = _dataset()
    x, y = trajectory.parcel_propagation(ds.u, ds.v, timestep=-6 * 3600, propdim='time', SETTLS_order=4, copy=True,
                                         return_traj=True, cyclic_xboundary=True, verbose=False, interp_order=1)
    g = np.load(os.path.join(GOLD, "g1_bwd_k4_o1.npz"))
    assert x.dims == ('time', 'latitude', 'longitude') and x.shape == (8, 89, 180)
    np.testing.assert_allclose(x.values[-1], g["x_dep"], rtol=0, atol=1e-12)
    assert x['time'].values[0] == times[-1] and x['time'].values[-1] == times[0]     # labels reversed (Q6)
    assert np.array_equal(x.values[0], np.meshgrid(lon, lat)[0])                      # entry 0 = seed grid
    xf, yf = trajectory.parcel_propagation(ds.u, ds.v, timestep=6 * 3600, SETTLS_order=2, cyclic_xboundary=True,
                                           verbose=False, interp_order=1)
    assert xf.dims == ('latitude', 'longitude') and xf.coords['time'] == times.tolist()[-1]


def test_lcs_return_variants_and_time_stamp():
    from LagrangianCoherence.LCS.LCS import LCS
    ds, times, lat, lon = _dataset()
    kw = dict(isglobal=True, interp_to_common_grid=False, truncation=None, verbose=False, traj_interp_order=1)
    out = LCS(timestep=6 * 3600, SETTLS_order=4, return_dpts=True)(ds, return_traj=True, **kw)
    assert len(out) == 5
    eig, xd, yd, xt, yt = out
    g = np.load(os.path.join(GOLD, "g1_fwd_k4_o1.npz"))
    np.testing.assert_allclose(eig.values[0], g["sigma"], rtol=1e-12)
    assert eig.dims == ('time', 'latitude', 'longitude') and eig['time'].values[0] == times[-1]   # fwd -> last
    assert xt.shape == (8, 89, 180) and np.array_equal(xt.values[-1], xd.values)
    assert len(LCS(timestep=6 * 3600, return_dpts=True)(ds, **kw)) == 3
    assert len(LCS(timestep=6 * 3600)(ds, return_traj=True, **kw)) == 3
    bwd = LCS(timestep=-6 * 3600, SETTLS_order=1)(ds, **kw)
    assert bwd['time'].values[0] == times[0]                                          # bwd -> first (LCS.py:158)
    e2 = LCS(timestep=6 * 3600, SETTLS_order=4)(u=ds.u, v=ds.v, **kw)                 # u=, v= keywords
    assert np.array_equal(e2.values, eig.values)


def test_resample_keyword_without_xarray():
    """LCS.py:88-91: `resample='3h'` interpolates the 6-hourly series linearly to 3-hourly and re-derives |timestep|
    from the new axis (its sign is kept)."""
    from LagrangianCoherence.LCS.LCS import LCS
    ds, times, lat, lon = _dataset()
    kw = dict(isglobal=True, interp_to_common_grid=False, truncation=None, verbose=False, traj_interp_order=1)
    e = LCS(timestep=-999.0, SETTLS_order=2)(ds, resample='3h', **kw)
    u, v, _, _ = flows.config1()
    x = np.arange(u.shape[0]) * 6.0
    from scipy.interpolate import interp1d
    xn = np.arange(0, x[-1] + 1e-9, 3.0)
    u3, v3 = interp1d(x, u, axis=0)(xn), interp1d(x, v, axis=0)(xn)
    s, _, _ = O.lcs(u3, v3, lat, lon, timestep=-3 * 3600.0, SETTLS_order=2, interp_order=1, cyclic_xboundary=True)
    np.testing.assert_allclose(e.values[0], s, rtol=1e-9)
    assert e['time'].values[0] == times[0]


def test_unsorted_inputs_are_sorted_and_dims_are_checked():
    from LagrangianCoherence.LCS import trajectory
    from LagrangianCoherence.LCS.LCS import LCS
    ds, times, lat, lon = _dataset()
    Ur, Vr = ds.u.isel(latitude=slice(None, None, -1)), ds.v.isel(latitude=slice(None, None, -1))
    kw = dict(timestep=6 * 3600, SETTLS_order=2, cyclic_xboundary=True, verbose=False, interp_order=1)
    x, _ = trajectory.parcel_propagation(Ur, Vr, **kw)
    x0, _ = trajectory.parcel_propagation(ds.u, ds.v, **kw)
    assert np.array_equal(x.values, x0.values) and np.array_equal(x['latitude'].values, lat)
    bad = labelled.DataArray(ds.u.values, ['lat', 'longitude', 'time'],
                             {'lat': ds.u.coords['latitude'], 'longitude': ds.u.coords['longitude'],
                              'time': ds.u.coords['time']})
    with pytest.raises(AssertionError, match="u and v dims are different"):
        LCS()(u=bad, v=ds.v, verbose=False)
    with pytest.raises(AssertionError, match="array dims should be latitude and longitude only"):
        LCS()(u=bad, v=bad, verbose=False)
    with pytest.raises(ValueError):
        LCS()(ds, verbose=False, traj_interp_order=0)


def test_subdomain_crop_and_flowmap_gradient_labels():
    from LagrangianCoherence.LCS.LCS import LCS, flowmap_gradient
    ds, times, lat, lon = _dataset()
    sub = {'latitude': slice(-20, 20), 'longitude': slice(-60, -20)}
    e = LCS(timestep=6 * 3600, SETTLS_order=1, subdomain=sub)(ds, verbose=False, traj_interp_order=1)
    assert e['latitude'].values.min() == -18 and e['latitude'].values.max() == 18   # strict (tools.py:184-185)
    assert e['longitude'].values.min() == -58 and e['longitude'].values.max() == -22
    full, xd, yd = LCS(timestep=6 * 3600, SETTLS_order=1, return_dpts=True)(ds, verbose=False, traj_interp_order=1)
    ilat, ilon = (lat > -20) & (lat < 20), (lon > -60) & (lon < -20)
    assert np.array_equal(e.values[0], full.values[0][ilat][:, ilon])
    dt = flowmap_gradient(xd, yd)
    assert dt.dims == ('derivatives', 'latitude', 'longitude') and dt.shape == (9, 89, 180)
    assert list(dt['derivatives'].values) == ['dxdx', 'dxdy', 'dydx', 'dydy', 'dzdx', 'dzdy', 'dxdr', 'dydr', 'dzdr']
    np.testing.assert_allclose(dt.values, O.flowmap_gradient(xd.values, yd.values, lat, lon), rtol=1e-12, atol=1e-12)
