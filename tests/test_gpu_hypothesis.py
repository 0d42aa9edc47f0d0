"""Randomised parity: hypothesis draws grid shapes, seed grids, K, interpolation order, time-step sign,
boundary mode and a start window; the float64 HIP path must agree with the CPU oracle on every draw."""
import numpy as np
import pytest
from hypothesis import HealthCheck, given, settings
from hypothesis import strategies as st

pytestmark = pytest.mark.gpu

_ENG = {}


def _engine():
    if "e" not in _ENG:
        from lagrangiancoherence_amd.engine import Engine
        _ENG["e"] = Engine(0)
    return _ENG["e"]


@settings(max_examples=60, deadline=None, suppress_health_check=[HealthCheck.too_slow, HealthCheck.data_too_large])
@given(ny=st.integers(8, 30), nx=st.integers(8, 40), nt=st.integers(2, 5), K=st.integers(0, 4),
       order=st.sampled_from([1, 3]), dt=st.sampled_from([-5400.0, -600.0, 900.0, 7200.0]),
       cyclic=st.booleans(), seeds_on_nodes=st.booleans(), seed=st.integers(0, 2 ** 31 - 1),
       lat_hi=st.sampled_from([60.0, 85.0, 88.0]), scale=st.sampled_from([5.0, 25.0, 80.0]))
def test_float64_advect_and_sigma_match_oracle(ny, nx, nt, K, order, dt, cyclic, seeds_on_nodes, seed, lat_hi, scale):
    from oracle import lcs_oracle as O
    eng = _engine()
    rng = np.random.default_rng(seed)
    lat = np.linspace(-lat_hi, lat_hi, ny)
    lon = -180 + 360.0 / nx * np.arange(nx)
    u = scale * rng.standard_normal((nt, ny, nx))
    v = 0.5 * scale * rng.standard_normal((nt, ny, nx))
    if seeds_on_nodes:
        slat, slon = lat, lon
    else:
        slat = np.linspace(lat[0], lat[-1], int(rng.integers(7, 40)))
        slon = np.linspace(lon[0], lon[-1], int(rng.integers(7, 50)))
    t0 = int(rng.integers(0, nt - 1))
    nsteps = int(rng.integers(1, nt - t0))
    f = eng.prepare_field(u, v, lat, lon, order)
    x, y = eng.advect(f, slat, slon, dt, SETTLS_order=K, interp_order=order, cyclic_xboundary=cyclic, t0=t0,
                      nsteps=nsteps)
    xr_, yr_ = O.parcel_propagation(u, v, lat, lon, timestep=dt, SETTLS_order=K, interp_order=order,
                                    cyclic_xboundary=cyclic, seed_lat=slat, seed_lon=slon, t0=t0, nsteps=nsteps)
    xg, yg = x.cpu().numpy(), y.cpu().numpy()
    np.testing.assert_allclose(yg, yr_, rtol=0, atol=1e-9)
    np.testing.assert_allclose(xg, xr_, rtol=0, atol=1e-9)
    if slat.size >= 5 and slon.size >= 5:
        # sigma from the ORACLE's departure points on both sides: isolates the gradient/eigen kernel
        ref = O.sigma_max(O.flowmap_gradient(xr_, yr_, slat, slon))
        sig = eng.sigma(xr_, yr_, slat, slat[1] - slat[0], slon[1] - slon[0]).cpu().numpy()
        ok = np.isfinite(ref)
        np.testing.assert_allclose(sig[ok], ref[ok], rtol=1e-7)
