"""Input generators: the vectorised ideal vortex equals the reference's triple
loop (examples/ideal_vortex.py:159-201, restated in oracle.ideal_vortex_loops)."""
import numpy as np
import pytest

from lagrangiancoherence_amd import flows
from oracle import lcs_oracle as O


def test_ideal_vortex_matches_loop_form_config1():
    u, v, lat, lon = flows.config1()
    ul, vl, latl, lonl = O.ideal_vortex_loops(**flows.vortex_config_subtropical)
    assert u.shape == (8, 89, 180)
    assert np.array_equal(lat, latl) and np.array_equal(lon, lonl)
    assert np.array_equal(u, ul) and np.array_equal(v, vl)


def test_ideal_vortex_moving_k_positive():
    cfg = dict(lat_min=-20, lat_max=21, lon_min=-30, lon_max=30, dx=3, dy=2, nt=5, max_intensity=40,
               radius=2, center=[-5, 3], u_c=0.5, v_c=2.0, k=2, basic_zonal=1.5)
    u, v, _, _ = flows.ideal_vortex(**cfg)
    ul, vl, _, _ = O.ideal_vortex_loops(**cfg)
    assert np.array_equal(u, ul) and np.array_equal(v, vl)


def test_era5_like_is_deterministic_and_sane():
    u, v, lat, lon = flows.era5_like(nt=3, ny=72, nx=144)
    u2, v2, _, _ = flows.era5_like(nt=3, ny=72, nx=144)
    assert np.array_equal(u, u2) and np.array_equal(v, v2)
    assert u.dtype == np.float32 and lat.dtype == np.float32
    assert lat[0] == -88.75 and lat[-1] == 88.75 and lon[0] == -180 and lon[-1] == 177.5
    assert 20 < np.abs(u).max() < 120 and 5 < np.abs(v).max() < 80
    # time dependence is real
    assert np.abs(u[2] - u[0]).max() > 1e-3


def test_seed_grid_inclusive():
    _, _, lat, lon = flows.era5_like(nt=2, ny=72, nx=144)
    slat, slon = flows.seed_grid(96, 160, lat, lon)
    assert slat[0] == lat[0] and slat[-1] == lat[-1] and slon[0] == lon[0] and slon[-1] == lon[-1]


@pytest.mark.gpu
def test_config2_on_device_equals_the_numpy_generator():
    import torch
    from lagrangiancoherence_amd import flows
    u, v, lat, lon = flows.config2(n=96, nt=9)
    ud, vd, lat2, lon2 = flows.config2_on_device(torch, "cuda", n=96, nt=9)
    assert np.array_equal(lat, lat2) and np.array_equal(lon, lon2) and ud.dtype == torch.float64
    np.testing.assert_allclose(ud.cpu().numpy(), u, rtol=0, atol=1e-11)
    np.testing.assert_allclose(vd.cpu().numpy(), v, rtol=0, atol=1e-11)
