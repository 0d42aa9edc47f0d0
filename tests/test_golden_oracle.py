"""The committed golden fixtures are outputs of the CPU oracle (tests/golden/make_golden.py); this keeps
them honest on the CPU: the oracle, the input generators and the installed scipy/numpy must still reproduce
every vector, so a drift in any of them cannot slip in behind the GPU tests' back."""
import os

import numpy as np
import pytest
import scipy

from lagrangiancoherence_amd import flows
from oracle import lcs_oracle as O

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def _load(name):
    return np.load(os.path.join(GOLD, name + ".npz"))


def _same(a, b):
    # identical library versions reproduce bit for bit; allow rounding-level slack for other builds
    if np.array_equal(a, b, equal_nan=True):
        return True
    return np.allclose(a, b, rtol=1e-9, atol=1e-9, equal_nan=True)


def test_fixture_versions_are_recorded():
    g = _load("g1_bwd_k4_o3")
    assert str(g["scipy"]) and str(g["numpy"])
    if str(g["scipy"]) != scipy.__version__:
        pytest.skip(f"fixtures were made with scipy {g['scipy']}, this is {scipy.__version__}")


@pytest.mark.parametrize("tag,dt,K", [("bwd_k4", -21600, 4), ("fwd_k2", 21600, 2), ("fwd_k4", 21600, 4)])
@pytest.mark.parametrize("order", [3, 1])
def test_config1_fixtures(tag, dt, K, order):
    g = _load(f"g1_{tag}_o{order}")
    u, v, lat, lon = flows.config1()
    assert np.array_equal(g["input_checksum"], [u.sum(), v.sum(), np.abs(u).max(), np.abs(v).max()])
    s, x, y = O.lcs(u, v, lat, lon, timestep=dt, SETTLS_order=K, interp_order=order, cyclic_xboundary=True)
    assert _same(x, g["x_dep"]) and _same(y, g["y_dep"]) and _same(s, g["sigma"])


def test_config1_trajectory_fixture():
    g = _load("g1_traj_bwd_k4_o3")
    u, v, lat, lon = flows.config1()
    tx, ty = O.parcel_propagation(u, v, lat, lon, timestep=-21600, SETTLS_order=4, interp_order=3,
                                  cyclic_xboundary=True, return_traj=True)
    assert _same(tx, g["traj_x"]) and _same(ty, g["traj_y"])


def test_config2_downsampled_fixture():
    g = _load("g2_c2_128_k4_o1")
    u, v, lat, lon = flows.config2(n=128, nt=21)
    s, x, y = O.lcs(u, v, lat, lon, timestep=-900, SETTLS_order=4, interp_order=1, cyclic_xboundary=True)
    assert _same(x, g["x_dep"]) and _same(y, g["y_dep"]) and _same(s, g["sigma"])


@pytest.mark.parametrize("order", [1, 3])
def test_config3_miniature_fixture(order):
    g = _load(f"g3_c3mini_k4_o{order}")
    u, v, lat, lon = flows.era5_like(nt=13, ny=72, nx=144)
    slat, slon = flows.seed_grid(96, 160, lat, lon)
    s, x, y = O.lcs(u, v, lat, lon, timestep=-900, SETTLS_order=4, interp_order=order, cyclic_xboundary=True,
                    seed_lat=slat, seed_lon=slon)
    assert x.dtype == np.float32
    assert _same(x, g["x_dep"]) and _same(y, g["y_dep"])
    assert np.allclose(s, g["sigma"], rtol=1e-5)
    # the float32 answer sits where the fixture says it does relative to the float64 truth
    assert np.abs(x.astype(np.float64) - g["x_dep64"]).max() < 5e-4
