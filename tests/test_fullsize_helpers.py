"""The window / subset constructions of tests/_fullsize.py reproduce the full-grid oracle exactly (CPU)."""
import numpy as np
import pytest

from lagrangiancoherence_amd import flows
from oracle import lcs_oracle as O
from tests._fullsize import oracle_subset, oracle_window, subset


@pytest.mark.parametrize("order", [1, 3])
def test_window_and_subset_equal_the_full_grid_oracle(order):
    u, v, lat, lon = flows.era5_like(nt=5, ny=36, nx=72)
    slat, slon = flows.seed_grid(60, 90, lat, lon)
    kw = dict(timestep=-1800.0, SETTLS_order=2, cyclic_xboundary=True)
    s, x, y = O.lcs(u, v, lat, lon, interp_order=order, seed_lat=slat, seed_lon=slon, **kw)
    for (r0, r1, c0, c1) in ((0, 12, 5, 20), (20, 33, 40, 60), (44, 60, 2, 88)):
        xw, yw, sw = oracle_window(O, u, v, lat, lon, slat, slon, r0, r1, c0, c1, np.float32, order, **kw)
        assert np.array_equal(xw, x[r0:r1, c0:c1]) and np.array_equal(yw, y[r0:r1, c0:c1])
        assert np.array_equal(sw, s[r0:r1, c0:c1], equal_nan=True)
    rows, cols = subset(60, 12, order, must=[29, 30]), subset(90, 10, 0)
    assert {29, 30} <= set(rows.tolist()) and rows[0] == 0 and rows[-1] == 59
    xs, ys = oracle_subset(O, u, v, lat, lon, slat, slon, rows, cols, np.float32, interp_order=order, **kw)
    assert np.array_equal(xs, x[rows][:, cols]) and np.array_equal(ys, y[rows][:, cols])
