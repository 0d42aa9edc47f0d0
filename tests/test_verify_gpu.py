"""lc_ctx_set_verify: the wave-state audit of the one-seed LDS kernels.

One process, one GPU: the audit must see every wave-level, find nothing, and leave the results bit-identical to the
plain kernels; with the injected corruption (mode 2) it must fire exactly where the corruption was put."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def setup():
    from lagrangiancoherence_amd import flows
    from lagrangiancoherence_amd.engine import Engine
    eng = Engine(0)
    u, v, lat, lon = flows.era5_like(nt=9, ny=180, nx=360)
    slat, slon = flows.seed_grid(256, 512, lat, lon)     # sparse seeds: a wave's patch is wider than its tile, both paths run
    yield eng, u, v, lat, lon, slat, slon
    eng.close()


@pytest.mark.parametrize("order", [1, 3])
def test_audit_sees_every_wave_level_and_changes_nothing(setup, order):
    eng, u, v, lat, lon, slat, slon = setup
    f = eng.prepare_field(u, v, lat, lon, order)
    eng.set_verify(0)
    x0, y0 = eng.advect(f, slat, slon, -900.0, 4, order)
    plain = eng.last_advect_kernel()
    assert plain == f"advect_lds_kernel<{order}, 4, true>"
    eng.set_verify(1)
    try:
        x1, y1 = eng.advect(f, slat, slon, -900.0, 4, order)
        assert eng.last_advect_kernel() == f"advect_lds_kernel<{order}, 4, true, verify>"
        a = eng.read_verify()
        assert torch.equal(x0, x1) and torch.equal(y0, y1)
        n_wave_levels = a["audited"]
        # every wave that holds an interior seed audits each of the 8 levels once: 32 x 64 waves minus none (pole rows are
        # rows of waves that also hold interior rows)
        assert n_wave_levels == (256 // 8) * (512 // 8) * 8, a
        assert a["tile_changed"] == 0 and a["entries_changed"] == 0 and "first_event" not in a, a
        assert eng.read_verify()["audited"] == 0                  # read_verify(reset=True) zeroed them
    finally:
        eng.set_verify(0)


def test_audit_fires_on_an_injected_corruption(setup):
    eng, u, v, lat, lon, slat, slon = setup
    f = eng.prepare_field(u, v, lat, lon, 1)
    x0, y0 = eng.advect(f, slat, slon, -900.0, 4, 1)
    eng.set_verify(2)
    try:
        x1, y1 = eng.advect(f, slat, slon, -900.0, 4, 1)
        a = eng.read_verify()
        assert a["tile_changed"] == 1 and a["entries_changed"] == 1, a
        ev = a["first_event"]
        assert (ev["tile"], ev["wave"], ev["level"]) == (5, 1, 1) and ev["hw_id_before"] == ev["hw_id_after"], a
        # node (row 3, column 7) of the 8 x 16 tile -- the centre lane's window origin -- was staged by lane 3 * 8 + 3
        assert int(ev["lane_mask"], 16) == 1 << 27, a
        # and the corruption is what a lost tile looks like from outside: a few seeds of that one wave differ
        d = ((x0 != x1) | (y0 != y1)).nonzero()
        assert 0 < len(d) <= 64 and d[:, 0].min() >= 8 and d[:, 0].max() < 16 and d[:, 1].min() >= 40 and d[:, 1].max() < 48, d
    finally:
        eng.set_verify(0)
    with pytest.raises(ValueError):
        eng.read_verify()                                         # the audit is off: nothing to read
