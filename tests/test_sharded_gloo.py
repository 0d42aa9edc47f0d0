"""N>1 path on CPU: world_size 2 and 3 over gloo.

The HIP kernels cannot run here, so the per-rank engine is a stand-in that answers
advect()/sigma() from the CPU oracle (tests may use the oracle; the product never
does).  What is under test is the product's decomposition logic in
lagrangiancoherence_amd/sharded.py: row partition, global-row offsets, the 2-row
halo exchange over torch.distributed, the sigma input window, ensemble sharding.
"""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from lagrangiancoherence_amd import sharded


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


class OracleEngine:
    """advect/sigma with the Engine's signatures, answered by the oracle on the global grid."""

    def __init__(self, u, v, lat, lon, seed_lat_global, seed_lon):
        self.args = (u, v, lat, lon)
        self.slat, self.slon = seed_lat_global, seed_lon

    reducer = None       # what set_flag_allreduce last installed: ("on", group, comm) or None
    log = ()

    def set_flag_allreduce(self, group=None, comm=None, enable=True):
        self.reducer = ("on", group, comm) if enable else None
        self.log = self.log + (("install" if enable else "remove"),)

    def advect(self, field, seed_lat, seed_lon, timestep, SETTLS_order, interp_order, cyclic, t0, nsteps, row0=0,
               ny_global=None, halo=None, noncyclic_clamp=None):
        from oracle import lcs_oracle as O
        u, v, lat, lon = self.args
        assert ny_global == self.slat.size and np.array_equal(seed_lat, self.slat[row0:row0 + len(seed_lat)])
        if not cyclic and noncyclic_clamp in (None, "reference_outer") and len(seed_lat) != ny_global:
            # the real lc_advect refuses the reference's clamp on a row block without the ranks' flag all-reduce
            assert self.reducer is not None, "row block with the reference's non-cyclic clamp but no flag all-reduce"
        x, y = O.parcel_propagation(u, v, lat, lon, timestep=timestep, SETTLS_order=SETTLS_order,
                                    interp_order=interp_order, cyclic_xboundary=cyclic, seed_lat=self.slat,
                                    seed_lon=self.slon, t0=t0, nsteps=nsteps,
                                    **({} if cyclic else {"noncyclic_clamp": noncyclic_clamp or "reference_outer"}))
        n = len(seed_lat)
        n_lo, n_hi = halo if halo else (0, 0)
        xe = torch.full((n_lo + n + n_hi, x.shape[1]), float("nan"), dtype=torch.float64)
        ye = torch.full((n_lo + n + n_hi, x.shape[1]), float("nan"), dtype=torch.float64)
        xe[n_lo:n_lo + n] = torch.from_numpy(x[row0:row0 + n].copy())
        ye[n_lo:n_lo + n] = torch.from_numpy(y[row0:row0 + n].copy())
        return xe, ye

    def sigma(self, x_ext, y_ext, lat_rows, dlat, dlon, ny_global, in_row0, out_row0, n_out_rows, fd_fp32_cast,
              tensor_layout):
        from oracle import lcs_oracle as O
        nx = x_ext.shape[1]
        gx = np.full((ny_global, nx), np.nan)
        gy = np.full((ny_global, nx), np.nan)
        gx[in_row0:in_row0 + x_ext.shape[0]] = x_ext.numpy()
        gy[in_row0:in_row0 + x_ext.shape[0]] = y_ext.numpy()
        assert np.array_equal(lat_rows, self.slat[in_row0:in_row0 + x_ext.shape[0]])
        with np.errstate(invalid="ignore"):
            s = O.sigma_max(O.flowmap_gradient(gx, gy, self.slat, self.slon, fd_fp32_cast=fd_fp32_cast), tensor_layout)
        return torch.from_numpy(s[out_row0:out_row0 + n_out_rows].copy())


class _Field:
    dtype = np.dtype(np.float64)


def _worker(rank, world, port, ny, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import lcs_oracle as O
        rng = np.random.default_rng(123)
        lat = np.linspace(-75, 75, 16)
        lon = -180 + 15.0 * np.arange(24)
        u = 15 * rng.standard_normal((4, 16, 24))
        v = 8 * rng.standard_normal((4, 16, 24))
        slat = np.linspace(-75, 75, ny)
        slon = np.linspace(-180, 165, 40)
        # 1. raw halo exchange on a labelled array
        lo, hi = sharded.row_partition(ny, world, rank)
        g = torch.arange(ny * 40, dtype=torch.float64).reshape(ny, 40)
        xe, ye, r0 = sharded.halo_exchange(g[lo:hi].clone(), -g[lo:hi].clone(), rank, world, ny, lo, hi)
        n_lo, n_hi = sharded.halo_rows(ny, lo, hi)
        assert r0 == lo - n_lo
        assert torch.equal(xe, g[lo - n_lo:hi + n_hi]) and torch.equal(ye, -g[lo - n_lo:hi + n_hi])
        # 2. whole sharded path vs the unsharded oracle
        eng = OracleEngine(u, v, lat, lon, slat, slon)
        out = sharded.sharded_lcs(eng, _Field(), slat, slon, -3600.0, rank, world, SETTLS_order=2, interp_order=1)
        s_ref, x_ref, y_ref = O.lcs(u, v, lat, lon, timestep=-3600.0, SETTLS_order=2, interp_order=1,
                                    cyclic_xboundary=True, seed_lat=slat, seed_lon=slon)
        assert out["rows"] == (lo, hi)
        assert np.array_equal(out["x_dep"].numpy(), x_ref[lo:hi])
        assert np.array_equal(out["sigma"].numpy(), s_ref[lo:hi]), "sharded sigma differs (halo wrong?)"
        # 3. redundant-halo variant gives the same rows without communication
        out2 = sharded.sharded_lcs(eng, _Field(), slat, slon, -3600.0, rank, world, SETTLS_order=2, interp_order=1,
                                   redundant_halo=True)
        assert np.array_equal(out2["sigma"].numpy(), out["sigma"].numpy())
        assert eng.log == ()                       # cyclic: no flag all-reduce is ever installed
        # 4. cyclic_xboundary=False (the reference's DEFAULT call form): the reference's outer-product clamp couples the
        # rows through the offending columns, so the sharded driver must install the ranks' flag all-reduce around the
        # advection (the real lc_advect refuses a row block without it; the stand-in asserts the same) and remove it
        # afterwards; the result equals the unsharded oracle's with that clamp.  'pointwise' needs no communication.
        u2 = u + 60.0                              # strong westerlies: parcels leave the longitude box
        eng2 = OracleEngine(u2, v, lat, lon, slat, slon)
        out3 = sharded.sharded_lcs(eng2, _Field(), slat, slon, 3600.0, rank, world, SETTLS_order=2, interp_order=1,
                                   cyclic_xboundary=False)
        x3, _ = O.parcel_propagation(u2, v, lat, lon, timestep=3600.0, SETTLS_order=2, interp_order=1, cyclic_xboundary=False,
                                     seed_lat=slat, seed_lon=slon, noncyclic_clamp="reference_outer")
        xp, _ = O.parcel_propagation(u2, v, lat, lon, timestep=3600.0, SETTLS_order=2, interp_order=1, cyclic_xboundary=False,
                                     seed_lat=slat, seed_lon=slon, noncyclic_clamp="pointwise")
        assert not np.array_equal(x3, xp), "test flow too weak: no parcel left the box"
        assert np.array_equal(out3["x_dep"].numpy(), x3[lo:hi]) and eng2.log == ("install", "remove") and eng2.reducer is None
        out4 = sharded.sharded_lcs(eng2, _Field(), slat, slon, 3600.0, rank, world, SETTLS_order=2, interp_order=1,
                                   cyclic_xboundary=False, noncyclic_clamp="pointwise")
        assert np.array_equal(out4["x_dep"].numpy(), xp[lo:hi]) and eng2.log == ("install", "remove")
        q.put((rank, "ok"))
    except Exception as e:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,ny", [(2, 33), (3, 31)])
def test_sharded_path_over_gloo(world, ny):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, ny, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}: {msg}"


def test_row_partition_covers_grid():
    for ny, world in ((4096, 8), (8192, 8), (33, 2), (31, 3), (17, 8)):
        cuts = [sharded.row_partition(ny, world, r) for r in range(world)]
        assert cuts[0][0] == 0 and cuts[-1][1] == ny
        assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
        assert max(h - l for l, h in cuts) - min(h - l for l, h in cuts) <= 1
    with pytest.raises(ValueError):
        sharded.row_partition(7, 8, 0)


def test_halo_rows_at_global_edges():
    assert sharded.halo_rows(100, 0, 50) == (0, 2)
    assert sharded.halo_rows(100, 50, 100) == (2, 0)
    assert sharded.halo_rows(100, 1, 99) == (1, 1)


def test_ensemble_partition():
    got = sum((sharded.ensemble_partition(64, 8, r) for r in range(8)), [])
    assert got == list(range(64))
    assert [len(sharded.ensemble_partition(10, 4, r)) for r in range(4)] == [3, 3, 2, 2]
