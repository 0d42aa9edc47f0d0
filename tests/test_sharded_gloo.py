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
               ny_global=None, halo=None, noncyclic_clamp=None, global_rows=None):
        from oracle import lcs_oracle as O
        u, v, lat, lon = self.args
        if global_rows is not None:          # an ascending selection of the global rows (the interleaved chunks' windows)
            from lagrangiancoherence_amd.engine import Engine
            g = np.asarray(global_rows)
            assert ny_global == self.slat.size and row0 == 0 and halo is None and np.array_equal(seed_lat, self.slat[g])
            Engine.pole_window(g, len(g), ny_global, interp_order)      # what the real advect does with them: must not refuse
            x, y = O.parcel_propagation(u, v, lat, lon, timestep=timestep, SETTLS_order=SETTLS_order, interp_order=interp_order,
                                        cyclic_xboundary=cyclic, seed_lat=self.slat, seed_lon=self.slon, t0=t0, nsteps=nsteps,
                                        **({} if cyclic else {"noncyclic_clamp": noncyclic_clamp}))
            self.calls = getattr(self, "calls", 0) + 1
            return torch.from_numpy(x[g].copy()), torch.from_numpy(y[g].copy())
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


def _worker_interleaved(rank, world, port, ny, q):
    """sharded_lcs(partition="interleaved"): the rank's chunks, advected in ONE call, their halo rows exchanged with the
    previous / next rank in one batch (a ring), equal the unsharded oracle's rows bit for bit."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import lcs_oracle as O
        rng = np.random.default_rng(321)
        lat = np.linspace(-75, 75, 16)
        lon = -180 + 15.0 * np.arange(24)
        u = 15 * rng.standard_normal((4, 16, 24))
        v = 8 * rng.standard_normal((4, 16, 24))
        slat = np.linspace(-75, 75, ny)
        slon = np.linspace(-180, 165, 40)
        eng = OracleEngine(u, v, lat, lon, slat, slon)
        s_ref, x_ref, y_ref = O.lcs(u, v, lat, lon, timestep=-3600.0, SETTLS_order=2, interp_order=1,
                                    cyclic_xboundary=True, seed_lat=slat, seed_lon=slon)
        sent = []
        real = dist.batch_isend_irecv
        dist.batch_isend_irecv = lambda ops: sent.append(len(ops)) or real(ops)
        out = sharded.sharded_lcs(eng, _Field(), slat, slon, -3600.0, rank, world, SETTLS_order=2, interp_order=1,
                                  partition="interleaved", window=32)
        dist.batch_isend_irecv = real
        chunks = sharded.interleaved_partition(ny, world, rank, 32)
        # one advect call over the concatenated chunks; ONE batch of at most 2 sends + 2 receives (a ring: previous / next rank)
        assert len(chunks) >= 2 and out["rows"] == chunks and eng.calls == 1 and sent in ([4], [3]), (chunks, eng.calls, sent)
        g = np.asarray(out["global_rows"])
        assert g.tolist() == [r for lo, hi in chunks for r in range(lo, hi)]
        assert np.array_equal(out["x_dep"].numpy(), x_ref[g]) and np.array_equal(out["y_dep"].numpy(), y_ref[g])
        assert np.array_equal(out["sigma"].numpy(), s_ref[g]), "interleaved sigma differs (halo rows from the wrong chunk?)"
        # the halo rows advected redundantly instead of exchanged: the same values, no communication
        red = sharded.sharded_lcs(eng, _Field(), slat, slon, -3600.0, rank, world, SETTLS_order=2, interp_order=1,
                                  partition="interleaved", window=32, redundant_halo=True)
        assert np.array_equal(red["sigma"].numpy(), out["sigma"].numpy()) and np.array_equal(red["x_dep"].numpy(), out["x_dep"].numpy())
        # the exchange on a labelled array: every chunk's window arrives whole
        lab = torch.arange(ny * 40, dtype=torch.float64).reshape(ny, 40)
        own = torch.as_tensor(g)
        xw, yw = sharded.chunk_halo_exchange(lab[own].contiguous(), (-lab[own]).contiguous(), chunks, ny, rank, world)
        win = torch.as_tensor(sharded.interleaved_rows(ny, chunks, with_halo=True))
        assert torch.equal(xw, lab[win]) and torch.equal(yw, -lab[win])
        # the ranks' chunks tile the grid
        mine = torch.zeros(ny, dtype=torch.int64)
        mine[g] = 1
        dist.all_reduce(mine)
        assert bool((mine == 1).all())
        # the reference's non-cyclic outer-product clamp couples all rows through the ranks' flags: contiguous blocks stay
        u2 = u + 60.0
        eng2 = OracleEngine(u2, v, lat, lon, slat, slon)
        out3 = sharded.sharded_lcs(eng2, _Field(), slat, slon, 3600.0, rank, world, SETTLS_order=2, interp_order=1,
                                   cyclic_xboundary=False, partition="interleaved", window=32)
        assert out3["rows"] == sharded.row_partition(ny, world, rank) and eng2.log == ("install", "remove")
        # ... the per-point clamp does not, and takes the interleaved chunks
        out4 = sharded.sharded_lcs(eng2, _Field(), slat, slon, 3600.0, rank, world, SETTLS_order=2, interp_order=1,
                                   cyclic_xboundary=False, noncyclic_clamp="pointwise", partition="interleaved", window=32)
        xp, _ = O.parcel_propagation(u2, v, lat, lon, timestep=3600.0, SETTLS_order=2, interp_order=1, cyclic_xboundary=False,
                                     seed_lat=slat, seed_lon=slon, noncyclic_clamp="pointwise")
        assert out4["rows"] == chunks and np.array_equal(out4["x_dep"].numpy(), xp[g])
        q.put((rank, "ok"))
    except Exception:  # pragma: no cover
        import traceback
        q.put((rank, traceback.format_exc()))
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,ny", [(2, 192), (3, 200)])
def test_interleaved_partition_over_gloo(world, ny):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker_interleaved, args=(r, world, port, ny, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=180) for _ in procs]
    for p in procs:
        p.join(timeout=60)
    for rank, msg in res:
        assert msg == "ok", f"rank {rank}: {msg}"


def test_interleaved_chunks_tile_the_grid_aligned_and_even():
    """The interleaved partition (strong scaling): chunks tile [0, ny) with boundaries on the multiples of the chunk size (the
    two-seed kernel's 16-row patches and 64-row workgroups never straddle two chunks of a rank's concatenated rows); a rank's
    chunks lie `world` chunks apart (every rank samples every latitude band); on the BASELINE grids every rank's rows are
    the same multiple of 512; grids that do not deal out evenly, or hold fewer than two chunks per rank, fall back to the
    contiguous blocks."""
    for ny, world, chunk in ((8192, 8, 256), (8192, 4, 256), (8192, 2, 256), (4096, 8, 256), (4096, 4, 256), (5000, 7, 256),
                             (192, 2, 32), (200, 3, 32), (1000, 3, 64)):
        ch = sharded.interleaved_chunks(ny, world, chunk)
        assert ch[0][0] == 0 and ch[-1][1] == ny and all(ch[i][1] == ch[i + 1][0] for i in range(len(ch) - 1))
        assert len(ch) >= 2 * world and all(lo % chunk == 0 for lo, hi in ch) and all(hi - lo == chunk for lo, hi in ch[:-1])
        assert ch[-1][1] - ch[-1][0] >= 2 * sharded.HALO
        per = [sum(hi - lo for lo, hi in ch[r::world]) for r in range(world)]
        assert max(per) <= 1.1 * ny / world, (ny, world, per)
        for r in range(world):
            mine = sharded.interleaved_partition(ny, world, r, chunk)
            assert mine == ch[r::world]
            rows = sharded.interleaved_rows(ny, mine)
            assert rows == sorted(set(rows)) and len(rows) == per[r]
            win = sharded.interleaved_rows(ny, mine, with_halo=True)
            assert win == sorted(set(win)) and set(rows) <= set(win) and len(win) <= per[r] + 2 * sharded.HALO * len(mine)
    for ny, world in ((8192, 8), (8192, 4), (8192, 2), (4096, 8), (4096, 4)):            # BASELINE configs[3] / configs[2] --scaling strong
        per = {sum(hi - lo for lo, hi in sharded.interleaved_partition(ny, world, r)) for r in range(world)}
        assert per == {ny // world} and (ny // world) % 512 == 0
    # too few chunks per rank / an uneven deal / one rank: the contiguous blocks
    for ny, world, chunk in ((600, 2, 256), (2048, 8, 256), (150, 2, 32), (203, 3, 32), (8192, 1, 256)):
        assert sharded.interleaved_chunks(ny, world, chunk) == [sharded.row_partition(ny, world, r) for r in range(world)]
    with pytest.raises(ValueError):
        sharded.interleaved_chunks(100, 2, 100)


def test_pole_window_marks_exactly_the_selected_pole_rows():
    from lagrangiancoherence_amd.engine import Engine
    for ny, world, window, halo in ((8192, 8, 256, False), (8192, 8, 256, True), (200, 3, 32, False), (200, 3, 32, True), (192, 2, 32, True)):
        for order in (1, 2, 3, 5):
            for r in range(world):
                g = np.asarray(sharded.interleaved_rows(ny, sharded.interleaved_partition(ny, world, r, window), with_halo=halo))
                row0, nyg = Engine.pole_window(g, len(g), ny, order)
                i = np.arange(len(g)) + row0
                assert np.array_equal((i < order) | (i >= nyg - order), (g < order) | (g >= ny - order))
    assert Engine.pole_window(np.arange(10), 10, 10, 3) == (0, 10)           # the whole grid is itself
    assert Engine.pole_window(np.arange(4, 9), 5, 20, 3) == (3, 11)          # an interior block: no pole rows
    for bad in ([1, 2, 3], [0, 5, 6], [0, 1, 2, 4, 9], [3, 2, 1]):              # part of a pole's rows / out of place / not ascending
        with pytest.raises(ValueError):
            Engine.pole_window(np.asarray(bad), len(bad), 10, 3)


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
