"""What each rank of an N-GPU run would take, measured on ONE GPU: the prediction DESIGN.md section 5 states before any
multi-GPU hardware has run.   python tools/shard_costs.py [c3w|c3s|c4|c5 ...]  ->  one JSON object per workload.

For the row-sharded workloads every rank's block of seed rows is advected (and its sigma rows computed) on its own, with
HIP events, exactly as bench.py's member_pass does it (halo buffers included, exchange excluded); the pack is timed once
(every rank packs the whole replicated wind).  For the ensemble workload each rank's members go through
sharded.ensemble_advect as they would there.  Predicted step(N) = pack + max over ranks (advect + sigma) + exchange
latency (a constant the caller adds); efficiency = step(1) / (N * step(N)) for strong scaling, step(1) / step(N) for weak."""
import json
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lagrangiancoherence_amd import flows, sharded                      # noqa: E402
from lagrangiancoherence_amd.engine import Engine                       # noqa: E402

eng = Engine(0)
K, dt = 4, -900.0
REPS = 3

# ---- a cost-weighted row partition: measured in round 5 and NOT adopted (profiles/r05/shard_costs_balanced.txt) ------------
# Cutting the rows by measured per-band cost equalises the ranks' times -- and raises every one of them: a rank's cost is
# not proportional to its rows at this granularity (1024 rows of C4 are 16 tile rows = 2 per XCD and 2^23 seeds, the two-seed
# kernel's threshold; 1078 rows are 17 tile rows, 959 rows fall to the one-seed kernel).  C4 on 8 ranks: 0.70 uniform, 0.67 cut by cost.


def balanced_bounds(weights, world: int):
    """``world`` contiguous row blocks [(lo, hi), ...] of a grid with per-row costs ``weights`` whose total costs are as
    equal as whole rows allow: the k-th cut is the row at which the running cost passes k / world of the total.  Every
    block keeps at least HALO rows (what the neighbour exchange needs).  A row's advection cost is not uniform over
    latitude -- jets and the polar convergence of the meridians stretch a wave's patch beyond its LDS tile there -- and
    the step time of a row-sharded run is its slowest rank's (C4 on 8 ranks 11.3 ... 14.8 ms)."""
    import numpy as np
    w = np.asarray(weights, dtype=np.float64)
    ny = int(w.size)
    if ny < world * sharded.HALO or not np.all(np.isfinite(w)) or not np.all(w > 0):
        raise ValueError("weights: one positive finite cost per row, at least HALO rows per rank")
    c = np.concatenate([[0.0], np.cumsum(w)])
    cuts = [0]
    for k in range(1, world):
        r = int(np.searchsorted(c, c[-1] * k / world, side="left"))
        r = min(max(r, cuts[-1] + sharded.HALO), ny - (world - k) * sharded.HALO)    # room for this block and for every block still to come
        cuts.append(r)
    cuts.append(ny)
    return [(cuts[k], cuts[k + 1]) for k in range(world)]


def band_costs(engine, field, seed_lat_global, seed_lon, timestep, SETTLS_order=0, interp_order=1, cyclic_xboundary=True,
               n_bands: int = 64, levels: int = 32, t0: int = 0, rank: int = 0, world: int = 1, group=None):
    """Per-row advection cost of a seed grid, measured: the grid is cut into ``n_bands`` equal bands of rows, each band is
    advected through the first ``levels`` time levels on its own (HIP events; the kernels the real call would take, by
    ``Engine.concurrent_calls``), and a band's time is spread evenly over its rows.  With ``world`` > 1 the ranks share
    the bands (band b is timed by rank b % world) and all-gather the times, so every rank returns the SAME array -- what a
    cost-weighted partition needs.  A pilot outside any timed region: n_bands launches of a few levels each."""
    import numpy as np
    torch = engine.torch
    seed_lat_global = np.asarray(seed_lat_global, dtype=field.dtype)
    seed_lon = np.asarray(seed_lon, dtype=field.dtype)
    nyg = seed_lat_global.size
    n_bands = max(1, min(int(n_bands), nyg // max(sharded.HALO, 1)))
    edges = [round(b * nyg / n_bands) for b in range(n_bands + 1)]
    levels = max(1, min(int(levels), field.nt - 1 - t0))
    slat_d, slon_d = engine.to_device(seed_lat_global, field.dtype), engine.to_device(seed_lon, field.dtype)
    mine = {}
    with engine.concurrent_calls(nyg * seed_lon.size // max(world, 1), 1):
        for b in range(rank, n_bands, max(world, 1)):
            lo, hi = edges[b], edges[b + 1]
            call = lambda: engine.advect(field, slat_d[lo:hi], slon_d, timestep, SETTLS_order, interp_order, cyclic_xboundary,
                                         t0, levels, row0=lo, ny_global=nyg)
            call()                                      # (first touch of the band's tiles, allocator warm)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            call()
            e1.record()
            e1.synchronize()
            mine[b] = e0.elapsed_time(e1)
    if world > 1:
        import torch.distributed as dist
        parts = [None] * world
        dist.all_gather_object(parts, mine, group=group)
        mine = {k: v for d in parts for k, v in d.items()}
    w = np.empty(nyg, dtype=np.float64)
    for b in range(n_bands):
        w[edges[b]:edges[b + 1]] = max(mine[b], 1e-6) / (edges[b + 1] - edges[b])
    return w



def timed(fn, reps=REPS):
    fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def row_sharded(name, ny_global, nx, nt, weak, balance_levels=0):
    """balance_levels > 0: the rows are cut by measured cost (sharded.band_costs over that many levels, 64 bands)."""
    u, v, lat, lon = flows.era5_like_on_device(torch, eng.device, nt=nt)
    out = {"workload": name, "nt": nt, "nx": nx, "K": K, "order": 1, "per_N": {}}
    out["pack_ms"] = timed(lambda: eng.prepare_field(u, v, lat, lon, 1))
    field = eng.prepare_field(u, v, lat, lon, 1)
    for N in (1, 2, 4, 8):
        nyg = ny_global * N if weak else ny_global
        slat, slon = flows.seed_grid(nyg, nx, lat, lon)
        slat_d, slon_d = eng.to_device(slat, np.float32), eng.to_device(slon, np.float32)
        dlat, dlon = float(slat[1] - slat[0]), float(slon[1] - slon[0])
        weights = None
        if balance_levels and N > 1:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            weights = band_costs(eng, field, slat, slon, dt, K, 1, True, n_bands=64, levels=balance_levels)
            e1.record()
            torch.cuda.synchronize()
            out.setdefault("pilot_ms_all_bands_on_one_gpu", {})[N] = round(e0.elapsed_time(e1), 3)
        ranks = []
        for r in range(N):
            lo, hi = balanced_bounds(weights, N)[r] if weights is not None else sharded.row_partition(nyg, N, r)
            n_lo, n_hi = sharded.halo_rows(nyg, lo, hi) if N > 1 else (0, 0)
            box = {}

            def adv():
                box["r"] = eng.advect(field, slat_d[lo:hi], slon_d, dt, K, 1, True, 0, nt - 1, row0=lo, ny_global=nyg, halo=(n_lo, n_hi))
            a = timed(adv)
            x_ext, y_ext = box["r"]
            # (the halo rows are NaN here -- no exchange --: sigma's rows next to them come out NaN, its cost is the same)
            s = timed(lambda: eng.sigma(x_ext, y_ext, slat_d[lo - n_lo:lo - n_lo + x_ext.shape[0]], dlat, dlon, ny_global=nyg,
                                        in_row0=lo - n_lo, out_row0=lo, n_out_rows=hi - lo))
            ranks.append({"rank": r, "rows": [lo, hi], "advect_ms": round(a, 4), "sigma_ms": round(s, 4), "kernel": eng.last_advect_kernel()})
            del box, x_ext, y_ext
        worst = max(q["advect_ms"] + q["sigma_ms"] for q in ranks)
        out["per_N"][N] = {"ranks": ranks, "step_ms_without_exchange": round(out["pack_ms"] + worst, 4)}
        torch.cuda.empty_cache()
    s1 = out["per_N"][1]["step_ms_without_exchange"]
    for N, d in out["per_N"].items():
        d["efficiency_without_exchange"] = round(s1 / d["step_ms_without_exchange"] if weak else s1 / (N * d["step_ms_without_exchange"]), 4)
    return out


def interleaved(name, ny_global, nx, nt, redundant=False):
    """The PRODUCT's strong-scaling partition (sharded.interleaved_chunks, round 6), every rank's share on this one GPU as
    bench.py's member_pass_chunks issues it: ONE lc_advect over the rank's concatenated chunks (Engine.advect(global_rows=...)),
    lc_sigma once per chunk on its window; the ring exchange of the chunks' halo rows is the one term that cannot be measured
    here (`redundant`: the first cut instead -- every chunk's halo rows advected redundantly, no exchange: 1040 rows per rank
    of C4 where the exchange form has 1024).  step(N) = pack + max over ranks (advect + sigma); efficiency = step(1) / (N step(N))."""
    u, v, lat, lon = flows.era5_like_on_device(torch, eng.device, nt=nt)
    out = {"workload": name, "ny_global": ny_global, "nx": nx, "nt": nt,
           "partition": "sharded.interleaved_chunks" + (", halo rows advected redundantly" if redundant else ", halo rows exchanged"), "per_N": {}}
    out["pack_ms"] = timed(lambda: eng.prepare_field(u, v, lat, lon, 1))
    field = eng.prepare_field(u, v, lat, lon, 1)
    slat, slon = flows.seed_grid(ny_global, nx, lat, lon)
    slat_d, slon_d = eng.to_device(slat, np.float32), eng.to_device(slon, np.float32)
    dlat, dlon = float(slat[1] - slat[0]), float(slon[1] - slon[0])
    for N in (1, 2, 4, 8):
        ranks = []
        for r in range(N):
            chunks = sharded.interleaved_partition(ny_global, N, r)
            rows = np.asarray(sharded.interleaved_rows(ny_global, chunks, with_halo=redundant), dtype=np.int64)
            sl = slat_d[torch.as_tensor(rows, device=slat_d.device)].contiguous()
            box = {}

            def adv():
                box["xy"] = eng.advect(field, sl, slon_d, dt, K, 1, True, 0, nt - 1, ny_global=ny_global, global_rows=rows)
            a = timed(adv, 2 if N == 1 else REPS)
            x_all, y_all = box["xy"]
            if not redundant:      # the windows the exchange would deliver: here the chunk rows with NaN halo rows (timing only)
                nanrows = torch.full((2, 2 * sharded.HALO * len(chunks), nx), float("nan"), dtype=x_all.dtype, device=x_all.device)
                x_all, y_all = sharded._assemble_windows(x_all, y_all, chunks, ny_global, nanrows, nanrows)

            def sig():
                off = 0
                for lo, hi in chunks:
                    w0, w1 = sharded.chunk_window(ny_global, lo, hi)
                    eng.sigma(x_all[off:off + w1 - w0], y_all[off:off + w1 - w0], slat_d[w0:w1], dlat, dlon, ny_global=ny_global,
                              in_row0=w0, out_row0=lo, n_out_rows=hi - lo)
                    off += w1 - w0
            sg = timed(sig)
            ranks.append({"rank": r, "chunks": len(chunks), "own_rows": int(sum(h - l for l, h in chunks)), "advected_rows": int(rows.size),
                          "advect_ms": round(a, 4), "sigma_ms": round(sg, 4), "kernel": eng.last_advect_kernel()})
            del box, x_all, y_all
        worst = max(q["advect_ms"] + q["sigma_ms"] for q in ranks)
        adv_ms = [q["advect_ms"] for q in ranks]
        out["per_N"][N] = {"ranks": ranks, "step_ms_without_exchange": round(out["pack_ms"] + worst, 4),
                           "advect_spread": round((max(adv_ms) - min(adv_ms)) / (sum(adv_ms) / len(adv_ms)), 4)}
        torch.cuda.empty_cache()
    s1 = out["per_N"][1]["step_ms_without_exchange"]
    for N, d in out["per_N"].items():
        d["efficiency"] = round(s1 / (N * d["step_ms_without_exchange"]), 4)
    return out


def ensemble(name, n_members, seeds, nt, nsteps):
    u, v, lat, lon = flows.era5_like_on_device(torch, eng.device, nt=nt)
    out = {"workload": name, "members": n_members, "seeds": seeds, "nsteps": nsteps, "per_N": {}}
    out["pack_ms"] = timed(lambda: eng.prepare_field(u, v, lat, lon, 1))
    field = eng.prepare_field(u, v, lat, lon, 1)
    slat, slon = flows.seed_grid(seeds, seeds, lat, lon)
    slat_d, slon_d = eng.to_device(slat, np.float32), eng.to_device(slon, np.float32)
    dlat, dlon = float(slat[1] - slat[0]), float(slon[1] - slon[0])
    for N in (1, 2, 4, 8):
        ranks = []
        for r in sorted({0, N // 2, N - 1}):            # first, middle and last rank: the members differ in start level only
            mine = sharded.ensemble_partition(n_members, N, r)
            box = {}

            def adv():
                box["p"] = sharded.ensemble_advect(eng, field, slat_d, slon_d, dt, mine, nsteps, K, 1, True)
            a = timed(adv, 2 if N == 1 else REPS)
            pos = box["p"]
            s = timed(lambda: [eng.sigma(x, y, slat_d, dlat, dlon) for x, y in pos], 2)
            ranks.append({"rank": r, "members": [mine[0], mine[-1]], "advect_ms": round(a, 3), "sigma_ms": round(s, 3), "kernel": eng.last_advect_kernel(),
                          "launches": eng.last_advect_launches()})
            del box, pos
        worst = max(q["advect_ms"] + q["sigma_ms"] for q in ranks)
        out["per_N"][N] = {"ranks": ranks, "step_ms": round(out["pack_ms"] + worst, 3)}
        torch.cuda.empty_cache()
    s1 = out["per_N"][1]["step_ms"]
    for N, d in out["per_N"].items():
        d["efficiency"] = round(s1 / (N * d["step_ms"]), 4)
    return out


which = sys.argv[1:] or ["c3w", "c3s", "c4", "c5"]
for w in which:
    if w == "c3w":
        r = row_sharded("c3 weak (4096 rows per rank)", 4096, 4096, 97, True)
    elif w == "c3s":
        r = row_sharded("c3 strong (4096^2 in all)", 4096, 4096, 97, False)
    elif w == "c4":
        r = row_sharded("c4 strong (8192^2 x 384)", 8192, 8192, 385, False)
    elif w in ("c4p", "c4pr"):      # the product's interleaved chunks (strong scaling), 1 / 2 / 4 / 8 ranks; c4pr: halo rows redundant
        r = interleaved("c4 strong (8192^2 x 384), interleaved 256-row chunks", 8192, 8192, 385, w.endswith("r"))
    elif w in ("c3sp", "c3spr"):
        r = interleaved("c3 strong (4096^2 x 96), interleaved 256-row chunks", 4096, 4096, 97, w.endswith("r"))
    elif w.startswith("c4b"):       # c4b32: rows cut by the cost measured over the first 32 levels
        r = row_sharded(f"c4 strong, rows cut by measured cost ({w[3:]} pilot levels)", 8192, 8192, 385, False, int(w[3:]))
    elif w.startswith("c3sb"):
        r = row_sharded(f"c3 strong, rows cut by measured cost ({w[4:]} pilot levels)", 4096, 4096, 97, False, int(w[4:]))
    elif w.startswith("c3wb"):
        r = row_sharded(f"c3 weak, rows cut by measured cost ({w[4:]} pilot levels)", 4096, 4096, 97, True, int(w[4:]))
    else:
        r = ensemble("c5 strong (64 x 2048^2 x 200)", 64, 2048, 264, 200)
    print(json.dumps(r), flush=True)
