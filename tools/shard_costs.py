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


def row_sharded(name, ny_global, nx, nt, weak):
    u, v, lat, lon = flows.era5_like_on_device(torch, eng.device, nt=nt)
    out = {"workload": name, "nt": nt, "nx": nx, "K": K, "order": 1, "per_N": {}}
    out["pack_ms"] = timed(lambda: eng.prepare_field(u, v, lat, lon, 1))
    field = eng.prepare_field(u, v, lat, lon, 1)
    for N in (1, 2, 4, 8):
        nyg = ny_global * N if weak else ny_global
        slat, slon = flows.seed_grid(nyg, nx, lat, lon)
        slat_d, slon_d = eng.to_device(slat, np.float32), eng.to_device(slon, np.float32)
        dlat, dlon = float(slat[1] - slat[0]), float(slon[1] - slon[0])
        ranks = []
        for r in range(N):
            lo, hi = sharded.row_partition(nyg, N, r)
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
    else:
        r = ensemble("c5 strong (64 x 2048^2 x 200)", 64, 2048, 264, 200)
    print(json.dumps(r), flush=True)
