#!/usr/bin/env python
"""Benchmark of the FTLE hot path on MI355X.

    python bench.py [--gpus N] [--steps K] [--warmup W]

One "step" = one pass of the whole hot path over one batch of synthetic input:
pack the wind series into the gather image, advect every seed through all time
levels (fused kernel), exchange the 2-row halo (N>1), compute sigma_max.  Inputs
(u, v, seeds) are resident in HBM before the timed region.

Workload (BASELINE.json configs[2], the 4096^2 grid the metric is quoted on):
4096x4096 seeds per GPU on a 720x1440 synthetic ERA5-like field, 97 time levels
(96 steps of 15 min), float32, SETTLS_order K=4, interp_order=1, cyclic.
N>1: the seed grid is (4096*N) x 4096, row-sharded (weak scaling), wind replicated.

Prints ONE JSON line on rank 0.
"""
from __future__ import annotations

import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBPS = 8000.0  # MI355X_MICROARCH.md: 8.0 TB/s spec (6.29 TB/s measured float4 copy)


def b_adv(K: int, order: int, s_p: int, s_f: int) -> int:
    """Algorithmic bytes per particle-timestep."""
    taps = 4 if order == 1 else 16
    return 4 * s_p + taps * (2 + 4 * K) * s_f


def measured_copy_peak(torch, nbytes: int = 1 << 30, reps: int = 10) -> float:
    """Device copy bandwidth of this box in GB/s (read + write bytes), the second HBM figure SURVEY 8d asks for."""
    src = torch.empty(nbytes // 4, dtype=torch.float32, device="cuda").normal_()
    dst = torch.empty_like(src)
    dst.copy_(src)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        dst.copy_(src)
    e1.record()
    torch.cuda.synchronize()
    return 2.0 * nbytes * reps / (e0.elapsed_time(e1) / 1e3) / 1e9


def pmc_traffic(kernel_prefix: str, workload: dict):
    """HBM bytes per launch of a kernel from the committed rocprofv3 --pmc summaries (profiles/*/
    *_pmc_traffic.json, written by profiles/summarize.py from separate FETCH_SIZE / WRITE_SIZE passes of
    this same command).  Counters cannot be read from inside the timed run; None if no summary matches."""
    import glob
    best = None
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "*_pmc_traffic.json"))):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("workload") != workload:
            continue
        for k, v in d.get("kernels", {}).items():
            if k.startswith(kernel_prefix) and "hbm_bytes_per_launch" in v:
                best = (v["hbm_bytes_per_launch"], os.path.relpath(f, ROOT))
    return best


def run_c2(args, torch, flows, Engine, local_rank):
    """BASELINE configs[1] on one GPU (a parity config; reported for the float64 path's rate)."""
    K, order = args.settls, args.order
    u, v, lat, lon = flows.config2()
    nt, ny, nx = u.shape
    eng = Engine(local_rank)
    ud, vd = eng.to_device(u, np.float64), eng.to_device(v, np.float64)
    lat_d, lon_d = eng.to_device(lat, np.float64), eng.to_device(lon, np.float64)
    dlat, dlon = float(lat[1] - lat[0]), float(lon[1] - lon[0])

    def one():
        m = [torch.cuda.Event(enable_timing=True) for _ in range(4)]
        m[0].record()
        f = eng.prepare_field(ud, vd, lat, lon, order, fuse_levels=args.fuse_levels)
        m[1].record()
        x, y = eng.advect(f, lat_d, lon_d, -900.0, K, order, True)
        m[2].record()
        s = eng.sigma(x, y, lat_d, dlat, dlon)
        m[3].record()
        return s, m
    for _ in range(args.warmup):
        one()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    marks = []
    for _ in range(args.steps):
        s, m = one()
        marks.append(m)
    torch.cuda.synchronize()
    el = time.perf_counter() - t0
    ms = {k: float(np.mean([m[i].elapsed_time(m[i + 1]) for m in marks])) for i, k in enumerate(("pack", "advect", "sigma"))}
    pts = ny * nx * (nt - 1)
    bytes_pts = b_adv(K, order, 8, 8)
    ach = pts * bytes_pts / (ms["advect"] / 1e3) / 1e9
    print(json.dumps({
        "metric": "particle-timesteps/sec, BASELINE configs[1] (float64)", "value": pts * args.steps / el,
        "unit": "particle-timesteps/s", "n_gpus": 1, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": el / args.steps * 1e3, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "f64", "data": "synthetic",
        "config": {"workload": f"BASELINE configs[1]: {ny}x{nx} seeds = field nodes, moving ideal vortex, {nt} levels, "
                               f"dt=-900 s, fp64", "SETTLS_order": K, "interp_order": order,
                   **({"fuse_levels": True} if args.fuse_levels else {})},
        "kernel_ms": ms,
        "roofline": {"bound": "hbm", "kernel": "advect_kernel<double,%d%s>" % (order, ",fused" if args.fuse_levels else ""), "achieved": ach,
                     "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": ach / HBM_PEAK_GBPS, "traffic": None,
                     "algorithmic_bytes_per_particle_timestep": bytes_pts},
    }), flush=True)


def issue_counters(kernel_prefix: str, workload: dict):
    """SQ/TCP-derived occupancy of the units that actually bound the kernel (committed --pmc summary)."""
    import glob
    for f in sorted(glob.glob(os.path.join(ROOT, "profiles", "*", "*_pmc_sq_tcp.json")), reverse=True):
        try:
            d = json.load(open(f))
        except Exception:
            continue
        if d.get("workload") != workload:
            continue
        for k, v in d.get("kernels", {}).items():
            if k.startswith(kernel_prefix) and "derived" in v:
                return {**{kk: round(vv, 4) for kk, vv in v["derived"].items()}, "source": os.path.relpath(f, ROOT)}
    return None


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--seeds", type=int, default=4096, help="seed rows per GPU and seed columns")
    ap.add_argument("--nt", type=int, default=97)
    ap.add_argument("--settls", type=int, default=4)
    ap.add_argument("--order", type=int, default=1)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--traj", action="store_true",
                    help="return_traj=True: also store the positions after every step (not the headline)")
    ap.add_argument("--wind-scale", type=float, default=1.0,
                    help="multiply the synthetic wind (stress case: stronger stretching; not the headline)")
    ap.add_argument("--fuse-levels", action="store_true",
                    help="c2 only: sample the fused image 2F[t]-F[t+1] once per SETTLS iteration in float64 too "
                         "(rounding-level differences from the reference's two-sample order; default keeps that order)")
    ap.add_argument("--field", default=None, metavar="NY,NX",
                    help="resolution of the synthetic wind field (default 720,1440 = 0.25 degrees; not the "
                         "headline when changed: probes other seed-to-node density ratios)")
    ap.add_argument("--workload", default="c3", choices=["c3", "c2"],
                    help="c3 (default, the headline): 4096^2 seeds on the 720x1440 fp32 flow; "
                         "c2: BASELINE configs[1], 1024^2 nodes, moving ideal vortex, 200 steps, fp64, N=1 only")
    args = ap.parse_args()

    import torch
    import torch.distributed as dist
    from lagrangiancoherence_amd import flows, sharded
    from lagrangiancoherence_amd.engine import Engine

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus:
        if world == 1 and args.gpus > 1:
            raise SystemExit("--gpus N>1 must be launched with torch.distributed.run (one rank per GPU)")
        args.gpus = world
    # LCS_BENCH_BACKEND=gloo + LCS_BENCH_ONE_GPU=1: rehearsal of the N>1 path with every rank on GPU 0
    # (RCCL refuses two ranks on one device); the driver's multi-GPU runs use nccl = RCCL over xGMI.
    backend = os.environ.get("LCS_BENCH_BACKEND", "nccl")
    if os.environ.get("LCS_BENCH_ONE_GPU"):
        local_rank = 0
    torch.cuda.set_device(local_rank)
    if world > 1:
        if backend == "nccl":
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend)

    if args.workload == "c2":
        return run_c2(args, torch, flows, Engine, local_rank)
    K, order, nt = args.settls, args.order, args.nt
    nsteps = nt - 1
    ny_local = nx = args.seeds
    ny_global = ny_local * world
    dt = -900.0

    # ---- synthetic input, then resident in HBM -------------------------------------------
    fny, fnx = (int(t) for t in args.field.split(",")) if args.field else (720, 1440)
    u, v, lat, lon = flows.era5_like(nt=nt, ny=fny, nx=fnx)       # float32, 720x1440 unless --field
    if args.wind_scale != 1.0:
        u, v = (u * np.float32(args.wind_scale)), (v * np.float32(args.wind_scale))
    slat, slon = flows.seed_grid(ny_global, nx, lat, lon)
    eng = Engine(local_rank)
    ud = eng.to_device(u, np.float32)
    vd = eng.to_device(v, np.float32)
    lo, hi = sharded.row_partition(ny_global, world, rank)
    n_lo, n_hi = sharded.halo_rows(ny_global, lo, hi)
    slat_d = eng.to_device(slat, np.float32)      # seeds resident too: the event brackets hold kernels only
    slon_d = eng.to_device(slon, np.float32)
    dlat, dlon = float(slat[1] - slat[0]), float(slon[1] - slon[0])
    torch.cuda.synchronize()

    # LCS_NATIVE_HALO=1: halo exchange through the C ABI (lc_halo_exchange, RCCL directly) instead of
    # torch.distributed point-to-point (the default; both are RCCL over xGMI with the nccl backend)
    comm = sharded.native_comm(eng, rank, world) if (world > 1 and os.environ.get("LCS_NATIVE_HALO")) else None
    ev = {k: [] for k in ("pack", "advect", "halo", "sigma")}

    def one_step(record: bool):
        marks = [torch.cuda.Event(enable_timing=True) for _ in range(5)]
        marks[0].record()
        field = eng.prepare_field(ud, vd, lat, lon, order)
        marks[1].record()
        res = eng.advect(field, slat_d[lo:hi], slon_d, dt, K, order, True, 0, nsteps, row0=lo,
                         ny_global=ny_global, halo=(n_lo, n_hi), return_traj=args.traj)
        x_ext, y_ext = res[0], res[1]
        marks[2].record()
        sharded.halo_exchange_into(x_ext, y_ext, n_lo, n_hi, rank, world, engine=eng, comm=comm)
        in_row0 = lo - n_lo
        marks[3].record()
        sig = eng.sigma(x_ext, y_ext, slat_d[in_row0:in_row0 + x_ext.shape[0]], dlat, dlon, ny_global=ny_global,
                        in_row0=in_row0, out_row0=lo, n_out_rows=hi - lo)
        marks[4].record()
        if record:
            ev_marks.append(marks)
        return sig

    ev_marks = []
    for _ in range(args.warmup):
        one_step(False)

    def fence():
        torch.cuda.synchronize()
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        sig = one_step(True)
    fence()
    elapsed = time.perf_counter() - t0
    if world > 1:
        tmax = torch.tensor([elapsed], dtype=torch.float64, device="cuda" if backend == "nccl" else "cpu")
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        elapsed = float(tmax.item())
    for marks in ev_marks:
        for i, k in enumerate(("pack", "advect", "halo", "sigma")):
            ev[k].append(marks[i].elapsed_time(marks[i + 1]))     # ms, on the launch stream
    ms = {k: float(np.mean(vv)) for k, vv in ev.items()}
    assert bool(torch.isfinite(sig).all()), "non-finite sigma in the benchmark output"

    pts_per_step = ny_global * nx * nsteps
    value = pts_per_step * args.steps / elapsed
    copy_gbps = measured_copy_peak(torch)                                       # after the timed region
    s_f = s_p = 4
    bytes_pts = b_adv(K, order, s_p, s_f)
    adv_s = ms["advect"] / 1e3
    achieved = (ny_local * nx * nsteps) * bytes_pts / adv_s / 1e9            # per GPU, dominant kernel
    sig_s = ms["sigma"] / 1e3
    sigma_gbps = (ny_local * nx) * 3 * s_p / sig_s / 1e9

    wl = {"seeds": args.seeds, "nt": nt, "order": order, "K": K, "dtype": "f32"}
    if args.field or args.wind_scale != 1.0 or args.traj:
        wl["variant"] = True     # no committed counter summary matches a non-headline variant
    tr_adv = pmc_traffic("advect_", wl) if world == 1 else None
    tr_sig = pmc_traffic("sigma_kernel", wl) if world == 1 else None
    out = {
        "metric": "particle-timesteps/sec (+ FTLE Mcells/sec) at 4096^2 seeds per GPU",
        "value": value,
        "unit": "particle-timesteps/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": elapsed / args.steps * 1e3,
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic",
        "config": {
            "workload": f"BASELINE configs[2]: {ny_local}x{nx} seeds per GPU (global {ny_global}x{nx}, row-sharded) on a "
                        f"{fny}x{fnx} synthetic ERA5-like wind series, {nt} levels ({nsteps} steps, dt=-900 s), fp32",
            "SETTLS_order": K, "interp_order": order, "cyclic_xboundary": True,
            **({"wind_scale": args.wind_scale} if args.wind_scale != 1.0 else {}),
            **({"field": [fny, fnx]} if args.field else {}),
            **({"return_traj": True} if args.traj else {}),
            "step": "pack + fused advect + halo exchange + sigma; u/v/seeds resident in HBM",
        },
        "advect_particle_timesteps_per_s": ny_global * nx * nsteps / adv_s,
        "ftle_mcells_per_s": ny_global * nx / sig_s / 1e6,
        "kernel_ms": ms,
        "roofline": {
            "bound": "hbm", "kernel": "advect_lds_kernel<%d,%d,true>" % (order, 4 if K == 4 else -1),
            "achieved": achieved, "peak": HBM_PEAK_GBPS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBPS,
            "measured_copy_peak": copy_gbps, "frac_of_measured_copy_peak": achieved / copy_gbps,
            "traffic": tr_adv[0] if tr_adv else None,
            "traffic_source": tr_adv[1] if tr_adv else None,
            "algorithmic_bytes_per_launch": (ny_local * nx * nsteps) * bytes_pts,
            "limiting_unit": issue_counters("advect_lds_kernel<%d," % order, wl) if world == 1 else None,
            "algorithmic_bytes_per_particle_timestep": bytes_pts,
            "note": "achieved = B_adv(K,order) x seeds x steps / HIP-event duration of the fused advect launch "
                    "(per GPU); the taps are served from L2/Infinity Cache, so this is an algorithmic, not an "
                    "HBM-traffic, figure (SURVEY 8d)",
        },
        "roofline_sigma": {
            "bound": "hbm", "kernel": "sigma_kernel_f32", "achieved": sigma_gbps, "peak": HBM_PEAK_GBPS,
            "unit": "GB/s", "frac": sigma_gbps / HBM_PEAK_GBPS, "frac_of_measured_copy_peak": sigma_gbps / copy_gbps,
            "traffic": tr_sig[0] if tr_sig else None,
            "algorithmic_bytes_per_cell": 3 * s_p,
        },
    }

    # ---- CPU baseline: the oracle (numpy+scipy port) on a bounded sample, rank 0, N=1 only ----
    if world == 1 and rank == 0 and not args.no_cpu_baseline:
        from oracle import lcs_oracle as O
        n_s, st_s = 1024, min(32, nsteps)
        sl, so = flows.seed_grid(n_s, n_s, lat, lon)
        c0 = time.perf_counter()
        O.lcs(u[:st_s + 1], v[:st_s + 1], lat, lon, timestep=dt, SETTLS_order=K, interp_order=order,
              cyclic_xboundary=True, seed_lat=sl, seed_lon=so)
        c1 = time.perf_counter()
        out["cpu_baseline"] = {
            "value": n_s * n_s * st_s / (c1 - c0), "unit": "particle-timesteps/s", "cores": 1, "kind": "port",
            "sample": f"{n_s}x{n_s} seeds x {st_s} steps of the same field and settings, advect+sigma, "
                      f"oracle/lcs_oracle.py (scipy.ndimage.map_coordinates + LAPACK SVD, single thread), "
                      f"{c1 - c0:.1f} s on a host with {os.cpu_count()} cores",
        }
    if rank == 0:
        print(json.dumps(out), flush=True)
    if world > 1:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
