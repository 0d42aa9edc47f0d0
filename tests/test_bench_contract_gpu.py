"""bench.py prints exactly one JSON line carrying the driver's contract fields plus `roofline` and
`cpu_baseline` (miniature workloads here; the headline sizes are bench.py's defaults)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _bench(*args, env=None, launcher=()):
    r = subprocess.run([sys.executable, *launcher, os.path.join(ROOT, "bench.py"), *args], capture_output=True, text=True,
                       timeout=900, cwd=ROOT, env={**os.environ, **(env or {})})
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip().startswith("{")]
    assert len(lines) == 1, r.stdout
    return json.loads(lines[0])


def test_bench_prints_one_json_line_with_the_contract_fields():
    d = _bench("--gpus", "1", "--steps", "2", "--warmup", "1", "--seeds", "256", "--nt", "5")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
              "vs_baseline", "dtype", "data", "config", "roofline", "cpu_baseline"):
        assert k in d, k
    assert d["n_gpus"] == 1 and d["steps"] == 2 and d["warmup"] == 1 and d["higher_is_better"] is True
    assert d["unit"] == "particle-timesteps/s" and d["value"] > 0 and d["vs_baseline"] is None
    assert d["scaling"] == "weak" and d["dtype"] == "f32" and d["data"] == "synthetic"
    # the label says what actually ran: a 256^2 x 4-step variant, not the headline configuration
    assert d["config"]["workload"].startswith("variant of BASELINE configs[2]: 256x256 seeds")
    rf = d["roofline"]
    for k in ("bound", "kernel", "achieved", "peak", "unit", "frac", "traffic", "hbm", "algorithmic_GBps", "csrc_hash"):
        assert k in rf, k
    # the fused LDS-tile kernel is bound by vector-ALU instruction throughput; the fraction is a physical one
    # (a 256^2 miniature is far below the 2^23 seeds from which the two-seed kernel is the default: one seed per lane)
    assert rf["bound"] == "valu" and rf["kernel"] == "advect_lds_kernel<1, 4, true>"
    assert rf["unit"] == "TFLOP/s" and rf["peak"] == 157.3 and abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-12
    assert 0 < rf["frac"] <= 1 and 0 < rf["hbm"]["compulsory_frac"] <= 1
    # no committed counter summary matches a miniature variant: replayed fields are null, never stale numbers
    assert rf["traffic"] is None and rf["limiting_unit"] is None and rf["hbm"]["hbm_traffic_frac"] is None
    assert d["roofline_sigma"]["bound"] == "hbm" and 0 < d["roofline_sigma"]["frac"] <= 1
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["cores"] == 1 and cb["value"] > 0 and "sample" in cb
    # value = seeds x steps-per-pass x passes / wall time
    assert abs(d["value"] - 256 * 256 * 4 * 2 / (d["ms_per_step"] * 2 / 1e3)) / d["value"] < 1e-6


def test_bench_direct_kernel_is_named_as_such():
    d = _bench("--steps", "1", "--warmup", "0", "--seeds", "256", "--nt", "5", "--no-cpu-baseline", env={"LCS_LDS_TILES": "1"})
    assert d["roofline"]["kernel"] == "advect_lds2_kernel<4, true, 0>"      # forced: two seeds per lane whatever the size
    d = _bench("--steps", "1", "--warmup", "0", "--seeds", "256", "--nt", "5", "--settls", "0", "--no-cpu-baseline")
    assert d["roofline"]["kernel"] == "advect_kernel_f32<1>" and d["roofline"]["bound"] == "tcp"
    d = _bench("--steps", "1", "--warmup", "0", "--seeds", "256", "--nt", "5", "--no-cpu-baseline", env={"LCS_LDS_TILES": "0"})
    assert d["roofline"]["kernel"] == "advect_kernel_f32<1>"


@pytest.mark.parametrize("wk,extra,units", [
    ("c4", ["--seeds", "512", "--nt", "9"], 512 * 512 * 8),
    ("c5", ["--seeds", "256", "--nt", "14", "--members", "4"], 4 * 256 * 256 * 10),
])
def test_bench_c4_c5_workloads(wk, extra, units):
    d = _bench("--workload", wk, "--steps", "2", "--warmup", "1", "--no-cpu-baseline", *extra)
    assert d["scaling"] == "strong" and wk in d["metric"]
    assert abs(d["value"] - units * 2 / (d["ms_per_step"] * 2 / 1e3)) / d["value"] < 1e-6
    assert d["config"]["workload"].startswith("variant of BASELINE configs[%d]" % (3 if wk == "c4" else 4))


@pytest.mark.parametrize("wk,extra,units,scaling", [
    ("c3", ["--scaling", "weak", "--seeds", "256", "--nt", "5"], 512 * 256 * 4, "weak"),
    ("c3", ["--scaling", "strong", "--seeds", "256", "--nt", "5"], 256 * 256 * 4, "strong"),
    ("c4", ["--seeds", "512", "--nt", "9"], 512 * 512 * 8, "strong"),
    ("c5", ["--seeds", "256", "--nt", "14", "--members", "4"], 4 * 256 * 256 * 10, "strong"),
])
def test_bench_two_ranks_rehearsal_carries_a_halo_check(wk, extra, units, scaling):
    """N=2 over gloo with both ranks on GPU 0 (RCCL refuses two ranks on one device): the JSON line carries
    halo_check for the row-sharded workloads, and the exchanged rows equal the redundantly advected ones bit for
    bit; the ensemble workload shards members and exchanges nothing."""
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    d = _bench("--gpus", "2", "--steps", "1", "--warmup", "1", "--workload", wk, *extra,
               launcher=("-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr",
                         "127.0.0.1", "--master-port", str(port)),
               env={"LCS_BENCH_BACKEND": "gloo", "LCS_BENCH_ONE_GPU": "1"})
    if wk != "c5":
        # No second chance (round 4 repeated once here).  Ranks time-sharing one GPU run with the wave-state audit on
        # (lc_ctx_set_verify), so a mismatch arrives with its own explanation: the seeds, the call that is off, and whether
        # any wave's LDS tile or slot changed under it.
        from tests._multiproc import judge_halo_check
        judge_halo_check(d["halo_check"])
        assert d["halo_check"]["wave_state_audit"]["rank0"]["audited"] > 0      # the audit ran on the kernel that was timed
    assert d["n_gpus"] == 2 and d["scaling"] == scaling
    if wk == "c5":
        assert "halo_check" not in d
    else:
        hc = d["halo_check"]
        assert hc["timed_path"] == "torch.distributed" and hc["timed_path_ok"] is True
    assert abs(d["value"] - units / (d["ms_per_step"] / 1e3)) / d["value"] < 1e-6


def test_bench_two_ranks_strong_scaling_takes_the_interleaved_chunks():
    """--partition interleaved, strong scaling over N > 1 ranks whose 256-row chunks deal out evenly, two or more per rank (here
    1024 x 1024 seeds over 2 ranks: chunks 0, 2 and 1, 3): bench.py deals the interleaved chunks (sharded.interleaved_chunks), advects a rank's chunks in
    one lc_advect and exchanges every chunk's halo rows in one batch (a ring); the halo check -- every chunk's window equals the
    same rows advected redundantly, bit for bit -- is in the line, and a corrupted row is reported there.  The default
    (measured faster at 8 ranks: profiles/r06/shard_costs_*.jsonl) stays the row blocks and the line exchange."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(LCS_BENCH_BACKEND="gloo", LCS_BENCH_ONE_GPU="1")
    base = [sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1", "--workload", "c4",
            "--seeds", "1024", "--nt", "5"]

    def run(extra=(), **more):
        r = subprocess.run(base + list(extra), capture_output=True, text=True, timeout=900, cwd=ROOT, env={**env, **more})
        assert r.returncode == 0, r.stderr[-3000:]
        return json.loads([ln for ln in r.stdout.splitlines() if ln.strip()][-1]), r.stderr
    d, _ = run(["--partition", "interleaved"])
    hc = d["halo_check"]
    from tests._multiproc import judge_halo_check
    judge_halo_check(hc)
    assert "interleaved chunks" in hc["timed_path"] and hc["timed_path_ok"] is True and hc["chunks_per_rank"] == 2
    assert d["config"]["partition"].startswith("interleaved") and "interleaved chunks" in d["config"]["workload"]
    assert d["per_rank"][0]["rows"] == [[0, 256], [512, 768]] and d["per_rank"][1]["rows"] == [[256, 512], [768, 1024]]
    assert abs(d["value"] - 1024 * 1024 * 4 / (d["ms_per_step"] / 1e3)) / d["value"] < 1e-6
    d2, err = run(["--partition", "interleaved"], LCS_BENCH_CORRUPT_HALO_CHECK="1")
    m = d2["halo_check"]["mismatch_rank0"]
    assert d2["halo_check"]["timed_path_ok"] is False and m["n_seeds"] == 1 and m["window_rows"] == [3] and "halo check failed" in err
    d3, _ = run()                                         # the default: contiguous row blocks + the line exchange
    assert d3["halo_check"]["timed_path"] == "torch.distributed" and d3["per_rank"][1]["rows"] == [512, 1024]


def test_bench_reports_a_failed_halo_check_with_its_diagnosis():
    """A halo check that fails is REPORTED in the JSON line next to the number it discredits (not an assert, not a crash),
    with what differs and which of the two calls is unstable: here one interior element of rank 0's timed result is changed
    after the fact, so both re-runs agree with the redundant advect and not with the timed block."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(LCS_BENCH_BACKEND="gloo", LCS_BENCH_ONE_GPU="1", LCS_BENCH_CORRUPT_HALO_CHECK="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
                        "--seeds", "256", "--nt", "5"], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    d = json.loads([ln for ln in r.stdout.splitlines() if ln.strip()][-1])
    hc = d["halo_check"]
    assert hc["timed_path_ok"] is False
    m = hc["mismatch_rank0"]
    assert m["rows_of_extended_block"] == [3] and m["n_rows"] == 1 and m["nan"] is False and abs(m["max_abs_dx"] - 1.0) < 1e-3
    assert m["redundant_repeats"] is True and m["block_again_equals_redundant"] is True and m["block_again_equals_timed"] is False
    # seed by seed: which one, where it was computed in either call, and the direct-gather kernel as the arbiter
    assert m["n_seeds"] == 1 and m["timed_seeds_off_the_arbiter"] == 1 and m["redundant_seeds_off_the_arbiter"] == 0
    sd = m["seeds"][0]
    assert (sd["row_ext"], sd["col"]) == (3, 5) and sd["redundant_equals_direct"] and not sd["timed_equals_direct"]
    assert sd["in_timed_call"] == {"tile_row": 0, "tile_col": 0, "wave": 0, "lane_row": 3, "lane_col": 5}
    assert m["repeats_part_at"] is None and m["arbiter"].startswith("advect_kernel")
    assert m["wave_state_audit"]["tile_changed"] == 0 and m["wave_state_audit"]["audited"] > 0
    full = json.load(open(os.path.join(ROOT, m["file"])))
    assert full["seeds"][0]["row_ext"] == 3 and full["rank"] == 0
    from tests._multiproc import judge_halo_check
    with pytest.raises(AssertionError, match="no wave's state changed"):     # an injected corruption is not excused
        judge_halo_check(hc)
    assert "halo check failed" in r.stderr


@pytest.mark.parametrize("wk,extra,units", [
    ("c3", ["--seeds", "256", "--nt", "5"], 512 * 256 * 4),
    ("c5", ["--seeds", "256", "--nt", "14", "--members", "4"], 4 * 256 * 256 * 10),
])
def test_bench_gpus_2_launches_its_own_ranks(wk, extra, units):
    """Exactly the driver's command form, `python bench.py --gpus 2 ...` with NO launcher around it: bench.py starts the
    two ranks itself as fresh children (rehearsed over gloo with both on GPU 0), relays rank 0's single JSON line and
    reports every rank's own kernel times."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(LCS_BENCH_BACKEND="gloo", LCS_BENCH_ONE_GPU="1")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "1",
                        "--workload", wk, *extra], capture_output=True, text=True, timeout=900, cwd=ROOT, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [ln for ln in r.stdout.splitlines() if ln.strip()]
    assert len(lines) == 1, r.stdout                       # stdout carries the one JSON line and nothing else
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["backend"] == "gloo" and d["rccl_ranks"] is None   # (nccl: the world size RCCL spanned)
    assert [p["rank"] for p in d["per_rank"]] == [0, 1]
    for p in d["per_rank"]:
        assert p["kernel_ms"]["advect"] > 0 and 0 < p["roofline_frac"] <= 1
    assert d["halo_ms"] >= 0
    if wk == "c3":
        assert d["halo_check"]["timed_path_ok"] is True and d["per_rank"][1]["rows"] == [256, 512]
    assert abs(d["value"] - units / (d["ms_per_step"] / 1e3)) / d["value"] < 1e-6


def test_bench_gpus_2_fails_loudly_when_a_rank_cannot_start():
    """A rank that dies (here: an init limit of 0 seconds fires the watchdog) must end the whole run non-zero with a
    message, not hang it."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(LCS_BENCH_BACKEND="gloo", LCS_BENCH_ONE_GPU="1", LCS_BENCH_INIT_TIMEOUT="0.001")
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0",
                        "--seeds", "256", "--nt", "5"], capture_output=True, text=True, timeout=600, cwd=ROOT, env=env)
    assert r.returncode != 0
    assert "did not finish within" in r.stderr and not [ln for ln in r.stdout.splitlines() if ln.startswith("{")]


def test_secondary_workloads_block_of_the_default_run():
    """What the default one-GPU run appends under "secondary" (here on miniature inputs): the reference's default
    interpolation order and its trajectory output on configs[2]'s field, configs[1] (float64) at orders 1 and 3 -- each with
    its rate, stage times, the advect kernel that ran and a flop fraction."""
    import numpy as np
    import torch
    sys.path.insert(0, ROOT)
    import bench
    from lagrangiancoherence_amd import flows
    from lagrangiancoherence_amd.engine import Engine
    eng = Engine(0)
    u, v, lat, lon = flows.era5_like(nt=5, ny=72, nx=144)
    slat, slon = flows.seed_grid(160, 256, lat, lon)
    sec = bench.secondary_workloads(torch, flows, eng, eng.to_device(u, np.float32), eng.to_device(v, np.float32), lat, lon,
                                    eng.to_device(slat, np.float32), eng.to_device(slon, np.float32), float(slat[1] - slat[0]),
                                    float(slon[1] - slon[0]), steps=2, warmup=1, c2_n=128, c2_nt=9,
                                    long_nt=25, c3_long_nt=11, c4_n=320, c5=(128, 4, 20))
    # ... and (round 5) north_star's own target shape and the other BASELINE configurations on one GPU
    # ... and (round 6) SETTLS_order 0 (the library default: the direct-gather kernel), 1 and 2 on the headline field
    assert list(sec) == ["c3 K=0", "c3 K=1", "c3 K=2", "c3 order 3", "c3 order 3 K=0", "c3 return_traj", "c2", "c2 order 3", "c3 x 10 steps",
                         "c4 on one GPU", "c5 on one GPU"]
    assert sec["c3 K=0"]["kernel"] == "advect_kernel_f32<1>", sec["c3 K=0"]      # K = 0 has nothing to stage a tile for
    for K in (0, 1, 2):
        e = sec[f"c3 K={K}"]
        assert e["algorithmic_bytes_per_particle_timestep"] == bench.b_adv(K, 1, 4, 4) == 48 + 64 * K
        assert abs(e["algorithmic_over_hbm_peak"] * bench.HBM_PEAK_GBPS - e["algorithmic_GBps"]) < 1e-6 * e["algorithmic_GBps"]
    for name, d in sec.items():
        assert "error" not in d, (name, d)
        assert d["value"] > 0 and d["ms_per_step"] > 0 and 0 < d["frac"] <= 1 and d["advect_launches"] >= 1
        assert set(d["kernel_ms"]) == {"pack", "advect", "sigma"} and abs(d["advect_kernel_ms"] * d["advect_launches"] - d["kernel_ms"]["advect"]) < 1e-3
    assert sec["c5 on one GPU"]["advect_launches"] == 2         # 20 steps + 3 levels of stagger between the members in chunks of 16 levels
    assert "o3" in sec["c3 order 3"]["kernel"] or "<3," in sec["c3 order 3"]["kernel"]
    assert sec["c2"]["kernel"].startswith("advect_lds64_kernel") and sec["c2 order 3"]["kernel"].startswith("advect_lds64_o3_kernel")
    eng.close()


def test_host_route_and_drop_in_cases_of_the_default_run():
    """The default run's reference-facing cases on miniature inputs: the headline through lc_lcs_host (numpy in / out, the
    pipelined form against the serial one, with the C side's marks), BASELINE configs[0] and a reanalysis-shaped slab through
    the drop-in surface."""
    sys.path.insert(0, ROOT)
    import bench
    import numpy as np
    from lagrangiancoherence_amd import flows
    u, v, lat, lon = flows.era5_like(nt=41, ny=72, nx=144)
    slat, slon = flows.seed_grid(256, 256, lat, lon)
    h = bench.host_route_case(u, v, lat, lon, slat, slon, -900.0, 4, 1, 40, 0, reps=2)
    assert h["value"] > 0 and h["ms_per_call"] > 0 and h["serial_form_ms_per_call"] > 0 and len(h["calls_ms"]) == 3
    m = h["marks_ms"]
    assert 0 <= m["buffers_allocated"] <= m["uploads_and_launches_issued"] <= m["kernels_done"] <= m["results_down"] <= h["ms_per_call"] + 1.0
    assert abs(h["upload_MB"] - 2 * 41 * 72 * 144 * 4 / 1e6) < 1e-9 and abs(h["download_MB"] - 3 * 256 * 256 * 4 / 1e6) < 1e-9
    c1 = bench.config1_dropin(flows, with_oracle=False, reps=2)
    assert c1["LCS_call_ms"] > 0 and c1["parcel_propagation_return_traj_ms"] > 0 and "cpu_oracle_lcs_ms" not in c1
    slab = bench.era5_slab_dropin(flows, nt=4, reps=1)
    assert slab["parcel_propagation_ms"] > 0 and slab["particle_timesteps_per_s"] > 0 and abs(slab["input_MB"] - 2 * 4 * 720 * 1440 * 4 / 1e6) < 1e-9


def test_live_counter_passes_of_the_default_run():
    """What the default one-GPU run adds to `roofline` from its own `rocprofv3 --pmc` child passes (here on a miniature
    workload): the dispatched kernel's HBM-side bytes per launch and its per-unit figures, counters only, no tracing."""
    import shutil
    if not shutil.which("rocprofv3"):
        pytest.skip("no rocprofv3 on this box")
    sys.path.insert(0, ROOT)
    import bench
    small = ["--seeds", "256", "--nt", "5"]
    k = "advect_lds_kernel<1, 4, true>"
    t = bench.live_traffic([k], small)
    if "exited" in t.get("error", "") or "exceeded" in t.get("error", ""):
        pytest.skip("rocprofv3 --pmc cannot collect on this box (%s): bench.py then replays the committed summaries" % t["error"])
    assert "error" not in t, t
    # 2 steps of one launch each; at least the seeds' outputs (256^2 x 2 planes x 4 bytes) leave the chip, and not a whole GB
    assert t[k]["launches"] == 2 and 256 * 256 * 8 <= t[k]["traffic"] < 1e9
    u = bench.live_limiting_unit(k, 4, small)
    assert "error" not in u, u
    lu = u["limiting_unit"]
    assert 100 < lu["valu_instr_per_wave_timestep"] < 1000 and 0 < lu["valu_issue_frac"] <= 1 and 0 <= lu["lds_bank_conflict_frac"] <= 1
