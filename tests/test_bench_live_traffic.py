"""bench.py's live_traffic(): the plumbing around the two `rocprofv3 --pmc` child passes, against a stand-in `rocprofv3`
on PATH that writes the counter CSV a real pass writes (columns as in rocprofv3's counter_collection.csv).  No GPU."""
import importlib.util
import os
import stat
import sys
import textwrap

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def bench():
    spec = importlib.util.spec_from_file_location("bench_under_test", os.path.join(ROOT, "bench.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def _fake_rocprof(tmp_path, monkeypatch, body):
    d = tmp_path / "bin"
    d.mkdir()
    exe = d / "rocprofv3"
    exe.write_text("#!" + sys.executable + "\n" + textwrap.dedent(body))
    exe.chmod(exe.stat().st_mode | stat.S_IXUSR)
    monkeypatch.setenv("PATH", str(d) + os.pathsep + os.environ["PATH"])


WRITER = """
    import os, sys
    a = sys.argv
    counter, out = a[a.index("--pmc") + 1], a[a.index("-d") + 1]
    if counter not in ("FETCH_SIZE", "WRITE_SIZE"):          # a pass over a set of SQ / TCP counters: one value each, two launches
        names = a[a.index("--pmc") + 1:a.index("--output-format")]
        fixed = {"SQ_WAVES": 100.0, "GRBM_GUI_ACTIVE": 8000.0, "SQ_INSTS_VALU": 832000.0, "SQ_INSTS_SALU": 300000.0, "SQ_INSTS_BRANCH": 20000.0,
                 "SQ_ACTIVE_INST_VALU": 128000.0, "SQ_LDS_IDX_ACTIVE": 64000.0, "SQ_LDS_BANK_CONFLICT": 16000.0, "TCC_HIT_sum": 3.0, "TCC_MISS_sum": 1.0}
        os.makedirs(os.path.join(out, "host"), exist_ok=True)
        with open(os.path.join(out, "host", "9_counter_collection.csv"), "w") as f:
            f.write('"Dispatch_Id","Kernel_Name","Counter_Name","Counter_Value"\\n')
            for launch in (1, 2):
                for n in names:
                    f.write('%d,"void (anonymous namespace)::advect_lds2_kernel<4, true, 0>((anonymous namespace)::AdvectArgs<float>)","%s",%f\\n'
                            % (launch, n, fixed.get(n, 0.0)))
        sys.exit(0)
    assert "--" in a and "--no-live-counters" in a and "--kernel-trace" not in a and os.getcwd() == "/tmp"
    os.makedirs(os.path.join(out, "host"), exist_ok=True)
    adv = "void (anonymous namespace)::advect_lds2_kernel<4, true, 0>((anonymous namespace)::AdvectArgs<float>)"
    sig = "void (anonymous namespace)::sigma_kernel<float, 1>((anonymous namespace)::SigmaArgs<float>)"
    vals = {"FETCH_SIZE": {adv: [100.0, 300.0], sig: [10.0]}, "WRITE_SIZE": {adv: [50.0, 70.0], sig: [4.0]}}[counter]
    with open(os.path.join(out, "host", "7_counter_collection.csv"), "w") as f:
        f.write('"Dispatch_Id","Kernel_Name","Counter_Name","Counter_Value"\\n')
        i = 0
        for k, vs in vals.items():
            for v in vs:
                i += 1
                f.write('%d,"%s","%s",%f\\n' % (i, k, counter, v))
        f.write('%d,"__amd_rocclr_copyBuffer","%s",1.0\\n' % (i + 1, counter))
"""


def test_name_matching(bench):
    m = bench.kernel_name_matches
    assert m("advect_lds2_kernel<4, true, 0>", "advect_lds2_kernel<4, true, 0>")
    assert m("advect_lds64_o3_kernel<4, true, false>", "advect_lds64_o3_kernel<4, true>")      # a defaulted trailing argument
    assert m("sigma_kernel<float, 1>", "sigma_kernel")
    assert not m("advect_lds2_kernel<4, true, 2>", "advect_lds2_kernel<4, true, 0>")
    assert not m("advect_lds2_o3_kernel<4, true, 0>", "advect_lds2_kernel<4, true, 0>")


def test_live_traffic_sums_two_passes(bench, tmp_path, monkeypatch):
    _fake_rocprof(tmp_path, monkeypatch, WRITER)
    r = bench.live_traffic(["advect_lds2_kernel<4, true, 0>", "sigma_kernel<float, 1>"], [])
    assert r["advect_lds2_kernel<4, true, 0>"] == {"traffic": (2 * 200.0 + 60.0) * 1024, "launches": 2}
    assert r["sigma_kernel<float, 1>"]["traffic"] == (2 * 10.0 + 4.0) * 1024
    assert r["source"].startswith("live:")


def test_live_traffic_reports_a_failed_or_hung_pass(bench, tmp_path, monkeypatch):
    _fake_rocprof(tmp_path, monkeypatch, "import sys\nsys.exit(3)\n")
    assert "exited 3" in bench.live_traffic(["k"], [])["error"]


def test_live_traffic_stops_a_hung_pass(bench, tmp_path, monkeypatch):
    _fake_rocprof(tmp_path, monkeypatch, "import time\ntime.sleep(60)\n")
    assert "exceeded" in bench.live_traffic(["k"], [], timeout_s=1.0)["error"]


def test_live_traffic_without_the_kernel_or_the_profiler(bench, tmp_path, monkeypatch):
    _fake_rocprof(tmp_path, monkeypatch, WRITER)
    assert "no launch" in bench.live_traffic(["advect_kernel"], [])["error"]
    monkeypatch.setenv("PATH", str(tmp_path))
    assert "not on PATH" in bench.live_traffic(["k"], [])["error"]


def test_live_limiting_unit_from_four_more_passes(bench, tmp_path, monkeypatch):
    _fake_rocprof(tmp_path, monkeypatch, WRITER)
    r = bench.live_limiting_unit("advect_lds2_kernel<4, true, 0>", 32, [])
    u = r["limiting_unit"]
    # 100 waves x 32 levels per launch; 8000 / 8 = 1000 cycles on 256 CUs
    assert u["valu_instr_per_wave_timestep"] == 260.0 and u["salu_instr_per_wave_timestep"] == 100.0
    assert u["valu_issue_frac"] == 0.5 and u["scalar_issue_frac"] == 1.25 and u["lds_active_frac"] == 0.25
    assert u["lds_bank_conflict_frac"] == 0.25 and u["l2_hit_frac"] == 0.75
    assert "error" in bench.live_limiting_unit("sigma_kernel", 1, [])


KT_WRITER = """
    import os, sys, json
    a = sys.argv
    assert "--kernel-trace" in a and "--stats" in a and "--pmc" not in a and os.getcwd() == "/tmp"
    assert a[a.index("--") + 1].endswith("python") or "python" in a[a.index("--") + 1]         # the interpreter itself after `--`
    out = a[a.index("-d") + 1]
    steps, warmup = int(a[a.index("--steps") + 1]), int(a[a.index("--warmup") + 1])
    os.makedirs(os.path.join(out, "host"), exist_ok=True)
    adv = "void (anonymous namespace)::advect_lds2_kernel<4, true, 0>((anonymous namespace)::AdvectArgs<float>)"
    pk = "void (anonymous namespace)::pack_fused_kernel<float>(float const*, float const*, float*, float*, int)"
    sg = "void (anonymous namespace)::sigma_march_kernel_f32<20, 0>((anonymous namespace)::SigmaArgs<float>)"
    rows, t = [], 1000
    for s in range(steps + warmup):
        slow = 2 if s < warmup else 1                      # the warm-up step's dispatches take twice as long
        for name, n, dur in ((pk, 1, 500000), (adv, 3, 2000000), (sg, 1, 60000)):
            for _ in range(n):
                rows.append((name, t, t + dur * slow)); t += dur * slow + 1000
    with open(os.path.join(out, "host", "5_kernel_trace.csv"), "w") as f:
        f.write('"Kind","Agent_Id","Kernel_Name","Start_Timestamp","End_Timestamp"\\n')
        for name, b, e in rows:
            f.write('"KERNEL_DISPATCH",1,"%s",%d,%d\\n' % (name, b, e))
        f.write('"KERNEL_DISPATCH",1,"void at::native::vectorized_elementwise_kernel<4>(int)",1,2\\n')
    with open(os.path.join(out, "host", "5_kernel_stats.csv"), "w") as f:
        f.write('"Name","Calls","TotalDurationNs","AverageNs","Percentage","MinNs","MaxNs","StdDev"\\n')
        for name in (adv, pk, sg):
            d = [e - b for k, b, e in rows if k == name]
            f.write('"%s",%d,%d,%f,1.0,%d,%d,0\\n' % (name, len(d), sum(d), sum(d) / len(d), min(d), max(d)))
        f.write('"void at::native::vectorized_elementwise_kernel<4>(int)",1,1,1.0,0.0,1,1,0\\n')
    print(json.dumps({"ms_per_step": 6.9, "steps": steps}))
"""


def test_kernel_trace_pass_reads_statistics_and_drops_the_warm_up_dispatches(bench, tmp_path, monkeypatch):
    """The live `rocprofv3 --kernel-trace --stats` child pass: per-kernel statistics as --stats gives them, and the average over
    the dispatches of the child's TIMED steps from the dispatch trace (the first `warmup` steps' dispatches dropped)."""
    _fake_rocprof(tmp_path, monkeypatch, KT_WRITER)
    kt, child, err = bench.kernel_trace_pass([], steps=5, warmup=1)
    assert err is None and child == {"ms_per_step": 6.9, "steps": 5}
    assert set(kt) == {"advect_lds2_kernel<4, true, 0>", "pack_fused_kernel<float>", "sigma_march_kernel_f32<20, 0>"}
    a = kt["advect_lds2_kernel<4, true, 0>"]
    assert a["calls"] == 18 and a["launches_per_step"] == 3 and a["timed_calls"] == 15
    assert abs(a["timed_avg_ms"] - 2.0) < 1e-9 and abs(a["avg_ms"] - (15 * 2.0 + 3 * 4.0) / 18) < 1e-6 and a["max_ms"] == 4.0
    assert abs(kt["pack_fused_kernel<float>"]["timed_avg_ms"] - 0.5) < 1e-9


def test_kernel_trace_pass_reports_a_failure(bench, tmp_path, monkeypatch):
    _fake_rocprof(tmp_path, monkeypatch, "import sys; sys.exit(134)\n")
    kt, child, err = bench.kernel_trace_pass([])
    assert kt == {} and child is None and "exited 134" in err


def test_binding_and_saved_profiles_round_trip(bench, tmp_path):
    """roofline.binding names the unit that limits the kernel with ITS fraction; --save-profiles writes summaries that
    stamped_counters() (the replay path) reads back."""
    lu = {"valu_issue_frac": 0.89, "tcp_lookups_per_cu_cycle": 0.7}
    assert bench.binding_of("valu", lu)["unit"] == "valu_issue" and bench.binding_of("valu", lu)["frac"] == 0.89
    assert bench.binding_of("tcp", lu)["unit"] == "vector_l1_lookups" and bench.binding_of("valu", None) is None
    wl = {"workload": "c3", "seeds": 4096, "nt": 97, "order": 1, "K": 4, "dtype": "f32"}
    k = "advect_lds2_kernel<4, true, 0>"
    out = tmp_path / "prof"
    bench.save_profiles(str(out), {"value": 1.0}, wl, "abc123", {k: {"calls": 18, "avg_ms": 2.1, "min_ms": 1.8, "max_ms": 2.6, "timed_avg_ms": 2.0}},
                        {k: {"traffic": 1.7e9, "launches": 6}, "source": "live"},
                        {"limiting_unit": {"valu_issue_frac": 0.89}, "counters": {"SQ_WAVES": 131072.0}}, k, "sigma_march_kernel_f32")
    assert sorted(os.listdir(out)) == ["c3_o1_bench_stdout.json", "c3_o1_kernel_stats.csv", "c3_o1_pmc_sq_tcp.json", "c3_o1_pmc_traffic.json"]
    import json
    t = json.load(open(out / "c3_o1_pmc_traffic.json"))
    assert t["csrc_hash"] == "abc123" and t["workload"] == wl and t["kernels"][k]["hbm_bytes_per_launch"] == 1.7e9
    q = json.load(open(out / "c3_o1_pmc_sq_tcp.json"))
    assert q["kernels"][k]["derived"] == {"valu_issue_frac": 0.89} and q["kernels"][k]["SQ_WAVES"] == 131072.0
    assert "timed_avg_ms" in open(out / "c3_o1_kernel_stats.csv").readline()
