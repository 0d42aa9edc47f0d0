"""No profiler invocation of this repository may sit on a GPU lease.

Round 5 lost metered GPU time three times to `rocprofv3 --pmc <too many SQ counters>` -> "error code 38: Request exceeds the
capabilities of the hardware to collect" -> `rocprofv3 caught signal 6` -> a stuck child.  Held here, on the CPU:
  * every script under tools/ starts rocprofv3 through tools/rocprof_guarded.sh (a `timeout -k` around rocprofv3 itself);
  * bench.py's own child passes carry a wall-clock limit and kill the child's process group;
  * every counter set anyone may hand to `--pmc` -- tools/pmc_sets.txt, bench.py's TRAFFIC_SETS / UNIT_SETS -- stays inside what
    one pass accepted on gfx950 this round: <= 8 SQ / GRBM counters, <= 5 TCP / TCC / TA counters, FETCH_SIZE and
    WRITE_SIZE alone in their passes.
"""
import glob
import inspect
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
MAX_SQ, MAX_TC = 8, 5


def _sets():
    import bench
    out = [("tools/pmc_sets.txt:%d" % (i + 1), ln.split())
           for i, ln in enumerate(open(os.path.join(ROOT, "tools", "pmc_sets.txt"))) if ln.strip()]
    out += [("bench.TRAFFIC_SETS[%d]" % i, list(s)) for i, s in enumerate(bench.TRAFFIC_SETS)]
    out += [("bench.UNIT_SETS[%d]" % i, list(s)) for i, s in enumerate(bench.UNIT_SETS)]
    return out


def test_every_counter_set_fits_one_pass():
    sets = _sets()
    assert len(sets) >= 8
    for where, names in sets:
        sq = [n for n in names if n.startswith(("SQ_", "GRBM_"))]
        tc = [n for n in names if n.startswith(("TCP_", "TCC_", "TA_", "TD_"))]
        size = [n for n in names if n in ("FETCH_SIZE", "WRITE_SIZE")]
        other = [n for n in names if n not in sq + tc + size]
        assert not other, f"{where}: counters of a block no pass of this repository has collected: {other}"
        assert len(sq) <= MAX_SQ, f"{where}: {len(sq)} SQ/GRBM counters (a pass took {MAX_SQ})"
        assert len(tc) <= MAX_TC, f"{where}: {len(tc)} TCP/TCC/TA counters (a pass took {MAX_TC})"
        assert not (sq and tc), f"{where}: SQ and TCP/TCC counters share a pass"
        assert not size or names == size and len(size) == 1, f"{where}: FETCH_SIZE / WRITE_SIZE go alone (TCC slots)"
        assert len(set(names)) == len(names), f"{where}: a counter twice"


def test_every_script_starts_rocprofv3_under_the_guard():
    guard = open(os.path.join(ROOT, "tools", "rocprof_guarded.sh")).read()
    assert re.search(r"^exec timeout -k \d+ .* rocprofv3 \"\$@\"$", guard, re.M), "the guard must wrap rocprofv3 itself in `timeout -k`"
    scripts = sorted(glob.glob(os.path.join(ROOT, "tools", "*.sh")))
    assert scripts
    for path in scripts:
        if path.endswith("rocprof_guarded.sh"):
            continue
        for n, line in enumerate(open(path), 1):
            code = line.split("#", 1)[0]
            if re.search(r"(^|[\s;(&|])rocprofv3(\s|$)", code):
                raise AssertionError(f"{os.path.relpath(path, ROOT)}:{n}: rocprofv3 started outside tools/rocprof_guarded.sh")
            if "rocprof_guarded.sh" in code:       # the program after `--` is the program itself (no env / bash -c / taskset hop)
                m = re.search(r"\s--\s+(\S+)", code)
                assert m and m.group(1) in ("python3", "python"), f"{os.path.relpath(path, ROOT)}:{n}: {m and m.group(1)!r} after `--`"


def test_no_python_tool_starts_rocprofv3_without_a_limit():
    for path in glob.glob(os.path.join(ROOT, "tools", "*.py")) + glob.glob(os.path.join(ROOT, "profiles", "*.py")):
        src = open(path).read()
        if "rocprofv3" in src and "subprocess" in src:
            assert "timeout" in src, f"{os.path.relpath(path, ROOT)} runs rocprofv3 as a child without a timeout"


def test_bench_child_passes_are_limited_and_killed_by_group():
    import bench
    for fn in (bench.pmc_passes, bench.kernel_trace_pass):
        src = inspect.getsource(fn)
        sig = inspect.signature(fn)
        assert 0 < sig.parameters["timeout_s"].default <= 300
        assert "p.wait(timeout=timeout_s)" in src and "os.killpg(p.pid, signal.SIGKILL)" in src and "start_new_session=True" in src
        # the program after `--` is the interpreter itself
        assert '"--", sys.executable' in src
    # tracing and counters never share a pass
    assert "--pmc" not in inspect.getsource(bench.kernel_trace_pass).split('"""')[2]
    assert "--kernel-trace" not in inspect.getsource(bench.pmc_passes).split('"""')[2]
