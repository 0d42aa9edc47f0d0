"""bench.py's own rank launcher (`python bench.py --gpus N` with no torch.distributed.run around it), on CPU:
rank 0's JSON line is relayed alone, a failing rank takes the run down with its exit code, a run past its wall-clock
limit is stopped, and the init watchdog turns a stuck call into a message and a non-zero exit."""
import io
import os
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def _child(code: str):
    return [sys.executable, "-c", code]


def test_ranks_get_the_torchrun_environment_and_rank0_line_is_relayed_alone():
    out, err = io.StringIO(), io.StringIO()
    code = ("import os, json; r = int(os.environ['RANK']);"
            "print('noise from', r);"
            "print(json.dumps({'rank': r, 'world': os.environ['WORLD_SIZE'], 'local': os.environ['LOCAL_RANK'],"
            " 'addr': os.environ['MASTER_ADDR'], 'port': int(os.environ['MASTER_PORT']) > 0}))")
    rc = bench.spawn_ranks(3, _child(code), 60, out=out, err=err)
    assert rc == 0
    lines = [ln for ln in out.getvalue().splitlines() if ln.strip()]
    assert len(lines) == 1
    import json
    d = json.loads(lines[0])
    assert d == {"rank": 0, "world": "3", "local": "0", "addr": "127.0.0.1", "port": True}
    # the other ranks' output (and rank 0's non-JSON lines) went to the error stream, labelled
    assert "[rank 1]" in err.getvalue() and "[rank 2]" in err.getvalue() and "[rank 0] noise" in err.getvalue()


def test_a_failing_rank_stops_the_others_and_its_code_is_returned():
    out, err = io.StringIO(), io.StringIO()
    code = ("import os, sys, time; r = int(os.environ['RANK']);\n"
            "if r == 1: sys.exit(7)\n"
            "time.sleep(120)")
    t0 = time.monotonic()
    rc = bench.spawn_ranks(2, _child(code), 100, out=out, err=err)
    assert rc == 7 and time.monotonic() - t0 < 30
    assert "rank 1 exited with 7" in err.getvalue() and out.getvalue() == ""


def test_wall_clock_limit_stops_a_hung_run():
    out, err = io.StringIO(), io.StringIO()
    t0 = time.monotonic()
    rc = bench.spawn_ranks(2, _child("import time; time.sleep(120)"), 2, out=out, err=err)
    assert rc == 124 and time.monotonic() - t0 < 30
    assert "did not finish within 2 s" in err.getvalue()


def test_watchdog_exits_nonzero_with_a_message_and_can_be_cancelled():
    code = ("import sys, time; sys.path.insert(0, %r); import bench;"
            "w = bench.watchdog(30, 'quick call', 0); w.cancel();"
            "bench.watchdog(0.5, 'ncclCommInitRank', 3); time.sleep(30)" % ROOT)
    r = subprocess.run(_child(code), capture_output=True, text=True, timeout=60)
    assert r.returncode == 70
    assert "rank 3: ncclCommInitRank did not finish within" in r.stderr


def test_gpus_greater_than_one_without_enough_devices_is_refused_before_any_launch():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "LCS_BENCH_ONE_GPU")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "64"], capture_output=True, text=True,
                       timeout=300, env=env)
    assert r.returncode != 0 and "one rank per GPU" in r.stderr
