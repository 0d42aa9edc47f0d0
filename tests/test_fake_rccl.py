"""The C ABI's RCCL entry points (csrc/halo.hip: lc_comm_*, lc_halo_exchange, lc_comm_flag_allreduce) without a second
GPU: a recording loopback stand-in for librccl.so.1 (tests/c/fake_rccl.c) is loaded into a torch-free child process
before liblcs_hip.so, and the child (tests/fake_rccl_driver.py) runs ranks 0 / 1 / 2 of 1-, 2- and 3-rank
communicators one after the other, in float32 and float64: the exact (peer, byte offset, count, dtype, stream) of every
ncclSend / ncclRecv inside one GroupStart / GroupEnd, the exchanged rows against the unsharded result bit for bit,
GroupEnd reached on an injected error, and the all-reduce's (ncclUint32, ncclMax, context stream).  The real RCCL
over xGMI is what `tests/test_sharded_gpu.py::test_native_rccl_halo_exchange_two_gpus` runs wherever two GPUs exist."""
import os
import re
import shutil
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c", "fake_rccl.c")


def _build(tmp_path):
    gcc = shutil.which("gcc")
    if gcc is None or not os.path.exists("/opt/rocm/include/hip/hip_runtime_api.h"):
        pytest.skip("gcc or the HIP headers are not available")
    out = str(tmp_path / "librccl.so.1")
    subprocess.run([gcc, "-std=gnu99", "-O1", "-Wall", "-Wextra", "-Werror", "-Wno-unused-parameter", "-fPIC", "-shared",
                    "-I/opt/rocm/include", SRC, "-o", out, "-Wl,-soname,librccl.so.1", "-L/opt/rocm/lib", "-lamdhip64",
                    "-Wl,-rpath,/opt/rocm/lib"], check=True)
    return out


def test_stand_in_exports_every_rccl_symbol_halo_hip_resolves(tmp_path):
    lib = _build(tmp_path)
    halo = open(os.path.join(ROOT, "lagrangiancoherence_amd", "csrc", "halo.hip")).read()
    wanted = re.findall(r'LC_SYM\(\w+, "(nccl\w+)"\)', halo)
    assert len(wanted) == 11, wanted
    syms = subprocess.run(["nm", "-D", "--defined-only", lib], capture_output=True, text=True, check=True).stdout
    for s in wanted:
        assert re.search(rf"\bT {s}\b", syms), s
    soname = subprocess.run(["readelf", "-d", lib], capture_output=True, text=True, check=True).stdout
    assert "librccl.so.1" in soname


@pytest.mark.gpu
def test_rccl_entry_points_as_ranks_of_one_process(tmp_path):
    lib = _build(tmp_path)
    env = {k: v for k, v in os.environ.items() if k != "LD_PRELOAD"}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "fake_rccl_driver.py"), lib], capture_output=True, text=True,
                       timeout=600, env=env)
    print(r.stdout, r.stderr[-4000:])
    assert r.returncode == 0, r.stdout + r.stderr[-4000:]
    assert "all checks passed" in r.stdout
