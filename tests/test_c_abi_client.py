"""The C ABI used from plain C: tests/c/abi_smoke.c is compiled with gcc against include/lcs_hip.h and
liblcs_hip.so (no Python, no torch in that process).  Compiling and linking runs on CPU; running needs a GPU."""
import os
import shutil
import subprocess

import pytest

from lagrangiancoherence_amd import build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c", "abi_smoke.c")


def _compile(tmp_path):
    build.build_library(verbose=False)
    exe = str(tmp_path / "abi_smoke")
    libdir = os.path.join(ROOT, "lagrangiancoherence_amd")
    cmd = [shutil.which("gcc") or "gcc", "-std=c99", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), SRC,
           "-o", exe, "-L", libdir, "-llcs_hip", "-lm", f"-Wl,-rpath,{libdir}", "-Wl,-rpath,/opt/rocm/lib"]
    subprocess.run(cmd, check=True)
    return exe


def test_header_compiles_as_c99_and_links(tmp_path):
    assert os.path.exists(_compile(tmp_path))


@pytest.mark.gpu
def test_c_client_runs_known_answer(tmp_path):
    exe = _compile(tmp_path)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    print(r.stdout, r.stderr)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "0 mismatches" in r.stdout and "status -2" in r.stdout
    assert "continuation: 0 differences" in r.stdout and "lc_build_id = " in r.stdout
    assert "lc_advect_ex with raw planes: status 0, 0 differences from the packed_lin form, wrong struct_size refused: 1" in r.stdout
