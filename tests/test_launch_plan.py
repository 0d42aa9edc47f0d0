"""lc_advect's launch bookkeeping (lagrangiancoherence_amd/csrc/launch_plan.h: level chunks, member-pair windows, XCD
tile order, pole blocks, outer-clamp restart) as a plain C++ program under AddressSanitizer + UBSan on the CPU
(the GPU pool has no sanitizer).  The header is the one advect.hip's launcher and kernels include."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
SRC = os.path.join(ROOT, "tests", "c", "launch_plan_test.cpp")


def test_launch_plan_invariants_under_asan_ubsan(tmp_path):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("g++ not available")
    exe = str(tmp_path / "launch_plan_test")
    subprocess.run([gxx, "-std=c++17", "-O1", "-g", "-Wall", "-Wextra", "-Werror", "-fsanitize=address,undefined",
                    "-fno-sanitize-recover=all", SRC, "-o", exe], check=True)
    r = subprocess.run([exe], capture_output=True, text=True, timeout=600,
                       env={**os.environ, "ASAN_OPTIONS": "detect_leaks=1", "UBSAN_OPTIONS": "print_stacktrace=1"})
    print(r.stdout, r.stderr)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "all checks passed" in r.stdout


def test_advect_hip_uses_the_tested_header():
    """The kernels and the launcher call the header's functions (a copy of the arithmetic kept beside it would make the
    CPU test vacuous)."""
    src = open(os.path.join(ROOT, "lagrangiancoherence_amd", "csrc", "advect.hip")).read()
    for fn in ("lcplan::tile_of_block", "lcplan::xcd_grid", "lcplan::pole_rows", "lcplan::pole_row", "lcplan::level_chunk",
               "lcplan::n_chunks", "lcplan::chunk_levels", "lcplan::member_groups", "lcplan::member_window",
               "lcplan::member_steps", "lcplan::group_levels", "lcplan::outer_restart"):
        assert fn in src, fn
