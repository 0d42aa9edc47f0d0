"""CPU-side checks of the C-ABI boundary: the library builds/loads here (hipcc
cross-compiles gfx950 without a GPU), exports every symbol include/lcs_hip.h
declares, and the ctypes prototypes cover exactly that set.  No compute calls."""
import os
import re

import pytest

from lagrangiancoherence_amd import _capi, build

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "lcs_hip.h")


def declared_symbols():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(lc_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
    build.build_library(verbose=False)
    return _capi.load()


def test_header_and_prototypes_agree():
    assert declared_symbols() == sorted(_capi.PROTOTYPES)


def test_library_exports_every_declared_symbol(lib):
    for name in declared_symbols():
        assert hasattr(lib, name), name


def test_version_and_error_string(lib):
    assert lib.lc_version() == 104 == _capi.LC_VERSION
    assert isinstance(lib.lc_last_error(), bytes)


def test_packed_elems_is_pure_arithmetic(lib):
    assert lib.lc_packed_elems(97, 720, 1440) == 97 * 723 * 1443 * 2
    assert lib.lc_packed_elems(0, 4, 4) == 0


def test_argument_validation_needs_no_gpu(lib):
    # null context is rejected before any HIP call
    assert lib.lc_sync(None) == _capi.LC_EINVAL
    assert b"null context" in lib.lc_last_error()
    with pytest.raises(ValueError):
        _capi.check(lib.lc_advect(None, None, None, None, 0, 2, 4, 4, 0., 1., 0., 1., None, 1, None, 1, 0, 1,
                                  1.0, 0, 1, 1, 0, 1, None, None, None, None), lib)


def test_advect_args_structure_matches_the_library(lib):
    """lc_advect_ex's argument structure: the ctypes mirror has the size the library was compiled with (the call checks
    struct_size before anything else), and a wrong size is refused with both numbers in the message."""
    import ctypes as C
    a = _capi.AdvectArgs(struct_size=C.sizeof(_capi.AdvectArgs))
    assert lib.lc_advect_ex(None, C.byref(a)) == _capi.LC_EINVAL and b"null context" in lib.lc_last_error()
    a.struct_size = C.sizeof(_capi.AdvectArgs) - 8
    assert lib.lc_advect_ex(None, C.byref(a)) == _capi.LC_EINVAL and b"struct_size" in lib.lc_last_error()
    assert lib.lc_advect_ex(None, None) == _capi.LC_EINVAL


def test_missing_library_fails_loudly(tmp_path):
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        _capi.load(str(tmp_path / "liblcs_hip.so"))


def test_engine_refuses_to_run_without_gpu():
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    from lagrangiancoherence_amd.engine import Engine
    with pytest.raises(RuntimeError, match="no CPU path"):
        Engine(0)


def test_local_rccl_declarations_match_the_installed_header():
    """csrc/halo.hip states the few RCCL declarations it uses (so the library builds without RCCL headers);
    where <rccl/rccl.h> is installed, the values must agree."""
    hdr = "/opt/rocm/include/rccl/rccl.h"
    if not os.path.exists(hdr):
        pytest.skip("no RCCL header on this machine")
    text = open(hdr).read()
    assert re.search(r"#define\s+NCCL_UNIQUE_ID_BYTES\s+128\b", text)
    assert re.search(r"ncclSuccess\s*=\s*0\b", text)
    assert re.search(r"ncclFloat32\s*=\s*7\b", text) and re.search(r"ncclFloat64\s*=\s*8\b", text)
    src = open(os.path.join(ROOT, "lagrangiancoherence_amd", "csrc", "halo.hip")).read()
    assert "#include <rccl" not in src
    assert "ncclFloat32 = 7, ncclFloat64 = 8" in src and "char internal[128]" in src


def test_context_options_reject_bad_arguments(lib):
    assert lib.lc_ctx_set_lds_tiles(None, 1) == _capi.LC_EINVAL
    assert lib.lc_ctx_last_advect_kernel(None) == b""


def test_common_grid_is_numpy_linspace(lib):
    """LCS/LCS.py:107-108: lats = np.linspace(-89.75, 89.75, 360), lons = np.linspace(-180, 179.5, 721), bit for bit."""
    import ctypes as C
    import numpy as np
    ny, nx = C.c_int(), C.c_int()
    assert lib.lc_common_grid(C.byref(ny), C.byref(nx), None, None) == 0 and (ny.value, nx.value) == (360, 721)
    lat, lon = np.empty(360), np.empty(721)
    lib.lc_common_grid(None, None, lat.ctypes.data_as(C.c_void_p), lon.ctypes.data_as(C.c_void_p))
    assert np.array_equal(lat, np.linspace(-89.75, 89.75, 180 * 2)) and np.array_equal(lon, np.linspace(-180, 179.5, 360 * 2 + 1))


def test_new_entry_points_validate_before_touching_a_device(lib):
    import ctypes as C
    assert lib.lc_regrid_common_grid(None, None, 0, 1, 2, 2, None, None, None, 1, None, 1, None) == _capi.LC_EINVAL
    assert lib.lc_spectral_truncate(None, None, 0, 1, 8, 16, 4, 0, None) == _capi.LC_EINVAL
    assert lib.lc_lcs_global_host(None, None, None, 0, 2, 4, 4, None, None, 1, 20, -900.0, 4, 3, 0.0, 1, 0,
                                  None, None, None) == _capi.LC_EINVAL
    assert b"null context" in lib.lc_last_error()


def test_x_boundary_mode_defaults():
    """cyclic_xboundary=False means the reference's own outer-product clamp (Q9), for a row block too (lc_advect then
    needs the flag all-reduce of the sharded driver and refuses the call without it: nothing falls back silently to
    the per-point clamp); explicit choices are honoured."""
    from lagrangiancoherence_amd.engine import x_boundary_mode
    assert x_boundary_mode(True) == _capi.LC_X_CYCLIC == 1
    assert x_boundary_mode(False) == _capi.LC_X_CLAMP_REFERENCE_OUTER == 2
    assert x_boundary_mode(False, whole_grid=False) == _capi.LC_X_CLAMP_REFERENCE_OUTER
    assert x_boundary_mode(False, "pointwise") == _capi.LC_X_CLAMP_POINT == 0 and x_boundary_mode(False, "reference_outer", whole_grid=False) == 2
    with pytest.raises(ValueError):
        x_boundary_mode(False, "nearest")


def test_csrc_hash_follows_code_not_comments(tmp_path, monkeypatch):
    """profiles/ summaries are stamped with build.csrc_hash(); bench.py replays their counters only on a match.
    The hash covers the code of csrc/ + the public header: comments and white space do not change it, code does."""
    import shutil
    from lagrangiancoherence_amd import build
    h0 = build.csrc_hash()
    assert len(h0) == 16 and h0 == build.csrc_hash()
    assert build._strip_comments('a = 1; // x\n/* y */ s = "// kept";   b') == 'a = 1; s = "// kept"; b'
    csrc = tmp_path / "pkg" / "csrc"
    shutil.copytree(build.CSRC, csrc)
    (tmp_path / "include").mkdir()
    shutil.copy(os.path.join(os.path.dirname(build.HERE), "include", "lcs_hip.h"), tmp_path / "include" / "lcs_hip.h")
    monkeypatch.setattr(build, "CSRC", str(csrc))
    monkeypatch.setattr(build, "HERE", str(tmp_path / "pkg"))
    assert build.csrc_hash() == h0
    f = csrc / "sigma.hip"
    f.write_text(f.read_text() + "\n// a trailing remark\n\n")
    assert build.csrc_hash() == h0
    f.write_text(f.read_text() + "\nstatic int lc_unused_marker = 1;\n")
    assert build.csrc_hash() != h0


def test_bench_replays_counters_only_from_summaries_of_the_running_code(tmp_path, monkeypatch):
    """bench.py's `traffic` / `limiting_unit` are replayed from profiles/*/*_pmc_*.json -- only when the summary's
    csrc_hash equals the running code's and the workload matches; a kernel name matches its templated spelling."""
    import json
    import sys
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    prof = tmp_path / "profiles" / "rXX"
    prof.mkdir(parents=True)
    wl = {"workload": "c3", "seeds": 64, "nt": 5, "order": 1, "K": 4, "dtype": "f32"}
    (prof / "x_pmc_traffic.json").write_text(json.dumps({"workload": wl, "csrc_hash": "abc", "kernels": {
        "advect_lds2_kernel<4, true>": {"hbm_bytes_per_launch": 123.0},
        "sigma_march_kernel_f32<20, 0>": {"hbm_bytes_per_launch": 7.0}}}))
    (prof / "x_pmc_sq_tcp.json").write_text(json.dumps({"workload": wl, "csrc_hash": "abc", "kernels": {
        "advect_lds2_kernel<4, true>": {"derived": {"valu_issue_frac": 0.9, "none": None}}}}))
    monkeypatch.setattr(bench, "ROOT", str(tmp_path))
    got = bench.stamped_counters("advect_lds2_kernel<4, true>", wl, "abc")
    assert got["traffic"] == 123.0 and got["limiting_unit"] == {"valu_issue_frac": 0.9}
    assert got["traffic_source"].endswith("x_pmc_traffic.json")
    assert bench.stamped_counters("sigma_march_kernel_f32", wl, "abc")["traffic"] == 7.0      # name<template args>
    assert bench.stamped_counters("advect_lds2_kernel<4, true>", wl, "other-code") == {}       # code changed
    assert bench.stamped_counters("advect_lds2_kernel<4, true>", dict(wl, seeds=128), "abc") == {}   # another workload
    assert bench.stamped_counters("advect_lds_kernel<1, 4, true>", wl, "abc") == {}            # another kernel
