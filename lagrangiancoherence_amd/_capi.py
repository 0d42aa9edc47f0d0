"""ctypes binding of liblcs_hip.so -- one prototype per symbol of include/lcs_hip.h.

The library is the product; there is no Python or CPU fallback.  If it is not
built, loading fails loudly with the command that builds it.
"""
from __future__ import annotations

import ctypes as C
import os

HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(HERE, "liblcs_hip.so")

LC_VERSION = 104     # include/lcs_hip.h: the ABI these prototypes describe (checked against lc_version() in load())
LC_F32, LC_F64, LC_F64_WIND_F32, LC_F64_WIND_F32_LIN32 = 0, 1, 2, 3
LC_OK, LC_EINVAL, LC_EUNSUPPORTED, LC_EHIP, LC_ENOMEM, LC_ERCCL = 0, -1, -2, -3, -4, -5
LC_LAYOUT_REFERENCE, LC_LAYOUT_PHYSICAL = 0, 1
LC_X_CLAMP_POINT, LC_X_CYCLIC, LC_X_CLAMP_REFERENCE_OUTER = 0, 1, 2
LC_GRID_REGULAR, LC_GRID_GAUSSIAN = 0, 1
LC_F64_AUTO, LC_F64_EXACT_ORDER, LC_F64_FAST = 0, 1, 2
LC_EXACT_ORDER_MAX_SEEDS = 1 << 18

_vp, _i, _d, _sz = C.c_void_p, C.c_int, C.c_double, C.c_size_t

# name -> (restype, argtypes); mirrors include/lcs_hip.h declaration by declaration
PROTOTYPES = {
    "lc_version": (_i, []),
    "lc_last_error": (C.c_char_p, []),
    "lc_build_id": (C.c_char_p, []),
    "lc_ctx_create": (_i, [_i, C.POINTER(_vp)]),
    "lc_ctx_destroy": (_i, [_vp]),
    "lc_ctx_set_stream": (_i, [_vp, _vp]),
    "lc_ctx_use_own_stream": (_i, [_vp]),
    "lc_sync": (_i, [_vp]),
    "lc_ctx_set_lds_tiles": (_i, [_vp, _i]),
    "lc_ctx_set_sigma_march": (_i, [_vp, _i]),
    "lc_ctx_set_level_chunk": (_i, [_vp, _i]),
    "lc_ctx_get_level_chunk": (_i, [_vp, C.POINTER(_i)]),
    "lc_ctx_set_f64_fidelity": (_i, [_vp, _i]),
    "lc_ctx_get_f64_fidelity": (_i, [_vp, C.POINTER(_i)]),
    "lc_ctx_set_host_pipeline": (_i, [_vp, _i]),
    "lc_ctx_last_host_marks": (_i, [_vp, C.POINTER(C.c_double)]),
    "lc_copy_to_device": (_i, [_vp, _vp, _vp, C.c_size_t]),
    "lc_copy_to_host": (_i, [_vp, _vp, _vp, C.c_size_t]),
    "lc_ctx_set_host_cache": (_i, [_vp, _i]),
    "lc_ctx_trim": (_i, [_vp]),
    "lc_ctx_set_xcd_split": (_i, [_vp, _i]),
    "lc_ctx_set_flag_allreduce": (_i, [_vp, _vp, _vp]),
    "lc_ctx_last_advect_kernel": (C.c_char_p, [_vp]),
    "lc_ctx_last_advect_launches": (_i, [_vp]),
    "lc_ctx_last_sigma_kernel": (C.c_char_p, [_vp]),
    "lc_ctx_last_pack_kernel": (C.c_char_p, [_vp]),
    "lc_ctx_set_verify": (_i, [_vp, _i]),
    "lc_ctx_read_verify": (_i, [_vp, C.POINTER(C.c_uint), _i]),
    "lc_malloc": (_i, [_vp, _sz, C.POINTER(_vp)]),
    "lc_free": (_i, [_vp, _vp]),
    "lc_memcpy_h2d": (_i, [_vp, _vp, _vp, _sz]),
    "lc_memcpy_d2h": (_i, [_vp, _vp, _vp, _sz]),
    "lc_packed_elems": (_sz, [_i, _i, _i]),
    "lc_field_pack": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "lc_field_extrapolate": (_i, [_vp, _vp, _i, _i, _i, _i, _vp]),
    "lc_regrid_common_grid": (_i, [_vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _i, _vp]),
    "lc_spectral_truncate": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _i, _vp]),
    "lc_inspect_gridtype": (_i, [_vp, _i, C.POINTER(_i)]),
    "lc_advect": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _d, _d, _d, _d, _vp, _i, _vp, _i, _i, _i,
                       _d, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "lc_advect_from": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _d, _d, _d, _d, _vp, _i, _vp, _i, _i, _i, _vp, _vp,
                            _d, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "lc_advect_batch": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _d, _d, _d, _d, _vp, _i, _vp, _i, _i, _i, _vp, _vp,
                             _d, _i, _i, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp]),
    "lc_sample": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _d, _d, _d, _d, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "lc_sample_raw": (_i, [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _d, _d, _d, _d, _i, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp]),
    "lc_sigma": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _d, _d, _i, _i, _i, _i, _vp]),
    "lc_flowmap_gradient": (_i, [_vp, _vp, _vp, _i, _i, _i, _vp, _d, _d, _i, _vp]),
    "lc_fourth_order_derivative": (_i, [_vp, _vp, _i, _i, _i, _i, _i, _vp]),
    "lc_gaussian_filter": (_i, [_vp, _vp, _i, _i, _i, _d, _vp, _vp]),
    "lc_ridge_classify": (_i, [_vp, _vp, _vp, _vp, _vp, _vp, _sz, _d, _vp, _vp, _vp, _vp]),
    "lc_comm_unique_id": (_i, [_vp, _sz]),
    "lc_comm_create": (_i, [_vp, _i, _i, _vp, _sz, C.POINTER(_vp)]),
    "lc_comm_destroy": (_i, [_vp]),
    "lc_comm_count": (_i, [_vp, C.POINTER(_i), C.POINTER(_i)]),
    "lc_comm_flag_allreduce": (_i, [_vp, _vp, _sz]),
    "lc_halo_exchange": (_i, [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i]),
    "lc_common_grid": (_i, [C.POINTER(_i), C.POINTER(_i), _vp, _vp]),
    "lc_lcs_global_host": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _i, _i, _d, _i, _i, _d, _i, _i, _vp, _vp, _vp]),
    "lc_lcs_host": (_i, [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _i, _vp, _i,
                         _d, _i, _i, _i, _i, _i, _d, _i, _i, _vp, _vp, _vp, _vp, _vp]),
}

FLAG_ALLREDUCE_FN = C.CFUNCTYPE(_i, _vp, _vp, _sz)   # lc_flag_allreduce_fn of include/lcs_hip.h


class AdvectArgs(C.Structure):
    """``lc_advect_args`` of include/lcs_hip.h, field for field."""
    _fields_ = [("struct_size", _sz),
                ("packed_lin", _vp), ("packed_cub", _vp), ("packed_ext", _vp),
                ("u_raw", _vp), ("v_raw", _vp),
                ("dtype", _i), ("nt", _i), ("ny_f", _i), ("nx_f", _i),
                ("lat_min", _d), ("lat_max", _d), ("lon_min", _d), ("lon_max", _d),
                ("seed_lat_dev", _vp), ("ny", _i), ("seed_lon_dev", _vp), ("nx", _i),
                ("row0", _i), ("ny_global", _i),
                ("x_start", _vp), ("y_start", _vp),
                ("timestep", _d),
                ("settls_order", _i), ("interp_order", _i), ("cyclic_x", _i),
                ("t0", _i), ("nsteps", _i), ("n_members", _i), ("t0_stride", _i),
                ("x_out", _vp), ("y_out", _vp), ("traj_x", _vp), ("traj_y", _vp),
                ("fuse_levels_raw", _i)]


PROTOTYPES["lc_advect_ex"] = (_i, [_vp, C.POINTER(AdvectArgs)])

_lib = None


class LCSError(RuntimeError):
    """A C-ABI call returned LC_EHIP / LC_ERCCL."""


def load(path: str | None = None, import_torch: bool = True):
    """Load liblcs_hip.so and attach prototypes.  Raises if it is not built.  ``import_torch=False``: a torch-free host
    process (ctypes only, e.g. the one-call routes or tests/fake_rccl_driver.py) skips the torch import below."""
    global _lib
    if _lib is not None and path is None:
        return _lib
    # LCS_LIB: an experiment build of the same library (python -m lagrangiancoherence_amd.build --out ...)
    p = path or os.environ.get("LCS_LIB") or LIB_PATH
    # One HIP runtime per process: torch wheels bundle their own ROCm libraries, and loading ours
    # (linked against /opt/rocm) first leaves the process with two HSA runtimes and "no ROCm-capable
    # device".  Importing torch first makes our DT_NEEDED entries resolve to the copies torch loaded.
    if import_torch:
        try:
            import torch  # noqa: F401
        except ImportError:  # torch-free hosts use lc_lcs_host only
            pass
    if not os.path.exists(p):
        raise RuntimeError(
            f"{p} is missing: the HIP extension is the only compute path of this package "
            "(no CPU fallback). Build it with `python -m lagrangiancoherence_amd.build`.")
    lib = C.CDLL(p)
    for name, (res, args) in PROTOTYPES.items():
        fn = getattr(lib, name)  # AttributeError if the .so lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    if lib.lc_version() != LC_VERSION:   # an argument list changed between ABI versions: never call across them
        raise RuntimeError(f"{p} is ABI version {lib.lc_version()}, these bindings are for {LC_VERSION}: rebuild it with "
                           "`python -m lagrangiancoherence_amd.build --force`")
    if path is None:
        _lib = lib
    return lib


def check(status: int, lib=None):
    """Translate an lc_status into the exception the reference would raise."""
    if status == LC_OK:
        return
    lib = lib or load()
    msg = lib.lc_last_error().decode("utf-8", "replace")
    if status in (LC_EINVAL, LC_EUNSUPPORTED):
        raise ValueError(msg)
    if status == LC_ENOMEM:
        raise MemoryError(msg)
    raise LCSError(msg)
