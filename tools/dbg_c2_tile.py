"""Diagnostic build (-DLCS_STAMPS, LCS_LIB=...): how often the windows of configs[1] (float64, seeds = nodes) leave their LDS tile,
for the per-wave tiles and the workgroup-shared tile (LCS_F64_WG_TILE=1), with the ext image and from the raw planes."""
import ctypes as C, os, sys
import numpy as np
sys.path.insert(0, '.')
import torch
from lagrangiancoherence_amd import flows
from lagrangiancoherence_amd.engine import Engine
lib = C.CDLL(os.environ["LCS_LIB"])
for wg in ("0", "1"):
    os.environ["LCS_F64_WG_TILE"] = wg
    eng = Engine(0)
    ud, vd, lat, lon = flows.config2_on_device(torch, eng.device)
    for ext_image in (True, False):
        f = eng.prepare_field(ud, vd, lat, lon, 1, ext_image=ext_image)
        out, cause = (C.c_ulonglong * 8)(), (C.c_ulonglong * 4)()
        lib.lc_debug_read_stamps(out, 1); lib.lc_debug_read_cause(cause, 1)
        eng.advect(f, lat, lon, -900.0, 4, 1, True); torch.cuda.synchronize()
        lib.lc_debug_read_stamps(out, 1); lib.lc_debug_read_cause(cause, 1)
        n = np.array(list(out)[4:7], dtype=np.float64); c = np.array(list(cause), dtype=np.float64)
        line = f"{eng.last_advect_kernel():34s} iteration wave-samples with a lane outside the tile {100 * n[1] / max(n[0], 1):5.1f} %, lanes outside {100 * n[2] / max(64 * n[0], 1):5.2f} %"
        if c[0]:
            line += f"; Euler samples: {100 * c[1] / c[0]:5.1f} % of the wave-samples, {100 * c[2] / (64 * c[0]):5.2f} % of the lanes outside the kept tile"
        print(line, flush=True)
    eng.close()
