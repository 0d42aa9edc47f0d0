"""Diagnostic build (-DLCS_STAMPS): how often the float64 LDS-tile kernel leaves its tile on BASELINE config 2."""
import ctypes as C, os, sys, numpy as np
sys.path.insert(0, '.')
import torch
from lagrangiancoherence_amd import flows
from lagrangiancoherence_amd.engine import Engine
eng = Engine(0)
lib = C.CDLL(os.environ["LCS_LIB"])
u, v, lat, lon = flows.config2()
ud, vd = eng.to_device(u, np.float64), eng.to_device(v, np.float64)
lat_d, lon_d = eng.to_device(lat, np.float64), eng.to_device(lon, np.float64)
f = eng.prepare_field(ud, vd, lat, lon, 1)
cause = (C.c_ulonglong * 4)()
eng.advect(f, lat_d, lon_d, -900.0, 4, 1, True); torch.cuda.synchronize()
print("kernel", eng.last_advect_kernel())
lib.lc_debug_read_cause(cause, 1)
c = np.array(list(cause), dtype=np.float64)
print(f"wave-samples {c[0]:.0f}; with a lane outside the tile {100 * c[1] / c[0]:.1f} %; lanes outside {100 * c[2] / (64 * c[0]):.2f} %; "
      f"re-anchors per wave-level {c[3] / (c[0] / 4):.3f}")
