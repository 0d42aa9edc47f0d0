"""Diagnostic build (-DLCS_STAMPS): share of a wave's cycles per phase of the two-seed order-1 kernel."""
import ctypes as C, os, sys, numpy as np
sys.path.insert(0, '.')
import torch
from lagrangiancoherence_amd import flows, _capi
from lagrangiancoherence_amd.engine import Engine
eng = Engine(0)
lib = C.CDLL(os.environ["LCS_LIB"])
u, v, lat, lon = flows.era5_like(nt=97)
slat, slon = flows.seed_grid(4096, 4096, lat, lon)
f = eng.prepare_field(u, v, lat, lon, 1)
out = (C.c_ulonglong * 8)()
eng.advect(f, slat, slon, -900.0, 4, 1, True); torch.cuda.synchronize()
lib.lc_debug_read_stamps(out, 1)
eng.advect(f, slat, slon, -900.0, 4, 1, True); torch.cuda.synchronize()
lib.lc_debug_read_stamps(out, 1)
t = np.array(list(out)[:4], dtype=np.float64)
waves = 4096 * 4096 / 128
print("kernel", eng.last_advect_kernel())
for n, x in zip(("anchor+tile load issue", "Euler sample (2 gathers, 2 seeds)", "tile wait + LDS write", "4 iterations"), t):
    print(f"{n:38s} {x / t.sum() * 100:5.1f} %   {x / waves / 96:8.0f} cycles per wave-level")
print("total per wave-level", t.sum() / waves / 96)
