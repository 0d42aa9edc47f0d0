"""Diagnostic build (-DLCS_STAMPS): share of a wave's cycles per phase of the two-seed order-1 kernel."""
import ctypes as C, os, sys, numpy as np
sys.path.insert(0, '.')
import torch
from lagrangiancoherence_amd import flows, _capi
from lagrangiancoherence_amd.engine import Engine
eng = Engine(0)
lib = C.CDLL(os.environ["LCS_LIB"])
NS = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
NT = int(sys.argv[2]) if len(sys.argv) > 2 else 97
u, v, lat, lon = flows.era5_like(nt=NT)
slat, slon = flows.seed_grid(NS, NS, lat, lon)
f = eng.prepare_field(u, v, lat, lon, 1)
out = (C.c_ulonglong * 8)()
redo = (C.c_ulonglong * 27)()
eng.advect(f, slat, slon, -900.0, 4, 1, True); torch.cuda.synchronize()
lib.lc_debug_read_stamps(out, 1)
lib.lc_debug_read_redo(redo, 1)
eng.advect(f, slat, slon, -900.0, 4, 1, True); torch.cuda.synchronize()
lib.lc_debug_read_stamps(out, 1)
t = np.array(list(out)[:4], dtype=np.float64)
waves = NS * NS / 128
cnt = np.array(list(out)[4:7], dtype=np.float64)
print(f"iteration wave-samples with a redo: {cnt[1] / cnt[0] * 100:.1f} %; seed-samples redone: {cnt[2] / (cnt[0] * 128) * 100:.2f} %")
print("kernel", eng.last_advect_kernel())
for n, x in zip(("anchor+tile load issue", "Euler sample (2 gathers, 2 seeds)", "tile wait + LDS write", "4 iterations"), t):
    print(f"{n:38s} {x / t.sum() * 100:5.1f} %   {x / waves / 96:8.0f} cycles per wave-level")
print("total per wave-level", t.sum() / waves / (NT - 1))

cause = (C.c_ulonglong * 4)()
lib.lc_debug_read_cause(cause, 1)
cz = np.array(list(cause), dtype=np.float64) / 2   # two launches accumulated
tot = NS * NS * (NT - 1) * 4.0
print(f"seed-samples outside the tile: in x {cz[0] / tot * 100:.2f} % (of which below {cz[2] / max(cz[0], 1) * 100:.0f} %), in y {cz[1] / tot * 100:.2f} % (below {cz[3] / max(cz[1], 1) * 100:.0f} %)")
lib.lc_debug_read_redo(redo, 1)
r = np.array(list(redo), dtype=np.float64).reshape(3, 3, 3)
print("share of iteration wave-samples with a redo (rows: |lat| 0-30, 30-60, 60-90; columns: first / middle / last third of the levels)")
print(np.round(100 * r[:, :, 1] / np.maximum(r[:, :, 0], 1), 1))
print("share of seed-samples redone"); print(np.round(100 * r[:, :, 2] / np.maximum(r[:, :, 0] * 128, 1), 2))

hist = (C.c_ulonglong * 10)()
if hasattr(lib, "lc_debug_read_hist") and lib.lc_debug_read_hist(hist, 1) == 0:
    h = np.array(list(hist), dtype=np.float64).reshape(2, 5)
    print("wave-levels by the number of their 4 iterations with a redo (0..4):", np.round(100 * h[0] / max(h[0].sum(), 1), 1), "%")
    print("... of the wave-levels whose previous level had >= 3 (%.1f %% of all):" % (100 * h[1].sum() / max(h[0].sum(), 1)), np.round(100 * h[1] / max(h[1].sum(), 1), 1), "%")
