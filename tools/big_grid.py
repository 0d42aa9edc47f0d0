"""One-off: the largest seed grid exercised, 16384^2 seeds (268 M, 4 x config 4's) x 24 steps on the C3 wind series:
the engine's answer on a subset of seeds against the oracle on exactly those seeds, and sigma finite everywhere."""
import os, sys, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lagrangiancoherence_amd import flows
from lagrangiancoherence_amd.engine import Engine
from oracle import lcs_oracle as O
from tests import _fullsize as F
n, nt = int(sys.argv[1]) if len(sys.argv) > 1 else 16384, 25
eng = Engine(0)
u, v, lat, lon = flows.era5_like(nt=nt)
slat, slon = flows.seed_grid(n, n, lat, lon)
f = eng.prepare_field(u, v, lat, lon, 1)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
x, y = eng.advect(f, slat, slon, -900.0, 4, 1, True)
torch.cuda.synchronize(); e0.record()
x, y = eng.advect(f, slat, slon, -900.0, 4, 1, True)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1)
print(f"{n}^2 seeds x {nt - 1} steps: {ms:.1f} ms, {n * n * (nt - 1) / ms / 1e6:.1f} G particle-timesteps/s, kernel {eng.last_advect_kernel()}")
rows, cols = F.subset(n, 40, 1), F.subset(n, 48, 0)
xs, ys = x[rows][:, cols].cpu().numpy(), y[rows][:, cols].cpu().numpy()
kw = dict(timestep=-900.0, SETTLS_order=4, interp_order=1, cyclic_xboundary=True)
x64, y64 = F.oracle_subset(O, u, v, lat, lon, slat, slon, rows, cols, np.float64, **kw)
x32, y32 = F.oracle_subset(O, u, v, lat, lon, slat, slon, rows, cols, np.float32, **kw)
eg = np.maximum(F.lon_err(xs, x64), np.abs(ys - y64)); eo = np.maximum(F.lon_err(x32, x64), np.abs(y32 - y64))
print(f"subset {len(rows)} x {len(cols)}: engine vs float64 oracle median {np.median(eg):.2e} p99 {np.percentile(eg, 99):.2e} max {eg.max():.2e} deg; "
      f"float32 oracle itself median {np.median(eo):.2e} p99 {np.percentile(eo, 99):.2e} max {eo.max():.2e}")
dlat, dlon = float(slat[1] - slat[0]), float(slon[1] - slon[0])
sig = eng.sigma(x, y, eng.to_device(slat, np.float32), dlat, dlon)
print("sigma finite:", bool(torch.isfinite(sig).all()), "kernel", eng.last_sigma_kernel(), "max", float(sig.max()))
