"""A/B helper: positions of the headline workload (configs[2]: 4096^2 seeds, 96 steps, K = 4, order 1) from the library named
by LCS_LIB, saved as a SHA-256 of all 2 x 4096^2 floats + a 64 x 64 sample, so that two library variants can be compared
bit for bit across processes.
usage: LCS_LIB=$PWD/build/ab/variant.so [LCS_DUMP_K=0] python tools/dump_positions.py out.npz      (LCS_DUMP_K: SETTLS_order, default 4)
       python tools/dump_positions.py --compare a.npz b.npz"""
import hashlib
import os
import sys

import numpy as np

if len(sys.argv) > 1 and sys.argv[1] == "--compare":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    same = str(a["sha"]) == str(b["sha"])
    nx, ny = int((a["x"] != b["x"]).sum()), int((a["y"] != b["y"]).sum())
    print("bit-identical" if same else f"DIFFERENT: {nx} + {ny} of 2 x {a['x'].size} sampled coordinates differ, "
          f"max |dx| {np.abs(a['x'] - b['x']).max():.3g} deg, max |dy| {np.abs(a['y'] - b['y']).max():.3g} deg")
    sys.exit(0 if same else 1)

import torch                                                  # noqa: E402

sys.path.insert(0, ".")
from lagrangiancoherence_amd import flows                    # noqa: E402
from lagrangiancoherence_amd.engine import Engine            # noqa: E402

u, v, lat, lon = flows.era5_like(nt=97, ny=720, nx=1440)
slat, slon = flows.seed_grid(4096, 4096, lat, lon)
eng = Engine(0)
f = eng.prepare_field(eng.to_device(u, np.float32), eng.to_device(v, np.float32), lat, lon, 1)
K = int(os.environ.get("LCS_DUMP_K", "4"))
x, y = eng.advect(f, eng.to_device(slat, np.float32), eng.to_device(slon, np.float32), -900.0, K, 1, True)[:2]
torch.cuda.synchronize()
xh, yh = x.cpu().numpy(), y.cpu().numpy()
h = hashlib.sha256(xh.tobytes() + yh.tobytes()).hexdigest()
print(eng.last_advect_kernel(), h)
np.savez(sys.argv[1], sha=h, x=xh[::64, ::64], y=yh[::64, ::64])
