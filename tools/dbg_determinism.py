"""Is lc_advect deterministic call to call (and with another process on the same GPU)?
    python tools/dbg_determinism.py [reps] [--bg]     --bg: also keep a second process busy on the GPU"""
import subprocess
import sys
import numpy as np
import torch
sys.path.insert(0, ".")
from lagrangiancoherence_amd import flows
from lagrangiancoherence_amd.engine import Engine

reps = int(sys.argv[1]) if len(sys.argv) > 1 and sys.argv[1].isdigit() else 50
bg = None
if "--bg" in sys.argv:
    bg = subprocess.Popen([sys.executable, __file__, "400", "--quiet"], stdout=subprocess.DEVNULL)
eng = Engine(0)
u, v, lat, lon = flows.era5_like(nt=9)
slat, slon = flows.seed_grid(512, 512, lat, lon)
ud, vd = eng.to_device(u, np.float32), eng.to_device(v, np.float32)
sl, so = eng.to_device(slat, np.float32), eng.to_device(slon, np.float32)
bad = 0
ref = None
for i in range(reps):
    f = eng.prepare_field(ud, vd, lat, lon, 1)
    x0, y0 = eng.advect(f, sl[0:256], so, -900.0, 4, 1, True, row0=0, ny_global=512, halo=(0, 2))
    x1, y1 = eng.advect(f, sl[0:258], so, -900.0, 4, 1, True, row0=0, ny_global=512)
    torch.cuda.synchronize()
    same = torch.equal(x0[:256], x1[:256]) and torch.equal(y0[:256], y1[:256])
    if ref is None:
        ref = (x1.clone(), y1.clone())
    stable = torch.equal(x1, ref[0]) and torch.equal(y1, ref[1])
    if not (same and stable):
        bad += 1
        rows = ((x0[:256] != x1[:256]) | (y0[:256] != y1[:256])).any(dim=1).nonzero().flatten().tolist()
        rows2 = ((x1 != ref[0]) | (y1 != ref[1])).any(dim=1).nonzero().flatten().tolist()
        if "--quiet" not in sys.argv:
            print(f"rep {i}: block-vs-extended rows {rows[:10]} | extended-vs-first rows {rows2[:10]} kernel {eng.last_advect_kernel()}")
if "--quiet" not in sys.argv:
    print(f"{bad} of {reps} repetitions differ" + (" (second process active)" if bg else ""))
if bg:
    bg.wait()
