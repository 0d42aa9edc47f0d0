"""Time lc_field_pack alone on the BASELINE fields (HIP events, 20 repeats):  python tools/pack_bench.py [c3|c2] [order]"""
import sys
import numpy as np
import torch
import os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lagrangiancoherence_amd import flows
from lagrangiancoherence_amd.engine import Engine

which = sys.argv[1] if len(sys.argv) > 1 else "c3"
order = int(sys.argv[2]) if len(sys.argv) > 2 else 1
eng = Engine(0)
if which == "c3":
    u, v, lat, lon = flows.era5_like(nt=97)
    dt = np.float32
else:
    u, v, lat, lon = flows.config2_on_device(torch, eng.device)   # (the host generator takes a minute)
    dt = np.float64
ud, vd = eng.to_device(u, dt), eng.to_device(v, dt)
for _ in range(3):
    f = eng.prepare_field(ud, vd, lat, lon, order)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
n = int(sys.argv[3]) if len(sys.argv) > 3 else 20
e0.record()
for _ in range(n):
    f = eng.prepare_field(ud, vd, lat, lon, order)
e1.record()
torch.cuda.synchronize()
print(f"pack {which} order {order}: {e0.elapsed_time(e1) / n:.3f} ms  (build {eng.lib.lc_build_id().decode()})")
