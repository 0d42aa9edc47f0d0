"""Time the two float32 sigma kernels on 4096^2 departure points (HIP events, 20 launches each)."""
import sys, numpy as np
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lagrangiancoherence_amd.engine import Engine
eng = Engine(0)
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
lat = np.linspace(-89.875, 89.875, n); lon = np.linspace(-180, 179.75, n)
g = torch.Generator(device="cuda").manual_seed(1)
X = torch.tensor(lon, dtype=torch.float32, device="cuda")[None, :].expand(n, n)
Y = torch.tensor(lat, dtype=torch.float32, device="cuda")[:, None].expand(n, n)
xd = (X + torch.rand(n, n, device="cuda", generator=g) * 0.1).contiguous()
yd = (Y + torch.rand(n, n, device="cuda", generator=g) * 0.05).clamp(-90, 90).contiguous()
slat = torch.tensor(lat, dtype=torch.float32, device="cuda")
dlat, dlon = float(lat[1] - lat[0]), float(lon[1] - lon[0])
res = {}
for mode in (1, 0):
    eng.set_sigma_march(mode)
    for _ in range(3):
        s = eng.sigma(xd, yd, slat, dlat, dlon)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(20):
        s = eng.sigma(xd, yd, slat, dlat, dlon)
    e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 20
    res[mode] = s.clone()
    print(f"march={mode}: {ms*1e3:.1f} us per launch (incl. launch overhead), {n*n/ms/1e3:.0f} Mcells/s, {12*n*n/ms/1e6:.0f} GB/s algorithmic")
print("bitwise equal:", bool(torch.equal(res[0], res[1])))
