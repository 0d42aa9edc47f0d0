"""How long does the reference's DEFAULT boundary mode (cyclic_xboundary=False, outer-product clamp) take on a regional
grid whose parcels leave the box -- the sub-step path, 2 (K + 1) launches per time level?"""
import sys, time
sys.path.insert(0, ".")
import numpy as np
import torch
from lagrangiancoherence_amd.engine import Engine
eng = Engine(0)
rng = np.random.default_rng(5)
for n, nt in ((200, 51), (400, 51), (89, 8)):
    lat = np.linspace(-40, 40, n)
    lon = np.linspace(-60, 50, n + 16)
    u = 30 + 25 * rng.standard_normal((nt, n, n + 16))
    v = 8 * rng.standard_normal((nt, n, n + 16))
    for order in (1, 3):
        f = eng.prepare_field(u, v, lat, lon, order, fuse_levels=False)
        for cyc in (False, True):
            for _ in range(2):
                eng.advect(f, lat, lon, 3600.0, 4, order, cyc)
            torch.cuda.synchronize()
            t = time.perf_counter()
            for _ in range(5):
                eng.advect(f, lat, lon, 3600.0, 4, order, cyc)
            torch.cuda.synchronize()
            print(f"{n}x{n + 16} nt={nt} order {order} cyclic={cyc}: {(time.perf_counter() - t) / 5 * 1e3:.2f} ms  kernel {eng.last_advect_kernel()}")
