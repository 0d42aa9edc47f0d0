import sys, numpy as np
sys.path.insert(0, '.')
import torch
from lagrangiancoherence_amd import flows
from lagrangiancoherence_amd.engine import Engine
eng = Engine(0)
for dtype in (np.float32, np.float64):
    for order in (1, 3):
        u, v, lat, lon = flows.era5_like(nt=9, ny=90, nx=180)
        u = u.astype(dtype); v = v.astype(dtype); lat = lat.astype(dtype); lon = lon.astype(dtype)
        u[3, 40, 60] = np.nan; v[5, 20, 100] = np.inf; u[2, 70, 10] = -np.inf; v[1, 5, 5] = 1e30; u[6, 80, 170] = -1e30
        slat, slon = flows.seed_grid(300, 400, lat, lon)
        f = eng.prepare_field(u, v, lat, lon, order)
        for cyc in (True, False):
            x, y = eng.advect(f, slat, slon, -1800.0, 4, order, cyc)
            torch.cuda.synchronize()
            xn, yn = x.cpu().numpy(), y.cpu().numpy()
            bad = ~np.isfinite(xn) | ~np.isfinite(yn)
            print(np.dtype(dtype).name, "order", order, "cyclic", cyc, "non-finite outputs: %.3f%%" % (bad.mean()*100), "y range", np.nanmin(yn), np.nanmax(yn))
            assert np.nanmin(yn) >= lat.min() - 1e-4 and np.nanmax(yn) <= lat.max() + 1e-4
            r = eng.sigma(x, y, slat, slat[1]-slat[0], slon[1]-slon[0])
            torch.cuda.synchronize()
print("completed")
