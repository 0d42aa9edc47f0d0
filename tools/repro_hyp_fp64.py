import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from oracle import lcs_oracle as O
from lagrangiancoherence_amd.engine import Engine
eng = Engine(0)
ny, nx, nt, K, order, dt, cyclic, seed, lat_hi, scale = 8, 32, 4, 4, 1, -5400.0, True, 1244, 85.0, 80.0
if len(sys.argv) > 1:   # ny nx nt K order dt cyclic seed lat_hi scale
    a = sys.argv[1:]
    ny, nx, nt, K, order, dt, cyclic, seed, lat_hi, scale = int(a[0]), int(a[1]), int(a[2]), int(a[3]), int(a[4]), float(a[5]), a[6] == "1", int(a[7]), float(a[8]), float(a[9])
rng = np.random.default_rng(seed)
lat = np.linspace(-lat_hi, lat_hi, ny)
lon = -180 + 360.0 / nx * np.arange(nx)
u = scale * rng.standard_normal((nt, ny, nx))
v = 0.5 * scale * rng.standard_normal((nt, ny, nx))
slat = np.linspace(lat[0], lat[-1], int(rng.integers(7, 40)))
slon = np.linspace(lon[0], lon[-1], int(rng.integers(7, 50)))
t0 = int(rng.integers(0, nt - 1)); nsteps = int(rng.integers(1, nt - t0))
print("t0", t0, "nsteps", nsteps, slat.size, slon.size)
xr, yr = O.parcel_propagation(u, v, lat, lon, timestep=dt, SETTLS_order=K, interp_order=order, cyclic_xboundary=cyclic, seed_lat=slat, seed_lon=slon, t0=t0, nsteps=nsteps)
for name, kw in (("fused", {}), ("exact", dict(fuse_levels=False)), ("stored", dict(fuse_levels=True))):
    for mode in (-1, 0):
        eng.set_lds_tiles(mode)
        f = eng.prepare_field(u, v, lat, lon, order, **kw)
        x, y = eng.advect(f, slat, slon, dt, SETTLS_order=K, interp_order=order, cyclic_xboundary=cyclic, t0=t0, nsteps=nsteps)
        dx = np.abs(x.cpu().numpy() - xr); dx = np.minimum(dx, np.abs(dx - 360)); dy = np.abs(y.cpu().numpy() - yr)
        i = np.unravel_index(np.argmax(dx), dx.shape)
        print(name, mode, eng.last_advect_kernel(), "max dx %.3e dy %.3e at %s (x=%.6f y=%.6f; got x=%.6f y=%.6f; seed lat %.6f lon %.6f)" % (dx.max(), dy.max(), i, xr[i], yr[i], x.cpu().numpy()[i], y.cpu().numpy()[i], slat[i[0]], slon[i[1]]), "n bad", int((dx > 1e-9).sum()))
# sensitivity of the oracle itself: seeds moved by 1e-12 degrees
xr2, yr2 = O.parcel_propagation(u, v, lat, lon, timestep=dt, SETTLS_order=K, interp_order=order, cyclic_xboundary=cyclic, seed_lat=slat + 1e-12, seed_lon=slon + 1e-12, t0=t0, nsteps=nsteps)
d = np.abs(xr2 - xr); d = np.minimum(d, np.abs(d - 360))
print("oracle, seeds + 1e-12 deg: max |dx| %.3e (amplification %.1e), at the failing point %.3e" % (d.max(), d.max() / 1e-12, d[i]))
