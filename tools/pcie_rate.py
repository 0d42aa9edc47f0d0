import time, sys, numpy as np
sys.path.insert(0, '.')
from lagrangiancoherence_amd import flows
from lagrangiancoherence_amd.engine import lcs_host
u, v, lat, lon = flows.era5_like(nt=97)
slat, slon = flows.seed_grid(4096, 4096, lat, lon)
for i in range(3):
    t = time.perf_counter()
    out = lcs_host(u, v, lat, lon, -900.0, SETTLS_order=4, interp_order=1, cyclic_xboundary=True, seed_lat=slat, seed_lon=slon)
    dt = time.perf_counter() - t
    print(f"lc_lcs_host pass {i}: {dt*1e3:.1f} ms -> {4096*4096*96/dt:.3e} particle-timesteps/s (host buffers in, host buffers out)")
