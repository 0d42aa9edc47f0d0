"""lc_lcs_host on BASELINE configs[2] (4096^2 seeds x 96 steps, float32): numpy arrays in, numpy arrays out -- the call a
reference-side binding makes in place of LCS/LCS.py:129-157.  Wall time of the Python call (results kept alive, so that freeing
the previous call's 201 MB is not inside the next call's bracket); LCS_HOST_TIMING=1 adds the C side's own marks on stderr."""
import time, sys, numpy as np
sys.path.insert(0, '.')
from lagrangiancoherence_amd import flows
from lagrangiancoherence_amd.engine import lcs_host
u, v, lat, lon = flows.era5_like(nt=97)
slat, slon = flows.seed_grid(4096, 4096, lat, lon)
keep = []
for i in range(5):
    t = time.perf_counter()
    out = lcs_host(u, v, lat, lon, -900.0, SETTLS_order=4, interp_order=1, cyclic_xboundary=True, seed_lat=slat, seed_lon=slon)
    dt = time.perf_counter() - t
    keep.append(out)
    print(f"lc_lcs_host pass {i}: {dt*1e3:.1f} ms -> {4096*4096*96/dt:.3e} particle-timesteps/s (host buffers in, host buffers out)", flush=True)
