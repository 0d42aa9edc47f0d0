import sys, time, numpy as np
sys.path.insert(0, '.')
import torch
from lagrangiancoherence_amd import flows, sharded
from lagrangiancoherence_amd.engine import Engine
eng = Engine(0)
# C4 shard: rank 3 of 8 of an 8192^2 seed grid, nt=385
u, v, lat, lon = flows.era5_like(nt=385)
slat, slon = flows.seed_grid(8192, 8192, lat, lon)
t = time.perf_counter(); f = eng.prepare_field(u, v, lat, lon, 1); torch.cuda.synchronize(); print("C4 upload+pack %.1f ms" % ((time.perf_counter()-t)*1e3))
lo, hi = sharded.row_partition(8192, 8, 3)
for rep in range(2):
    t = time.perf_counter()
    x, y = eng.advect(f, slat[lo:hi], slon, -900.0, 4, 1, True, row0=lo, ny_global=8192)
    torch.cuda.synchronize(); dt = time.perf_counter()-t
    print("C4 shard %dx%d seeds x 384 steps: %.2f ms -> %.3e particle-timesteps/s, finite=%s" % (hi-lo, 8192, dt*1e3, (hi-lo)*8192*384/dt, bool(torch.isfinite(x).all() and torch.isfinite(y).all())))
print("max mem GB", torch.cuda.max_memory_allocated()/1e9)
del f, x, y
# C5: 8 members (one rank's share of 64) x 2048^2 x 200 steps on nt=264 levels
u, v, lat, lon = flows.era5_like(nt=264)
slat, slon = flows.seed_grid(2048, 2048, lat, lon)
f = eng.prepare_field(u, v, lat, lon, 1)
for rep in range(2):
    t = time.perf_counter()
    mine, sig = sharded.ensemble_lcs(eng, f, slat, slon, -900.0, 64, 200, rank=3, world=8, SETTLS_order=4, interp_order=1)
    torch.cuda.synchronize(); dt = time.perf_counter()-t
    print("C5 rank share: members %s, %.2f ms -> %.3e particle-timesteps/s, sigma finite=%s" % (mine, dt*1e3, len(mine)*2048*2048*200/dt, bool(torch.isfinite(sig).all())))
