"""How much of a small grid's advect time is the pole rows' generic path (one dependent gather per sample)?
Times lc_advect on N x N seeds with the first / last seed row treated as global pole rows (row0=0, ny_global=N: the
default) and with no pole row in the block (row0=1, ny_global=N+2: every row takes the LDS-tile path)."""
import sys, os
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from lagrangiancoherence_amd import flows
from lagrangiancoherence_amd.engine import Engine

eng = Engine(0)
u, v, lat, lon = flows.era5_like(nt=97)
f = eng.prepare_field(u, v, lat, lon, 1)
for n in (256, 512, 1024, 2048):
    slat, slon = flows.seed_grid(n, n, lat, lon)
    for name, kw in (("pole rows", dict()), ("no pole rows", dict(row0=1, ny_global=n + 2))):
        for _ in range(3):
            eng.advect(f, slat, slon, -900.0, 4, 1, True, **kw)
        torch.cuda.synchronize()
        ev = [torch.cuda.Event(enable_timing=True) for _ in range(2)]
        ev[0].record()
        for _ in range(10):
            eng.advect(f, slat, slon, -900.0, 4, 1, True, **kw)
        ev[1].record()
        torch.cuda.synchronize()
        print(f"{n}^2 seeds, 96 levels, K=4, {name}: {ev[0].elapsed_time(ev[1]) / 10:.3f} ms  ({eng.last_advect_kernel()})")
