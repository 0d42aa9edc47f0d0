"""Advect time of seed grids whose number of tile rows is / is not a multiple of the 8 XCDs, with whole tile rows dealt to the
XCDs (product until round 5) against eighths of a tile row (LCS_XCD_SPLIT=8): python tools/xcd_split_probe.py  (run twice, env set / unset)"""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lagrangiancoherence_amd import flows
from lagrangiancoherence_amd.engine import Engine

eng = Engine(0)
u, v, lat, lon = flows.era5_like_on_device(torch, eng.device, nt=97)
field = eng.prepare_field(u, v, lat, lon, 1)
fo3 = eng.prepare_field(u, v, lat, lon, 3)


def t(fn, reps=3):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


print("LCS_XCD_SPLIT =", os.environ.get("LCS_XCD_SPLIT"))
for ny, nx, order in [(4096, 4096, 1), (4096, 4096, 3), (1024, 8192, 1), (1084, 8192, 1), (1088, 8192, 1), (1152, 8192, 1), (970, 8192, 1), (512, 4096, 1), (576, 4096, 1),
                      (721, 1440, 1), (721, 1440, 3), (2048, 2048, 1), (2100, 2048, 1), (3000, 3000, 1), (3000, 3000, 3)]:
    slat, slon = flows.seed_grid(ny, nx, lat, lon)
    sl, so = eng.to_device(slat, np.float32), eng.to_device(slon, np.float32)
    f = field if order == 1 else fo3
    ms = t(lambda: eng.advect(f, sl, so, -900.0, 4, order, True))
    print(f"{ny:5d} x {nx:5d} order {order}: {ms:8.3f} ms  {eng.last_advect_kernel()}")
