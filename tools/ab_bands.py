"""A/B: the headline advect (configs[2]) as latitude BANDS of seed rows, each band its own lc_advect (row0 / ny_global: the
sharded path, bit-identical), on the library named by LCS_LIB.  Prints ms per band (two seeds per lane forced) and for the bands
run back to back on separate streams.  usage: LCS_LIB=... python tools/ab_bands.py [edge_rows ...]"""
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from lagrangiancoherence_amd import flows                    # noqa: E402
from lagrangiancoherence_amd.engine import Engine            # noqa: E402

edges = [int(a) for a in sys.argv[1:]] or [512]
u, v, lat, lon = flows.era5_like(nt=97, ny=720, nx=1440)
n = 4096
slat, slon = flows.seed_grid(n, n, lat, lon)
eng = Engine(0)
eng.set_lds_tiles(1)                    # two seeds per lane whatever the size
f32 = np.float32
f = eng.prepare_field(eng.to_device(u, f32), eng.to_device(v, f32), lat, lon, 1)
slat_d, slon_d = eng.to_device(slat, f32), eng.to_device(slon, f32)


def timed(fn, reps=10):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(reps):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps


def band(lo, hi):
    return eng.advect(f, slat_d[lo:hi], slon_d, -900.0, 4, 1, True, row0=lo, ny_global=n)


print("whole", round(timed(lambda: band(0, n)), 3), eng.last_advect_kernel(), flush=True)
for e in edges:
    t = [timed(lambda lo=lo, hi=hi: band(lo, hi)) for lo, hi in ((0, e), (e, n - e), (n - e, n))]
    print(f"edge {e}: south {t[0]:.3f} middle {t[1]:.3f} north {t[2]:.3f} sum {sum(t):.3f}", eng.last_advect_kernel(), flush=True)
