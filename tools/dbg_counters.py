import sys, ctypes as C, numpy as np
sys.path.insert(0, '.')
import torch
from lagrangiancoherence_amd import flows, _capi
from lagrangiancoherence_amd.engine import Engine
scale = float(sys.argv[1]) if len(sys.argv) > 1 else 1.0
u, v, lat, lon = flows.era5_like(nt=97)
u, v = u*np.float32(scale), v*np.float32(scale)
slat, slon = flows.seed_grid(4096, 4096, lat, lon)
eng = Engine(0)
f = eng.prepare_field(u, v, lat, lon, 1)
lib = eng.lib
buf = (C.c_ulonglong * 96)()
import struct
for kp in (1.5, 2.0, 2.5, 3.0, 3.5):
    import os
    os.environ["LCS_KP"] = str(kp)
    lib.lc_debug_counters(buf, 1)
    x, y = eng.advect(f, slat, slon, -900.0, 4, 1, True)
    torch.cuda.synchronize()
    lib.lc_debug_counters(buf, 0)
    wl = 262144 * 96
    print("kp", kp, "tile-x %", [round(buf[2+k]/wl*100,2) for k in range(4)], "tile-y %", [round(buf[6+k]/wl*100,2) for k in range(4)], "any-bad per iteration %", round(buf[11]/wl*25,2))
sys.exit(0)
names = ["euler range bad", "euler xcare", "it0 tile-x", "it1 tile-x", "it2 tile-x", "it3 tile-x", "it0 tile-y", "it1 tile-y", "it2 tile-y", "it3 tile-y", "iter xcare", "iter bad(any)"]
wl = 262144 * 96
for i, n in enumerate(names):
    print(f"{n:18s} waves {buf[i]:12d} ({buf[i]/wl*100:6.2f}% of wave-levels)  lanes {buf[16+i]:14d} ({buf[16+i]/(wl*64)*100:6.3f}%)")

print("it0 tile-y misses by latitude band (32 bands, south to north), % of that band's wave-levels; second number: misses with ry<0")
for b in range(32):
    print(b, round(buf[32+b]/(wl/32)*100,2), round(buf[64+b]/(wl/32)*100,2))
