import sys, numpy as np
sys.path.insert(0, '.')
from lagrangiancoherence_amd import flows
from lagrangiancoherence_amd.engine import Engine
from oracle import lcs_oracle as O
from tests._fullsize import oracle_window, lon_err
eng = Engine(0)
u, v, lat, lon = flows.era5_like(nt=97)
slat, slon = flows.seed_grid(4096, 4096, lat, lon)
f = eng.prepare_field(u, v, lat, lon, 1)
r = eng.lcs(f, slat, slon, -900.0, SETTLS_order=4, interp_order=1, cyclic_xboundary=True)
g = {k: r[k].cpu().numpy() for k in ("sigma", "x_dep", "y_dep")}
KW = dict(timestep=-900.0, SETTLS_order=4, cyclic_xboundary=True)
r0, r1, c0, c1 = 2016, 2080, 1000, 1064
x32, y32, s32 = oracle_window(O, u, v, lat, lon, slat, slon, r0, r1, c0, c1, np.float32, 1, **KW)
x64, y64, s64 = oracle_window(O, u, v, lat, lon, slat, slon, r0, r1, c0, c1, np.float64, 1, **KW)
sg = g["sigma"][r0:r1, c0:c1].astype(np.float64)
e = np.abs(sg / s64 - 1)
idx = np.argwhere(e > 1e-2)
print("cells with sigma rel err > 1e-2:", len(idx))
for (i, j) in idx[:20]:
    print(i + r0, j + c0, "sig gpu %.6g o32 %.6g o64 %.6g" % (sg[i, j], s32[i, j], s64[i, j]))
ex = lon_err(g["x_dep"][r0:r1, c0:c1], x64); ey = np.abs(g["y_dep"][r0:r1, c0:c1] - y64)
print("pos err max", ex.max(), ey.max(), "at", np.unravel_index(ex.argmax(), ex.shape), np.unravel_index(ey.argmax(), ey.shape))
i, j = idx[0] if len(idx) else (32, 32)
sl = (slice(max(i - 3, 0), i + 4), slice(max(j - 3, 0), j + 4))
np.set_printoptions(precision=7, linewidth=200)
print("y gpu\n", g["y_dep"][r0:r1, c0:c1][sl]); print("y o64\n", y64[sl])
print("x gpu\n", g["x_dep"][r0:r1, c0:c1][sl]); print("x o64\n", x64[sl])
print("sig gpu\n", sg[sl]); print("sig o64\n", s64[sl]); print("sig o32\n", s32[sl])
# sigma of the GPU kernel applied to the ORACLE's float32 positions: isolates K3 from K1
import torch
ra, rb = r0 - 2, r1 + 2
xo, yo, _ = oracle_window(O, u, v, lat, lon, slat, slon, ra, rb, c0 - 2, c1 + 2, np.float32, 1, **KW)
