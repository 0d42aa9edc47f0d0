"""Whole-grid check of the LDS-tile advect kernel against the direct-gather kernel (same arithmetic, so the two
must agree bit for bit on every seed): python tools/dbg_lds_vs_direct.py [order] [nt]"""
import os, sys, numpy as np
sys.path.insert(0, '.')
import torch
from lagrangiancoherence_amd import flows
from lagrangiancoherence_amd.engine import Engine
order = int(sys.argv[1]) if len(sys.argv) > 1 else 1
nt = int(sys.argv[2]) if len(sys.argv) > 2 else 97
eng = Engine(0)
u, v, lat, lon = flows.era5_like(nt=nt)
slat, slon = flows.seed_grid(4096, 4096, lat, lon)
f = eng.prepare_field(u, v, lat, lon, order)
eng.set_lds_tiles(1)
x1, y1 = eng.advect(f, slat, slon, -900.0, 4, order, True)
x1b, y1b = eng.advect(f, slat, slon, -900.0, 4, order, True)
eng.set_lds_tiles(0)
x0, y0 = eng.advect(f, slat, slon, -900.0, 4, order, True)
torch.cuda.synchronize()
print("LDS run-to-run identical:", bool(torch.equal(x1, x1b) and torch.equal(y1, y1b)))
dx = (x1 - x0).abs(); dx = torch.minimum(dx, (dx - 360).abs()); dy = (y1 - y0).abs()
d = torch.maximum(dx, dy)
print("seeds differing at all:", int((d > 0).sum()), " > 1e-3 deg:", int((d > 1e-3).sum()), " > 1 deg:", int((d > 1).sum()), "max", float(d.max()))
bad = torch.nonzero(d > 1e-3)
for r, c in bad[:20].tolist():
    print("row", r, "col", c, "lds", float(x1[r, c]), float(y1[r, c]), "direct", float(x0[r, c]), float(y0[r, c]))
