"""lc_lcs_host on C3 inside different process states (profiles/r06/host_route.txt, "one ring per DEVICE"): what else the
process did with the GPU before the calls.  usage (GPU box): python tools/host_ring_probe.py <mode>
  plain            nothing else
  torch_first      torch initialised before the arrays exist
  bigalloc         48 GB of torch tensors and 4 GB of host memory held
  engine_only      an Engine alive that staged nothing (its upload is below Engine.STAGED_COPY_FROM)
  engine_small     an Engine that staged one 33 MB upload
  engine_upload    an Engine that staged the whole 805 MB
  engine_closed    ... and was closed before the calls
  host_then_engine one lc_lcs_host call first, then the Engine's staged upload"""
import sys
import time

import numpy as np

sys.path.insert(0, '.')
mode = sys.argv[1] if len(sys.argv) > 1 else "plain"
from lagrangiancoherence_amd import flows  # noqa: E402

if mode != "plain":
    import torch
    torch.zeros(1, device="cuda")
    torch.cuda.synchronize()
u, v, lat, lon = flows.era5_like(nt=97)
slat, slon = flows.seed_grid(4096, 4096, lat, lon)
from lagrangiancoherence_amd.engine import Engine, lcs_host  # noqa: E402


def call():
    return lcs_host(u, v, lat, lon, -900.0, SETTLS_order=4, interp_order=1, cyclic_xboundary=True, seed_lat=slat, seed_lon=slon)


keep, ts = [], []
if mode == "host_then_engine":
    keep.append(call())
if mode.startswith("engine") or mode == "host_then_engine":
    eng = Engine(0)
    what = {"engine_only": lat, "engine_small": u[:8], "host_then_engine": u[:8]}.get(mode, u)
    held = [eng.to_device(what, np.float32)]
    if what is u:
        held.append(eng.to_device(v, np.float32))
    torch.cuda.synchronize()
    if mode == "engine_closed":
        eng.close()
if mode == "bigalloc":
    big = [torch.empty(8 << 30, dtype=torch.uint8, device="cuda") for _ in range(6)]
    torch.cuda.synchronize()
    junk = np.ones(4 << 30, dtype=np.uint8)
for i in range(5):
    t = time.perf_counter()
    out = call()
    ts.append((time.perf_counter() - t) * 1e3)
    keep.append(out)       # (freeing 201 MB of results is not the route's time)
print(mode, [round(t, 1) for t in ts], {k: round(x, 1) for k, x in keep[-1]["host_marks_ms"].items()}, flush=True)
