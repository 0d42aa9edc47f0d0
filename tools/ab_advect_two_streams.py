"""Does a single lc_advect of BASELINE configs[1] lose time to the tails of its launches?  The same advect as ONE call and as
row blocks on separate HIP streams (independent chains of level-chunk launches: one block's last workgroups overlap the other's):
    python tools/ab_advect_two_streams.py [order] [parts ...]"""
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lagrangiancoherence_amd import flows
from lagrangiancoherence_amd.engine import Engine

order = int(sys.argv[1]) if len(sys.argv) > 1 else 1
parts_list = [int(a) for a in sys.argv[2:]] or [1, 2, 3, 4]
eng = Engine(0)
u, v, lat, lon = flows.config2_on_device(torch, eng.device)
f = eng.prepare_field(u, v, lat, lon, order)
la, lo = eng.to_device(lat, np.float64), eng.to_device(lon, np.float64)
ny, nx = la.numel(), lo.numel()
x = torch.empty((ny, nx), dtype=torch.float64, device=eng.device)
y = torch.empty_like(x)
streams = [torch.cuda.Stream() for _ in range(max(parts_list))]


def run(parts):
    if parts == 1:
        eng.advect(f, la, lo, -900.0, 4, order, True, out=(x, y))
        return
    cur = torch.cuda.current_stream()
    ready = torch.cuda.Event()
    ready.record(cur)
    step = -(-ny // parts // 8) * 8
    for p in range(parts):
        r0, r1 = p * step, min(ny, (p + 1) * step)
        with torch.cuda.stream(streams[p]):
            streams[p].wait_event(ready)
            eng.advect(f, la[r0:r1], lo, -900.0, 4, order, True, row0=r0, ny_global=ny, out=(x[r0:r1], y[r0:r1]))
        done = torch.cuda.Event()
        done.record(streams[p])
        cur.wait_event(done)


ref = None
for parts in parts_list:
    for _ in range(2):
        run(parts)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        run(parts)
    e1.record()
    torch.cuda.synchronize()
    if ref is None:
        ref = (x.clone(), y.clone())
    same = bool((x == ref[0]).all()) and bool((y == ref[1]).all())
    print(f"order {order}: {parts} part(s) on {parts} stream(s): {e0.elapsed_time(e1) / 5:.3f} ms per advect, kernel {eng.last_advect_kernel()}, bit-identical to one call: {same}")
