"""Does having used a second HIP stream slow later single-stream kernels?  (round 4: the serial advect of config 2 measured
3.13-3.16 ms in processes that never used a side stream and 3.37-3.39 ms after Engine.pack_and_advect had used one.)"""
import sys
sys.path.insert(0, ".")
import numpy as np
import torch
from lagrangiancoherence_amd import flows
from lagrangiancoherence_amd.engine import Engine

eng = Engine(0)
ud, vd, lat, lon = flows.config2_on_device(torch, eng.device)
sl, so = eng.to_device(lat, np.float64), eng.to_device(lon, np.float64)
f = eng.prepare_field(ud, vd, lat, lon, 1)
x = torch.empty((1024, 1024), dtype=torch.float64, device="cuda")
y = torch.empty_like(x)


def t_advect(label):
    for _ in range(2):
        eng.advect(f, sl, so, -900.0, 4, 1, True, out=(x, y))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        eng.advect(f, sl, so, -900.0, 4, 1, True, out=(x, y))
    e1.record()
    torch.cuda.synchronize()
    print(f"{label}: advect {e0.elapsed_time(e1) / 10:.3f} ms")


t_advect("before any side stream")
side = torch.cuda.Stream()
with torch.cuda.stream(side):
    z = torch.zeros(1 << 20, device="cuda") + 1
torch.cuda.synchronize()
t_advect("after a side stream ran one small kernel")
_, xx, yy = eng.pack_and_advect(ud, vd, lat, lon, sl, so, -900.0, 4, 1, True, pipeline=True, chunk=24)
torch.cuda.synchronize()
t_advect("after one pipelined pack_and_advect")
del side, z
eng._side_stream = None
torch.cuda.synchronize()
t_advect("after dropping the side streams")
f2 = eng.prepare_field(ud, vd, lat, lon, 1)
f = f2
t_advect("with a freshly prepared field")
