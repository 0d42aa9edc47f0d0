"""A/B: configs[2] (float32, order 1, K = 4, 4096^2 seeds, 97 levels): pack-then-advect against the pack of level chunk k+1
on a side stream beside the advect of chunk k (the form Engine.pack_and_advect has for the other families; float32 at order 1
was never measured in it because its pack writes two images).  Bit-identity of the positions is asserted.
Usage: python tools/ab_pipe_c3_o1.py [reps]"""
import ctypes as C
import json
import sys
import time

import numpy as np
import torch

sys.path.insert(0, ".")
from lagrangiancoherence_amd import _capi, flows                    # noqa: E402
from lagrangiancoherence_amd.engine import Engine, PackedField      # noqa: E402

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 10
K, order, dt, nt, n = 4, 1, -900.0, 97, 4096
u, v, lat, lon = flows.era5_like(nt=nt, ny=720, nx=1440)
slat, slon = flows.seed_grid(n, n, lat, lon)
eng = Engine(0)
f32 = np.float32
ud, vd = eng.to_device(u, f32), eng.to_device(v, f32)
slat_d, slon_d = eng.to_device(slat, f32), eng.to_device(slon, f32)
ny_f, nx_f = 720, 1440
le = eng.lib.lc_packed_elems(1, ny_f, nx_f)
la, lo = lat.astype(f32), lon.astype(f32)
side = torch.cuda.Stream()
side_hi = torch.cuda.Stream(priority=-1)


def serial():
    f = eng.prepare_field(ud, vd, lat, lon, order)
    return eng.advect(f, slat_d, slon_d, dt, K, order, True)


def piped(chunk, advect_stream=None):
    cur = torch.cuda.current_stream()
    lin = eng._empty((le * nt,), f32)
    ext = eng._empty((le * (nt - 1),), f32)
    field = PackedField(lin, None, ext, nt, ny_f, nx_f, float(la[0]), float(la[-1]), float(lo[0]), float(lo[-1]), np.dtype(f32),
                        False, 1, False, None, None, None)
    x, y = eng._empty((n, n), f32), eng._empty((n, n), f32)
    side.wait_stream(cur)
    starts = list(range(0, nt - 1, chunk))
    events = []
    with torch.cuda.stream(side):
        eng._use_current_stream()
        for t0 in starts:
            m = min(chunk, nt - 1 - t0)
            _capi.check(eng.lib.lc_field_pack(eng.ctx, C.c_void_p(ud[t0:].data_ptr()), C.c_void_p(vd[t0:].data_ptr()), _capi.LC_F32,
                                              m + 1, ny_f, nx_f, 1, C.c_void_p(lin[le * t0:].data_ptr()),
                                              C.c_void_p(ext[le * t0:].data_ptr())), eng.lib)
            e = torch.cuda.Event()
            e.record(side)
            events.append(e)
    for e, t0 in zip(events, starts):
        cur.wait_event(e)
        m = min(chunk, nt - 1 - t0)
        eng.advect(field, slat_d, slon_d, dt, K, order, True, t0=t0, nsteps=m, start=(x, y) if t0 else None, out=(x, y))
    cur.wait_stream(side)
    return x, y


def timeit(fn, reps):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    a.record()
    for _ in range(reps):
        r = fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / reps, (time.perf_counter() - t0) * 1e3 / reps, r


out = {}
for rnd in range(2):
    ms, wall, ref = timeit(serial, reps)
    out[f"serial_{rnd}"] = (round(ms, 3), round(wall, 3))
    print("serial", ms, wall, flush=True)
    for chunk in (32, 16, 48, 8):
        ms, wall, r = timeit(lambda: piped(chunk), reps)
        same = bool(torch.equal(r[0], ref[0]) and torch.equal(r[1], ref[1]))
        out[f"piped{chunk}_{rnd}"] = (round(ms, 3), round(wall, 3), same)
        print("piped", chunk, ms, wall, same, eng.last_advect_kernel(), flush=True)
        assert same
print(json.dumps(out))
