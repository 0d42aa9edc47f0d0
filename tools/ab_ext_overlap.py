"""float64 order 3 on BASELINE configs[1]: the ext image built chunk by chunk on a side stream (lc_field_extrapolate) WHILE the
advect kernel runs, after a pack that only filters (lc_field_pack without ext) -- against the serial product:
    python tools/ab_ext_overlap.py [chunk ...]"""
import ctypes as C
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from lagrangiancoherence_amd import _capi, flows
from lagrangiancoherence_amd.engine import Engine, PackedField, _NP2LC

chunks = [int(a) for a in sys.argv[1:]] or [16, 32, 64]
eng = Engine(0)
u, v, lat, lon = flows.config2_on_device(torch, eng.device)
dtype = np.dtype(np.float64)
la, lo = eng.to_device(lat, dtype), eng.to_device(lon, dtype)
nt, ny_f, nx_f = (int(s) for s in u.shape)
le = eng.lib.lc_packed_elems(1, ny_f, nx_f)
side = torch.cuda.Stream(eng.device)


def serial():
    f = eng.prepare_field(u, v, lat, lon, 3)
    return eng.advect(f, la, lo, -900.0, 4, 3, True)


def overlapped(chunk):
    cur = torch.cuda.current_stream(eng.device)
    cub = eng._empty((le * nt,), dtype)
    ext = eng._empty((le * (nt - 1),), dtype)
    eng._use_current_stream()
    _capi.check(eng.lib.lc_field_pack(eng.ctx, eng._ptr(u), eng._ptr(v), _NP2LC[dtype], nt, ny_f, nx_f, 3, eng._ptr(cub), None), eng.lib)
    lat64, lon64 = np.asarray(lat, dtype), np.asarray(lon, dtype)
    field = PackedField(None, cub, ext, nt, ny_f, nx_f, float(lat64[0]), float(lat64[-1]), float(lon64[0]), float(lon64[-1]), dtype,
                        False, 3, False, u, v, (u._version, v._version))
    x, y = eng._empty((ny_f, nx_f), dtype), eng._empty((ny_f, nx_f), dtype)
    side.wait_stream(cur)
    starts, events = list(range(0, nt - 1, chunk)), []
    with torch.cuda.stream(side):
        eng._use_current_stream()
        for t0 in starts:
            n = min(chunk, nt - 1 - t0)
            _capi.check(eng.lib.lc_field_extrapolate(eng.ctx, C.c_void_p(cub[le * t0:].data_ptr()), _NP2LC[dtype], n + 1, ny_f, nx_f,
                                                     C.c_void_p(ext[le * t0:].data_ptr())), eng.lib)
            e = torch.cuda.Event()
            e.record(side)
            events.append(e)
    for e, t0 in zip(events, starts):
        cur.wait_event(e)
        n = min(chunk, nt - 1 - t0)
        eng.advect(field, la, lo, -900.0, 4, 3, True, t0=t0, nsteps=n, start=(x, y) if t0 else None, out=(x, y))
    cur.wait_stream(side)
    return x, y


def timed(fn, *a):
    for _ in range(2):
        r = fn(*a)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        r = fn(*a)
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 5, r


t, ref = timed(serial)
print(f"serial product (pack with ext, then advect): {t:.3f} ms")
for c in chunks:
    t, r = timed(overlapped, c)
    print(f"filter-only pack, then ext by chunks of {c} levels on a side stream beside the advect: {t:.3f} ms, bit-identical: "
          f"{bool((r[0] == ref[0]).all()) and bool((r[1] == ref[1]).all())}")
t, _ = timed(serial)
print(f"serial product again: {t:.3f} ms")
