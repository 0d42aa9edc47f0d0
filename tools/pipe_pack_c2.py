"""Experiment (float64 config 2): pack the fused-level image chunk by chunk on a side stream while the advect kernel
works on the previous chunk -- the pack is HBM-bound (6.7 TB/s), the float64 advect kernel latency-bound (HBM at 0.43).
    python tools/pipe_pack_c2.py [order]"""
import ctypes as C
import sys
import time
sys.path.insert(0, '.')
import numpy as np
import torch
from lagrangiancoherence_amd import flows, _capi
from lagrangiancoherence_amd.engine import Engine, PackedField

order = int(sys.argv[1]) if len(sys.argv) > 1 else 1
CH = int(sys.argv[2]) if len(sys.argv) > 2 else 32
which = sys.argv[3] if len(sys.argv) > 3 else "c2"
eng = Engine(0)
if which == "c2":
    ud, vd, lat, lon = flows.config2_on_device(torch, eng.device)
    slat, slon, npdt, tdt, lc = lat, lon, np.float64, torch.float64, _capi.LC_F64
else:   # c3: float32, order 3 (order 1 keeps its lin image: not this script)
    u, v, lat, lon = flows.era5_like(nt=97)
    ud, vd = eng.to_device(u, np.float32), eng.to_device(v, np.float32)
    slat, slon = flows.seed_grid(4096, 4096, lat, lon)
    npdt, tdt, lc = np.float32, torch.float32, _capi.LC_F32
NT, ny_f, nx_f = (int(n) for n in ud.shape)
sl, so = eng.to_device(slat, npdt), eng.to_device(slon, npdt)
le = eng.lib.lc_packed_elems(1, ny_f, nx_f)
ext = torch.empty(le * (NT - 1), dtype=tdt, device='cuda')
cub = torch.empty(le * NT, dtype=tdt, device='cuda') if order == 3 else None
x = torch.empty((len(slat), len(slon)), dtype=tdt, device='cuda')
y = torch.empty_like(x)
f = PackedField(None, cub, ext, NT, ny_f, nx_f, float(lat[0]), float(lat[-1]), float(lon[0]), float(lon[-1]), np.dtype(npdt),
                False, order, False, ud, vd)
A, B = torch.cuda.current_stream(), torch.cuda.Stream()
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None


def pack_range(t0, t1):   # image levels [t0, t1], ext levels [t0, t1)
    eng._use_current_stream()
    _capi.check(eng.lib.lc_field_pack(eng.ctx, P(ud[t0:]), P(vd[t0:]), lc, t1 - t0 + 1, ny_f, nx_f, order,
                                      P(cub[le * t0:]) if order == 3 else None, P(ext[le * t0:])), eng.lib)


def serial():
    pack_range(0, NT - 1)
    eng.set_level_chunk(CH)
    eng.advect(f, sl, so, -900.0, 4, order, True, out=(x, y))


def piped():
    eng.set_level_chunk(0)
    evs = []
    B.wait_stream(A)
    with torch.cuda.stream(B):
        for t0 in range(0, NT - 1, CH):
            pack_range(t0, min(t0 + CH, NT - 1))
            e = torch.cuda.Event()
            e.record(B)
            evs.append(e)
    for c, t0 in enumerate(range(0, NT - 1, CH)):
        A.wait_event(evs[c])
        n = min(CH, NT - 1 - t0)
        eng.advect(f, sl, so, -900.0, 4, order, True, t0=t0, nsteps=n, start=(x, y) if t0 else None, out=(x, y))
    B.wait_stream(A)


ref = None
for name, fn in (("serial", serial), ("piped", piped), ("serial", serial), ("piped", piped)):
    fn()
    torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    s = float(x.sum())
    ref = s if ref is None else ref
    print(f"{which} order {order} chunk {CH} {name}: {(time.perf_counter() - t) * 100:.3f} ms per pack+advect, checksum equal {s == ref}")
