"""Experiment: pack the wind series level chunk by level chunk on a side stream while the advect kernel works on the
previous chunk (the pack is HBM-bound, the advect VALU-bound).  Prints ms per step for the serial and the pipelined form."""
import sys, time
sys.path.insert(0, '.')
import numpy as np, torch
from lagrangiancoherence_amd import flows, _capi
from lagrangiancoherence_amd.engine import Engine, PackedField, _NP2LC
NS, NT, CH = 4096, 97, 32
u, v, lat, lon = flows.era5_like(nt=NT)
slat, slon = flows.seed_grid(NS, NS, lat, lon)
eng = Engine(0)
ud, vd = eng.to_device(u, np.float32), eng.to_device(v, np.float32)
sl, so = eng.to_device(slat, np.float32), eng.to_device(slon, np.float32)
ny_f, nx_f = u.shape[1:]
le = eng.lib.lc_packed_elems(1, ny_f, nx_f)
lin = torch.empty(le * NT, dtype=torch.float32, device='cuda'); ext = torch.empty(le * (NT - 1), dtype=torch.float32, device='cuda')
x = torch.empty((NS, NS), dtype=torch.float32, device='cuda'); y = torch.empty_like(x)
f = PackedField(lin, None, ext, NT, ny_f, nx_f, float(lat[0]), float(lat[-1]), float(lon[0]), float(lon[-1]), np.dtype(np.float32))
A, B = torch.cuda.current_stream(), torch.cuda.Stream()
P = lambda t: __import__('ctypes').c_void_p(t.data_ptr())
def pack_range(t0, t1):   # lin levels [t0, t1], ext levels [t0, t1)
    eng._use_current_stream()
    _capi.check(eng.lib.lc_field_pack(eng.ctx, P(ud[t0:]), P(vd[t0:]), 0, t1 - t0 + 1, ny_f, nx_f, 1, P(lin[le * t0:]), P(ext[le * t0:])), eng.lib)
def serial():
    pack_range(0, NT - 1)
    eng.set_level_chunk(CH)
    eng.advect(f, sl, so, -900.0, 4, 1, True, out=(x, y))
def piped():
    eng.set_level_chunk(0)
    evs = []
    B.wait_stream(A)
    with torch.cuda.stream(B):
        for t0 in range(0, NT - 1, CH):
            pack_range(t0, min(t0 + CH, NT - 1))
            e = torch.cuda.Event(); e.record(B); evs.append(e)
    for c, t0 in enumerate(range(0, NT - 1, CH)):
        A.wait_event(evs[c])
        n = min(CH, NT - 1 - t0)
        eng.advect(f, sl, so, -900.0, 4, 1, True, t0=t0, nsteps=n, start=(x, y) if t0 else None, out=(x, y))
    B.wait_stream(A)
for name, fn in (("serial", serial), ("piped", piped), ("serial", serial), ("piped", piped)):
    fn(); torch.cuda.synchronize()
    t = time.perf_counter()
    for _ in range(10): fn()
    torch.cuda.synchronize()
    print(name, "%.3f ms per pack+advect" % ((time.perf_counter() - t) * 100), float(x.sum()))
