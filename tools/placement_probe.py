"""Does the float64 advect kernel's time depend on WHERE its images sit in memory?  (round 4: the same advect measured 3.13-3.16 ms
in one allocation pattern and 3.37-3.39 in another.)  Places the ext image at byte offsets inside a larger buffer and times
lc_advect on BASELINE configs[1]:  python tools/placement_probe.py [order]"""
import sys
sys.path.insert(0, ".")
import ctypes as C
import numpy as np
import torch
from lagrangiancoherence_amd import flows, _capi
from lagrangiancoherence_amd.engine import Engine, PackedField

order = int(sys.argv[1]) if len(sys.argv) > 1 else 1
eng = Engine(0)
ud, vd, lat, lon = flows.config2_on_device(torch, eng.device)
NT, ny_f, nx_f = (int(n) for n in ud.shape)
sl, so = eng.to_device(lat, np.float64), eng.to_device(lon, np.float64)
le = eng.lib.lc_packed_elems(1, ny_f, nx_f)
PADE = 1 << 21      # elements of slack (16 MB)
big_ext = torch.empty(le * (NT - 1) + PADE, dtype=torch.float64, device="cuda")
big_cub = torch.empty(le * NT + PADE, dtype=torch.float64, device="cuda") if order == 3 else None
x = torch.empty((ny_f, nx_f), dtype=torch.float64, device="cuda")
y = torch.empty_like(x)
P = lambda t: C.c_void_p(t.data_ptr()) if t is not None else None
print("base addresses: u %x v %x ext %x" % (ud.data_ptr(), vd.data_ptr(), big_ext.data_ptr()))
for off_bytes in (0, 256, 1024, 4096, 8192, 65536, 1 << 20, (1 << 20) + 4096, 2 << 20, (2 << 20) + 65536, 4 << 20, 8 << 20):
    o = off_bytes // 8
    ext = big_ext[o:o + le * (NT - 1)]
    cub = big_cub[o:o + le * NT] if order == 3 else None
    eng._use_current_stream()
    _capi.check(eng.lib.lc_field_pack(eng.ctx, P(ud), P(vd), _capi.LC_F64, NT, ny_f, nx_f, order, P(cub), P(ext)), eng.lib)
    f = PackedField(None, cub, ext, NT, ny_f, nx_f, float(lat[0]), float(lat[-1]), float(lon[0]), float(lon[-1]), np.dtype(np.float64),
                    False, order, False, ud, vd)
    for _ in range(2):
        eng.advect(f, sl, so, -900.0, 4, order, True, out=(x, y))
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(8):
        eng.advect(f, sl, so, -900.0, 4, order, True, out=(x, y))
    e1.record()
    torch.cuda.synchronize()
    print(f"order {order} ext offset {off_bytes:>9d} B: advect {e0.elapsed_time(e1) / 8:.3f} ms")
