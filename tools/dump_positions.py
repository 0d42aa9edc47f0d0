"""Results of a fixed set of float32 order-1 calls (two-seed kernels forced), for bit-for-bit comparison of library variants:
    LCS_LIB=a.so python tools/dump_positions.py a.npz ; LCS_LIB=b.so python tools/dump_positions.py b.npz
    python tools/dump_positions.py --compare a.npz b.npz"""
import sys
import numpy as np
sys.path.insert(0, ".")
if sys.argv[1] == "--compare":
    a, b = np.load(sys.argv[2]), np.load(sys.argv[3])
    bad = 0
    for k in a.files:
        if k.startswith("kernel"):
            print(k, str(a[k]), "|", str(b[k]))
            continue
        same = a[k].tobytes() == b[k].tobytes()
        n = int((a[k].view(np.uint32) != b[k].view(np.uint32)).sum())
        print(f"{k:28s} {'identical' if same else 'DIFFERENT: %d of %d words' % (n, a[k].size)}")
        bad += not same
    sys.exit(1 if bad else 0)
from lagrangiancoherence_amd import flows
from lagrangiancoherence_amd.engine import Engine
eng = Engine(0)
eng.set_lds_tiles(1)  # two seeds per lane at order 1 whatever the size
out = {}
u, v, lat, lon = flows.era5_like(nt=25)
f = eng.prepare_field(eng.to_device(u, np.float32), eng.to_device(v, np.float32), lat, lon, 1)
for name, (ny, nx) in {"dense": (1400, 2000), "ragged": (701, 1003), "sparse": (300, 520)}.items():
    slat, slon = flows.seed_grid(ny, nx, lat, lon)
    for K in (4, 2):
        x, y = eng.advect(f, slat, slon, -900.0, K, 1, True)
        out[f"{name}_k{K}_x"], out[f"{name}_k{K}_y"] = x.cpu().numpy(), y.cpu().numpy()
        out[f"kernel_{name}_k{K}"] = eng.last_advect_kernel()
    x, y = eng.advect(f, slat, slon, 3600.0, 4, 1, False, noncyclic_clamp="pointwise")
    out[f"{name}_noncyclic_x"], out[f"{name}_noncyclic_y"] = x.cpu().numpy(), y.cpu().numpy()
    out[f"kernel_{name}_noncyclic"] = eng.last_advect_kernel()
slat, slon = flows.seed_grid(704, 1000, lat, lon)
x, y, tx, ty = eng.advect(f, slat, slon, -900.0, 4, 1, True, return_traj=True)
out["traj_x"], out["traj_y"] = tx.cpu().numpy(), ty.cpu().numpy()
out["kernel_traj"] = eng.last_advect_kernel()
slat, slon = flows.seed_grid(512, 512, lat, lon)
x, y = eng.advect_batch(f, slat, slon, -900.0, 5, 16, 4, 1, True, t0=0, t0_stride=2)
out["batch_x"], out["batch_y"] = x.cpu().numpy(), y.cpu().numpy()
out["kernel_batch"] = eng.last_advect_kernel()
np.savez(sys.argv[1], **out)
print({k: str(v) for k, v in out.items() if k.startswith("kernel")})
