import sys, numpy as np
sys.path.insert(0, '.')
from lagrangiancoherence_amd import flows
from lagrangiancoherence_amd.engine import Engine
from oracle import lcs_oracle as O
np.set_printoptions(precision=6, linewidth=220, suppress=True)
eng = Engine(0)
u, v, lat, lon = flows.era5_like(nt=97)
slat, slon = flows.seed_grid(4096, 4096, lat, lon)
rows = np.array([0, 2027, 2028, 2029, 4095]); cols = np.array([1039, 1040, 1041])
f = eng.prepare_field(u, v, lat, lon, 1)
x, y, tx, ty = eng.advect(f, slat[rows], slon[cols], -900.0, 4, 1, True, return_traj=True)
tx, ty = tx.cpu().numpy(), ty.cpu().numpy()
ox, oy = O.parcel_propagation(u, v, lat, lon, timestep=-900.0, SETTLS_order=4, interp_order=1, cyclic_xboundary=True,
                              seed_lat=slat[rows], seed_lon=slon[cols], return_traj=True)
d = np.abs(tx - ox); d = np.minimum(d, np.abs(d - 360))
first = np.argwhere(d > 1e-2)
print("first divergences (step,row,col):", first[:5].tolist())
if len(first):
    s, i, j = first[0]
    for k in range(max(s - 3, 0), min(s + 3, 97)):
        print(k, "gpu", tx[k, i, j], ty[k, i, j], "oracle", ox[k, i, j], oy[k, i, j])
