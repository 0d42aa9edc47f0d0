"""Generate the golden fixtures under tests/golden/ from the CPU oracle.

    python tests/golden/make_golden.py

The reference itself cannot run in this image, so these
vectors are outputs of ``oracle/lcs_oracle.py`` -- the numpy+scipy restatement
that calls the same scipy/numpy kernels at the same call sites -- with
scipy/numpy versions recorded in each file.  They freeze the oracle's answers
so a later edit to the oracle or to the input generators cannot drift silently,
and they are what the GPU parity tests compare against on the GPU box (where
neither /root/reference nor a second opinion exists).

Cases (SURVEY.md section 8c G-1/G-2 plus a small fp32 seed-grid case):
  g1_*  config 1: examples/ideal_vortex.py:220-223 inputs, 89x180, nt=8, fp64
  g2    config 2 downsampled: 128x128 nodes, 20 steps, dt=-900, K=4, order 1
  g3    config 3 miniature: 72x144 fp32 field, 96x160 seeds, 12 steps, K=4
"""
import os
import sys

import numpy as np
import scipy

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))

from lagrangiancoherence_amd import flows  # noqa: E402
from oracle import lcs_oracle as O  # noqa: E402

META = dict(scipy=scipy.__version__, numpy=np.__version__)


def save(name, **arrs):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **arrs, **{k: np.array(v) for k, v in META.items()})
    print(f"{name}: {os.path.getsize(path) / 1024:.0f} KiB")


def g1():
    u, v, lat, lon = flows.config1()
    chk = np.array([u.sum(), v.sum(), np.abs(u).max(), np.abs(v).max()])
    # the four call forms of examples/ideal_vortex.py:262-288 (isglobal=True ->
    # cyclic; interp_to_common_grid=False, truncation=None skip the out-of-scope
    # regrid/truncation so the seeds stay 89x180), orders 3 (default) and 1
    for tag, dt, K in (("bwd_k4", -6 * 3600, 4), ("fwd_k2", 6 * 3600, 2), ("fwd_k4", 6 * 3600, 4)):
        for order in (3, 1):
            s, x, y = O.lcs(u, v, lat, lon, timestep=dt, SETTLS_order=K, interp_order=order,
                            cyclic_xboundary=True)
            save(f"g1_{tag}_o{order}", x_dep=x, y_dep=y, sigma=s, input_checksum=chk,
                 timestep=dt, SETTLS_order=K, interp_order=order)
    tx, ty = O.parcel_propagation(u, v, lat, lon, timestep=-6 * 3600, SETTLS_order=4, interp_order=3,
                                  cyclic_xboundary=True, return_traj=True)
    save("g1_traj_bwd_k4_o3", traj_x=tx, traj_y=ty, input_checksum=chk)


def g2():
    u, v, lat, lon = flows.config2(n=128, nt=21)
    chk = np.array([u.sum(), v.sum(), np.abs(u).max(), np.abs(v).max()])
    s, x, y = O.lcs(u, v, lat, lon, timestep=-900, SETTLS_order=4, interp_order=1, cyclic_xboundary=True)
    save("g2_c2_128_k4_o1", x_dep=x, y_dep=y, sigma=s, input_checksum=chk)


def g3():
    u, v, lat, lon = flows.era5_like(nt=13, ny=72, nx=144)
    slat, slon = flows.seed_grid(96, 160, lat, lon)
    chk = np.array([u.astype(np.float64).sum(), v.astype(np.float64).sum()])
    for order in (1, 3):
        # fp32 field and coordinates: the oracle follows numpy promotion (all fp32)
        s, x, y = O.lcs(u, v, lat, lon, timestep=-900, SETTLS_order=4, interp_order=order,
                        cyclic_xboundary=True, seed_lat=slat, seed_lon=slon)
        # and the same inputs widened to fp64: the "true" answer the fp32 tolerance is judged by
        s64, x64, y64 = O.lcs(u.astype(np.float64), v.astype(np.float64), lat.astype(np.float64),
                              lon.astype(np.float64), timestep=-900, SETTLS_order=4, interp_order=order,
                              cyclic_xboundary=True, seed_lat=slat.astype(np.float64),
                              seed_lon=slon.astype(np.float64))
        save(f"g3_c3mini_k4_o{order}", x_dep=x, y_dep=y, sigma=s, x_dep64=x64, y_dep64=y64, sigma64=s64,
             input_checksum=chk)


if __name__ == "__main__":
    g1()
    g2()
    g3()
