"""The reference's examples/ideal_vortex.py (lines 211-288) on the HIP engine, without the plotting.

    python examples/ideal_vortex_hip.py [out.npz]

Builds the "unsteady" subtropical vortex of the example (89 x 180 nodes, 8 six-hourly levels), then runs
the same four computations through the drop-in import paths: backward and forward trajectories with
``return_traj=True`` and the repelling / attracting FTLE fields ``log(sigma)/2`` of
``LCS(...)(ds, isglobal=True)`` (0.5 degree regrid + T20 truncation included, as in the reference's defaults).
Uses xarray objects when xarray is installed, the tests' labelled stand-ins (tests/labelled.py) otherwise.
"""
import os
import sys

import numpy as np
import pandas as pd

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))

from LagrangianCoherence.LCS import LCS, trajectory  # noqa: E402
from lagrangiancoherence_amd import flows  # noqa: E402


def dataset():
    u, v, lat, lon = flows.ideal_vortex(**flows.vortex_config_subtropical)
    coords = {'latitude': lat, 'longitude': lon, 'time': pd.date_range('2000-01-01', periods=u.shape[0], freq='6h')}
    dims = ['latitude', 'longitude', 'time']
    try:
        import xarray as xr
        return xr.Dataset({'u': xr.DataArray(u.transpose(1, 2, 0), dims=dims, coords=coords),
                           'v': xr.DataArray(v.transpose(1, 2, 0), dims=dims, coords=coords)})
    except ImportError:
        from tests import labelled          # xarray-free stand-in: the drop-in returns what it is given
        coords['time'] = coords['time'].values
        return labelled.Dataset({'u': labelled.DataArray(u.transpose(1, 2, 0), dims, coords, name='u'),
                                 'v': labelled.DataArray(v.transpose(1, 2, 0), dims, coords, name='v')})


def main():
    ds = dataset()
    x_dye, y_dye = trajectory.parcel_propagation(ds.u, ds.v, timestep=-6 * 3600, propdim='time', SETTLS_order=4,
                                                 copy=True, return_traj=True, cyclic_xboundary=True, verbose=False)
    x, y = trajectory.parcel_propagation(ds.u, ds.v, timestep=6 * 3600, propdim='time', SETTLS_order=2, copy=True,
                                         return_traj=True, cyclic_xboundary=True, verbose=False)
    rcs = LCS.LCS(timestep=6 * 3600, timedim='time', SETTLS_order=4)
    ftle_r = np.log(np.asarray(rcs(ds.copy(), isglobal=True, verbose=False).values)) / 2
    acs = LCS.LCS(timestep=-6 * 3600, timedim='time', SETTLS_order=4)
    ftle_a = np.log(np.asarray(acs(ds.copy(), isglobal=True, verbose=False).values)) / 2
    print(f"backward trajectories {tuple(x_dye.shape)}, forward {tuple(x.shape)}")
    print(f"repelling FTLE  {ftle_r.shape}: min {np.nanmin(ftle_r):.4f} max {np.nanmax(ftle_r):.4f}")
    print(f"attracting FTLE {ftle_a.shape}: min {np.nanmin(ftle_a):.4f} max {np.nanmax(ftle_a):.4f}")
    if len(sys.argv) > 1:
        np.savez_compressed(sys.argv[1], x_dye=np.asarray(x_dye.values), y_dye=np.asarray(y_dye.values),
                            ftle_r=ftle_r, ftle_a=ftle_a)
        print("wrote", sys.argv[1])


if __name__ == "__main__":
    main()
