"""``from LagrangianCoherence.LCS.tools import ...`` -- the hot-path helpers, HIP-backed."""
from lagrangiancoherence_amd.tools import (derivative_spherical_coords, find_ridges_spherical_hessian,  # noqa: F401
                                           fourth_order_derivative, xr_map_coordinates)
