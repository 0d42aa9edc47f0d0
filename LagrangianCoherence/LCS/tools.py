"""``from LagrangianCoherence.LCS.tools import ...`` -- the hot-path helpers, HIP-backed."""
from lagrangiancoherence_amd.tools import (derivative_spherical_coords, fourth_order_derivative,  # noqa: F401
                                           xr_map_coordinates)
