"""``from LagrangianCoherence.LCS.LCS import LCS`` -- HIP-backed (see lagrangiancoherence_amd.dropin)."""
from lagrangiancoherence_amd.dropin import LCS, flowmap_gradient, parcel_propagation  # noqa: F401
from lagrangiancoherence_amd.tools import derivative_spherical_coords, fourth_order_derivative  # noqa: F401
