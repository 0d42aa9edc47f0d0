"""``from LagrangianCoherence.LCS import trajectory`` -- HIP-backed parcel_propagation."""
from lagrangiancoherence_amd.dropin import parcel_propagation  # noqa: F401
from lagrangiancoherence_amd.tools import xr_map_coordinates  # noqa: F401
