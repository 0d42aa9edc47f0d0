"""MI355X-native FTLE engine: HIP kernels behind a C ABI, drop-in for LagrangianCoherence's
parcel-advection -> flow-map-gradient -> sigma_max path.

    from LagrangianCoherence.LCS import LCS, trajectory        # the reference's import paths
    from lagrangiancoherence_amd.engine import Engine          # array-level API on device tensors

See DESIGN.md (what is built and what bounds it) and INTEGRATION.md (how to bind the C ABI).
"""
__version__ = "0.1.0"
