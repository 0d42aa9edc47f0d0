"""Import-path shim: the reference is imported as ``LagrangianCoherence.LCS.*``
(examples/ideal_vortex.py:5,8; LCS/LCS.py:12,15).  Everything here re-exports the
MI355X engine's drop-in surface from ``lagrangiancoherence_amd.dropin``."""
