"""fast_amd: MI355X-native FastS structured-grid compressible Navier-Stokes time step.

Host side mirrors the reference's FastS.PyTree / FastC.PyTree entry points; all numerical
work of the time step runs in hand-written HIP kernels behind the C-ABI declared in
include/fasts_hip.h (library: fast_amd/csrc/libfasts_hip.so).
"""
__version__ = "0.1.0"
