"""jrl-walkgen_amd -- MI355X-native ZMP-MPC hot path of jrl-walkgen.

The product is the C-ABI shared library `lib/libwg_mpc.so` (HIP kernels for
gfx950 + thin C++ host layer, see include/wg_mpc.h).  This package only holds
the ctypes binding used by the tests and the benchmark; it contains no compute
of its own and no CPU fallback: if the library or a GPU is missing, calls fail.

The directory name carries a hyphen (it mirrors the reference's name), so import
it with  importlib.import_module("jrl-walkgen_amd").
"""
from .wgmpc import *  # noqa: F401,F403
