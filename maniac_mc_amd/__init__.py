"""MI355X-native (gfx950, hand-written HIP) energy engine for MANIAC's per-move GCMC hot path.

csrc/      HIP kernels + the C ABI declared in include/maniac_gpu.h  -> libmaniac_hip.so
fortran/   ISO_C_BINDING interface module and the Fortran host Metropolis driver
engine.py  ctypes mirror of the C ABI (plumbing for tests, smoke and bench)
system.py  the reference's state (residue types, com + offsets, force field) in numpy
synth.py   synthetic benchmark / parity systems
"""
from . import _lib, synth, system  # noqa: F401

__all__ = ["_lib", "synth", "system"]
