"""trust4_amd -- MI355X-native engine for TRUST4's stage-1 seed -> chain -> extend hot path.

The product is the C-ABI library ``trust4_amd/libt4hip.so`` (include/trust4_hip.h), built by
``trust4_amd.build`` with hipcc for gfx950. This package only holds the thin ctypes mirror of that
ABI used by tests and bench.py. There is NO CPU fallback: importing works anywhere, but creating an
``Engine`` raises if the HIP library is missing or no GPU is present.
"""
from .api import Engine, Index, Batch, Assembler, CellSet, T4Error, OV_DTYPE, HIT_DTYPE, lib_path  # noqa: F401
