"""spartan-bn254_amd — ctypes loader for libsbn254_hip.so (the C ABI in include/sbn254.h).

The product is the shared library; this module only lets Python tests and bench.py call it the way the
Rust shim would.  It contains no arithmetic and no fallback: if the HIP library or a
gfx950 device is missing, loading / context creation raises.
"""
from .binding import (  # noqa: F401
    Context, Group, Bases, Table, SbnError, lib, lib_path, build_library,
    SBN_SCALARS_MONT, SBN_POINTS_MONT, g1_compress, g1_sum, unipoly_from_evals, unipoly_eval, factored_lens, EXPORTED_SYMBOLS,
)
