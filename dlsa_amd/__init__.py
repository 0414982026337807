"""dlsa_amd -- MI355X (gfx950) Smith-Waterman engine behind DLSA's worker C ABI.

The product is the shared library dlsa_amd/lib/libsw_hip.so (C ABI in include/sw.h and
include/sw_hip.h).  This package only holds the build script and a ctypes binding that mirrors the Go
worker's cgo bridge so that tests and bench.py can call the library the way the worker does.
"""
from .bridge import (  # noqa: F401
    AlignmentScore, GoResult, SwHipError, find_alignment_with_fallback, find_rust_alignment_sequential,
    find_rust_alignment_simd, find_rust_alignment_simd_low_mem, lib, lib_path)
