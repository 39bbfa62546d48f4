"""myscaledb_amd -- MI355X-native vector-scan / BM25 hot path for MyScaleDB.

The product is the C-ABI shared library `libmsvs.so` (include/msvs.h, sources in
myscaledb_amd/csrc/).  This Python package is only plumbing around it: a ctypes
binding (capi.py) used by the tests, bench.py and the multi-GPU driver (sharded.py).
There is no CPU fallback: importing `capi` without the built library raises.
"""
__all__ = ["capi"]
