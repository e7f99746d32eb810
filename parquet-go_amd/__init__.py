"""MI355X-native Parquet page-value decode: Python plumbing over libpqd.so.

The product is the HIP library behind the C ABI of include/pqd.h; this package
only loads it with ctypes, moves bytes to and from HBM with torch, and mirrors
the reference's codec objects (`encoding.py`) so tests and benchmarks read like
parquet-go's own.  The directory name carries a hyphen, so import it through
`__graft_entry__.load_package()` (registers it as `parquet_go_amd`).
"""
from . import _lib  # noqa: F401
from ._lib import (Context, Page, PqdError, STATUS, KIND, ENC, lib, library_path)  # noqa: F401
from .runtime import Batch, DeviceBuffer, decode_pages, to_device  # noqa: F401
from . import chunk, encoding, file, levels, sharding  # noqa: F401
