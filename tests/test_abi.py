"""CPU-side checks of the boundary: the library builds, loads and exports every
symbol include/pqd.h declares; no compute call is made here."""
import ctypes
import os
import re

import pytest

import __graft_entry__ as entry

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    text = open(os.path.join(ROOT, "include", "pqd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(pqd_[a-z0-9_]+)\s*\(", text)))


def test_header_symbols_are_exported():
    entry.build()
    pkg = entry.load_package()
    so = ctypes.CDLL(pkg.library_path())
    names = _declared_symbols()
    assert len(names) >= 20
    for name in names:
        assert hasattr(so, name), f"{name} declared in include/pqd.h but not exported"
    # and the Python binding table covers exactly the same set
    from parquet_go_amd._lib import SYMBOLS
    assert sorted(SYMBOLS) == names


def test_abi_version_and_status_names():
    pkg = entry.load_package()
    L = pkg.lib()
    assert L.pqd_abi_version() == 2
    assert L.pqd_status_name(0) == b"OK"
    assert L.pqd_status_name(9) == b"MISSING_VALUES"
    assert L.pqd_status_name(19) == b"CHECKSUM"
    assert L.pqd_can_decode_in_place(0) == 1 and L.pqd_can_decode_in_place(6) == 1
    assert L.pqd_can_decode_in_place(5) == 0


def test_page_struct_layout_matches_header():
    pkg = entry.load_package()
    # 10 x 8-byte fields, 4 x int32, 2 x int32, int64, 3 x uint64
    assert ctypes.sizeof(pkg.Page) == 10 * 8 + 4 * 4 + 2 * 4 + 8 + 3 * 8


def test_no_gpu_fails_loudly():
    import torch
    if torch.cuda.is_available():
        pytest.skip("a GPU is present")
    pkg = entry.load_package()
    with pytest.raises(pkg.PqdError):
        pkg.Context(0)


def test_product_does_not_import_the_oracle():
    pkg_dir = os.path.join(ROOT, "parquet-go_amd")
    for dirpath, _, files in os.walk(pkg_dir):
        for f in files:
            if f.endswith((".py", ".hip", ".h", ".hpp", ".cpp")):
                text = open(os.path.join(dirpath, f)).read()
                assert "pqoracle" not in text and "pq_oracle" not in text, f"{f} references the checker"
