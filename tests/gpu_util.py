"""Shared helpers of the GPU parity tests: one context per session, batched decode
through the C ABI, and comparison with the CPU checker."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

import __graft_entry__ as entry  # noqa: E402

_state = {}


def pkg():
    if "pkg" not in _state:
        _state["pkg"] = entry.load_package()
    return _state["pkg"]


def ctx():
    if "ctx" not in _state:
        _state["ctx"] = pkg().Context(0)
    return _state["ctx"]


def spec(*a, **k):
    from parquet_go_amd.runtime import PageSpec
    pkg()
    return PageSpec(*a, **k)


def decode(specs, grow=True):
    from parquet_go_amd.runtime import decode_pages
    return decode_pages(ctx(), specs, grow=grow)


def assert_same(out, want, dtype, what="", values=True):
    """GPU result `out` (PageOut) against the checker's `want` (oracle Result)."""
    assert out.status == want.status, f"{what}: status {out.status} != {want.status} (args {out.arg}, {want.arg})"
    if want.status != 0:
        assert out.arg == want.arg, f"{what}: status arg {out.arg} != {want.arg}"
    if values:
        got = out.data.view(dtype) if out.data.size % np.dtype(dtype).itemsize == 0 else out.data
        exp = want.values
        assert got.size == exp.size, f"{what}: {got.size} values, want {exp.size}"
        if not np.array_equal(got, exp):
            bad = np.nonzero(got != exp)[0]
            raise AssertionError(f"{what}: {bad.size} mismatches, first at {bad[0]}: got {got[bad[0]]} want {exp[bad[0]]}")
