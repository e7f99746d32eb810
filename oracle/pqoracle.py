"""ctypes front end of the CPU checker (oracle/libpqoracle.so).

TEST INFRASTRUCTURE ONLY.  Imported by tests/, the smoke check and the
CPU-baseline leg of the benchmark; the product package never imports it.
Each wrapper returns a `Result` holding the status pair and numpy views of
what the decoder produced, so tests can compare it with the GPU path on the
same input.
"""
from __future__ import annotations

import ctypes as C
import os
import subprocess
from dataclasses import dataclass
from typing import Optional

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "libpqoracle.so")


class _Res(C.Structure):
    _fields_ = [("status", C.c_int32), ("arg", C.c_int64), ("n_values", C.c_uint64),
                ("n_bytes", C.c_uint64), ("consumed", C.c_uint64)]


@dataclass
class Result:
    status: int
    arg: int
    n_values: int
    n_bytes: int
    consumed: int
    values: Optional[np.ndarray] = None
    offsets: Optional[np.ndarray] = None


def build(force: bool = False) -> str:
    srcs = [os.path.join(_HERE, f) for f in ("pq_oracle.c", "pq_port_opt.c", "pq_oracle.h")]
    if force or not os.path.exists(_SO) or any(os.path.getmtime(_SO) < os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-B", "libpqoracle.so"], stdout=subprocess.DEVNULL)
    return _SO


_lib = None


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = C.CDLL(_SO)
        u8p, u32p, i32p, i64p = (C.POINTER(C.c_uint8), C.POINTER(C.c_uint32), C.POINTER(C.c_int32), C.POINTER(C.c_int64))
        vp, u64, u32, i64 = C.c_void_p, C.c_uint64, C.c_uint32, C.c_int64
        sig = {
            "pqo_rle_decode_levels": (_Res, [vp, u64, vp, u64, u32]),
            "pqo_rle_decode_boolean": (_Res, [vp, u64, vp, u64]),
            "pqo_rle_decode_int32": (_Res, [vp, u64, vp, u64, u32]),
            "pqo_rle_dict_decode_int32": (_Res, [vp, u64, vp, u64]),
            "pqo_bitpacked_decode_levels": (_Res, [vp, u64, vp, u64, u32]),
            "pqo_delta_decode_int32": (_Res, [vp, u64, vp, u64]),
            "pqo_delta_decode_int64": (_Res, [vp, u64, vp, u64]),
            "pqo_dlba_decode": (_Res, [vp, u64, vp, u64, vp, u64]),
            "pqo_dba_decode": (_Res, [vp, u64, vp, u64, vp, u64]),
            "pqo_dba_decode_flba": (_Res, [vp, u64, vp, u64, i64]),
            "pqo_bss_decode": (_Res, [vp, u64, vp, u64, i64, C.c_int]),
            "pqo_plain_decode_fixed": (_Res, [vp, u64, vp, u64, i64, C.c_int]),
            "pqo_plain_decode_byte_array": (_Res, [vp, u64, vp, u64, vp, u64]),
            "pqo_dlba_estimate": (i64, [vp, u64]),
            "pqo_dba_estimate": (i64, [vp, u64]),
            "pqo_dict_lookup": (_Res, [vp, u64, vp, u64, u32, vp, u64]),
            "pqopt_delta_decode_int64": (_Res, [vp, u64, vp, u64]),
            "pqopt_rle_dict_lookup_int64": (_Res, [vp, u64, vp, u64, vp, u64]),
            "pqo_indexed_page_fixup": (None, [vp, u64, u64]),
            "pqo_rle_encode_levels": (u64, [vp, u64, vp, u64, u32]),
            "pqo_rle_encode_int32": (u64, [vp, u64, vp, u64, u32]),
            "pqo_rle_encode_boolean": (u64, [vp, u64, vp, u64]),
            "pqo_rle_dict_encode_int32": (u64, [vp, u64, vp, u64]),
            "pqo_bitpacked_encode_levels": (u64, [vp, u64, vp, u64, u32]),
            "pqo_delta_encode_int32": (u64, [vp, u64, vp, u64, u32, u32]),
            "pqo_delta_encode_int64": (u64, [vp, u64, vp, u64, u32, u32]),
            "pqo_dlba_encode": (u64, [vp, u64, vp, vp, u64]),
            "pqo_dba_encode": (u64, [vp, u64, vp, vp, u64]),
            "pqo_bss_encode": (u64, [vp, vp, u64, u64]),
            "pqo_plain_encode_byte_array": (u64, [vp, u64, vp, vp, u64]),
            "pqo_rle_encode_levels_mr": (u64, [vp, u64, vp, u64, u32]),
            "pqo_selftest_unpack": (C.c_int, []),
        }
        for name, (res, args) in sig.items():
            f = getattr(L, name)
            f.restype = res
            f.argtypes = args
        _lib = L
    return _lib


def _src(buf) -> np.ndarray:
    a = np.frombuffer(bytes(buf), dtype=np.uint8) if not isinstance(buf, np.ndarray) else np.ascontiguousarray(buf).view(np.uint8).reshape(-1)
    return a


def _p(a: Optional[np.ndarray]):
    return None if a is None else a.ctypes.data


def _mk(res: _Res, values=None, offsets=None) -> Result:
    return Result(res.status, res.arg, res.n_values, res.n_bytes, res.consumed, values, offsets)


def _retry(call, dtype, cap_bytes):
    """Run `call(dst, cap_bytes)`; grow and retry once on DST_TOO_SMALL like a Go caller would."""
    for _ in range(48):
        dst = np.zeros(max(cap_bytes, 8), dtype=np.uint8)
        res = call(dst, cap_bytes)
        if res.status == 16 and res.arg > cap_bytes:
            cap_bytes = max(int(res.arg), 2 * cap_bytes)
            continue
        return res, dst[: res.n_bytes].view(dtype) if res.n_bytes % np.dtype(dtype).itemsize == 0 else dst[: res.n_bytes]
    raise RuntimeError("oracle retry loop did not converge")


# ---- decoders ---------------------------------------------------------------

def rle_decode_levels(src, bit_width: int, cap: int = 0) -> Result:
    s = _src(src)
    res, v = _retry(lambda d, c: lib().pqo_rle_decode_levels(_p(d), c, _p(s), s.size, bit_width), np.uint8, cap or 8 * s.size + 64)
    return _mk(res, v)


def rle_decode_boolean(src, cap: int = 0) -> Result:
    s = _src(src)
    res, v = _retry(lambda d, c: lib().pqo_rle_decode_boolean(_p(d), c, _p(s), s.size), np.uint8, cap or s.size + 64)
    return _mk(res, v)


def rle_decode_int32(src, bit_width: int, cap: int = 0) -> Result:
    s = _src(src)
    res, v = _retry(lambda d, c: lib().pqo_rle_decode_int32(_p(d), c, _p(s), s.size, bit_width), np.int32, cap or 64 * s.size + 64)
    return _mk(res, v)


def rle_dict_decode_int32(src, cap: int = 0) -> Result:
    s = _src(src)
    res, v = _retry(lambda d, c: lib().pqo_rle_dict_decode_int32(_p(d), c, _p(s), s.size), np.int32, cap or 64 * s.size + 64)
    return _mk(res, v)


def bitpacked_decode_levels(src, bit_width: int, cap: int = 0) -> Result:
    s = _src(src)
    res, v = _retry(lambda d, c: lib().pqo_bitpacked_decode_levels(_p(d), c, _p(s), s.size, bit_width), np.uint8, cap or 8 * s.size + 64)
    return _mk(res, v)


def delta_decode_int32(src, cap: int = 0) -> Result:
    s = _src(src)
    res, v = _retry(lambda d, c: lib().pqo_delta_decode_int32(_p(d), c, _p(s), s.size), np.int32, cap or 64)
    return _mk(res, v)


def delta_decode_int64(src, cap: int = 0) -> Result:
    s = _src(src)
    res, v = _retry(lambda d, c: lib().pqo_delta_decode_int64(_p(d), c, _p(s), s.size), np.int64, cap or 64)
    return _mk(res, v)


def _byte_array(fn, src, cap, ocap) -> Result:
    s = _src(src)
    cap = cap or max(64, 4 * s.size)
    ocap = ocap or 64
    for _ in range(48):
        dst = np.zeros(cap, dtype=np.uint8)
        off = np.zeros(ocap, dtype=np.uint32)
        res = fn(_p(dst), cap, _p(off), ocap, _p(s), s.size)
        if res.status == 16 and res.arg > cap:
            cap = max(int(res.arg), 2 * cap)
            continue
        if res.status == 18 and res.arg > ocap:
            ocap = max(int(res.arg), 2 * ocap)
            continue
        no = res.n_values + 1 if res.status == 0 else res.n_values
        return _mk(res, dst[: res.n_bytes], off[:no])
    raise RuntimeError("oracle retry loop did not converge")


def dlba_decode(src, cap: int = 0, ocap: int = 0) -> Result:
    return _byte_array(lib().pqo_dlba_decode, src, cap, ocap)


def dba_decode(src, cap: int = 0, ocap: int = 0) -> Result:
    return _byte_array(lib().pqo_dba_decode, src, cap, ocap)


def plain_decode_byte_array(src, cap: int = 0, ocap: int = 0) -> Result:
    return _byte_array(lib().pqo_plain_decode_byte_array, src, cap, ocap)


def dba_decode_flba(src, size: int, cap: int = 0) -> Result:
    s = _src(src)
    res, v = _retry(lambda d, c: lib().pqo_dba_decode_flba(_p(d), c, _p(s), s.size, size), np.uint8, cap or max(64, 4 * s.size))
    return _mk(res, v)


def bss_decode(src, size: int, is_flba: bool = False) -> Result:
    s = _src(src)
    res, v = _retry(lambda d, c: lib().pqo_bss_decode(_p(d), c, _p(s), s.size, size, int(is_flba)), np.uint8, s.size + 8)
    return _mk(res, v)


def plain_decode_fixed(src, width: int, is_flba: bool = False) -> Result:
    s = _src(src)
    res, v = _retry(lambda d, c: lib().pqo_plain_decode_fixed(_p(d), c, _p(s), s.size, width, int(is_flba)), np.uint8, s.size + 8)
    return _mk(res, v)


def dlba_estimate(src) -> int:
    s = _src(src)
    return lib().pqo_dlba_estimate(_p(s), s.size)


def dba_estimate(src) -> int:
    s = _src(src)
    return lib().pqo_dba_estimate(_p(s), s.size)


def dict_lookup(dictionary: np.ndarray, indexes: np.ndarray, row_stride: int = 0) -> Result:
    d = np.ascontiguousarray(dictionary)
    width = d.dtype.itemsize
    idx = np.ascontiguousarray(indexes, dtype=np.int32)
    stride = row_stride or width
    rows = np.zeros(max(1, idx.size * stride), dtype=np.uint8)
    res = lib().pqo_dict_lookup(_p(rows), stride, _p(d), d.size, width, _p(idx), idx.size)
    return _mk(res, rows[: idx.size * stride])


def opt_delta_decode_int64(src, cap: int = 0) -> Result:
    """The tuned port (pq_port_opt.c) of delta_decode_int64: same results, for the benchmark's CPU leg."""
    s = _src(src)
    res, v = _retry(lambda d, c: lib().pqopt_delta_decode_int64(_p(d), c, _p(s), s.size), np.int64, cap or 64)
    return _mk(res, v)


def opt_rle_dict_lookup_int64(src, num_values: int, dictionary: np.ndarray) -> Result:
    """The tuned port of rle_dict_decode_int32 + indexed_page_fixup + dict_lookup for an int64 dictionary."""
    s = _src(src)
    d = np.ascontiguousarray(dictionary, dtype=np.int64)
    dst = np.zeros(max(1, num_values), np.int64)
    res = lib().pqopt_rle_dict_lookup_int64(_p(dst), num_values, _p(s), s.size, _p(d), d.size)
    return _mk(res, dst[:num_values] if res.status == 0 else dst[:0])


def indexed_page_fixup(values: np.ndarray, num_values: int) -> np.ndarray:
    """dictionary.go:136-162: truncate to num_values or zero-extend."""
    out = np.zeros(num_values, dtype=np.int32)
    k = min(num_values, values.size)
    out[:k] = values[:k]
    return out


# ---- encoders ---------------------------------------------------------------

def _enc(fn, cap, *args) -> bytes:
    dst = np.zeros(cap, dtype=np.uint8)
    n = fn(_p(dst), cap, *args)
    if n == 2**64 - 1:
        raise ValueError("oracle encoder rejected the input")
    return dst[:n].tobytes()


def rle_encode_levels(values, bit_width: int) -> bytes:
    v = np.ascontiguousarray(values, dtype=np.uint8)
    return _enc(lib().pqo_rle_encode_levels, 2 * v.size + 64, _p(v), v.size, bit_width)


def rle_encode_levels_mr(values, bit_width: int) -> bytes:
    v = np.ascontiguousarray(values, dtype=np.uint8)
    return _enc(lib().pqo_rle_encode_levels_mr, 2 * v.size + 128, _p(v), v.size, bit_width)


def rle_encode_int32(values, bit_width: int) -> bytes:
    v = np.ascontiguousarray(values, dtype=np.int32)
    return _enc(lib().pqo_rle_encode_int32, 5 * v.size + 128, _p(v), v.size, bit_width)


def rle_encode_boolean(packed_bits) -> bytes:
    v = np.ascontiguousarray(packed_bits, dtype=np.uint8)
    return _enc(lib().pqo_rle_encode_boolean, 2 * v.size + 64, _p(v), v.size)


def rle_dict_encode_int32(values) -> bytes:
    v = np.ascontiguousarray(values, dtype=np.int32)
    return _enc(lib().pqo_rle_dict_encode_int32, 5 * v.size + 128, _p(v), v.size)


def bitpacked_encode_levels(values, bit_width: int) -> bytes:
    v = np.ascontiguousarray(values, dtype=np.uint8)
    return _enc(lib().pqo_bitpacked_encode_levels, v.size * 2 + 16, _p(v), v.size, bit_width)


def delta_encode_int32(values, block_size: int = 128, num_miniblocks: int = 4) -> bytes:
    v = np.ascontiguousarray(values, dtype=np.int32)
    cap = 128 + (v.size // block_size + 2) * (16 + num_miniblocks + block_size * 4)
    return _enc(lib().pqo_delta_encode_int32, cap, _p(v), v.size, block_size, num_miniblocks)


def delta_encode_int64(values, block_size: int = 128, num_miniblocks: int = 4) -> bytes:
    v = np.ascontiguousarray(values, dtype=np.int64)
    cap = 128 + (v.size // block_size + 2) * (16 + num_miniblocks + block_size * 8)
    return _enc(lib().pqo_delta_encode_int64, cap, _p(v), v.size, block_size, num_miniblocks)


def _flatten(values):
    offs = np.zeros(len(values) + 1, dtype=np.uint32)
    if len(values):
        offs[1:] = np.cumsum([len(v) for v in values], dtype=np.uint64).astype(np.uint32)
    data = np.frombuffer(b"".join(bytes(v) for v in values), dtype=np.uint8).copy() if len(values) else np.zeros(0, np.uint8)
    if data.size == 0:
        data = np.zeros(1, np.uint8)[:0]
    return data, offs


def dlba_encode(values=None, data=None, offsets=None) -> bytes:
    if values is not None:
        data, offsets = _flatten(values)
    data = np.ascontiguousarray(data, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint32)
    cap = data.size + 8 * offsets.size + 4096
    return _enc(lib().pqo_dlba_encode, cap, _p(data), _p(offsets), offsets.size)


def dba_encode(values=None, data=None, offsets=None) -> bytes:
    if values is not None:
        data, offsets = _flatten(values)
    data = np.ascontiguousarray(data, dtype=np.uint8)
    offsets = np.ascontiguousarray(offsets, dtype=np.uint32)
    cap = data.size + 16 * offsets.size + 8192
    return _enc(lib().pqo_dba_encode, cap, _p(data), _p(offsets), offsets.size)


def plain_encode_byte_array(values) -> bytes:
    data, offsets = _flatten(values)
    cap = data.size + 4 * offsets.size + 16
    return _enc(lib().pqo_plain_encode_byte_array, cap, _p(data), _p(offsets), offsets.size)


def bss_encode(data, size: int) -> bytes:
    d = _src(data) if isinstance(data, (bytes, bytearray, memoryview)) else np.ascontiguousarray(data).view(np.uint8).reshape(-1)
    dst = np.zeros(max(d.size, 1), dtype=np.uint8)
    lib().pqo_bss_encode(_p(dst), _p(d), d.size, size)
    return dst[: d.size].tobytes()
