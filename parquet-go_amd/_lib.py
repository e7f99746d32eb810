"""ctypes binding of libpqd.so (the C ABI declared in include/pqd.h).

There is no CPU fallback: if the library is missing, or no GPU is usable, the
calls fail loudly.
"""
from __future__ import annotations

import ctypes as C
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.environ.get("PQD_LIBRARY") or os.path.join(_HERE, "libpqd.so")


class PqdError(RuntimeError):
    def __init__(self, status, arg=0, what=""):
        self.status, self.arg = status, arg
        super().__init__(f"{what}: {status_name(status)} (arg={arg})")


class STATUS:
    OK = 0
    NOT_SUPPORTED = 1
    INVALID_ARGUMENT = 2
    INVALID_BIT_WIDTH = 3
    UNEXPECTED_EOF = 4
    VARINT_OVERFLOW = 5
    RUN_TOO_LARGE = 6
    BAD_DELTA_HEADER = 7
    FIRST_VALUE_RANGE = 8
    MISSING_VALUES = 9
    NEGATIVE_LENGTH = 10
    LENGTH_OUT_OF_BOUNDS = 11
    PREFIX_SUFFIX_MISMATCH = 12
    NEGATIVE_PREFIX = 13
    PREFIX_OUT_OF_BOUNDS = 14
    INDEX_OUT_OF_BOUNDS = 15
    DST_TOO_SMALL = 16
    DEVICE = 17
    OFFSETS_TOO_SMALL = 18
    CHECKSUM = 19


class ENC:
    PLAIN = 0
    PLAIN_DICTIONARY = 2
    RLE = 3
    BIT_PACKED = 4
    DELTA_BINARY_PACKED = 5
    DELTA_LENGTH_BYTE_ARRAY = 6
    DELTA_BYTE_ARRAY = 7
    RLE_DICTIONARY = 8
    BYTE_STREAM_SPLIT = 9


class KIND:
    LEVELS = 0
    BOOLEAN = 1
    INT32 = 2
    INT64 = 3
    INT96 = 4
    FLOAT = 5
    DOUBLE = 6
    BYTE_ARRAY = 7
    FIXED_LEN_BYTE_ARRAY = 8


class Page(C.Structure):
    """struct pqd_page"""
    _fields_ = [
        ("src", C.c_void_p), ("src_len", C.c_uint64),
        ("dst", C.c_void_p), ("dst_cap", C.c_uint64),
        ("offsets", C.c_void_p), ("offsets_cap", C.c_uint64),
        ("dict", C.c_void_p), ("dict_len", C.c_uint64), ("dict_offsets", C.c_void_p),
        ("num_values", C.c_int64),
        ("encoding", C.c_int32), ("kind", C.c_int32), ("bit_width", C.c_int32), ("dst_stride", C.c_int32),
        ("status", C.c_int32), ("reserved", C.c_int32), ("status_arg", C.c_int64),
        ("out_values", C.c_uint64), ("out_bytes", C.c_uint64), ("consumed", C.c_uint64),
    ]


class EncodePage(C.Structure):
    """struct pqd_encode_page"""
    _fields_ = [
        ("src", C.c_void_p), ("n_values", C.c_uint64), ("dst", C.c_void_p), ("dst_cap", C.c_uint64),
        ("encoding", C.c_int32), ("kind", C.c_int32), ("bit_width", C.c_int32), ("status", C.c_int32),
        ("status_arg", C.c_int64), ("out_bytes", C.c_uint64),
        ("offsets", C.c_void_p), ("src_bytes", C.c_uint64),
    ]


class PageHeader(C.Structure):
    """struct pqd_page_header"""
    _fields_ = [(n, C.c_int32) for n in (
        "type", "uncompressed_page_size", "compressed_page_size", "has_crc", "crc", "num_values", "encoding",
        "definition_level_encoding", "repetition_level_encoding", "num_nulls", "num_rows", "definition_levels_byte_length",
        "repetition_levels_byte_length", "is_compressed", "is_sorted", "reserved")] + [
        ("header_offset", C.c_uint64), ("data_offset", C.c_uint64)]


class ColumnInfo(C.Structure):
    """struct pqd_column_info"""
    _fields_ = [("kind", C.c_int32), ("type_length", C.c_int32), ("max_definition_level", C.c_int32),
                ("max_repetition_level", C.c_int32)]


class PagePart(C.Structure):
    """struct pqd_page_part"""
    _fields_ = [("page", C.c_uint64), ("role", C.c_int32), ("reserved", C.c_int32), ("src_offset", C.c_uint64)]


class FileInfo(C.Structure):
    """struct pqd_file_info"""
    _fields_ = [("version", C.c_int32), ("is_encrypted", C.c_int32), ("num_rows", C.c_int64), ("n_schema", C.c_uint64),
                ("n_columns", C.c_uint64), ("n_row_groups", C.c_uint64), ("n_chunks", C.c_uint64), ("names_bytes", C.c_uint64),
                ("footer_bytes_read", C.c_uint64)]


class FileColumn(C.Structure):
    """struct pqd_file_column"""
    _fields_ = [("kind", C.c_int32), ("type_length", C.c_int32), ("repetition", C.c_int32), ("converted_type", C.c_int32),
                ("max_definition_level", C.c_uint8), ("max_repetition_level", C.c_uint8), ("depth", C.c_uint8),
                ("reserved0", C.c_uint8), ("schema_index", C.c_uint32), ("name_offset", C.c_uint32), ("name_len", C.c_uint32)]


class FileRowGroup(C.Structure):
    """struct pqd_file_row_group"""
    _fields_ = [("num_rows", C.c_int64), ("total_byte_size", C.c_int64), ("file_offset", C.c_int64),
                ("total_compressed_size", C.c_int64)]


class FileChunk(C.Structure):
    """struct pqd_file_chunk"""
    _fields_ = [("num_values", C.c_int64), ("offset", C.c_int64), ("total_compressed_size", C.c_int64),
                ("total_uncompressed_size", C.c_int64), ("data_page_offset", C.c_int64), ("dictionary_page_offset", C.c_int64),
                ("kind", C.c_int32), ("codec", C.c_int32), ("encodings", C.c_uint32), ("has_metadata", C.c_int32),
                ("is_encrypted", C.c_int32), ("reserved", C.c_int32)]


# every symbol include/pqd.h declares: name -> (restype, argtypes)
_vp, _u64, _i64, _int = C.c_void_p, C.c_uint64, C.c_int64, C.c_int
_pu64, _pi64 = C.POINTER(C.c_uint64), C.POINTER(C.c_int64)
SYMBOLS = {
    "pqd_abi_version": (_int, []),
    "pqd_status_name": (C.c_char_p, [_int]),
    "pqd_ctx_create": (_int, [_int, C.POINTER(_vp)]),
    "pqd_ctx_destroy": (None, [_vp]),
    "pqd_batch_create": (_int, [_vp, C.POINTER(Page), _u64, C.POINTER(_vp)]),
    "pqd_batch_launch": (_int, [_vp, _vp]),
    "pqd_batch_finish": (_int, [_vp, _vp, C.POINTER(Page), _u64]),
    "pqd_batch_destroy": (None, [_vp]),
    "pqd_decode_batch": (_int, [_vp, C.POINTER(Page), _u64, _vp]),
    "pqd_decode_levels": (_int, [_vp, _int, _int, _vp, _u64, _vp, _u64, _pu64, _pi64]),
    "pqd_decode_boolean": (_int, [_vp, _int, _vp, _u64, _vp, _u64, _pu64, _pi64]),
    "pqd_decode_int32": (_int, [_vp, _int, _int, _vp, _u64, _vp, _u64, _pu64, _pi64]),
    "pqd_decode_int64": (_int, [_vp, _int, _vp, _u64, _vp, _u64, _pu64, _pi64]),
    "pqd_decode_int96": (_int, [_vp, _int, _vp, _u64, _vp, _u64, _pu64, _pi64]),
    "pqd_decode_float": (_int, [_vp, _int, _vp, _u64, _vp, _u64, _pu64, _pi64]),
    "pqd_decode_double": (_int, [_vp, _int, _vp, _u64, _vp, _u64, _pu64, _pi64]),
    "pqd_decode_byte_array": (_int, [_vp, _int, _vp, _u64, _vp, _u64, _vp, _u64, _pu64, _pu64, _pi64]),
    "pqd_decode_fixed_len_byte_array": (_int, [_vp, _int, _int, _vp, _u64, _vp, _u64, _pu64, _pi64]),
    "pqd_estimate_decode_byte_array_size": (_i64, [_vp, _int, _vp, _u64]),
    "pqd_can_decode_in_place": (_int, [_int]),
    "pqd_dict_create": (_int, [_vp, _vp, _u64, C.c_uint32, C.POINTER(_vp)]),
    "pqd_dict_destroy": (None, [_vp]),
    "pqd_dict_lookup": (_int, [_vp, _vp, _u64, _vp, _u64, _pi64]),
    "pqd_dict_lookup_device": (_int, [_vp, _vp, _u64, C.c_uint32, _vp, _u64, _vp, _u64, _vp, _vp]),
    "pqd_dict_lookup_bytes_device": (_int, [_vp, C.c_int32, _vp, _vp, _u64, C.c_uint32, _vp, _u64, _vp, _u64, _vp, _u64, _vp, _vp]),
    "pqd_encode_bound": (_u64, [_int, _int, _int, _u64]),
    "pqd_host_register": (_int, [_vp, _vp, _u64]),
    "pqd_host_unregister": (_int, [_vp, _vp]),
    "pqd_host_alloc": (_int, [_vp, _u64, C.POINTER(C.c_void_p)]),
    "pqd_host_free": (_int, [_vp, _vp]),
    "pqd_encode_bound_byte_array": (_u64, [_int, _u64, _u64]),
    "pqd_encode_byte_array": (_int, [_vp, _int, _int, _int, _vp, _u64, _vp, _u64, _vp, _u64, _pu64, _pi64]),
    "pqd_encode_workspace_size": (_u64, [C.POINTER(EncodePage), _u64]),
    "pqd_encode_batch": (_int, [_vp, C.POINTER(EncodePage), _u64, _vp, _u64, _vp]),
    "pqd_encode_values": (_int, [_vp, _int, _int, _int, _vp, _u64, _vp, _u64, _pu64, _pi64]),
    "pqd_dict_bounds_device": (_int, [_vp, C.c_int32, C.c_int32, _vp, _u64, _vp, _u64, _vp, _vp, _u64, _vp, _vp]),
    "pqd_page_bounds_device": (_int, [_vp, C.c_int32, C.c_int32, _vp, _u64, _vp, _vp, _u64, _vp, _vp]),
    "pqd_levels_workspace_size": (_u64, [_u64]),
    "pqd_levels_count_equal_device": (_int, [_vp, _vp, _u64, C.c_uint8, _vp, _u64, _vp, _vp]),
    "pqd_optional_expand_device": (_int, [_vp, _vp, _u64, C.c_uint8, _vp, _u64, C.c_uint32, C.c_int32, _vp, _vp, _vp, _u64, _vp, _vp]),
    "pqd_optional_expand_noalloc": (_int, [_vp, _vp, _u64, C.c_uint8, _vp, _u64, C.c_uint32, C.c_int32, _vp, _vp, _vp, _vp]),
    "pqd_repeated_row_starts_device": (_int, [_vp, _vp, _u64, _vp, _u64, _vp, _u64, _vp, _vp]),
    "pqd_chunk_scan_pages": (_int, [_vp, _u64, _i64, C.POINTER(PageHeader), _u64, _pu64, _pi64]),
    "pqd_chunk_plan_pages": (_int, [C.POINTER(PageHeader), _u64, _vp, _u64, _vp, C.POINTER(ColumnInfo), C.POINTER(Page),
                                    C.POINTER(PagePart), _u64, _pu64, _pi64]),
    "pqd_file_footer_range": (_int, [_vp, _u64, _u64, _pu64, _pu64]),
    "pqd_file_parse_footer": (_int, [_vp, _u64, C.POINTER(FileInfo), C.POINTER(FileColumn), _u64, C.POINTER(FileRowGroup), _u64,
                                     C.POINTER(FileChunk), _u64, _vp, _u64, _pi64]),
}

_lib = None


def library_path() -> str:
    return _SO


def lib():
    """The loaded library; raises when the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            raise FileNotFoundError(f"{_SO} is missing: run `python -c 'import __graft_entry__ as g; g.build()'`")
        # torch bundles its own libamdhip64.so.7; whichever copy is loaded first serves
        # the whole process, so torch goes first and libpqd.so binds to the same runtime
        # (two HIP runtimes in one process would not see each other's allocations).
        import torch  # noqa: F401
        L = C.CDLL(_SO)
        for name, (res, args) in SYMBOLS.items():
            f = getattr(L, name)
            f.restype, f.argtypes = res, args
        _lib = L
    return _lib


def status_name(status: int) -> str:
    try:
        return lib().pqd_status_name(status).decode()
    except Exception:
        return str(status)


class Context:
    """pqd_ctx: one per process and device."""

    def __init__(self, device: int = 0):
        h = C.c_void_p()
        rc = lib().pqd_ctx_create(device, C.byref(h))
        if rc != STATUS.OK:
            raise PqdError(rc, 0, "pqd_ctx_create")
        self.handle, self.device = h, device

    def close(self):
        if self.handle:
            lib().pqd_ctx_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
