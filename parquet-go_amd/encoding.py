"""Host-side mirror of parquet-go's `encoding.Encoding` objects over the C ABI.

One object per codec, with the reference's method names and argument meaning
(encoding/encoding.go:20-72): `DecodeInt64(dst, src)` decodes the page payload
`src` and returns the values; like the Go methods it writes from index 0 of
`dst` and hands back a grown array when `dst` is too small (the wrappers answer
PQD_ERR_DST_TOO_SMALL with the size to retry with -- Go's "may grow dst",
encoding/rle/rle.go:510-522).  Methods a codec does not implement raise
`ErrNotSupported`, as `encoding.NotSupported` does (encoding/notsupported.go:116-205).

Every decode runs on the GPU through libpqd.so; there is no host fallback.
Errors follow the reference's taxonomy so that callers can test them the way Go
callers use `errors.Is`:

    EncodingError                  "<ENCODING>: <detail>"       encoding.Error, notsupported.go:31-39
      ErrNotSupported              encoding.ErrNotSupported     notsupported.go:11-19
      ErrInvalidArgument           encoding.ErrInvalidArgument  notsupported.go:20-27
      ErrUnexpectedEOF             io.ErrUnexpectedEOF          rle.go:255,274,303; binary_packed.go:403,471
"""
import ctypes as C
from typing import List, Optional, Tuple

import numpy as np

from ._lib import ENC, KIND, STATUS, Context, lib, status_name


class EncodingError(Exception):
    def __init__(self, encoding: str, status: int, arg: int, detail: str):
        super().__init__(f"{encoding}: {detail}")
        self.encoding, self.status, self.arg = encoding, status, arg


class ErrNotSupported(EncodingError):
    pass


class ErrInvalidArgument(EncodingError):
    pass


class ErrUnexpectedEOF(EncodingError):
    pass


class ErrIndexOutOfBounds(EncodingError, IndexError):
    """The reference panics with an index-out-of-bounds error here (errors.go:75-83)."""


_EOF_WHERE = {1: "decoding run-length block header", 2: "decoding bit-packed block", 3: "decoding run-length block",
              4: "decoding the length prefix", 5: "decoding the page header", 6: "decoding min delta",
              7: "input too short"}


def _error(name: str, status: int, arg: int) -> EncodingError:
    S = STATUS
    if status == S.NOT_SUPPORTED:
        return ErrNotSupported(name, status, arg, "encoding not supported")
    if status == S.INVALID_ARGUMENT:
        return ErrInvalidArgument(name, status, arg, f"cannot decode from input of size {arg}: invalid argument")
    if status == S.INVALID_BIT_WIDTH:
        return ErrInvalidArgument(name, status, arg, f"cannot decode values with a bit width of {arg}: invalid argument")
    if status == S.UNEXPECTED_EOF:
        return ErrUnexpectedEOF(name, status, arg, f"{_EOF_WHERE.get(arg, 'decoding')}: unexpected EOF")
    if status == S.MISSING_VALUES:
        return ErrUnexpectedEOF(name, status, arg, f"{arg} missing values: unexpected EOF")
    if status == S.INDEX_OUT_OF_BOUNDS:
        return ErrIndexOutOfBounds(name, status, arg, f"index out of bounds at position {arg}")
    return EncodingError(name, status, arg, f"{status_name(status)} ({arg})")


def _bytes(src) -> np.ndarray:
    if isinstance(src, np.ndarray):
        return np.ascontiguousarray(src).view(np.uint8).reshape(-1)
    return np.frombuffer(bytes(src), dtype=np.uint8)


def _ptr(a: Optional[np.ndarray]):
    return None if a is None or a.size == 0 else a.ctypes.data_as(C.c_void_p)


_default_ctx = {}


def default_context(device: int = 0) -> Context:
    if device not in _default_ctx:
        _default_ctx[device] = Context(device)
    return _default_ctx[device]


class Encoding:
    """Base: every Decode method answers ErrNotSupported (encoding.NotSupported)."""

    NAME = "NOT_SUPPORTED"
    CODE = -1

    def __init__(self, BitWidth: int = 0, ctx: Optional[Context] = None):
        self.BitWidth = BitWidth
        self._ctx = ctx

    # -- identity (encoding/encoding.go:22-27) --
    def String(self) -> str:
        return self.NAME

    def Encoding(self) -> int:
        return self.CODE

    def __repr__(self):
        return self.NAME

    # -- plumbing --
    def ctx(self) -> Context:
        return self._ctx or default_context()

    def _unsupported(self, what):
        raise ErrNotSupported(self.NAME, STATUS.NOT_SUPPORTED, 0, f"{what}: encoding not supported")

    def _fixed(self, fn_name: str, dtype, dst, src, lead_args: Tuple = ()):
        """Calls one of the fixed-width wrappers, growing dst as Go would."""
        L = lib()
        fn = getattr(L, fn_name)
        s = _bytes(src)
        dt = np.dtype(dtype)
        out = dst if isinstance(dst, np.ndarray) and dst.dtype == dt and dst.flags.c_contiguous else np.empty(max(64, s.size), dt)
        nbytes, arg = C.c_uint64(0), C.c_int64(0)
        for _ in range(3):
            rc = fn(self.ctx().handle, self.CODE, *lead_args, _ptr(s), s.size, _ptr(out), out.nbytes, C.byref(nbytes), C.byref(arg))
            if rc == STATUS.DST_TOO_SMALL and arg.value > out.nbytes:
                out = np.empty((arg.value + dt.itemsize - 1) // dt.itemsize, dt)
                continue
            break
        values = out.reshape(-1)[: nbytes.value // dt.itemsize]
        if rc != STATUS.OK:
            err = _error(self.NAME, rc, arg.value)
            err.values = values  # the part decoded before the failure, as the Go methods return it
            raise err
        return values

    # -- the Decode methods (encoding/encoding.go:48-56), all unsupported here --
    def DecodeLevels(self, dst, src):
        self._unsupported("LEVELS")

    def DecodeBoolean(self, dst, src):
        self._unsupported("BOOLEAN")

    def DecodeInt32(self, dst, src):
        self._unsupported("INT32")

    def DecodeInt64(self, dst, src):
        self._unsupported("INT64")

    def DecodeInt96(self, dst, src):
        self._unsupported("INT96")

    def DecodeFloat(self, dst, src):
        self._unsupported("FLOAT")

    def DecodeDouble(self, dst, src):
        self._unsupported("DOUBLE")

    def DecodeByteArray(self, dst, src, offsets):
        self._unsupported("BYTE_ARRAY")

    def DecodeFixedLenByteArray(self, dst, src, size: int):
        self._unsupported("FIXED_LEN_BYTE_ARRAY")

    def EstimateDecodeByteArraySize(self, src) -> int:
        return 0  # encoding/notsupported.go:199-201

    def CanDecodeInPlace(self) -> bool:
        return bool(lib().pqd_can_decode_in_place(self.CODE))

    # -- shared implementations the codecs below pick from --
    def _levels(self, dst, src):
        return self._fixed("pqd_decode_levels", np.uint8, dst, src, (int(self.BitWidth),))

    def _boolean(self, dst, src):
        return self._fixed("pqd_decode_boolean", np.uint8, dst, src)

    def _int32(self, dst, src):
        return self._fixed("pqd_decode_int32", np.int32, dst, src, (int(self.BitWidth),))

    def _int64(self, dst, src):
        return self._fixed("pqd_decode_int64", np.int64, dst, src)

    def _int96(self, dst, src):
        return self._fixed("pqd_decode_int96", np.uint8, dst, src)

    def _float(self, dst, src):
        return self._fixed("pqd_decode_float", np.float32, dst, src)

    def _double(self, dst, src):
        return self._fixed("pqd_decode_double", np.float64, dst, src)

    def _flba(self, dst, src, size):
        return self._fixed("pqd_decode_fixed_len_byte_array", np.uint8, dst, src, (int(size),))

    def _byte_array(self, dst, src, offsets):
        L = lib()
        s = _bytes(src)
        out = dst if isinstance(dst, np.ndarray) and dst.dtype == np.uint8 and dst.size else np.empty(max(64, 4 * s.size), np.uint8)
        off = offsets if isinstance(offsets, np.ndarray) and offsets.dtype == np.uint32 and offsets.size else np.empty(max(64, s.size + 8), np.uint32)
        nbytes, noff, arg = C.c_uint64(0), C.c_uint64(0), C.c_int64(0)
        for _ in range(4):
            rc = L.pqd_decode_byte_array(self.ctx().handle, self.CODE, _ptr(s), s.size, _ptr(out), out.size, _ptr(off), off.size,
                                         C.byref(nbytes), C.byref(noff), C.byref(arg))
            if rc == STATUS.DST_TOO_SMALL and arg.value > out.size:
                out = np.empty(arg.value, np.uint8)
                continue
            if rc == STATUS.OFFSETS_TOO_SMALL and arg.value > off.size:
                off = np.empty(arg.value, np.uint32)
                continue
            break
        values, offs = out[: nbytes.value], off[: noff.value]
        if rc != STATUS.OK:
            err = _error(self.NAME, rc, arg.value)
            err.values, err.offsets = values, offs
            raise err
        return values, offs

    # -- the Encode methods (encoding/encoding.go:29-41): dst is replaced, not appended to, as in the reference --
    def _encode(self, kind: int, dtype, src, size: int = 0, width_arg: int = 0) -> bytes:
        """One page through pqd_encode_values; the output buffer grows when the library asks (Go's append).
        `size` is a FIXED_LEN_BYTE_ARRAY's type length, `width_arg` the BitWidth of an RLE codec."""
        L = lib()
        v = np.ascontiguousarray(src, dtype=dtype).reshape(-1)
        n = v.size // size if size else v.size
        if size and v.size % size:
            raise ErrInvalidArgument(self.NAME, STATUS.INVALID_ARGUMENT, v.size, f"cannot encode FIXED_LEN_BYTE_ARRAY from input of size {v.size}: invalid argument")
        size = size or width_arg
        cap = max(64, int(L.pqd_encode_bound(self.CODE, kind, size, n)))
        nbytes, arg = C.c_uint64(0), C.c_int64(0)
        for _ in range(3):
            out = np.empty(cap, np.uint8)
            rc = L.pqd_encode_values(self.ctx().handle, self.CODE, kind, size, _ptr(v), n, _ptr(out), out.size, C.byref(nbytes), C.byref(arg))
            if rc == STATUS.DST_TOO_SMALL and arg.value > cap:
                cap = arg.value
                continue
            break
        if rc != STATUS.OK:
            raise _error(self.NAME, rc, arg.value)
        return out[: nbytes.value].tobytes()

    def EncodeInt32(self, dst, src):
        self._unsupported("INT32")

    def EncodeInt64(self, dst, src):
        self._unsupported("INT64")

    def EncodeFloat(self, dst, src):
        self._unsupported("FLOAT")

    def EncodeDouble(self, dst, src):
        self._unsupported("DOUBLE")

    def EncodeFixedLenByteArray(self, dst, src, size: int):
        self._unsupported("FIXED_LEN_BYTE_ARRAY")

    def _enc_int32(self, dst, src):
        return self._encode(KIND.INT32, np.int32, src)

    def _enc_int64(self, dst, src):
        return self._encode(KIND.INT64, np.int64, src)

    def _enc_float(self, dst, src):
        return self._encode(KIND.FLOAT, np.float32, src)

    def _enc_double(self, dst, src):
        return self._encode(KIND.DOUBLE, np.float64, src)

    def _enc_flba(self, dst, src, size):
        return self._encode(KIND.FIXED_LEN_BYTE_ARRAY, np.uint8, src, int(size))

    def EncodeLevels(self, dst, src):
        self._unsupported("LEVELS")

    def EncodeBoolean(self, dst, src):
        self._unsupported("BOOLEAN")

    def EncodeByteArray(self, dst, src, offsets):
        self._unsupported("BYTE_ARRAY")

    def _enc_levels(self, dst, src):
        return self._encode(KIND.LEVELS, np.uint8, src, width_arg=int(self.BitWidth))

    def _enc_boolean(self, dst, src):
        return self._encode(KIND.BOOLEAN, np.uint8, src)

    def _enc_int32_width(self, dst, src):
        return self._encode(KIND.INT32, np.int32, src, width_arg=int(self.BitWidth))

    def _enc_byte_array(self, dst, src, offsets, kind: int = KIND.BYTE_ARRAY, size: int = 0) -> bytes:
        """EncodeByteArray(dst, src, offsets): value k is src[offsets[k]:offsets[k + 1]] (encoding/values.go:47-52)."""
        L = lib()
        v = _bytes(src)
        off = None if offsets is None else np.ascontiguousarray(offsets, dtype=np.uint32).reshape(-1)
        if off is not None and off.size == 0:
            off = None  # len(offsets) == 0: no values at all
        n = (off.size - 1) if off is not None else (v.size // size if size else 0)
        cap = max(64, int(L.pqd_encode_bound_byte_array(self.CODE, n, v.size)))
        nbytes, arg = C.c_uint64(0), C.c_int64(0)
        for _ in range(3):
            out = np.empty(cap, np.uint8)
            rc = L.pqd_encode_byte_array(self.ctx().handle, self.CODE, kind, size, _ptr(v), v.size, None if off is None else _ptr(off), n, _ptr(out),
                                         out.size, C.byref(nbytes), C.byref(arg))
            if rc == STATUS.DST_TOO_SMALL and arg.value > cap:
                cap = arg.value
                continue
            break
        if rc != STATUS.OK:
            raise _error(self.NAME, rc, arg.value)
        return out[: nbytes.value].tobytes()

    def _enc_flba_prefix(self, dst, src, size):
        size = int(size)
        v = _bytes(src)
        if size < 0 or size > 32767:  # encoding.MaxFixedLenByteArraySize (byte_array.go:81-83)
            raise ErrInvalidArgument(self.NAME, STATUS.INVALID_ARGUMENT, size, "invalid argument")
        if size == 0 or v.size % size:
            raise ErrInvalidArgument(self.NAME, STATUS.INVALID_ARGUMENT, v.size, f"cannot encode FIXED_LEN_BYTE_ARRAY from input of size {v.size}: invalid argument")
        return self._enc_byte_array(dst, v, None, KIND.FIXED_LEN_BYTE_ARRAY, size)

    def _estimate(self, src) -> int:
        s = _bytes(src)
        return int(lib().pqd_estimate_decode_byte_array_size(self.ctx().handle, self.CODE, _ptr(s), s.size))


class Plain(Encoding):
    """plain.Encoding (encoding/plain/plain.go:24-112)."""
    NAME, CODE = "PLAIN", ENC.PLAIN
    DecodeBoolean = Encoding._boolean
    DecodeInt32 = Encoding._int32
    DecodeInt64 = Encoding._int64
    DecodeInt96 = Encoding._int96
    DecodeFloat = Encoding._float
    DecodeDouble = Encoding._double
    DecodeByteArray = Encoding._byte_array
    DecodeFixedLenByteArray = Encoding._flba
    EstimateDecodeByteArraySize = Encoding._estimate
    EncodeByteArray = Encoding._enc_byte_array


class RLE(Encoding):
    """rle.Encoding{BitWidth} (encoding/rle/rle.go:30-94)."""
    NAME, CODE = "RLE", ENC.RLE
    DecodeLevels = Encoding._levels
    DecodeBoolean = Encoding._boolean
    DecodeInt32 = Encoding._int32
    EncodeLevels = Encoding._enc_levels
    EncodeBoolean = Encoding._enc_boolean
    EncodeInt32 = Encoding._enc_int32_width


class RLEDictionary(Encoding):
    """rle.DictionaryEncoding (encoding/rle/dictionary.go:11-59): the bit width is the page's first byte."""
    NAME, CODE = "RLE_DICTIONARY", ENC.RLE_DICTIONARY
    DecodeInt32 = Encoding._int32
    EncodeInt32 = Encoding._enc_int32


class PlainDictionary(Encoding):
    """plain.DictionaryEncoding (encoding/plain/dictionary.go:9-27): DecodeInt32 is a PLAIN decode of raw
    int32 values.  (A data page marked PLAIN_DICTIONARY is decoded with the RLE_DICTIONARY codec: that
    rewrite is the caller's, column.go:850-857, and pqd_chunk_plan_pages makes it.)"""
    NAME, CODE = "PLAIN_DICTIONARY", ENC.PLAIN_DICTIONARY
    DecodeInt32 = Encoding._int32


class BitPacked(Encoding):
    """bitpacked.Encoding{BitWidth} (encoding/bitpacked/bitpacked.go:8-33)."""
    NAME, CODE = "BIT_PACKED", ENC.BIT_PACKED
    DecodeLevels = Encoding._levels


class DeltaBinaryPacked(Encoding):
    """delta.BinaryPackedEncoding (encoding/delta/binary_packed.go:16-46)."""
    NAME, CODE = "DELTA_BINARY_PACKED", ENC.DELTA_BINARY_PACKED
    DecodeInt32 = Encoding._int32
    DecodeInt64 = Encoding._int64
    EncodeInt32 = Encoding._enc_int32
    EncodeInt64 = Encoding._enc_int64


class DeltaLengthByteArray(Encoding):
    """delta.LengthByteArrayEncoding (encoding/delta/length_byte_array.go:8-74)."""
    NAME, CODE = "DELTA_LENGTH_BYTE_ARRAY", ENC.DELTA_LENGTH_BYTE_ARRAY
    DecodeByteArray = Encoding._byte_array
    EstimateDecodeByteArraySize = Encoding._estimate
    EncodeByteArray = Encoding._enc_byte_array


class DeltaByteArray(Encoding):
    """delta.ByteArrayEncoding (encoding/delta/byte_array.go:14-183)."""
    NAME, CODE = "DELTA_BYTE_ARRAY", ENC.DELTA_BYTE_ARRAY
    DecodeByteArray = Encoding._byte_array
    DecodeFixedLenByteArray = Encoding._flba
    EstimateDecodeByteArraySize = Encoding._estimate
    EncodeByteArray = Encoding._enc_byte_array
    EncodeFixedLenByteArray = Encoding._enc_flba_prefix


class ByteStreamSplit(Encoding):
    """bytestreamsplit.Encoding (encoding/bytestreamsplit/bytestreamsplit.go:11-105)."""
    NAME, CODE = "BYTE_STREAM_SPLIT", ENC.BYTE_STREAM_SPLIT
    DecodeInt32 = Encoding._int32
    DecodeInt64 = Encoding._int64
    DecodeFloat = Encoding._float
    DecodeDouble = Encoding._double
    DecodeFixedLenByteArray = Encoding._flba
    EncodeInt32 = Encoding._enc_int32
    EncodeInt64 = Encoding._enc_int64
    EncodeFloat = Encoding._enc_float
    EncodeDouble = Encoding._enc_double
    EncodeFixedLenByteArray = Encoding._enc_flba


def LookupEncoding(code: int) -> Encoding:
    """parquet.LookupEncoding (encoding.go:88-95) for the codes of the `encodings` table (encoding.go:50-60)."""
    table = {c.CODE: c for c in (Plain, PlainDictionary, RLE, BitPacked, DeltaBinaryPacked, DeltaLengthByteArray,
                                 DeltaByteArray, RLEDictionary, ByteStreamSplit)}
    return table[code]() if code in table else Encoding()


def byte_array_values(values: np.ndarray, offsets: np.ndarray) -> List[bytes]:
    """Splits a DecodeByteArray result into its values (encoding/values.go:236-276 keeps them joined)."""
    b = values.tobytes()
    return [b[int(offsets[i]): int(offsets[i + 1])] for i in range(len(offsets) - 1)]


class Dictionary:
    """The Lookup surface of parquet.Dictionary for fixed-width types (dictionary.go:64-69;
    int32Dictionary / int64Dictionary / floatDictionary / doubleDictionary / be128Dictionary twins)."""

    def __init__(self, values: np.ndarray, ctx: Optional[Context] = None):
        v = np.ascontiguousarray(values)
        if v.dtype.itemsize not in (4, 8, 12):
            raise ErrInvalidArgument("DICTIONARY", STATUS.INVALID_ARGUMENT, v.dtype.itemsize, "values must be 4, 8 or 12 bytes wide")
        self._ctx = ctx or default_context()
        self._values = v
        self._h = C.c_void_p()
        rc = lib().pqd_dict_create(self._ctx.handle, _ptr(v), v.size, v.dtype.itemsize, C.byref(self._h))
        if rc != STATUS.OK:
            raise _error("DICTIONARY", rc, 0)

    def Len(self) -> int:
        return int(self._values.size)

    def Lookup(self, indexes, rows: Optional[np.ndarray] = None, stride: int = 0) -> np.ndarray:
        """rows[i] = dict[indexes[i]]; `stride` bytes between rows (a `sparse.Array` in the reference)."""
        idx = np.ascontiguousarray(indexes, dtype=np.int32)
        width = self._values.dtype.itemsize
        stride = stride or width
        if rows is None:
            rows = np.zeros(idx.size * stride, np.uint8)
        arg = C.c_int64(0)
        rc = lib().pqd_dict_lookup(self._h, _ptr(idx), idx.size, _ptr(rows), stride, C.byref(arg))
        if rc != STATUS.OK:
            raise _error("DICTIONARY", rc, arg.value)
        if stride == width:
            return rows.view(self._values.dtype) if rows.dtype == np.uint8 else rows
        return rows

    def close(self):
        if self._h:
            lib().pqd_dict_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
