"""Input tables and known-answer vectors transcribed from the reference's tests.

These are data, not code: the value tables of encoding/encoding_test.go:31-193
and the byte-exact vectors of encoding/encoding_test.go:218-270,
encoding/bitpacked/bitpacked_test.go:18-56 and the fuzz seed corpora under
encoding/{delta,rle}/testdata/fuzz.
"""
import numpy as np

MAXI32 = 2**31 - 1
MAXI64 = 2**63 - 1
MINI64 = -2**63
MAXF32 = float(np.finfo(np.float32).max)
MAXF64 = float(np.finfo(np.float64).max)

# encoding/encoding_test.go:31-57
BOOLEAN_TESTS = [
    [],
    [True],
    [False],
    [True, False, True, False, True, True, True, False, False, True],
    [True] * 32,
    [True] * 33,
    [False, True] * 7 + [False],
    [False, True] * 8,
]

_NEVER = list(range(0x20))
_STREAKS = [0] * 4 + [1] * 4 + [2] * 4 + [3] * 4 + [4] * 4 + [5] * 4 + [6] * 3 + [7] * 3 + [8] * 3 + [9] * 3

# encoding/encoding_test.go:59-82
LEVELS_TESTS = [
    [],
    [0],
    [1],
    [0, 1, 0, 2, 3, 4, 5, 6, 127, 127, 0],
    [42] * 24,
    _NEVER,
    _STREAKS,
]

# encoding/encoding_test.go:84-110
INT32_TESTS = [
    [],
    [0],
    [1],
    [-1, 0, 1, 0, 2, 3, 4, 5, 6, MAXI32, MAXI32, 0],
    [42] * 24,
    _NEVER,
    _STREAKS,
    [24, 36, 47, 32, 29, 4, 9, 20, 2, 18],
]

# encoding/encoding_test.go:112-152
INT64_TESTS = [
    [],
    [0],
    [1],
    [-1, 0, 1, 0, 2, 3, 4, 5, 6, MAXI64, MAXI64, 0],
    [42] * 24,
    _NEVER,
    _STREAKS,
    _STREAKS,
    ([MINI64, MAXI64] * 4 + [0, MAXI64] + [MINI64, MAXI64] * 3) * 5,
]

# encoding/encoding_test.go:160-173
FLOAT_TESTS = [
    [],
    [0],
    [1],
    [0, 1, 0, 1, 0, 2, 3, 4, 5, 6, MAXF32, MAXF32, 0],
    [-1, 0, 1, 0, 2, 3, 4, 5, 6, MAXF32, MAXF32, 0],
]
DOUBLE_TESTS = [
    [],
    [0],
    [1],
    [-1, 0, 1, 0, 2, 3, 4, 5, 6, MAXF64, MAXF64, 0],
]

# encoding/encoding_test.go:175-193
BYTE_ARRAY_TESTS = [
    [],
    [b""],
    [b"A", b"B", b"C"],
    [b"hello world!", b"1234567890" * 100],
]
FLBA_TESTS = [
    (1, b""),
    (1, b"ABCDEFGH"),
    (2, b"ABCDEFGH"),
    (4, b"ABCDEFGH"),
    (8, b"ABCDEFGH"),
    (10, b"123456789" * 100),
    (16, b"1234567890" * 160),
]

# encoding/encoding_test.go:218-270 (TestRLEDecodeToleratesEmptyRuns)
RLE_EMPTY_RUN_KATS = [
    ("levels", 1, bytes([0x00, 0x08, 0x01]), np.array([1, 1, 1, 1], np.uint8)),
    ("int32", 16, bytes([0x00, 0x08, 0x07, 0x00]), np.array([7, 7, 7, 7], np.int32)),
    ("boolean", 1, bytes([0x03, 0, 0, 0, 0x00, 0x10, 0xFF]), np.array([0xFF], np.uint8)),
]

# encoding/bitpacked/bitpacked_test.go:18-56 (TestEncodeLevelsKnownBytes)
BITPACKED_KATS = [
    (3, [0, 1, 2, 3, 4, 5, 6, 7], bytes([0x05, 0x39, 0x77])),
    (1, [0, 1, 1, 0, 1, 0, 1, 1], bytes([0x6B])),
    (2, [0, 1, 2, 3, 0, 1, 2, 3], bytes([0x1B, 0x1B])),
]

# Fuzz seed corpora (go test fuzz v1 files): (input bytes, seed).
FUZZ_DELTA_BYTE_ARRAY = [
    (b"\x80\xf8\xa9\xaf\x14\xfc\r\rR1000", 13),
    (b"\xa1\xa1\xa1\xa1\xa1\xa1\xa1\xa1\xa100", -180),
    (b"\x8000000", -97),
    (b"\x800000" + b"\xc9" * 9 + b"0" * 746, -79),
]
FUZZ_RLE_BOOLEAN = [b"0\x00\x00", b"00000001"]
FUZZ_RLE_INT32 = [b"0000"]
FUZZ_RLE_LEVELS = [b"\xba\xba\xba\xba0\xba\xba\xba\xba\xba\xba"]

# internal/quick/quick.go:12-26 slice sizes used by the reference's property tests
QUICK_SIZES = [0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15, 16, 17, 18, 19, 20,
               21, 22, 23, 24, 25, 26, 27, 28, 29, 30, 31, 32, 33, 34, 35, 36, 37, 38, 39,
               99, 100, 101, 127, 128, 129, 255, 256, 257, 1000, 1023, 1024, 1025,
               2000, 2095, 2048, 2049, 4000, 4095, 4096, 4097]


def max_len(values, bits):
    """bits.Len of the widest value reinterpreted as unsigned (encoding_test.go maxLenInt*)."""
    m = 0
    for v in values:
        m = max(m, (int(v) & ((1 << bits) - 1)).bit_length())
    return m
