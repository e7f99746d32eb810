"""Pins the CPU checker (oracle/) against the reference's own known answers.

Mirrors encoding/encoding_test.go (TestEncoding tables, TestRLEDecodeToleratesEmptyRuns),
encoding/bitpacked/bitpacked_test.go (TestEncodeLevelsKnownBytes),
encoding/delta/binary_packed_test.go (TestEncodeMiniBlockInt32/64 bit layout),
encoding/delta/length_byte_array_test.go, encoding/delta/delta_test.go (width x N sweep),
encoding/bytestreamsplit/bytestreamsplit_test.go (stream layout) and
encoding/rle/rle_test.go.  No GPU involved.
"""
import numpy as np
import pytest

from oracle import pqoracle as O
from tests import refdata as R

OK = 0


def test_unpack_word_path_matches_bit_definition():
    assert O.lib().pqo_selftest_unpack() == 0


# ---- byte-exact vectors -------------------------------------------------------

def test_rle_decode_tolerates_empty_runs():
    name, bw, src, want = R.RLE_EMPTY_RUN_KATS[0]
    r = O.rle_decode_levels(src, bw)
    assert r.status == OK and np.array_equal(r.values, want)
    name, bw, src, want = R.RLE_EMPTY_RUN_KATS[1]
    r = O.rle_decode_int32(src, bw)
    assert r.status == OK and np.array_equal(r.values, want)
    name, bw, src, want = R.RLE_EMPTY_RUN_KATS[2]
    r = O.rle_decode_boolean(src)
    assert r.status == OK and np.array_equal(r.values, want)


@pytest.mark.parametrize("bw,values,want", R.BITPACKED_KATS)
def test_bitpacked_known_bytes(bw, values, want):
    assert O.bitpacked_encode_levels(values, bw) == want
    r = O.bitpacked_decode_levels(want, bw)
    assert r.status == OK
    assert list(r.values[: len(values)]) == values


def test_bitpacked_empty_and_zero_width():
    # bitpacked.go:76-78: bitWidth 0 or empty input decode to the single value 0
    for src, bw in ((b"", 3), (b"\x12\x34", 0)):
        r = O.bitpacked_decode_levels(src, bw)
        assert r.status == OK and list(r.values) == [0]


def _lsb_layout(values, bw):
    """The bit-by-bit layout TestEncodeMiniBlockInt32/64 spells out."""
    out = bytearray((len(values) * bw + 7) // 8)
    bit = 0
    for v in values:
        for b in range(bw):
            out[bit // 8] |= ((int(v) >> b) & 1) << (bit % 8)
            bit += 1
    return bytes(out)


def _zigzag_varint(v):
    u = (v << 1) ^ (v >> 63) if v >= 0 else ((-v) << 1) - 1
    out = bytearray()
    while u >= 0x80:
        out.append((u & 0x7F) | 0x80)
        u >>= 7
    out.append(u)
    return bytes(out)


def _miniblock_layout_case(bw, bits):
    """One full block whose stored (delta - minDelta) fields are i & mask.

    A full block of 128 deltas keeps the writer's zero padding out of the block
    minimum (binary_packed.go:139-146).  At the full value width the deltas are
    offset by the most negative value so that the minimum stays in range.
    """
    mask = (1 << bw) - 1
    half = 1 << (bits - 1)
    base = -half if bw == bits else 0
    fields = [i & mask for i in range(128)]
    fields[0] = 0
    fields[1] = mask

    def wrap(v):
        return (v + half) % (1 << bits) - half

    vals = [0]
    for f in fields:
        vals.append(wrap(vals[-1] + wrap(f + base)))
    dtype = np.int32 if bits == 32 else np.int64
    vals = np.array(vals, dtype=dtype)
    enc = (O.delta_encode_int32 if bits == 32 else O.delta_encode_int64)(vals)
    head = bytes([0x80, 0x01, 0x04, 0x81, 0x01, 0x00]) + _zigzag_varint(base)
    assert enc[: len(head)] == head
    assert enc[len(head)] == bw
    data = len(head) + 4
    assert enc[data: data + 4 * bw] == _lsb_layout(fields[:32], bw)
    r = (O.delta_decode_int32 if bits == 32 else O.delta_decode_int64)(enc)
    assert r.status == OK and np.array_equal(r.values, vals)


@pytest.mark.parametrize("bw", range(1, 33))
def test_miniblock_layout_int32(bw):
    # binary_packed_test.go:113-143: fields are laid out LSB first, bit by bit
    _miniblock_layout_case(bw, 32)


@pytest.mark.parametrize("bw", range(1, 65))
def test_miniblock_layout_int64(bw):
    # binary_packed_test.go:293-322
    _miniblock_layout_case(bw, 64)


def test_decode_byte_array_lengths():
    # length_byte_array_test.go:5-34: lengths 0..998 -> exclusive prefix sums
    lengths = np.arange(999, dtype=np.int32)
    payload = np.zeros(int(lengths.sum()), np.uint8)
    offs = np.zeros(1000, np.uint32)
    offs[1:] = np.cumsum(lengths)
    enc = O.dlba_encode(data=payload, offsets=offs)
    r = O.dlba_decode(enc)
    assert r.status == OK and np.array_equal(r.offsets, offs) and r.n_bytes == int(lengths.sum())


@pytest.mark.parametrize("size", [1, 2, 3, 4, 8, 16, 32])
def test_bss_stream_layout(size):
    # bytestreamsplit_test.go:94-132: buf[s*n+i] == src[i*size+s]
    n = 17
    src = np.arange(n * size, dtype=np.uint8)
    enc = np.frombuffer(O.bss_encode(src, size), np.uint8)
    for s in range(size):
        for i in range(n):
            assert enc[s * n + i] == src[i * size + s]
    r = O.bss_decode(enc, size, is_flba=True)
    assert r.status == OK and np.array_equal(r.values, src)


# ---- reference input tables, round trip (TestEncoding) --------------------------

def _pack_bools(bools):
    out = bytearray((len(bools) + 7) // 8)
    for i, b in enumerate(bools):
        if b:
            out[i // 8] |= 1 << (i % 8)
    return bytes(out)


@pytest.mark.parametrize("bools", R.BOOLEAN_TESTS)
def test_rle_boolean_roundtrip(bools):
    bits = _pack_bools(bools)
    r = O.rle_decode_boolean(O.rle_encode_boolean(np.frombuffer(bits, np.uint8)))
    assert r.status == OK and r.values.tobytes() == bits
    r = O.plain_decode_fixed(bits, 1)
    assert r.status == OK and r.values.tobytes() == bits


@pytest.mark.parametrize("levels", R.LEVELS_TESTS)
def test_levels_roundtrip(levels):
    bw = R.max_len(levels, 8)
    v = np.array(levels, np.uint8)
    r = O.rle_decode_levels(O.rle_encode_levels(v, bw), bw)
    assert r.status == OK and np.array_equal(r.values[: v.size], v)
    r = O.rle_decode_levels(O.rle_encode_levels_mr(v, max(bw, 1)), max(bw, 1))
    assert r.status == OK and np.array_equal(r.values[: v.size], v)
    r = O.bitpacked_decode_levels(O.bitpacked_encode_levels(v, bw), bw)
    assert r.status == OK and np.array_equal(r.values[: v.size], v)


@pytest.mark.parametrize("values", R.INT32_TESTS)
def test_int32_roundtrip(values):
    v = np.array(values, np.int32)
    bw = R.max_len(values, 32)
    r = O.rle_decode_int32(O.rle_encode_int32(v, bw), bw)
    assert r.status == OK and np.array_equal(r.values, v)  # exact length, as assertEqualInt32 demands
    r = O.rle_dict_decode_int32(O.rle_dict_encode_int32(v))
    assert r.status == OK and np.array_equal(r.values, v)
    r = O.delta_decode_int32(O.delta_encode_int32(v))
    assert r.status == OK and np.array_equal(r.values, v)
    r = O.bss_decode(O.bss_encode(v, 4), 4)
    assert r.status == OK and np.array_equal(r.values.view(np.int32), v)
    r = O.plain_decode_fixed(v.tobytes(), 4)
    assert r.status == OK and np.array_equal(r.values.view(np.int32), v)


@pytest.mark.parametrize("values", R.INT64_TESTS)
def test_int64_roundtrip(values):
    v = np.array(values, np.int64)
    r = O.delta_decode_int64(O.delta_encode_int64(v))
    assert r.status == OK and np.array_equal(r.values, v)
    r = O.bss_decode(O.bss_encode(v, 8), 8)
    assert r.status == OK and np.array_equal(r.values.view(np.int64), v)
    r = O.plain_decode_fixed(v.tobytes(), 8)
    assert r.status == OK and np.array_equal(r.values.view(np.int64), v)


@pytest.mark.parametrize("values", R.FLOAT_TESTS)
def test_float_roundtrip(values):
    v = np.array(values, np.float32)
    r = O.bss_decode(O.bss_encode(v, 4), 4)
    assert r.status == OK and r.values.tobytes() == v.tobytes()


@pytest.mark.parametrize("values", R.DOUBLE_TESTS)
def test_double_roundtrip(values):
    v = np.array(values, np.float64)
    r = O.bss_decode(O.bss_encode(v, 8), 8)
    assert r.status == OK and r.values.tobytes() == v.tobytes()


@pytest.mark.parametrize("values", R.BYTE_ARRAY_TESTS)
def test_byte_array_roundtrip(values):
    flat = b"".join(values)
    offs = np.cumsum([0] + [len(v) for v in values]).astype(np.uint32)
    for enc, dec, est in ((O.dlba_encode, O.dlba_decode, O.dlba_estimate),
                          (O.dba_encode, O.dba_decode, O.dba_estimate),
                          (O.plain_encode_byte_array, O.plain_decode_byte_array, len)):
        buf = enc(values)
        r = dec(buf)
        assert r.status == OK
        assert r.values.tobytes() == flat
        if len(values):
            assert np.array_equal(r.offsets, offs)
        # encoding_test.go:499-501: the estimate never undershoots
        assert len(flat) <= est(buf)


@pytest.mark.parametrize("size,data", R.FLBA_TESTS)
def test_flba_roundtrip(size, data):
    d = np.frombuffer(data, np.uint8)
    r = O.bss_decode(O.bss_encode(d, size), size, is_flba=True)
    assert r.status == OK and r.values.tobytes() == data
    n = len(data) // size
    vals = [data[i * size:(i + 1) * size] for i in range(n)]
    r = O.dba_decode_flba(O.dba_encode(vals), size)
    assert r.status == OK and r.values.tobytes() == data
    r = O.plain_decode_fixed(data, size, is_flba=True)
    assert r.status == OK and r.values.tobytes() == data


# ---- sweeps ---------------------------------------------------------------------

def _pm(i, bw, bits):
    v = -i if i % 2 else i
    return v & ((1 << bw) - 1)


@pytest.mark.parametrize("bw", range(0, 33))
def test_delta_sweep_int32(bw):
    # encoding/test/test.go:10-24 values, delta_test.go:33-44 N = 0..200
    for n in range(0, 201, 1 if bw in (0, 1, 31, 32) else 7):
        v = np.array([_pm(i, bw, 32) for i in range(n)], np.uint64).astype(np.uint32).view(np.int32)
        r = O.delta_decode_int32(O.delta_encode_int32(v))
        assert r.status == OK and np.array_equal(r.values, v), (bw, n)


@pytest.mark.parametrize("bw", range(0, 65))
def test_delta_sweep_int64(bw):
    for n in range(0, 201, 1 if bw in (0, 1, 63, 64) else 7):
        v = np.array([_pm(i, bw, 64) for i in range(n)], np.uint64).view(np.int64)
        r = O.delta_decode_int64(O.delta_encode_int64(v))
        assert r.status == OK and np.array_equal(r.values, v), (bw, n)


@pytest.mark.parametrize("n", range(0, 101, 3))
def test_bss_sweep(n):
    # bytestreamsplit_test.go:29-43
    f = np.arange(n, dtype=np.float32)
    d = np.arange(n, dtype=np.float64)
    assert O.bss_decode(O.bss_encode(f, 4), 4).values.tobytes() == f.tobytes()
    assert O.bss_decode(O.bss_encode(d, 8), 8).values.tobytes() == d.tobytes()


@pytest.mark.parametrize("block,mbs", [(128, 1), (128, 2), (256, 8), (256, 2), (384, 4), (1024, 4), (65536, 1)])
def test_delta_nonstandard_block_shapes(block, mbs):
    rng = np.random.default_rng(block * 31 + mbs)
    v = rng.integers(-2**40, 2**40, size=3 * block + 17, dtype=np.int64)
    r = O.delta_decode_int64(O.delta_encode_int64(v, block, mbs))
    assert r.status == OK and np.array_equal(r.values, v)


def test_rle_large_run():
    # rle_test.go:73-96 (TestDecodeInt32LargeRun), one run above the old 2^24 limit
    n = 16_777_217
    enc = O.rle_encode_int32(np.ones(n, np.int32), 1)
    assert len(enc) < 16
    r = O.rle_decode_int32(enc, 1, cap=4 * n)
    assert r.status == OK and r.n_values == n and int(r.values.min()) == 1 and int(r.values.max()) == 1


@pytest.mark.parametrize("data", R.FUZZ_RLE_LEVELS + R.FUZZ_RLE_BOOLEAN + R.FUZZ_RLE_INT32)
def test_fuzz_seeds_rle(data):
    # fuzz.go:125-155: encode -> decode equality on the seed inputs
    lv = np.frombuffer(data, np.uint8)
    r = O.rle_decode_levels(O.rle_encode_levels(lv, 8), 8)
    assert r.status == OK and np.array_equal(r.values[: lv.size], lv)
    r = O.rle_decode_boolean(O.rle_encode_boolean(lv))
    assert r.status == OK and np.array_equal(r.values, lv)
    if lv.size % 4 == 0:
        iv = lv.view(np.int32)
        r = O.rle_decode_int32(O.rle_encode_int32(iv, 32), 32)
        assert r.status == OK and np.array_equal(r.values, iv)


@pytest.mark.parametrize("data,seed", R.FUZZ_DELTA_BYTE_ARRAY)
def test_fuzz_seeds_delta_byte_array(data, seed):
    # The harness cuts the input into strings with Go's math/rand; the cut used
    # here is this test's own (numpy), the inputs are the reference's seeds.
    rng = np.random.default_rng(abs(seed))
    limit = len(data) // 10 + 1
    vals, i = [], 0
    while i < len(data):
        n = min(int(rng.integers(1, limit + 1)), len(data) - i)
        vals.append(data[i:i + n])
        i += n
    for enc, dec in ((O.dba_encode, O.dba_decode), (O.dlba_encode, O.dlba_decode)):
        r = dec(enc(vals))
        assert r.status == OK and r.values.tobytes() == data
        assert np.array_equal(np.diff(r.offsets.astype(np.int64)), [len(v) for v in vals])


# ---- error behaviour ---------------------------------------------------------------

def test_error_statuses():
    E = O
    assert E.rle_decode_levels(b"\x02", 9).status == 3  # INVALID_BIT_WIDTH
    assert E.rle_decode_int32(b"\x02", 33).status == 3
    assert E.rle_decode_levels(b"\x80", 1).status == 4  # header cut short
    assert E.rle_decode_levels(b"\x80" * 11, 1).status == 5  # varint overflow
    assert E.rle_decode_levels(b"\x03\xff", 3).status == 4  # bit-packed run needs 3 bytes
    assert E.rle_decode_levels(b"\x08", 1).status == 4  # run value missing
    assert E.rle_decode_levels(b"\x08", 0).status == 0  # width 0 carries no value byte
    assert list(E.rle_decode_levels(b"\x08", 0).values) == [0, 0, 0, 0]
    big = bytes([0x80, 0x80, 0x80, 0x80, 0x20])  # count = 2^32 -> above MaxInt32
    assert E.rle_decode_int32(big + b"\x01", 1).status == 6
    assert E.rle_decode_boolean(b"\x01\x02").status == 4
    assert E.rle_decode_boolean(b"\x09\x00\x00\x00\x00").status == 4
    assert E.rle_decode_boolean(b"\x00\x00\x00\x00").status == 0
    # DELTA header validation, binary_packed.go:437-449
    assert E.delta_decode_int64(b"").status == 4
    assert E.delta_decode_int64(bytes([0x80, 0x01, 0x00, 0x01, 0x00])).status == 7
    assert E.delta_decode_int64(bytes([0x80, 0x01, 0x00, 0x01, 0x00])).arg == 1
    assert E.delta_decode_int64(bytes([0x7f, 0x04, 0x01, 0x00])).arg == 2
    assert E.delta_decode_int64(bytes([0x80, 0x81, 0x04, 0x04, 0x01, 0x00])).arg == 3
    assert E.delta_decode_int64(bytes([0x80, 0x01, 0x03, 0x01, 0x00])).arg == 4
    r = E.delta_decode_int64(bytes([0x80, 0x01, 0x04, 0x05, 0x02]))  # 5 values, only the first present
    assert r.status == 9 and r.arg == 4 and r.n_values == 5 and list(r.values) == [1, 0, 0, 0, 0]
    assert E.delta_decode_int32(bytes([0x80, 0x01, 0x04, 0x01]) + b"\x80\x80\x80\x80\x20").status == 8
    assert E.delta_decode_int64(bytes([0x80, 0x01, 0x04, 0x00, 0x00])).n_values == 0
    # truncated miniblock is zero padded, not an error (binary_packed.go:380-389)
    full = O.delta_encode_int64(np.arange(0, 40, dtype=np.int64) ** 2)
    r = E.delta_decode_int64(full[:-3])
    assert r.status == 0 and r.n_values == 40
    # 256 miniblocks in a block of 128: the miniblocks hold no values (128 / 256 = 0, and 0 % 32 == 0 passes
    # binary_packed.go:443), every block yields nothing, the loop ends with the input: "4 missing values"
    # (binary_packed.go:367-404); widths of miniblocks that unpack nothing are not looked at
    r = E.delta_decode_int64(bytes([0x80, 0x01, 0x80, 0x02, 0x05, 0x00]) + bytes(range(256)) * 3)
    assert r.status == 9 and r.arg == 4
    # BSS / PLAIN size checks
    assert E.bss_decode(b"12345", 4).status == 2
    assert E.bss_decode(b"1234", 0, is_flba=True).status == 2
    assert E.bss_decode(b"1234", 40000, is_flba=True).status == 2
    assert E.plain_decode_fixed(b"12345", 4).status == 2
    assert E.plain_decode_byte_array(b"\x05\x00\x00\x00ab").status == 4
    assert E.plain_decode_byte_array(b"\x01\x00\x00").status == 4
    # dictionary lookup bounds
    d = np.arange(4, dtype=np.int64)
    assert E.dict_lookup(d, np.array([0, 3, 4], np.int32)).status == 15
    assert E.dict_lookup(d, np.array([0, -1], np.int32)).status == 15
    r = E.dict_lookup(d, np.array([3, 0, 2], np.int32), row_stride=24)
    assert r.status == 0 and list(r.values.reshape(3, 24)[:, :8].copy().view(np.int64).ravel()) == [3, 0, 2]


def test_dba_error_order():
    # byte_array_purego.go:13-24: per value n<0, n>len(src), p<0, p>len(last)
    def page(prefix, suffix, data):
        return O.delta_encode_int32(np.array(prefix, np.int32)) + O.delta_encode_int32(np.array(suffix, np.int32)) + data
    assert O.dba_decode(page([0, 1], [2, -1], b"abc")).status == 10
    assert O.dba_decode(page([0, 1], [2, 5], b"abc")).status == 11
    assert O.dba_decode(page([0, -1], [2, 1], b"abc")).status == 13
    assert O.dba_decode(page([0, 3], [2, 1], b"abc")).status == 14
    assert O.dba_decode(page([0, 1, 2], [2, 1], b"abc")).status == 12
    r = O.dba_decode(page([0, 1, 3], [2, 2, 1], b"abcdX"))
    assert r.status == 0 and r.values.tobytes() == b"ab" + b"acd" + b"acdX" and list(r.offsets) == [0, 2, 5, 9]


def test_indexed_page_fixup():
    # dictionary.go:136-162 / dictionary_test.go:346-352 (TestNewIndexedPage)
    v = np.array([5, 6, 7], np.int32)
    assert list(O.indexed_page_fixup(v, 2)) == [5, 6]
    assert list(O.indexed_page_fixup(v, 5)) == [5, 6, 7, 0, 0]
