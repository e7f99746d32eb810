"""GPU parity: DELTA_BINARY_PACKED through the C ABI against the CPU checker.

Mirrors the reference's delta tests (encoding/encoding_test.go tables,
encoding/delta/delta_test.go width x N sweep, the parquet test files) with the
GPU kernel in the place of the Go decoder.  Bit exact.
"""
import numpy as np
import pytest

from oracle import pqoracle as O
from tests import fixtures as F
from tests import refdata as R
from tests.gpu_util import assert_same, decode, pkg, spec

pytestmark = pytest.mark.gpu


def _run(cases, kind, dtype, oracle_fn, values=True):
    """cases: list of (label, encoded bytes, misalign)."""
    P = pkg()
    specs = [spec(P.ENC.DELTA_BINARY_PACKED, kind, enc, misalign=mis) for _, enc, mis in cases]
    outs = decode(specs)
    for (label, enc, _), out in zip(cases, outs):
        want = oracle_fn(enc)
        assert_same(out, want, dtype, label, values=values)
        if want.status in (0, 9):  # the remainder is only meaningful when the stream parsed
            assert out.consumed == want.consumed, f"{label}: consumed {out.consumed} != {want.consumed}"


def _pm(i, bw):
    v = -i if i % 2 else i
    return v & ((1 << bw) - 1)


def test_int64_reference_tables():
    P = pkg()
    cases = [(f"table{k}", O.delta_encode_int64(np.array(v, np.int64)), k % 16) for k, v in enumerate(R.INT64_TESTS)]
    _run(cases, P.KIND.INT64, np.int64, O.delta_decode_int64)


def test_int32_reference_tables():
    P = pkg()
    cases = [(f"table{k}", O.delta_encode_int32(np.array(v, np.int32)), k % 16) for k, v in enumerate(R.INT32_TESTS)]
    _run(cases, P.KIND.INT32, np.int32, O.delta_decode_int32)


def test_int64_width_sweep():
    # delta_test.go:46-57: every bit width 0..64, N over 0..200
    P = pkg()
    cases = []
    for bw in range(0, 65):
        for n in (0, 1, 2, 3, 31, 32, 33, 34, 127, 128, 129, 130, 161, 200):
            v = np.array([_pm(i, bw) for i in range(n)], np.uint64).view(np.int64)
            cases.append((f"bw{bw}-n{n}", O.delta_encode_int64(v), (bw + n) % 16))
    _run(cases, P.KIND.INT64, np.int64, O.delta_decode_int64)


def test_int32_width_sweep():
    P = pkg()
    cases = []
    for bw in range(0, 33):
        for n in (0, 1, 2, 3, 31, 32, 33, 34, 127, 128, 129, 130, 161, 200):
            v = np.array([_pm(i, bw) for i in range(n)], np.uint64).astype(np.uint32).view(np.int32)
            cases.append((f"bw{bw}-n{n}", O.delta_encode_int32(v), (bw * 3 + n) % 16))
    _run(cases, P.KIND.INT32, np.int32, O.delta_decode_int32)


def test_quick_sizes_random():
    # internal/quick sizes, random full-range values (wrapping arithmetic)
    P = pkg()
    rng = np.random.default_rng(0)
    cases = []
    for n in R.QUICK_SIZES:
        v = rng.integers(-2**63, 2**63 - 1, size=n, dtype=np.int64)
        cases.append((f"n{n}", O.delta_encode_int64(v), n % 16))
    _run(cases, P.KIND.INT64, np.int64, O.delta_decode_int64)
    cases = []
    for n in R.QUICK_SIZES:
        v = rng.integers(-2**31, 2**31 - 1, size=n, dtype=np.int32)
        cases.append((f"n{n}", O.delta_encode_int32(v), n % 16))
    _run(cases, P.KIND.INT32, np.int32, O.delta_decode_int32)


def test_parquet_file_pages():
    # pages cut from testdata/delta_binary_packed.parquet & co, expected values from pyarrow
    P = pkg()
    items = list(F.pages(kind=("v1", "v2"), encoding=(5,)))
    assert len(items) > 70
    specs = []
    for k, (rec, a) in enumerate(items):
        kind = P.KIND.INT64 if rec["physical_type"] == "INT64" else P.KIND.INT32
        specs.append(spec(P.ENC.DELTA_BINARY_PACKED, kind, a["values"].tobytes(), misalign=k % 16))
    outs = decode(specs)
    for (rec, a), out in zip(items, outs):
        dt = np.int64 if rec["physical_type"] == "INT64" else np.int32
        assert out.status == 0, F.page_id(rec)
        assert np.array_equal(out.data.view(dt), a["expected_values"]), F.page_id(rec)


def test_large_pages_and_shapes():
    P = pkg()
    rng = np.random.default_rng(1)
    cases = []
    # timestamps like the benchmark column, one full-size page
    ts = np.cumsum(rng.integers(1_000_000, 1_004_096, size=131072, dtype=np.int64)) + 1_600_000_000_000_000_000
    cases.append(("timestamps", O.delta_encode_int64(ts), 3))
    # widths changing from block to block
    parts = [rng.integers(0, 1 << int(b), size=300, dtype=np.int64) for b in rng.integers(1, 62, size=40)]
    cases.append(("mixed-widths", O.delta_encode_int64(np.cumsum(np.concatenate(parts))), 9))
    cases.append(("constant", O.delta_encode_int64(np.full(5000, 7, np.int64)), 0))
    cases.append(("minmax", O.delta_encode_int64(np.array(([-2**63, 2**63 - 1] * 700), np.int64)), 5))
    for block, mbs in [(128, 1), (128, 2), (256, 8), (256, 2), (384, 4), (1024, 4), (2048, 64), (4096, 128), (65536, 1), (65536, 4)]:
        v = rng.integers(-2**40, 2**40, size=3 * block + 17 if block < 65536 else 70000, dtype=np.int64)
        cases.append((f"block{block}-mb{mbs}", O.delta_encode_int64(v, block, mbs), (block + mbs) % 16))
    _run(cases, P.KIND.INT64, np.int64, O.delta_decode_int64)
    cases = []
    for block, mbs in [(128, 4), (256, 8), (384, 4), (1024, 2), (65536, 2)]:
        v = rng.integers(-2**31, 2**31 - 1, size=2 * block + 5 if block < 65536 else 66000, dtype=np.int32)
        cases.append((f"i32-block{block}-mb{mbs}", O.delta_encode_int32(v, block, mbs), mbs))
    _run(cases, P.KIND.INT32, np.int32, O.delta_decode_int32)


def test_block_sizes_that_vary():
    """Pages whose blocks differ in size from one to the next: min-delta varints of 1..10 bytes,
    miniblock widths that change inside and between blocks, single odd blocks in steady pages."""
    P = pkg()
    rng = np.random.default_rng(77)
    cases = []
    n = 40000
    # varint length jitter: the block minimum straddles a 7-bit boundary
    for bw in (7, 12, 14, 21, 48, 62):
        v = np.cumsum(rng.integers(0, 1 << bw, size=n, dtype=np.int64) >> rng.integers(0, 3, size=n))
        cases.append((f"jitter-bw{bw}", O.delta_encode_int64(v), bw % 16))
    # widths drawn per stretch of 32 values, so every miniblock has its own
    widths = rng.integers(0, 40, size=n // 32 + 1)
    d = np.concatenate([rng.integers(0, 1 << int(w), size=32, dtype=np.int64) for w in widths])[:n]
    cases.append(("width-per-miniblock", O.delta_encode_int64(np.cumsum(d)), 3))
    # steady page with a spike every few thousand values
    d = rng.integers(1000, 1016, size=n, dtype=np.int64)
    d[rng.integers(0, n, size=12)] = 1 << 40
    d[rng.integers(0, n, size=12)] = -(1 << 33)
    cases.append(("spikes", O.delta_encode_int64(np.cumsum(d)), 5))
    # sign changes: min deltas from -2**62 to 2**62 (ten-byte varints next to one-byte ones)
    d = rng.integers(-8, 8, size=n, dtype=np.int64)
    d[::517] = rng.integers(-2**62, 2**62, size=d[::517].size, dtype=np.int64)
    with np.errstate(over="ignore"):
        cases.append(("wide-minima", O.delta_encode_int64(np.cumsum(d)), 9))
    _run(cases, P.KIND.INT64, np.int64, O.delta_decode_int64)
    cases32 = []
    for bw in (7, 13, 20, 30):
        v = np.cumsum(rng.integers(0, 1 << bw, size=n, dtype=np.int64) >> rng.integers(0, 3, size=n)).astype(np.int32)
        cases32.append((f"jitter32-bw{bw}", O.delta_encode_int32(v), bw % 16))
    _run(cases32, P.KIND.INT32, np.int32, O.delta_decode_int32)


def test_errors_and_truncation():
    P = pkg()
    v = np.cumsum(np.arange(1000, dtype=np.int64) ** 2)
    full = O.delta_encode_int64(v)
    cases = [("empty", b"", 0)]
    for cut in (1, 2, 3, 4, 5, 6, 7, 20, 100, len(full) // 2, len(full) - 200, len(full) - 3, len(full) - 1):
        cases.append((f"cut{cut}", full[:cut], cut % 16))
    cases += [
        ("zero-miniblocks", bytes([0x80, 0x01, 0x00, 0x01, 0x00]), 0),
        ("block-127", bytes([0x7f, 0x04, 0x01, 0x00]), 1),
        ("block-too-large", bytes([0x80, 0x81, 0x04, 0x04, 0x01, 0x00]), 2),
        ("miniblock-not-32", bytes([0x80, 0x01, 0x03, 0x01, 0x00]), 3),
        ("too-many-values", bytes([0x80, 0x01, 0x04, 0x80, 0x80, 0x80, 0x80, 0x08, 0x00]), 4),
        ("negative-total", bytes([0x80, 0x01, 0x04]) + b"\xff" * 9 + b"\x01\x00", 5),
        ("header-overflow", b"\x80" * 11, 6),
        ("zero-values", bytes([0x80, 0x01, 0x04, 0x00, 0x00]), 7),
        ("one-value", bytes([0x80, 0x01, 0x04, 0x01, 0x05]), 8),
        ("missing", bytes([0x80, 0x01, 0x04, 0x05, 0x02]), 9),
        ("min-delta-overflow", bytes([0x80, 0x01, 0x04, 0x05, 0x02]) + b"\x80" * 11, 10),
        ("short-widths", bytes([0x80, 0x01, 0x04, 0x40, 0x02, 0x02, 0x03]), 11),
        ("more-miniblocks-than-values", bytes([0x80, 0x01, 0x80, 0x02, 0x05, 0x00]) + b"\x00" * 700, 12),
        # ... whose miniblocks hold no values, so that their widths (any byte) unpack nothing: "4 missing values"
        ("empty-miniblocks-any-width", bytes([0x80, 0x01, 0x80, 0x02, 0x05, 0x00]) + bytes(range(256)) * 3, 13),
    ]
    _run(cases, P.KIND.INT64, np.int64, O.delta_decode_int64)
    _run([(label, src, mis) for label, src, mis in cases[-2:]], P.KIND.INT32, np.int32, O.delta_decode_int32)
    # a width above the value width: status only (the reference hands it to bitpack.Unpack)
    bad = bytearray(O.delta_encode_int64(np.arange(300, dtype=np.int64)))
    hdr = 5 if bad[3] < 0x80 else 6
    bad[hdr + 1] = 65
    _run([("width65", bytes(bad), 0)], P.KIND.INT64, np.int64, O.delta_decode_int64, values=False)
    cases = [("i32-first-range", bytes([0x80, 0x01, 0x04, 0x01]) + b"\x80\x80\x80\x80\x20", 0)]
    v32 = np.cumsum(np.arange(500, dtype=np.int32))
    f32 = O.delta_encode_int32(v32)
    for cut in (3, 9, len(f32) // 2, len(f32) - 1):
        cases.append((f"i32-cut{cut}", f32[:cut], cut % 16))
    _run(cases, P.KIND.INT32, np.int32, O.delta_decode_int32)


def test_dst_too_small_reports_needed_size():
    P = pkg()
    v = np.arange(1000, dtype=np.int64)
    out = decode([spec(P.ENC.DELTA_BINARY_PACKED, P.KIND.INT64, O.delta_encode_int64(v), dst_cap=4000)], grow=False)[0]
    assert out.status == P.STATUS.DST_TOO_SMALL and out.arg == 8000
    out = decode([spec(P.ENC.DELTA_BINARY_PACKED, P.KIND.INT64, O.delta_encode_int64(v), dst_cap=8000)], grow=False)[0]
    assert out.status == 0 and np.array_equal(out.data.view(np.int64), v)


def _few(cases, kind, dtype, oracle_fn, **kw):
    # batches of at most 200 pages: the regime in which pages with blocks of 128 / 4 miniblocks and
    # steady block sizes are decoded tile by tile by many workgroups instead of one per page
    for k in range(0, len(cases), 200):
        _run(cases[k: k + 200], kind, dtype, oracle_fn, **kw)


def test_few_pages_steady_widths():
    P = pkg()
    rng = np.random.default_rng(41)
    c64, c32 = [], []
    sizes = (2, 3, 32, 33, 128, 129, 130, 160, 161, 1023, 1024, 1025, 1026, 1153, 2048, 2049, 4097, 131072, 131073, 300001)
    for k, n in enumerate(sizes):
        for bw in (0, 1, 7, 12, 31, 32, 33, 48, 63):
            # every miniblock holds its own extremes, so that all widths (and block sizes) are the same
            base = rng.integers(0, 1 << bw, size=n, dtype=np.uint64) if bw else np.zeros(n, np.uint64)
            if bw:
                base[0::32] = 0
                base[1::32] = (1 << bw) - 1
            d = base.astype(np.int64) + 5
            v = np.cumsum(d.astype(np.uint64), dtype=np.uint64).view(np.int64) + np.int64(-(1 << 40))
            if n > 4097 and bw not in (0, 12, 33):
                continue
            c64.append((f"i64-n{n}-bw{bw}", O.delta_encode_int64(v), (k + bw) % 16))
            if bw <= 31:
                v32 = (np.cumsum(d.astype(np.uint64), dtype=np.uint64) & 0xffffffff).astype(np.uint32).view(np.int32)
                c32.append((f"i32-n{n}-bw{bw}", O.delta_encode_int32(v32), (k + bw + 3) % 16))
    _few(c64, P.KIND.INT64, np.int64, O.delta_decode_int64)
    _few(c32, P.KIND.INT32, np.int32, O.delta_decode_int32)


def test_few_pages_odd_blocks_and_declined_pages():
    P = pkg()
    rng = np.random.default_rng(42)
    cases = []
    n = 50000
    # steady widths with an odd block now and then (a min delta whose varint is a byte longer, a narrower miniblock)
    d = rng.integers(100, 100 + 4096, size=n, dtype=np.int64)
    d[0::32] = 100
    d[1::32] = 100 + 4095
    for at in (128 * 7, 128 * 8, 128 * 50, 128 * 51, 128 * 200, 128 * 389):
        d[at: at + 128] += 1 << 20          # longer min delta
    d[128 * 100 + 32: 128 * 100 + 64] = 100  # a miniblock of width 0
    cases.append(("odd-blocks", O.delta_encode_int64(np.cumsum(d)), 5))
    # widths that change all the time: declined by the walk, decoded by the page-per-workgroup kernel
    w = rng.integers(0, 40, size=n // 32 + 1)
    d2 = np.concatenate([rng.integers(0, 1 << int(b), size=32, dtype=np.uint64) if b else np.zeros(32, np.uint64) for b in w])[:n]
    cases.append(("changing-widths", O.delta_encode_int64(np.cumsum(d2, dtype=np.uint64).view(np.int64)), 9))
    # geometry the tile-parallel path does not take
    cases.append(("blocks-of-256", O.delta_encode_int64(np.cumsum(d), 256, 8), 1))
    # a stream cut short: inside a header, inside a miniblock (zero padded), before the last block
    enc = O.delta_encode_int64(np.cumsum(d))
    for cut in (len(enc) - 1, len(enc) - 100, len(enc) // 2, 40, 7):
        cases.append((f"cut-{cut}", enc[:cut], cut % 16))
    # the total says more values than there are blocks for
    cases.append(("tail-garbage", enc + bytes(50), 3))
    _few(cases, P.KIND.INT64, np.int64, O.delta_decode_int64)
    # a destination that is too small is reported, not written
    out = decode([spec(P.ENC.DELTA_BINARY_PACKED, P.KIND.INT64, enc, dst_cap=8 * 1000)], grow=False)[0]
    assert out.status == P.STATUS.DST_TOO_SMALL and out.arg == 8 * n


def test_few_pages_length_streams_chain():
    # DELTA_LENGTH_BYTE_ARRAY with equal lengths: the length stream takes the tile-parallel path and
    # the bytes start where it ended
    P = pkg()
    rng = np.random.default_rng(43)
    for n in (1, 2, 129, 5000, 40000):
        vals = [bytes(rng.integers(0, 256, size=24, dtype=np.uint8)) for _ in range(n)]
        enc = O.dlba_encode(vals)
        out = decode([spec(P.ENC.DELTA_LENGTH_BYTE_ARRAY, P.KIND.BYTE_ARRAY, enc, misalign=n % 16)])[0]
        want = O.dlba_decode(enc)
        assert out.status == 0 and out.out_values == n and out.consumed == want.consumed
        assert out.data.tobytes() == b"".join(vals)
        assert np.array_equal(out.offsets[: n + 1], want.offsets)


def test_few_pages_min_delta_varint_flips():
    # deltas uniform below 2^w: the block's min delta sits near a varint length boundary, so one block
    # in a few is a byte longer or shorter than its neighbours (and for wide w the walk declines the page)
    P = pkg()
    rng = np.random.default_rng(44)
    c64, c32 = [], []
    for bw in (6, 7, 12, 13, 14, 16, 20, 21, 24, 28, 32, 40, 48, 63):
        for n in (5000, 131072):
            d = rng.integers(0, 1 << bw, size=n, dtype=np.uint64)
            c64.append((f"i64-w{bw}-n{n}", O.delta_encode_int64(np.cumsum(d, dtype=np.uint64).view(np.int64)), (bw + n) % 16))
            if bw <= 28:
                v32 = (np.cumsum(d, dtype=np.uint64) & 0xffffffff).astype(np.uint32).view(np.int32)
                c32.append((f"i32-w{bw}-n{n}", O.delta_encode_int32(v32), (bw + 7) % 16))
    _few(c64, P.KIND.INT64, np.int64, O.delta_decode_int64)
    _few(c32, P.KIND.INT32, np.int32, O.delta_decode_int32)
