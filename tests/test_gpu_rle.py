"""GPU parity: RLE / bit-packed hybrid (levels, booleans, int32 indexes) and the fused
RLE_DICTIONARY -> values path, through the C ABI, against the CPU checker.

Mirrors encoding/encoding_test.go (tables, TestRLEDecodeToleratesEmptyRuns),
encoding/rle/rle_test.go and dictionary_test.go (TestDictionary sizes,
TestNewIndexedPage) with the GPU kernels in the place of the Go decoders.
"""
import numpy as np
import pytest

from oracle import pqoracle as O
from tests import fixtures as F
from tests import refdata as R
from tests.gpu_util import assert_same, decode, pkg, spec

pytestmark = pytest.mark.gpu


def _levels_cases(rng):
    cases = []
    for k, lv in enumerate(R.LEVELS_TESTS):
        v = np.array(lv, np.uint8)
        bw = R.max_len(lv, 8)
        cases.append((f"table{k}", bw, O.rle_encode_levels(v, bw)))
        if bw:
            cases.append((f"table{k}-mr", bw, O.rle_encode_levels_mr(v, bw)))
    for bw in range(1, 9):
        for n in (1, 7, 8, 9, 63, 64, 65, 1000, 4097, 20000):
            # mixed run lengths: half short runs, half long ones
            runs = []
            while sum(len(r) for r in runs) < n:
                ln = int(rng.geometric(1 / 4)) if rng.random() < 0.5 else int(rng.geometric(1 / 300))
                runs.append(np.full(ln, rng.integers(0, 1 << bw), np.uint8))
            v = np.concatenate(runs)[:n]
            cases.append((f"bw{bw}-n{n}", bw, O.rle_encode_levels(v, bw)))
            cases.append((f"bw{bw}-n{n}-mr", bw, O.rle_encode_levels_mr(v, bw)))
        v = rng.integers(0, 1 << bw, size=5000, dtype=np.uint8)  # no runs at all: one long bit-packed run
        cases.append((f"bw{bw}-random", bw, O.rle_encode_levels(v, bw)))
    cases.append(("bw0", 0, O.rle_encode_levels(np.zeros(100, np.uint8), 0)))
    return cases


def test_levels():
    P = pkg()
    rng = np.random.default_rng(3)
    cases = _levels_cases(rng)
    outs = decode([spec(P.ENC.RLE, P.KIND.LEVELS, enc, bit_width=bw, misalign=k % 16) for k, (_, bw, enc) in enumerate(cases)])
    for (label, bw, enc), out in zip(cases, outs):
        assert_same(out, O.rle_decode_levels(enc, bw), np.uint8, label)


def test_levels_cut_to_num_values():
    # column.go:940-945: the caller keeps num_values levels
    P = pkg()
    v = np.array([1, 0, 1, 1, 0, 1, 1, 1, 0, 0, 1], np.uint8)
    enc = O.rle_encode_levels_mr(v, 1)
    full = O.rle_decode_levels(enc, 1)
    assert full.n_values == 16  # one padded bit-packed group
    out = decode([spec(P.ENC.RLE, P.KIND.LEVELS, enc, bit_width=1, num_values=11)])[0]
    assert out.status == 0 and out.out_values == 11 and np.array_equal(out.data, v)


def test_empty_runs_kat():
    P = pkg()
    (_, bw0, s0, w0), (_, bw1, s1, w1), (_, _, s2, w2) = R.RLE_EMPTY_RUN_KATS
    outs = decode([spec(P.ENC.RLE, P.KIND.LEVELS, s0, bit_width=bw0),
                   spec(P.ENC.RLE, P.KIND.INT32, s1, bit_width=bw1),
                   spec(P.ENC.RLE, P.KIND.BOOLEAN, s2, bit_width=1)])
    assert outs[0].status == 0 and np.array_equal(outs[0].data, w0)
    assert outs[1].status == 0 and np.array_equal(outs[1].data.view(np.int32), w1)
    assert outs[2].status == 0 and np.array_equal(outs[2].data, w2)


def test_int32():
    P = pkg()
    rng = np.random.default_rng(4)
    cases = []
    for k, vals in enumerate(R.INT32_TESTS):
        v = np.array(vals, np.int32)
        bw = R.max_len(vals, 32)
        cases.append((f"table{k}", bw, O.rle_encode_int32(v, bw)))
    for bw in range(0, 33):
        hi = 1 << bw
        for n in (5, 8, 64, 1000, 2048, 2049, 6000):
            v = rng.integers(0, hi, size=n, dtype=np.int64).astype(np.uint32).view(np.int32)
            if bw and n >= 64:  # splice in runs
                v[10:40] = v[10]
                v[n // 2: n // 2 + 200] = v[n // 2]
            cases.append((f"bw{bw}-n{n}", bw, O.rle_encode_int32(v, bw)))
    outs = decode([spec(P.ENC.RLE, P.KIND.INT32, enc, bit_width=bw, misalign=(k * 5) % 16) for k, (_, bw, enc) in enumerate(cases)])
    for (label, bw, enc), out in zip(cases, outs):
        assert_same(out, O.rle_decode_int32(enc, bw), np.int32, label)
    # the same streams framed as dictionary-index pages (first byte = width)
    dcases = [(label, bytes([bw]) + enc) for label, bw, enc in cases if bw > 0] + [("empty", b"")]
    outs = decode([spec(P.ENC.RLE_DICTIONARY, P.KIND.INT32, enc, misalign=k % 16) for k, (_, enc) in enumerate(dcases)])
    for (label, enc), out in zip(dcases, outs):
        assert_same(out, O.rle_dict_decode_int32(enc), np.int32, "dict-" + label)


def test_boolean():
    P = pkg()
    rng = np.random.default_rng(5)
    cases = []
    for k, bools in enumerate(R.BOOLEAN_TESTS):
        bits = np.packbits(np.array(bools, np.uint8), bitorder="little") if bools else np.zeros(0, np.uint8)
        cases.append((f"table{k}", O.rle_encode_boolean(bits)))
    for n in (1, 7, 100, 5000):
        b = rng.integers(0, 256, size=n, dtype=np.uint8)
        b[n // 3: n // 2] = 0xFF
        b[: n // 5] = 0
        cases.append((f"n{n}", O.rle_encode_boolean(b)))
    for data in R.FUZZ_RLE_BOOLEAN:
        cases.append(("fuzz", O.rle_encode_boolean(np.frombuffer(data, np.uint8))))
    outs = decode([spec(P.ENC.RLE, P.KIND.BOOLEAN, enc, bit_width=1, misalign=k % 16) for k, (_, enc) in enumerate(cases)])
    for (label, enc), out in zip(cases, outs):
        assert_same(out, O.rle_decode_boolean(enc), np.uint8, label)


def test_large_run():
    # rle_test.go:73-96: one run of 16,777,217 values
    P = pkg()
    n = 16_777_217
    enc = O.rle_encode_int32(np.ones(n, np.int32), 1)
    out = decode([spec(P.ENC.RLE, P.KIND.INT32, enc, bit_width=1, dst_cap=4 * n)])[0]
    assert out.status == 0 and out.out_values == n
    v = out.data.view(np.int32)
    assert v.size == n and int(v.min()) == 1 and int(v.max()) == 1


def _runs_of(rng, n, hi, short_mean, long_mean):
    """n values below hi in runs: half the runs of mean short_mean, half of mean long_mean."""
    out = np.empty(n, np.int64)
    i = 0
    while i < n:
        k = 2048
        lens = np.where(rng.random(k) < 0.5, rng.geometric(1 / short_mean, size=k), rng.geometric(1 / long_mean, size=k))
        rep = np.repeat(rng.integers(0, hi, size=k), lens)
        m = min(rep.size, n - i)
        out[i:i + m] = rep[:m]
        i += m
    return out


def test_long_pages():
    """Pages long enough to be indexed chunk by chunk (many workgroups per page): run headers
    crossing chunk boundaries, bit-packed runs longer than a chunk, empty runs, every mode;
    and pages the chunked kernels hand to the one-wavefront kernel (errors, too many values)."""
    P = pkg()
    rng = np.random.default_rng(91)
    L, I, B = P.KIND.LEVELS, P.KIND.INT32, P.KIND.BOOLEAN
    cases = []  # (label, kind, bit width, bytes)
    for bw, short, long_ in ((1, 2, 40), (3, 3, 60), (7, 1.5, 20), (8, 4, 100)):
        v = _runs_of(rng, 600_000, 1 << bw, short, long_).astype(np.uint8)
        cases.append((f"levels-go-bw{bw}", L, bw, O.rle_encode_levels(v, bw)))
        cases.append((f"levels-mr-bw{bw}", L, bw, O.rle_encode_levels_mr(v, bw)))
    for bw, short, long_ in ((1, 3, 20), (5, 2, 40), (13, 9, 200), (20, 5, 60), (32, 12, 12)):
        v = _runs_of(rng, 300_000, 1 << bw, short, long_)
        cases.append((f"int32-bw{bw}", I, bw, O.rle_encode_int32(v.astype(np.uint32).view(np.int32), bw)))
    # empty runs (a zero count and no value byte, rle.go:311-313) scattered through a long stream
    parts = []
    for k in range(12000):
        parts.append(b"\x00" if k % 3 == 0 else (bytes([2 * (1 + k % 50)]) + bytes([k % 4])))
        if k % 500 == 499:
            parts.append(bytes([(8 << 1) | 1]) + bytes(rng.integers(0, 256, size=8 * 2, dtype=np.uint8)))
    cases.append(("empty-runs", L, 2, b"".join(parts)))
    # one bit-packed run longer than several chunks between short runs
    body = bytes(rng.integers(0, 256, size=3 * 5000, dtype=np.uint8))
    varint = lambda x: bytes([(x & 0x7f) | 0x80, x >> 7]) if x >= 0x80 else bytes([x])
    long_bp = b"\x06\x05" * 700 + varint((5000 << 1) | 1) + body + b"\x08\x02" * 3000
    cases.append(("long-bit-packed", L, 3, long_bp))
    b = rng.integers(0, 256, size=60_000, dtype=np.uint8)
    b[10_000:30_000] = 0xFF
    b[40_000:41_000] = 0
    cases.append(("boolean", B, 1, O.rle_encode_boolean(b)))
    # errors deep inside long pages, and more values than a page may hold
    good = O.rle_encode_levels_mr(_runs_of(rng, 400_000, 8, 2, 30).astype(np.uint8), 3)
    cases.append(("truncated", L, 3, good[: len(good) - 1]))
    cases.append(("bad-varint", L, 3, good[:30_000] + b"\x80" * 11 + good[30_000:]))
    cases.append(("bit-packed-overruns", L, 3, good[:25_000] + b"\xff\xff\x03" + good[25_000:40_000]))
    huge = bytes([0xfe, 0xff, 0xff, 0xff, 0x07, 0x01])  # one run of 2**30 - 1 values
    cases.append(("too-many-values", L, 1, b"\x06\x01" * 9000 + huge * 3))
    specs = [spec(P.ENC.RLE, kind, enc, bit_width=bw, misalign=k % 16) for k, (_, kind, bw, enc) in enumerate(cases)]
    outs = decode(specs)
    for (label, kind, bw, enc), out in zip(cases, outs):
        assert len(enc) >= 16384, label
        if label == "too-many-values":
            # the one limit the reference does not have: a page decodes to at most 2**31 - 1 values here
            assert out.status == P.STATUS.RUN_TOO_LARGE, out.status
            continue
        if kind == L:
            want, dt = O.rle_decode_levels(enc, bw), np.uint8
        elif kind == I:
            want, dt = O.rle_decode_int32(enc, bw), np.int32
        else:
            want, dt = O.rle_decode_boolean(enc), np.uint8
        assert_same(out, want, dt, label, values=want.status == 0)


def test_long_dictionary_pages_with_runs():
    """RLE_DICTIONARY pages of a few hundred KB full of runs, gathered through the dictionary."""
    P = pkg()
    rng = np.random.default_rng(92)
    d = rng.integers(-2**62, 2**62, size=5000, dtype=np.int64)
    specs, wants = [], []
    for k in range(5):
        idx = _runs_of(rng, 400_000, d.size, 3, 20 + 15 * k).astype(np.int32)
        enc = O.rle_dict_encode_int32(idx)
        assert len(enc) >= 16384
        specs.append(spec(P.ENC.RLE_DICTIONARY, P.KIND.INT64, enc, dictionary=d, num_values=idx.size, misalign=3 * k))
        wants.append(d[idx])
    for out, want in zip(decode(specs), wants):
        assert out.status == 0
        assert np.array_equal(out.data.view(np.int64), want)


def test_errors():
    P = pkg()
    L, I, B = P.KIND.LEVELS, P.KIND.INT32, P.KIND.BOOLEAN
    big = bytes([0x80, 0x80, 0x80, 0x80, 0x20])
    cases = [
        (L, 9, b"\x02"), (I, 33, b"\x02"), (L, 1, b"\x80"), (L, 1, b"\x80" * 11), (L, 3, b"\x03\xff"),
        (L, 1, b"\x08"), (L, 0, b"\x08"), (I, 1, big + b"\x01"), (I, 16, b"\x08\x07"), (I, 7, b"\x03" + b"\xaa" * 6),
        (L, 2, b"\x04\x03\x03\x1b\x80"), (B, 1, b"\x01\x02"), (B, 1, b"\x09\x00\x00\x00\x00"),
        (B, 1, b"\x00\x00\x00\x00"), (B, 1, b"\x02\x00\x00\x00\x07\xaa"), (B, 1, b"\x01\x00\x00\x00\x10"),
    ]
    outs = decode([spec(P.ENC.RLE, k, src, bit_width=bw, misalign=i % 16) for i, (k, bw, src) in enumerate(cases)])
    for (k, bw, src), out in zip(cases, outs):
        if k == L:
            want, dt = O.rle_decode_levels(src, bw), np.uint8
        elif k == I:
            want, dt = O.rle_decode_int32(src, bw), np.int32
        else:
            want, dt = O.rle_decode_boolean(src), np.uint8
        assert_same(out, want, dt, f"kind{k}-bw{bw}-{src.hex()}")
    out = decode([spec(P.ENC.RLE_DICTIONARY, P.KIND.INT32, b"\x21\x02\x01")])[0]
    assert out.status == P.STATUS.INVALID_BIT_WIDTH and out.arg == 33
    out = decode([spec(P.ENC.RLE, P.KIND.LEVELS, b"\x10\x01", bit_width=1, dst_cap=4)], grow=False)[0]
    assert out.status == P.STATUS.DST_TOO_SMALL and out.arg == 8


@pytest.mark.parametrize("np_dtype,kind_name", [(np.int64, "INT64"), (np.int32, "INT32"), (np.float64, "DOUBLE"), (np.float32, "FLOAT")])
def test_fused_dictionary(np_dtype, kind_name):
    # dictionary_test.go:15-138 sizes; index streams as the reference writer cuts them
    P = pkg()
    kind = getattr(P.KIND, kind_name)
    rng = np.random.default_rng(6)
    cases = []
    for dn in (1, 2, 3, 10, 17, 100, 1000, 10000, 65536):
        d = rng.integers(-2**31, 2**31 - 1, size=dn).astype(np_dtype)
        for n in (1, 9, 100, 2048, 5000):
            idx = rng.integers(0, dn, size=n, dtype=np.int32)
            if n >= 100:
                idx[20:70] = idx[20]
            cases.append((f"d{dn}-n{n}", d, idx, n))
    # TestNewIndexedPage: a stream shorter than num_values is zero-extended, a longer one cut
    d = np.arange(100, 120).astype(np_dtype)
    cases.append(("zero-extend", d, np.array([5, 6, 7], np.int32), 8))
    cases.append(("cut", d, np.arange(16, dtype=np.int32), 11))
    cases.append(("all-zero-indexes", d, np.zeros(4000, np.int32), 4000))
    specs = []
    for k, (label, d, idx, nv) in enumerate(cases):
        enc = O.rle_dict_encode_int32(idx)
        specs.append(spec(P.ENC.RLE_DICTIONARY, kind, enc, num_values=nv, dictionary=d, dst_cap=nv * d.dtype.itemsize, misalign=k % 16))
    outs = decode(specs, grow=False)
    for (label, d, idx, nv), out in zip(cases, outs):
        want_idx = O.indexed_page_fixup(O.rle_dict_decode_int32(O.rle_dict_encode_int32(idx)).values, nv)
        want = O.dict_lookup(d, want_idx)
        assert out.status == 0 and out.out_values == nv, label
        assert out.data.tobytes() == want.values.tobytes(), label


def test_index_pages_without_a_dictionary_are_fixed_up_too():
    # indexedPageType.Decode + newIndexedPage (dictionary.go:136-162, 217-219; TestNewIndexedPage,
    # dictionary_test.go:346-352): the int32 indexes of a dictionary-encoded page come back cut to
    # num_values or zero-extended to it -- the only path a byte-array dictionary column can take
    P = pkg()
    rng = np.random.default_rng(61)
    cases = [("zero-extend", np.array([5, 6, 7], np.int32), 8), ("cut", np.arange(16, dtype=np.int32), 11),
             ("exact", np.arange(24, dtype=np.int32), 24), ("empty-stream", np.zeros(0, np.int32), 5),
             ("long-extend", rng.integers(0, 1000, size=3000, dtype=np.int32), 7001),
             ("long-cut", rng.integers(0, 1000, size=9000, dtype=np.int32), 4097)]
    specs = [spec(P.ENC.RLE_DICTIONARY, P.KIND.INT32, O.rle_dict_encode_int32(idx), num_values=nv, dst_cap=nv * 4, misalign=k % 16)
             for k, (_, idx, nv) in enumerate(cases)]
    outs = decode(specs, grow=False)
    for (label, idx, nv), out in zip(cases, outs):
        want = O.indexed_page_fixup(O.rle_dict_decode_int32(O.rle_dict_encode_int32(idx)).values, nv)
        assert out.status == 0 and out.out_values == nv, label
        assert np.array_equal(out.data.view(np.int32), want), label
    # plain RLE int32 values and levels are only ever cut (column.go:940-945)
    v = np.array([3, 3, 3, 1], np.int32)
    out = decode([spec(P.ENC.RLE, P.KIND.INT32, O.rle_encode_int32(v, 2), bit_width=2, num_values=9, dst_cap=64)], grow=False)[0]
    assert out.status == 0 and out.out_values == 4


def test_dictionary_length_without_a_pointer_is_rejected():
    P = pkg()
    from tests.gpu_util import ctx
    from parquet_go_amd.runtime import Batch, to_device, DeviceBuffer
    src = to_device(O.rle_dict_encode_int32(np.arange(8, dtype=np.int32)))
    dst = DeviceBuffer(64)
    p = P.Page()
    p.src, p.src_len, p.dst, p.dst_cap = src.ptr, src.nbytes, dst.ptr, 64
    p.dict, p.dict_len, p.num_values = None, 10, 8
    p.encoding, p.kind = P.ENC.RLE_DICTIONARY, P.KIND.INT64
    b = Batch(ctx(), [p])
    b.launch(0)
    res = b.finish(0)
    assert res[0].status == P.STATUS.INVALID_ARGUMENT
    # a kind outside pqd_kind is not supported, whatever the encoding
    p.dict_len, p.kind, p.encoding = 0, 9, P.ENC.PLAIN
    b = Batch(ctx(), [p])
    b.launch(0)
    assert b.finish(0)[0].status == P.STATUS.NOT_SUPPORTED


def test_fused_dictionary_strided_and_int96():
    P = pkg()
    rng = np.random.default_rng(8)
    d = rng.integers(-2**62, 2**62, size=300, dtype=np.int64)
    idx = rng.integers(0, 300, size=3000, dtype=np.int32)
    enc = O.rle_dict_encode_int32(idx)
    # rows 24 bytes apart, like []parquet.Value (value.go:23-47)
    out = decode([spec(P.ENC.RLE_DICTIONARY, P.KIND.INT64, enc, num_values=3000, dictionary=d, dst_stride=24, dst_cap=3000 * 24)], grow=False)[0]
    assert out.status == 0
    rows = out.raw[: 3000 * 24].reshape(3000, 24)
    assert np.array_equal(rows[:, :8].copy().view(np.int64).ravel(), d[idx])
    assert (rows[:, 8:] == 0xA5).all()  # bytes between rows untouched
    d96 = rng.integers(0, 2**32, size=(50, 3), dtype=np.uint32)
    idx = rng.integers(0, 50, size=777, dtype=np.int32)
    out = decode([spec(P.ENC.PLAIN_DICTIONARY, P.KIND.INT96, O.rle_dict_encode_int32(idx), num_values=777,
                       dictionary=d96.reshape(-1).view(np.dtype("V12")), dst_cap=777 * 12)], grow=False)[0]
    assert out.status == 0 and np.array_equal(out.data.view(np.uint32).reshape(777, 3), d96[idx])


def test_fused_dictionary_index_out_of_bounds():
    # the reference panics "index out of bounds" (errors.go:75-83)
    P = pkg()
    d = np.arange(10, dtype=np.int64)
    idx = np.array([1, 2, 3, 10, 4, 11], np.int32)
    out = decode([spec(P.ENC.RLE_DICTIONARY, P.KIND.INT64, O.rle_dict_encode_int32(idx), num_values=6, dictionary=d, dst_cap=48)], grow=False)[0]
    want = O.dict_lookup(d, idx)
    assert out.status == want.status == P.STATUS.INDEX_OUT_OF_BOUNDS and out.arg == want.arg == 3


def test_parquet_file_dictionary_pages():
    # alltypes_dictionary / alltypes_tiny_pages / datapage_v2: dictionary page (PLAIN) on the
    # host side here, index page + lookup on the GPU; expected values from pyarrow
    P = pkg()
    items = [(r, a) for r, a in F.pages(kind=("v1", "v2"), encoding=(2, 8))
             if r["physical_type"] in ("INT32", "INT64", "FLOAT", "DOUBLE") and "expected_values" in a]
    assert len(items) > 20
    specs = []
    for k, (rec, a) in enumerate(items):
        drec, dbytes = F.dictionary_of(rec)
        d = dbytes.view(F.NP[rec["physical_type"]])
        specs.append(spec(rec["encoding"], getattr(P.KIND, rec["physical_type"]), a["values"].tobytes(),
                          num_values=rec["num_non_null"], dictionary=d, dst_cap=max(8, rec["num_non_null"] * d.dtype.itemsize), misalign=k % 16))
    outs = decode(specs, grow=False)
    for (rec, a), out in zip(items, outs):
        assert out.status == 0, F.page_id(rec)
        assert out.data.tobytes() == np.ascontiguousarray(a["expected_values"]).tobytes(), F.page_id(rec)


def test_parquet_file_levels_and_booleans():
    P = pkg()
    lv = [(r, a) for r, a in F.pages(kind=("v1", "v2")) if r["max_def"] > 0 and "expected_def" in a]
    specs = [spec(P.ENC.RLE, P.KIND.LEVELS, a["def"].tobytes(), bit_width=r["def_bit_width"], num_values=r["num_values"],
                  dst_cap=r["num_values"] + 8) for r, a in lv]
    outs = decode(specs, grow=False)
    for (r, a), out in zip(lv, outs):
        assert out.status == 0 and np.array_equal(out.data, a["expected_def"]), F.page_id(r)
    bl = list(F.pages(kind=("v1", "v2"), encoding=(3,), physical=("BOOLEAN",)))
    outs = decode([spec(P.ENC.RLE, P.KIND.BOOLEAN, a["values"].tobytes(), bit_width=1) for r, a in bl])
    for (r, a), out in zip(bl, outs):
        bits = np.unpackbits(out.data, bitorder="little")[: r["num_non_null"]]
        assert out.status == 0 and np.array_equal(bits, a["expected_values"]), F.page_id(r)
