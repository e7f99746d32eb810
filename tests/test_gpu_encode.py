"""GPU: the page encoders against the CPU checker's encoders, byte for byte
(encoding/delta/binary_packed.go:77-277, encoding/rle/dictionary.go:23-59 + rle.go:203-249,
encoding/bytestreamsplit/bytestreamsplit_purego.go:7-45), and back through the GPU decoders.
The checker's encoders are pinned by the reference's own vectors (TestEncodeMiniBlockInt32/64
layouts, the RLE and BIT_PACKED known answers) in the CPU tests."""
import numpy as np
import pytest

from oracle import pqoracle as O
from tests.gpu_util import ctx, decode, pkg, spec

pytestmark = pytest.mark.gpu


def _encode(jobs, **k):
    from parquet_go_amd.runtime import encode_pages
    return encode_pages(ctx(), jobs, **k)


def _delta_inputs(rng, dtype):
    info = np.iinfo(dtype)
    out = []
    for n in (0, 1, 2, 3, 64, 127, 128, 129, 130, 255, 256, 257, 1000, 2048, 2049, 2050, 4097, 40000, 131073):
        out.append((f"steps-n{n}", np.cumsum(rng.integers(-50, 1000, size=n)).astype(dtype)))
    out.append(("constant", np.full(3000, 7, dtype)))
    out.append(("zeros", np.zeros(777, dtype)))
    out.append(("extremes", rng.choice(np.array([info.min, info.max, 0, -1, 1], dtype), size=5000)))        # differences wrap
    out.append(("random-wide", rng.integers(info.min, info.max, size=9999, dtype=dtype, endpoint=True)))
    out.append(("descending", (np.arange(6000, 0, -1) * 1000).astype(dtype)))
    widths = np.concatenate([rng.integers(0, 1 << w, size=32 * int(rng.integers(1, 9)), dtype=np.int64) for w in rng.integers(0, 28, size=60)])
    out.append(("widths-change", np.cumsum(widths).astype(dtype)))
    # one large difference in an otherwise flat block: a wide miniblock beside empty ones
    spike = np.zeros(1500, np.int64); spike[[100, 700, 1499]] = [1 << 29, -(1 << 28), 12345]
    out.append(("spikes", np.cumsum(spike).astype(dtype)))
    # the last block holds one value, two values, a full miniblock ...: the zero padding takes part in the minimum
    for tail in (1, 2, 31, 32, 33, 96, 127):
        out.append((f"tail{tail}", (rng.integers(100, 200, size=1 + 128 * 3 + tail)).astype(dtype)))
        out.append((f"tail{tail}-neg", -(rng.integers(100, 200, size=1 + 128 * 3 + tail)).astype(dtype)))
    return out


@pytest.mark.parametrize("np_dtype,kind_name", [(np.int32, "INT32"), (np.int64, "INT64")])
def test_delta_binary_packed_bytes(np_dtype, kind_name):
    P = pkg()
    kind = getattr(P.KIND, kind_name)
    rng = np.random.default_rng(11)
    cases = _delta_inputs(rng, np_dtype)
    enc = O.delta_encode_int32 if np_dtype == np.int32 else O.delta_encode_int64
    for misalign in (0, 3):
        outs = _encode([(P.ENC.DELTA_BINARY_PACKED, kind, v, 0) for _, v in cases], misalign=misalign)
        for (label, v), (status, arg, page, raw) in zip(cases, outs):
            want = bytes(enc(v))
            assert status == 0, (label, status, arg)
            assert page == want, (label, len(page), len(want), next((i for i, (a, b) in enumerate(zip(page, want)) if a != b), None))
            assert (raw[len(page):] == 0xA5).all(), label  # nothing written behind the page
    # and the GPU decoder reads them back
    specs = [spec(P.ENC.DELTA_BINARY_PACKED, kind, page, dst_cap=max(v.nbytes, 8)) for (_, v), (_, _, page, _) in zip(cases, outs)]
    for (label, v), out in zip(cases, decode(specs)):
        assert out.status == 0 and np.array_equal(out.data.view(np_dtype), v), label


def test_delta_destination_too_small():
    P = pkg()
    rng = np.random.default_rng(12)
    v = np.cumsum(rng.integers(0, 1 << 20, size=50000)).astype(np.int64)
    want = bytes(O.delta_encode_int64(v))
    from parquet_go_amd._lib import EncodePage
    from parquet_go_amd.runtime import DeviceBuffer, to_device
    import torch
    L = P.lib()
    src = to_device(v)
    for cap in (0, 5, len(want) - 1, len(want)):
        dst = DeviceBuffer(len(want) + 64, fill=0xA5)
        p = EncodePage()
        p.src, p.n_values, p.dst, p.dst_cap = src.ptr, v.size, dst.ptr, cap
        p.encoding, p.kind = P.ENC.DELTA_BINARY_PACKED, P.KIND.INT64
        arr = (EncodePage * 1)(p)
        ws = torch.empty(int(L.pqd_encode_workspace_size(arr, 1)) + 256, dtype=torch.uint8, device="cuda:0")
        assert L.pqd_encode_batch(ctx().handle, arr, 1, ws.data_ptr(), ws.numel(), None) == 0
        raw = dst.numpy()
        if cap < len(want):
            assert (arr[0].status, arr[0].status_arg, arr[0].out_bytes) == (P.STATUS.DST_TOO_SMALL, len(want), 0)
            assert (raw[cap:] == 0xA5).all()
        else:
            assert arr[0].status == 0 and raw[: arr[0].out_bytes].tobytes() == want


def test_byte_stream_split_bytes():
    P = pkg()
    rng = np.random.default_rng(13)
    jobs, wants = [], []
    for n in (0, 1, 3, 4, 5, 1023, 1024, 1025, 70001):
        for dtype, kind in ((np.float32, P.KIND.FLOAT), (np.float64, P.KIND.DOUBLE), (np.int32, P.KIND.INT32), (np.int64, P.KIND.INT64)):
            v = rng.integers(0, 256, size=n * np.dtype(dtype).itemsize, dtype=np.uint8).view(dtype)
            jobs.append((P.ENC.BYTE_STREAM_SPLIT, kind, v, 0))
            wants.append(bytes(O.bss_encode(v.tobytes(), np.dtype(dtype).itemsize)))
        for size in (1, 2, 3, 5, 16, 37):
            v = rng.integers(0, 256, size=n * size, dtype=np.uint8)
            jobs.append((P.ENC.BYTE_STREAM_SPLIT, P.KIND.FIXED_LEN_BYTE_ARRAY, v, size))
            wants.append(bytes(O.bss_encode(v.tobytes(), size)))
    for misalign in (0, 1):
        outs = _encode(jobs, misalign=misalign)
        for k, ((status, arg, page, raw), want) in enumerate(zip(outs, wants)):
            assert status == 0 and page == want, (k, status, arg, len(page), len(want))
            assert (raw[len(page):] == 0xA5).all(), k


def test_unsupported_combinations_and_the_codec_objects():
    P = pkg()
    from parquet_go_amd import encoding as E
    outs = _encode([(P.ENC.DELTA_BINARY_PACKED, P.KIND.DOUBLE, np.zeros(4), 0), (P.ENC.PLAIN, P.KIND.INT32, np.zeros(4, np.int32), 0),
                    (P.ENC.BYTE_STREAM_SPLIT, P.KIND.BYTE_ARRAY, np.zeros(4, np.uint8), 0)])
    assert [o[0] for o in outs] == [P.STATUS.NOT_SUPPORTED] * 3
    # typed values are aligned like a Go slice
    bad = _encode([(P.ENC.DELTA_BINARY_PACKED, P.KIND.INT64, np.arange(9, dtype=np.int64), 0)], src_misalign=4)
    assert bad[0][:2] == (P.STATUS.INVALID_ARGUMENT, 4)
    rng = np.random.default_rng(14)
    v64 = np.cumsum(rng.integers(-9, 99999, size=33333)).astype(np.int64)
    v32 = rng.integers(-2**31, 2**31 - 1, size=5000, dtype=np.int32)
    assert E.DeltaBinaryPacked().EncodeInt64(None, v64) == bytes(O.delta_encode_int64(v64))
    assert E.DeltaBinaryPacked().EncodeInt32(None, v32) == bytes(O.delta_encode_int32(v32))
    f = rng.random(7777).astype(np.float32)
    assert E.ByteStreamSplit().EncodeFloat(None, f) == bytes(O.bss_encode(f.tobytes(), 4))
    assert np.array_equal(E.ByteStreamSplit().DecodeFloat(None, E.ByteStreamSplit().EncodeFloat(None, f)), f)
    fl = rng.integers(0, 256, size=16 * 900, dtype=np.uint8)
    assert E.ByteStreamSplit().EncodeFixedLenByteArray(None, fl, 16) == bytes(O.bss_encode(fl.tobytes(), 16))
    with pytest.raises(E.ErrNotSupported):
        E.Plain().EncodeInt64(None, v64)
    with pytest.raises(E.ErrNotSupported):
        E.DeltaBinaryPacked().EncodeDouble(None, f)


def _index_inputs(rng):
    out = []
    for n in (0, 1, 7, 8, 9, 15, 16, 17, 63, 64, 65, 2047, 2048, 2049, 2055, 2056, 4096, 50000, 300007):
        out.append((f"random-n{n}", rng.integers(0, 1000, size=n, dtype=np.int32)))
    out.append(("zeros", np.zeros(5000, np.int32)))                       # bit width 0: one header, no value
    out.append(("zeros-tail", np.zeros(5003, np.int32)))
    out.append(("one-value", np.full(4099, 77, np.int32)))                 # one run over many tiles, then a tail
    out.append(("negative", np.array([-1, 5, 5, 5, 5, 5, 5, 5, 5] * 300, np.int32)))  # 32-bit fields
    out.append(("width-1", rng.integers(0, 2, size=30011, dtype=np.int32)))
    out.append(("width-31", rng.integers(0, 2**31 - 1, size=9000, dtype=np.int32)))
    # runs of repeats and bit-packed stretches of every length, crossing the 256-group tiles
    v = np.empty(0, np.int32)
    while v.size < 200000:
        k = int(rng.integers(1, 5000)) if rng.integers(0, 2) else int(rng.integers(1, 40))
        part = np.full(k, rng.integers(0, 300), np.int32) if rng.integers(0, 2) else rng.integers(0, 300, size=k, dtype=np.int32)
        v = np.concatenate([v, part])
    out.append(("writer-like", v))
    # neighbouring runs of repeats with different values, aligned to the groups of eight
    out.append(("aligned-runs", np.repeat(rng.integers(0, 50, size=3000, dtype=np.int32), 8)))
    out.append(("aligned-runs-16", np.repeat(rng.integers(0, 3, size=3000, dtype=np.int32), 16)))
    # a group of one value inside a bit-packed stretch ends it (rle.go:478-492)
    w = rng.integers(0, 9, size=8 * 1000, dtype=np.int32)
    w[8 * 10: 8 * 11] = 4
    w[8 * 500: 8 * 503] = 2
    out.append(("broadcast-inside", w))
    return out


def test_rle_dictionary_index_pages_bytes():
    P = pkg()
    rng = np.random.default_rng(15)
    cases = _index_inputs(rng)
    for misalign in (0, 1):
        outs = _encode([(P.ENC.RLE_DICTIONARY, P.KIND.INT32, v, 0) for _, v in cases], misalign=misalign, src_misalign=4 * misalign)
        for (label, v), (status, arg, page, raw) in zip(cases, outs):
            want = bytes(O.rle_dict_encode_int32(v))
            assert status == 0, (label, status, arg)
            assert page == want, (label, len(page), len(want), next((i for i, (a, b) in enumerate(zip(page, want)) if a != b), None))
            assert (raw[len(page):] == 0xA5).all(), label
    specs = [spec(P.ENC.RLE_DICTIONARY, P.KIND.INT32, page, dst_cap=max(v.nbytes, 8)) for (_, v), (_, _, page, _) in zip(cases, outs)]
    for (label, v), out in zip(cases, decode(specs)):
        assert out.status == 0 and np.array_equal(out.data.view(np.int32)[: v.size], v), label
    from parquet_go_amd import encoding as E
    v = cases[-4][1]
    assert E.RLEDictionary().EncodeInt32(None, v) == bytes(O.rle_dict_encode_int32(v))
    assert np.array_equal(E.RLEDictionary().DecodeInt32(None, E.RLEDictionary().EncodeInt32(None, v)), v)


def _bounds(d, idx):
    """dictionary_purego.go:68-170, as written."""
    lo = hi = d[idx[0]]
    for i in idx[1:]:
        v = d[i]
        if v < lo:
            lo = v
        if v > hi:
            hi = v
    return lo, hi


def test_dictionary_bounds():
    from parquet_go_amd.runtime import dict_bounds
    P = pkg()
    rng = np.random.default_rng(16)
    for dtype in (np.int32, np.int64, np.uint32, np.uint64, np.float32, np.float64):
        if np.dtype(dtype).kind == "f":
            d = (rng.standard_normal(5000) * 1e6).astype(dtype)
        else:
            info = np.iinfo(dtype)
            d = rng.integers(info.min, info.max, size=5000, dtype=dtype, endpoint=True)
        for n in (1, 2, 100, 5000, 400001):
            idx = rng.integers(0, d.size, size=n, dtype=np.int32)
            st, arg, lo, hi = dict_bounds(ctx(), d, idx)
            assert (st, arg) == (0, 0)
            assert (lo, hi) == (d[idx].min(), d[idx].max()), (dtype, n)
        st, arg, lo, hi = dict_bounds(ctx(), d, np.zeros(0, np.int32))
        assert (st, lo, hi) == (0, 0, 0)
        idx = rng.integers(0, d.size, size=70000, dtype=np.int32)
        idx[[123, 60000]] = [d.size, -1]
        st, arg, lo, hi = dict_bounds(ctx(), d, idx)
        assert (st, arg) == (P.STATUS.INDEX_OUT_OF_BOUNDS, 123)
    # NaN and signed zeros follow the comparisons of the reference's loop
    for dtype in (np.float32, np.float64):
        d = np.array([1.5, np.nan, -0.0, 0.0, -3.0, np.inf, -np.inf, 7.0], dtype)
        for idx in ([0, 1, 4], [1, 0, 4], [2, 3], [3, 2], [3, 2, 2, 3], [1], [1, 1], [0, 5, 6, 1], [2, 3, 0], [7, 1, 1, 0, 4] * 5000, [1] + [4, 7] * 40000):
            st, arg, lo, hi = dict_bounds(ctx(), d, np.array(idx, np.int32))
            wlo, whi = _bounds(d, idx)
            assert st == 0
            assert np.array([lo, hi], dtype).tobytes() == np.array([wlo, whi], dtype).tobytes(), (dtype, idx[:6], lo, hi, wlo, whi)


# ---- RLE levels / booleans / int32 with the width given, and the byte-array encoders ----

def _level_inputs(rng, bw):
    hi = 1 << bw if bw else 1
    out = []
    for n in (0, 1, 7, 8, 9, 15, 16, 17, 64, 2047, 2048, 2049, 2056, 5000, 40003):
        out.append((f"random-n{n}", rng.integers(0, hi, size=n, dtype=np.uint8)))
    # runs of words, words that are uniform inside bit-packed runs, a uniform word behind a word that starts with the same byte
    # (rle.go:176-178 compares a word with the broadcast of its predecessor's FIRST byte)
    for n in (64, 3000, 20000, 70000):
        v = np.empty(n, np.uint8)
        at = 0
        while at < n:
            k = int(rng.choice([1, 2, 3, 7, 8, 9, 16, 24, 40, 64, 200, 3000]))
            kind = int(rng.integers(0, 3))
            if kind == 0:
                v[at: at + k] = rng.integers(0, hi)
            elif kind == 1:
                v[at: at + k] = rng.integers(0, hi, size=min(k, n - at))
            else:
                a, b = rng.integers(0, hi, size=2)
                seg = np.full(min(k, n - at), a, np.uint8)
                seg[::8] = b  # every word starts with b and goes on with a
                v[at: at + k] = seg
            at += k
        out.append((f"runs-n{n}", v))
    w = np.zeros(8 * 600, np.uint8)
    for i in range(600):  # all four (uniform?, equals the broadcast of the byte before?) cases side by side
        first = rng.integers(0, hi)
        w[8 * i: 8 * i + 8] = first if rng.integers(0, 2) else rng.integers(0, hi, size=8)
        if rng.integers(0, 3) == 0 and i > 0:
            w[8 * i: 8 * i + 8] = w[8 * (i - 1)]
    out.append(("word-cases", w))
    out.append(("zeros", np.zeros(4100, np.uint8)))
    out.append(("ones-tail", np.concatenate([np.full(4096, hi - 1, np.uint8), rng.integers(0, hi, size=5, dtype=np.uint8)])))
    return out


@pytest.mark.parametrize("bw", [0, 1, 2, 3, 5, 8])
def test_rle_levels_bytes(bw):
    P = pkg()
    rng = np.random.default_rng(20 + bw)
    cases = _level_inputs(rng, bw)
    for misalign in (0, 5):
        outs = _encode([(P.ENC.RLE, P.KIND.LEVELS, v, bw) for _, v in cases], misalign=misalign, src_misalign=misalign)
        for (label, v), (status, arg, page, raw) in zip(cases, outs):
            want = bytes(O.rle_encode_levels(v, bw))
            assert status == 0, (label, status, arg)
            assert page == want, (label, len(page), len(want), next((i for i, (a, b) in enumerate(zip(page, want)) if a != b), None))
            assert (raw[len(page):] == 0xA5).all(), label
    specs = [spec(P.ENC.RLE, P.KIND.LEVELS, page, bit_width=bw, dst_cap=v.size + 64) for (_, v), (_, _, page, _) in zip(cases, outs) if v.size]
    for (label, v), out in zip([c for c in cases if c[1].size], decode(specs)):
        assert out.status == 0 and np.array_equal(out.data[: v.size], v & ((1 << bw) - 1)), label


def test_rle_levels_refused_widths():
    P = pkg()
    outs = _encode([(P.ENC.RLE, P.KIND.LEVELS, np.zeros(20, np.uint8), 9), (P.ENC.RLE, P.KIND.LEVELS, np.array([0, 0, 1] * 10, np.uint8), 0),
                    (P.ENC.RLE, P.KIND.INT32, np.zeros(20, np.int32), 33), (P.ENC.RLE, P.KIND.INT32, np.array([0, 5] * 10, np.int32), 0)])
    assert [o[:2] for o in outs] == [(P.STATUS.INVALID_BIT_WIDTH, 9), (P.STATUS.INVALID_BIT_WIDTH, 0), (P.STATUS.INVALID_BIT_WIDTH, 33),
                                     (P.STATUS.INVALID_BIT_WIDTH, 0)]


@pytest.mark.parametrize("bw", [0, 1, 4, 7, 8, 13, 16, 25, 32])
def test_rle_int32_with_the_width_given(bw):
    P = pkg()
    rng = np.random.default_rng(40 + bw)
    cases = []
    for n in (0, 1, 8, 9, 100, 2048, 2055, 30000):
        v = rng.integers(0, 1 << bw if bw else 1, size=n, dtype=np.int64).astype(np.uint32).view(np.int32)
        cases.append(v)
        r = v.copy()
        for k in range(0, n, 97):
            r[k: k + int(rng.integers(1, 60))] = r[k]
        cases.append(r)
    outs = _encode([(P.ENC.RLE, P.KIND.INT32, v, bw) for v in cases])
    for v, (status, arg, page, raw) in zip(cases, outs):
        assert status == 0 and page == bytes(O.rle_encode_int32(v, bw)), (v.size, status, arg)


def test_rle_boolean_bytes():
    P = pkg()
    rng = np.random.default_rng(50)
    cases = [np.zeros(0, np.uint8), np.zeros(1, np.uint8), np.full(300, 0xFF, np.uint8), np.zeros(5000, np.uint8)]
    for n in (1, 2, 3, 9, 100, 4097, 20000):
        cases.append(rng.integers(0, 256, size=n, dtype=np.uint8))
        # stretches of 0x00, 0xFF and of other repeated bytes (rle.go:125 treats 0xFF pairs unlike 0x00 pairs)
        v = rng.choice(np.array([0, 0, 0xFF, 0xFF, 0x55, 0x55, 1, 0x80], np.uint8), size=n)
        for k in range(0, n, 13):
            v[k: k + int(rng.integers(1, 12))] = v[k]
        cases.append(v)
        cases.append(np.repeat(rng.choice(np.array([0, 0xFF, 7], np.uint8), size=n // 3 + 1), 3)[:n])
    # stretches that cross the 4096-byte steps of the kernel: 0xFF behind zeros, 0xFF behind something else, long runs
    for lead in (4090, 4094, 4095, 4096, 8191):
        cases.append(np.concatenate([np.zeros(lead, np.uint8), np.full(9000, 0xFF, np.uint8), np.full(5, 3, np.uint8), np.zeros(4100, np.uint8)]))
        cases.append(np.concatenate([rng.integers(1, 255, size=lead - 1, dtype=np.uint8), np.array([0], np.uint8), np.full(4100, 0xFF, np.uint8), np.zeros(2, np.uint8),
                                     np.full(3, 0xFF, np.uint8), np.full(4097, 0x11, np.uint8)]))
    cases.append(rng.choice(np.array([0, 0xFF, 0x55], np.uint8), size=70000))
    outs = _encode([(P.ENC.RLE, P.KIND.BOOLEAN, v, 0) for v in cases], misalign=1)
    for k, (v, (status, arg, page, raw)) in enumerate(zip(cases, outs)):
        want = bytes(O.rle_encode_boolean(v))
        assert status == 0 and page == want, (k, v.size, status, arg, len(page), len(want))
        assert (raw[len(page):] == 0xA5).all(), k
    specs = [spec(P.ENC.RLE, P.KIND.BOOLEAN, page, dst_cap=v.size + 64) for v, (_, _, page, _) in zip(cases, outs)]
    for k, (v, out) in enumerate(zip(cases, decode(specs))):
        assert out.status == 0 and np.array_equal(out.data[: v.size], v), k


def _strings(rng, kind, n):
    if kind == "urls":
        hosts = [b"https://www.example.com/products/category/", b"https://cdn.example.org/static/images/thumbs/2024/", b"http://a.b/"]
        return sorted(hosts[int(rng.integers(0, 3))] + b"%08d" % int(rng.integers(0, 10**8)) for _ in range(n))
    if kind == "random":
        return [bytes(rng.integers(0, 256, size=int(rng.integers(0, 22)), dtype=np.uint8)) for _ in range(n)]
    if kind == "long":
        base = bytes(rng.integers(0, 256, size=3000, dtype=np.uint8))
        return [base[: int(rng.integers(0, 3000))] + bytes(rng.integers(0, 256, size=int(rng.integers(0, 40)), dtype=np.uint8)) for _ in range(n)]
    if kind == "empty":
        return [b""] * n
    if kind == "same":
        return [b"the same value every time"] * n
    raise ValueError(kind)


def _byte_array_cases(rng):
    cases = []
    for kind in ("urls", "random", "long", "empty", "same"):
        for n in (0, 1, 2, 63, 64, 65, 129, 1000, 5003):
            if kind == "long" and n > 1000:
                continue
            data, offs = O._flatten(_strings(rng, kind, n))
            cases.append((f"{kind}-n{n}", data, offs))
    return cases


@pytest.mark.parametrize("enc_name", ["PLAIN", "DELTA_LENGTH_BYTE_ARRAY", "DELTA_BYTE_ARRAY"])
def test_byte_array_encoders_bytes(enc_name):
    P = pkg()
    rng = np.random.default_rng(60)
    cases = _byte_array_cases(rng)
    enc = getattr(P.ENC, enc_name)
    oracle = {"PLAIN": lambda d, o: O._enc(O.lib().pqo_plain_encode_byte_array, d.size + 4 * o.size + 16, O._p(d), O._p(o), o.size),
              "DELTA_LENGTH_BYTE_ARRAY": lambda d, o: O.dlba_encode(data=d, offsets=o),
              "DELTA_BYTE_ARRAY": lambda d, o: O.dba_encode(data=d, offsets=o)}[enc_name]
    for misalign in (0, 7):
        outs = _encode([(enc, P.KIND.BYTE_ARRAY, d, 0, o) for _, d, o in cases], misalign=misalign, src_misalign=misalign)
        for (label, d, o), (status, arg, page, raw) in zip(cases, outs):
            want = bytes(oracle(d, o))
            assert status == 0, (label, status, arg)
            assert page == want, (label, len(page), len(want), next((i for i, (a, b) in enumerate(zip(page, want)) if a != b), None))
            assert (raw[len(page):] == 0xA5).all(), label
    # and the GPU decoders read them back
    specs = [spec(enc, P.KIND.BYTE_ARRAY, page, dst_cap=d.size + 64, offsets_cap=o.size + 1) for (_, d, o), (_, _, page, _) in zip(cases, outs)]
    for (label, d, o), out in zip(cases, decode(specs)):
        assert out.status == 0 and out.data.tobytes() == d.tobytes() and np.array_equal(out.offsets[: o.size], o), label


def test_byte_array_encoders_too_small_and_fixed_len():
    P = pkg()
    rng = np.random.default_rng(61)
    data, offs = O._flatten(_strings(rng, "urls", 3000))
    for enc, want in ((P.ENC.PLAIN, data.size + 4 * 3000), (P.ENC.DELTA_LENGTH_BYTE_ARRAY, len(O.dlba_encode(data=data, offsets=offs))),
                      (P.ENC.DELTA_BYTE_ARRAY, len(O.dba_encode(data=data, offsets=offs)))):
        bound = int(P.lib().pqd_encode_bound_byte_array(enc, 3000, data.size))
        assert bound >= want
        for short in (1, 40, want - 100, want):  # by how much the destination is too small: the size to come back with
            status, arg, page, raw = _encode([(enc, P.KIND.BYTE_ARRAY, data, 0, offs)], dst_slack=want - short - bound)[0]
            assert (status, arg, page) == (P.STATUS.DST_TOO_SMALL, want, b""), (enc, short, status, arg)
    # FIXED_LEN_BYTE_ARRAY through DELTA_BYTE_ARRAY (byte_array.go:76-122)
    for size in (1, 5, 16, 100):
        for n in (0, 1, 200, 3000):
            v = np.repeat(rng.integers(0, 256, size=(n // 4 + 1) * size, dtype=np.uint8).reshape(-1, size), 4, axis=0)[:n].copy()
            v[:, -1] = rng.integers(0, 256, size=n)
            flat = v.reshape(-1)
            want = bytes(O.dba_encode(data=flat, offsets=np.arange(n + 1, dtype=np.uint32) * size))
            status, arg, page, raw = _encode([(P.ENC.DELTA_BYTE_ARRAY, P.KIND.FIXED_LEN_BYTE_ARRAY, flat, size)])[0]
            assert status == 0 and page == want, (size, n, status, arg)
    # through the host entry point a cgo binding calls
    import ctypes as C
    L = P.lib()
    for enc, want in ((P.ENC.PLAIN, None), (P.ENC.DELTA_LENGTH_BYTE_ARRAY, bytes(O.dlba_encode(data=data, offsets=offs))),
                      (P.ENC.DELTA_BYTE_ARRAY, bytes(O.dba_encode(data=data, offsets=offs)))):
        cap = int(L.pqd_encode_bound_byte_array(enc, 3000, data.size))
        dst = np.zeros(cap, np.uint8)
        n_out, arg = C.c_uint64(0), C.c_int64(0)
        rc = L.pqd_encode_byte_array(ctx().handle, enc, P.KIND.BYTE_ARRAY, 0, data.ctypes.data, data.size, offs.ctypes.data, 3000, dst.ctypes.data, cap,
                                     C.byref(n_out), C.byref(arg))
        assert rc == 0
        if want is not None:
            assert dst[: n_out.value].tobytes() == want
        else:
            assert n_out.value == data.size + 4 * 3000


@pytest.mark.parametrize("np_dtype,kind_name,unsigned", [(np.int32, "INT32", 0), (np.int64, "INT64", 0), (np.uint32, "INT32", 1), (np.uint64, "INT64", 1),
                                                         (np.float32, "FLOAT", 0), (np.float64, "DOUBLE", 0)])
def test_page_bounds(np_dtype, kind_name, unsigned):
    """boundsInt32 ... boundsFloat64 (page_bounds_purego.go:9-107): min and max by the loop's comparisons."""
    import torch
    P = pkg()
    L = P.lib()
    rng = np.random.default_rng(70)
    kind = getattr(P.KIND, kind_name)
    ws = torch.zeros(64 << 10, dtype=torch.uint8, device="cuda")
    out = torch.zeros(2, dtype=torch.int64, device="cuda")
    st = torch.zeros(2, dtype=torch.int64, device="cuda")

    def reference(v):
        if v.size == 0:
            return np.zeros(2, np_dtype)
        lo = hi = v[0]
        for x in v[1:]:
            if x < lo:
                lo = x
            if x > hi:
                hi = x
        return np.array([lo, hi], np_dtype)

    cases = [np.zeros(0, np_dtype)]
    for n in (1, 2, 100, 2049, 100003):
        if np.issubdtype(np_dtype, np.floating):
            v = rng.standard_normal(n).astype(np_dtype)
            cases.append(v)
            w = v.copy(); w[rng.integers(0, n, size=max(1, n // 10))] = np.nan; w[0] = 1.5
            cases.append(w)                       # NaNs inside never win
            z = v.copy(); z[0] = np.nan
            cases.append(z)                       # a NaN in front stays
            cases.append(np.where(rng.integers(0, 2, size=n) == 1, np_dtype(0.0), np_dtype(-0.0)).astype(np_dtype))  # +0 and -0: the first stays
        else:
            info = np.iinfo(np_dtype)
            cases.append(rng.integers(info.min, info.max, size=n, dtype=np_dtype, endpoint=True))
            cases.append(np.full(n, info.max, np_dtype))
    for v in cases:
        t = torch.from_numpy(v.view(np.int32 if v.itemsize == 4 else np.int64).copy()).cuda() if v.size else torch.zeros(1, dtype=torch.int64, device="cuda")
        rc = L.pqd_page_bounds_device(ctx().handle, kind, unsigned, t.data_ptr(), v.size, out.data_ptr(), ws.data_ptr(), ws.numel(), st.data_ptr(), None)
        torch.cuda.synchronize()
        assert rc == 0 and st.tolist() == [0, 0]
        got = out.cpu().numpy().view(np.uint8)[: 2 * v.itemsize].view(np_dtype)
        want = reference(v[: min(v.size, 3000)]) if v.size <= 3000 else np.array([np.nanmin(v) if v[0] == v[0] else v[0], np.nanmax(v) if v[0] == v[0] else v[0]], np_dtype)
        assert got.tobytes() == want.tobytes() or (v.size > 3000 and np.array_equal(got, want)), (v.size, got, want)


def test_encode_methods_of_the_codec_objects_for_levels_booleans_and_byte_arrays():
    """The host entry points a cgo binding calls (pqd_encode_values, pqd_encode_byte_array) behind the mirror of
    encoding.Encoding's Encode methods (encoding/encoding.go:29-41)."""
    pkg()
    from parquet_go_amd import encoding as E
    rng = np.random.default_rng(80)
    lv = np.repeat(rng.integers(0, 4, size=700, dtype=np.uint8), rng.integers(1, 30, size=700))
    assert E.RLE(2).EncodeLevels(None, lv) == bytes(O.rle_encode_levels(lv, 2))
    with pytest.raises(E.EncodingError):
        E.RLE(9).EncodeLevels(None, lv)
    bits = np.repeat(rng.choice(np.array([0, 0xFF, 0x3C], np.uint8), size=900), rng.integers(1, 9, size=900))
    assert E.RLE().EncodeBoolean(None, bits) == bytes(O.rle_encode_boolean(bits))
    idx = rng.integers(0, 1 << 11, size=5001, dtype=np.int32)
    assert E.RLE(11).EncodeInt32(None, idx) == bytes(O.rle_encode_int32(idx, 11))
    data, offs = O._flatten(_strings(rng, "urls", 2500))
    assert E.DeltaLengthByteArray().EncodeByteArray(None, data, offs) == bytes(O.dlba_encode(data=data, offsets=offs))
    assert E.DeltaByteArray().EncodeByteArray(None, data, offs) == bytes(O.dba_encode(data=data, offsets=offs))
    assert E.Plain().EncodeByteArray(None, data, offs) == bytes(O.plain_encode_byte_array(_split(data, offs)))
    assert E.DeltaLengthByteArray().EncodeByteArray(None, b"", None) == b""
    fl = rng.integers(0, 4, size=16 * 999, dtype=np.uint8)
    assert E.DeltaByteArray().EncodeFixedLenByteArray(None, fl, 16) == bytes(O.dba_encode(data=fl, offsets=np.arange(1000, dtype=np.uint32) * 16))
    with pytest.raises(E.ErrInvalidArgument):
        E.DeltaByteArray().EncodeFixedLenByteArray(None, fl[:-1], 16)
    # and back
    v, o = E.DeltaByteArray().DecodeByteArray(None, E.DeltaByteArray().EncodeByteArray(None, data, offs), None)
    assert v.tobytes() == data.tobytes() and np.array_equal(o, offs)


def _split(data, offs):
    b = data.tobytes()
    return [b[int(offs[i]): int(offs[i + 1])] for i in range(len(offs) - 1)]
