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
