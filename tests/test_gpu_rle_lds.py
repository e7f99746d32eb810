"""GPU: RLE_DICTIONARY pages decoded straight to values, with each of the gather paths
forced in turn -- dictionary through the caches (PQD_DICT_LDS=0), dictionary resident in LDS
(PQD_DICT_LDS=2), only the first part of the dictionary in LDS with the rest read from
memory (PQD_DICT_LDS_CAP_BYTES, PQD_DICT_SLICES=0), and the dictionary sliced over groups of
workgroups (a family is sliced when all its dictionaries are larger than LDS and need no more
slices than allowed; PQD_DICT_SLICE_BYTES and PQD_DICT_LDS_CAP_BYTES bring that down to the
sizes of test_sliced_families) -- against the CPU checker's decode + newIndexedPage fix-up +
dictionary lookup (encoding/rle/dictionary.go:30-37, dictionary.go:136-162,
dictionary_purego.go:12-38).  Bit exact; same statuses and positions for indexes outside
the dictionary."""
import os

import numpy as np
import pytest

from oracle import pqoracle as O
from tests.gpu_util import pkg, spec

pytestmark = pytest.mark.gpu

_ctxs = {}


def _ctx(mode):
    """A context per gather path; the library reads the knobs when a context is made."""
    if mode not in _ctxs:
        env = {"cache": {"PQD_DICT_LDS": "0"},
               "lds": {"PQD_DICT_LDS": "2", "PQD_DICT_LDS_MAX_BYTES": str(1 << 30)},
               "part": {"PQD_DICT_LDS": "2", "PQD_DICT_LDS_MAX_BYTES": str(1 << 30), "PQD_DICT_LDS_CAP_BYTES": "4096",
                        "PQD_DICT_SLICES": "0"},
               "sliced": {"PQD_DICT_LDS": "2", "PQD_DICT_LDS_MAX_BYTES": str(1 << 30), "PQD_DICT_LDS_CAP_BYTES": "4096",
                          "PQD_DICT_SLICES": "8", "PQD_DICT_SLICE_BYTES": "2048"},
               "sliced3": {"PQD_DICT_LDS": "2", "PQD_DICT_LDS_MAX_BYTES": str(1 << 30), "PQD_DICT_LDS_CAP_BYTES": "65536",
                           "PQD_DICT_SLICES": "3", "PQD_DICT_SLICE_BYTES": "48000"}}[mode]
        keys = ("PQD_DICT_LDS", "PQD_DICT_LDS_MAX_BYTES", "PQD_DICT_LDS_CAP_BYTES", "PQD_DICT_SLICES", "PQD_DICT_SLICE_BYTES")
        old = {k: os.environ.get(k) for k in keys}
        for k in keys:
            os.environ.pop(k, None)
        os.environ.update(env)
        try:
            _ctxs[mode] = pkg().Context(0)
        finally:
            for k in keys:
                os.environ.pop(k, None)
                if old[k] is not None:
                    os.environ[k] = old[k]
    return _ctxs[mode]


MODES = ("cache", "lds", "part", "sliced", "sliced3")


def _decode(mode, specs):
    from parquet_go_amd.runtime import decode_pages
    return decode_pages(_ctx(mode), specs, grow=False)


def _want(d, enc, nv):
    idx = O.indexed_page_fixup(O.rle_dict_decode_int32(enc).values, nv)
    return O.dict_lookup(d, idx)


def _writer_runs(rng, n, dn, mean_run):
    """Indexes with runs of geometric length: the writer turns 8-aligned groups of eight equal values into
    run-length runs and the rest into bit-packed groups, so tiles meet many runs of both kinds."""
    out = np.empty(n, np.int32)
    at = 0
    while at < n:
        k = int(rng.geometric(1.0 / mean_run))
        out[at: at + k] = rng.integers(0, dn)
        at += k
    return out


@pytest.mark.parametrize("mode", MODES)
@pytest.mark.parametrize("np_dtype,kind_name", [(np.int64, "INT64"), (np.int32, "INT32"), (np.float64, "DOUBLE"), (np.float32, "FLOAT")])
def test_paths_agree_with_the_checker(mode, np_dtype, kind_name):
    P = pkg()
    kind = getattr(P.KIND, kind_name)
    rng = np.random.default_rng(hash((mode, kind_name)) % 2**32)
    cases = []
    for dn in (1, 3, 17, 100, 513, 1500, 10000, 16384, 70000):
        d = rng.integers(-2**31, 2**31 - 1, size=dn).astype(np_dtype)
        for n in (1, 9, 100, 2048, 2049, 5000, 40000):
            idx = rng.integers(0, dn, size=n, dtype=np.int32)
            if n >= 100:
                idx[20:70] = idx[20]
            cases.append((f"d{dn}-n{n}", d, idx, n))
        cases.append((f"d{dn}-runs40", d, _writer_runs(rng, 30000, dn, 40), 30000))
        cases.append((f"d{dn}-runs9", d, _writer_runs(rng, 30000, dn, 9), 30000))       # alternating short runs: > 100 runs per tile
        cases.append((f"d{dn}-runs3000", d, _writer_runs(rng, 50000, dn, 3000), 50000))  # tiles inside one repeated run
    d = np.arange(100, 120).astype(np_dtype)
    cases.append(("zero-extend", d, np.array([5, 6, 7], np.int32), 8))
    cases.append(("zero-extend-far", d, rng.integers(0, 20, size=3000, dtype=np.int32), 9000))
    cases.append(("cut", d, np.arange(16, dtype=np.int32), 11))
    cases.append(("cut-far", d, rng.integers(0, 20, size=9000, dtype=np.int32), 2047))
    cases.append(("all-zero-indexes", d, np.zeros(4000, np.int32), 4000))
    cases.append(("empty-stream", d, np.zeros(0, np.int32), 5))
    specs = []
    for k, (label, d, idx, nv) in enumerate(cases):
        specs.append(spec(P.ENC.RLE_DICTIONARY, kind, O.rle_dict_encode_int32(idx), num_values=nv, dictionary=d,
                          dst_cap=nv * d.dtype.itemsize, misalign=k % 16))
    outs = _decode(mode, specs)
    for (label, d, idx, nv), out in zip(cases, outs):
        want = _want(d, O.rle_dict_encode_int32(idx), nv)
        assert out.status == 0 and out.out_values == nv, (mode, label, out.status, out.arg)
        assert out.data.tobytes() == want.values.tobytes(), (mode, label)


@pytest.mark.parametrize("mode", MODES)
def test_many_pages_share_and_change_dictionaries(mode):
    # pages of one column chunk share a dictionary; a batch holds many chunks: the persistent
    # workgroups reload at every change, also in the middle of their stretch of tiles
    P = pkg()
    rng = np.random.default_rng(5)
    from parquet_go_amd.runtime import Batch, to_device, DeviceBuffer
    import torch
    dicts = [rng.integers(-2**62, 2**62, size=dn, dtype=np.int64) for dn in (700, 16384, 5, 1800, 3000, 40000)]
    d_dev = [to_device(d) for d in dicts]
    pages, keep, want = [], [], []
    for k in range(60):
        which = (k // 7) % len(dicts)
        d = dicts[which]
        n = int(rng.integers(1, 30000))
        idx = rng.integers(0, d.size, size=n, dtype=np.int32) if k % 3 else _writer_runs(rng, n, d.size, 25)
        enc = O.rle_dict_encode_int32(idx)
        src = to_device(enc, misalign=k % 16)
        dst = DeviceBuffer(n * 8 + 64, fill=0xA5)
        p = P.Page()
        p.src, p.src_len, p.dst, p.dst_cap = src.ptr, len(enc), dst.ptr, n * 8
        p.dict, p.dict_len, p.num_values = d_dev[which].ptr, d.size, n
        p.encoding, p.kind = P.ENC.RLE_DICTIONARY, P.KIND.INT64
        pages.append(p)
        keep.append((src, dst, n))
        want.append(d[idx])
    b = Batch(_ctx(mode), pages)
    stream = torch.cuda.current_stream(0).cuda_stream
    for _ in range(2):  # a batch may be launched again
        b.launch(stream)
    res = b.finish(stream)
    for k, (src, dst, n) in enumerate(keep):
        assert res[k].status == 0 and res[k].out_values == n, (mode, k)
        raw = dst.numpy()
        assert np.array_equal(raw[: n * 8].view(np.int64), want[k]), (mode, k)
        assert (raw[n * 8:] == 0xA5).all(), (mode, k)


@pytest.mark.parametrize("mode", MODES)
def test_index_out_of_bounds_and_capacity(mode):
    P = pkg()
    rng = np.random.default_rng(6)
    d = rng.integers(-2**62, 2**62, size=1000, dtype=np.int64)
    specs, wants = [], []
    # the first offender wins, wherever it lies: fast tiles, mixed tiles, repeated runs, the zero-extended tail
    for bad_at, n in ((3, 6), (2047, 5000), (2048, 5000), (4999, 5000), (12345, 40000)):
        idx = rng.integers(0, 1000, size=n, dtype=np.int32)
        idx[bad_at] = 1000 + int(rng.integers(0, 5))
        if n > bad_at + 50:
            idx[bad_at + 40] = 5000
        specs.append(spec(P.ENC.RLE_DICTIONARY, P.KIND.INT64, O.rle_dict_encode_int32(idx), num_values=n, dictionary=d, dst_cap=n * 8))
        wants.append((P.STATUS.INDEX_OUT_OF_BOUNDS, bad_at))
    runs = _writer_runs(rng, 30000, 1000, 30)
    runs[runs == runs[9000]] = 1000  # every repeat of one value, from its first occurrence on
    specs.append(spec(P.ENC.RLE_DICTIONARY, P.KIND.INT64, O.rle_dict_encode_int32(runs), num_values=30000, dictionary=d, dst_cap=30000 * 8))
    wants.append((P.STATUS.INDEX_OUT_OF_BOUNDS, int(np.nonzero(runs == 1000)[0][0])))
    # an empty dictionary: index 0 of the zero-extended part is outside it (position = end of the stream)
    specs.append(spec(P.ENC.RLE_DICTIONARY, P.KIND.INT64, O.rle_dict_encode_int32(np.zeros(0, np.int32)), num_values=10,
                      dictionary=np.zeros(0, np.int64), dst_cap=80))
    wants.append((P.STATUS.INDEX_OUT_OF_BOUNDS, 0))
    # destination too small: the size to come back with, and nothing written past the capacity
    idx = rng.integers(0, 1000, size=7000, dtype=np.int32)
    specs.append(spec(P.ENC.RLE_DICTIONARY, P.KIND.INT64, O.rle_dict_encode_int32(idx), num_values=7000, dictionary=d, dst_cap=7000 * 8 - 8))
    wants.append((P.STATUS.DST_TOO_SMALL, 7000 * 8))
    outs = _decode(mode, specs)
    for k, (out, (st, arg)) in enumerate(zip(outs, wants)):
        assert (out.status, out.arg) == (st, arg), (mode, k, out.status, out.arg)
    got = outs[-1]
    assert got.out_values == 6999 and np.array_equal(got.data.view(np.int64), d[idx[:6999]])


def test_width_byte_beyond_the_dictionary(monkeypatch):
    # a page whose width byte says 32 bits while the dictionary is small: every field is read in full
    P = pkg()
    d = np.arange(50, dtype=np.int64) * 3
    idx = np.array([1, 2, 3, 49, 0, 7, 7, 7] * 700, np.int32)
    n = idx.size
    for bw in (6, 13, 17, 25, 29, 32):
        packed = np.packbits(np.unpackbits(idx.astype("<u4").view(np.uint8).reshape(-1, 4), axis=1, bitorder="little")[:, :bw].reshape(-1),
                             bitorder="little").tobytes()
        enc = bytes([bw]) + bytes(_uvarint(((n // 8) << 1) | 1)) + packed
        for mode in MODES:
            for kind, dd in ((P.KIND.INT64, d), (P.KIND.INT32, d.astype(np.int32))):
                out = _decode(mode, [spec(P.ENC.RLE_DICTIONARY, kind, enc, num_values=n, dictionary=dd, dst_cap=n * dd.dtype.itemsize, misalign=3)])[0]
                assert out.status == 0 and np.array_equal(out.data.view(dd.dtype), dd[idx]), (mode, bw)


@pytest.mark.parametrize("mode,lo_bytes,hi_bytes", [("sliced", 4096, 8 * 2048), ("sliced3", 65536, 3 * 48000)])
@pytest.mark.parametrize("np_dtype,kind_name", [(np.int64, "INT64"), (np.int32, "INT32")])
def test_sliced_families(mode, lo_bytes, hi_bytes, np_dtype, kind_name):
    # every dictionary of the batch lies in the sliced range, so the launch is the sliced kernel's:
    # one to all slices in use, fast tiles, tiles of runs, cut and zero-extended pages, indexes
    # outside the dictionary in every kind of tile, a destination that is too small
    P = pkg()
    kind = getattr(P.KIND, kind_name)
    w = np.dtype(np_dtype).itemsize
    rng = np.random.default_rng(hash((mode, kind_name)) % 2**32)
    lo, hi = lo_bytes // w + 1, hi_bytes // w
    sizes = sorted({lo, lo + 1, (lo + hi) // 2, hi - 3, hi})
    cases = []
    for dn in sizes:
        d = rng.integers(-2**31, 2**31 - 1, size=dn).astype(np_dtype)
        for n in (1, 9, 100, 2048, 2049, 5000, 40000):
            idx = rng.integers(0, dn, size=n, dtype=np.int32)
            if n >= 100:
                idx[20:70] = idx[20]
            cases.append((f"d{dn}-n{n}", d, idx, n, None))
        cases.append((f"d{dn}-runs40", d, _writer_runs(rng, 30000, dn, 40), 30000, None))
        cases.append((f"d{dn}-runs9", d, _writer_runs(rng, 30000, dn, 9), 30000, None))
        cases.append((f"d{dn}-runs3000", d, _writer_runs(rng, 50000, dn, 3000), 50000, None))
        cases.append((f"d{dn}-zero-extend", d, rng.integers(0, dn, size=3000, dtype=np.int32), 9000, None))
        cases.append((f"d{dn}-cut", d, rng.integers(0, dn, size=9000, dtype=np.int32), 2047, None))
        cases.append((f"d{dn}-first-entries", d, rng.integers(0, 7, size=20000, dtype=np.int32), 20000, None))
        cases.append((f"d{dn}-last-entries", d, rng.integers(dn - 7, dn, size=20000, dtype=np.int32), 20000, None))
        for bad_at, n in ((3, 6), (2047, 5000), (2048, 5000), (4999, 5000), (12345, 40000)):
            idx = rng.integers(0, dn, size=n, dtype=np.int32)
            idx[bad_at] = dn + int(rng.integers(0, 5))
            if n > bad_at + 50:
                idx[bad_at + 40] = dn + 1000
            cases.append((f"d{dn}-bad{bad_at}", d, idx, n, bad_at))
    specs = []
    for k, (label, d, idx, nv, bad) in enumerate(cases):
        specs.append(spec(P.ENC.RLE_DICTIONARY, kind, O.rle_dict_encode_int32(idx), num_values=nv, dictionary=d,
                          dst_cap=nv * w, misalign=k % 16))
    d = rng.integers(-2**31, 2**31 - 1, size=hi).astype(np_dtype)
    idx_small = rng.integers(0, hi, size=7000, dtype=np.int32)
    specs.append(spec(P.ENC.RLE_DICTIONARY, kind, O.rle_dict_encode_int32(idx_small), num_values=7000, dictionary=d, dst_cap=7000 * w - w))
    outs = _decode(mode, specs)
    for (label, d_, idx, nv, bad), out in zip(cases, outs):
        if bad is not None:
            assert (out.status, out.arg) == (P.STATUS.INDEX_OUT_OF_BOUNDS, bad), (mode, label, out.status, out.arg)
            continue
        want = _want(d_, O.rle_dict_encode_int32(idx), nv)
        assert out.status == 0 and out.out_values == nv, (mode, label, out.status, out.arg)
        assert out.data.tobytes() == want.values.tobytes(), (mode, label)
    got = outs[-1]
    assert (got.status, got.arg) == (P.STATUS.DST_TOO_SMALL, 7000 * w)
    assert got.out_values == 6999 and np.array_equal(got.data.view(np_dtype), d[idx_small[:6999]])


def _uvarint(v):
    out = bytearray()
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)
    return out
