"""GPU: PLAIN byte-array pages through the many-workgroups-per-page path and through the
one-wavefront walker, against the CPU checker (plain.Encoding.DecodeByteArray,
encoding/plain/plain.go:76-94).  The first path guesses where records start and verifies the
guesses; the pages here are chosen to make guesses go wrong: values that look like length
fields, values that are pages themselves, zeros, phase shifts, records longer than a tile."""
import os

import numpy as np
import pytest

from oracle import pqoracle as O
from tests.gpu_util import pkg, spec

pytestmark = pytest.mark.gpu

_ctxs = {}


def _ctx(tiled):
    if tiled not in _ctxs:
        old = os.environ.get("PQD_PLAIN_TILED")
        os.environ["PQD_PLAIN_TILED"] = "1" if tiled else "0"
        try:
            _ctxs[tiled] = pkg().Context(0)
        finally:
            if old is None:
                os.environ.pop("PQD_PLAIN_TILED", None)
            else:
                os.environ["PQD_PLAIN_TILED"] = old
    return _ctxs[tiled]


def _decode(tiled, specs):
    from parquet_go_amd.runtime import decode_pages
    return decode_pages(_ctx(tiled), specs, grow=False)


def _page(values):
    return bytes(O.plain_encode_byte_array(values))


def _pages(rng):
    def rand(n, lo, hi):
        return [bytes(rng.integers(0, 256, size=int(k), dtype=np.uint8)) for k in rng.integers(lo, hi, size=n)]

    out = []
    out.append(("urls", [b"https://www.example.com/catalog/items/%06d" % int(v) for v in rng.integers(0, 10**6, size=40000)]))
    out.append(("random-short", rand(60000, 0, 22)))
    out.append(("random-mixed", rand(20000, 0, 300)))
    out.append(("all-empty", [b""] * 50000))
    out.append(("empty-after-odd", [b"x"] + [b""] * 30000 + [b"abcde"] + [b""] * 30000))          # the chain leaves the 4-byte grid
    out.append(("zeros", [bytes(int(k)) for k in rng.integers(0, 40, size=30000)]))                # every position looks like an empty value
    out.append(("small-ints", [np.array(rng.integers(0, 9, size=int(k)), "<u4").tobytes() for k in rng.integers(0, 12, size=20000)]))
    inner = _page([b"ab", b"", b"cdefg"] * 7)
    out.append(("pages-inside", [inner[int(a):] for a in rng.integers(0, 12, size=9000)]))          # values that are chains themselves
    out.append(("one-byte", [b"z"] * 100000))
    out.append(("long", rand(30, 5000, 70000) + [b"x"] + rand(4, 100000, 200000) + [b"tail"]))     # records longer than tiles
    out.append(("long-then-short", rand(3, 40000, 50000) + rand(5000, 0, 30) + rand(2, 17000, 33000) + rand(5000, 40, 60)))
    out.append(("tile-edges", [b"a" * 16380, b"b" * 16376, b"", b"c" * 3, b"d" * 16383, b"e" * 16384, b"f"]))
    out.append(("none", []))                                                                       # an empty page among the others
    out.append(("exactly-a-tile", [b"q" * 12] * 1024))                                              # 16 bytes a record: 16384 bytes
    out.append(("a-tile-and-a-byte", [b"q" * 12] * 1023 + [b"q" * 13]))
    out.append(("single", [b"only"]))
    out.append(("single-long", [bytes(rng.integers(0, 256, size=100001, dtype=np.uint8))]))
    return out


@pytest.mark.parametrize("tiled", [True, False])
def test_pages_agree_with_the_checker(tiled):
    P = pkg()
    rng = np.random.default_rng(21)
    cases = _pages(rng)
    specs = []
    for k, (label, values) in enumerate(cases):
        enc = _page(values)
        specs.append(spec(P.ENC.PLAIN, P.KIND.BYTE_ARRAY, enc, misalign=k % 16, dst_cap=len(enc), offsets_cap=len(values) + 1))
    outs = _decode(tiled, specs)
    for (label, values), out in zip(cases, outs):
        assert out.status == 0 and out.out_values == len(values), (label, out.status, out.arg, out.out_values)
        assert out.data.tobytes() == b"".join(values), label
        lens = np.fromiter(map(len, values), np.int64, len(values))
        assert np.array_equal(out.offsets[: len(values) + 1], np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32)), label
        assert out.consumed == len(_page(values)), label


@pytest.mark.parametrize("tiled", [True, False])
def test_damaged_pages_and_small_destinations(tiled):
    P = pkg()
    rng = np.random.default_rng(22)
    values = [bytes(rng.integers(0, 256, size=int(k), dtype=np.uint8)) for k in rng.integers(0, 90, size=3000)]
    page = _page(values)
    specs, wants = [], []
    # cut anywhere: inside a length field, inside a value, right behind a record
    cuts = [1, 2, 3, 4, 5, 6] + [int(c) for c in rng.integers(7, len(page), size=40)] + [len(page) - 1]
    for cut in cuts:
        specs.append(spec(P.ENC.PLAIN, P.KIND.BYTE_ARRAY, page[:cut], misalign=cut % 16, dst_cap=len(page), offsets_cap=len(values) + 1))
        wants.append(O.plain_decode_byte_array(page[:cut]))
    # a length field that points past the end, early and late; trailing bytes too few for a length field
    for at in (0, 1000, len(values) - 1):
        pos = sum(4 + len(v) for v in values[:at])
        bad = bytearray(page)
        bad[pos: pos + 4] = (len(page)).to_bytes(4, "little")
        specs.append(spec(P.ENC.PLAIN, P.KIND.BYTE_ARRAY, bytes(bad), dst_cap=len(page), offsets_cap=len(values) + 1))
        wants.append(O.plain_decode_byte_array(bytes(bad)))
    for extra in (b"\x00", b"\x01\x00", b"\x00\x00\x00"):
        specs.append(spec(P.ENC.PLAIN, P.KIND.BYTE_ARRAY, page + extra, dst_cap=len(page), offsets_cap=len(values) + 2))
        wants.append(O.plain_decode_byte_array(page + extra))
    outs = _decode(tiled, specs)
    for k, (out, want) in enumerate(zip(outs, wants)):
        assert (out.status, out.arg) == (want.status, want.arg), (k, out.status, out.arg, want.status, want.arg)
        assert out.out_values == want.n_values and out.data.tobytes() == want.values.tobytes(), k
        assert np.array_equal(out.offsets[: want.n_values], want.offsets[: want.n_values]), k
    # destinations and offset buffers that are too small: the record that does not fit decides (plain.go walks in order)
    total = sum(map(len, values))
    specs = []
    for cap, ocap in ((1, len(values) + 1), (total - 1, len(values) + 1), (total // 2, len(values) + 1), (17, len(values) + 1),
                      (total, len(values)), (total, 1), (total, 2), (total, 1500), (total // 3, 2000), (total // 3, 500), (total, len(values) + 1)):
        specs.append(spec(P.ENC.PLAIN, P.KIND.BYTE_ARRAY, page, dst_cap=cap, offsets_cap=ocap, misalign=cap % 16))
    outs = _decode(tiled, specs)
    for k, (out, s) in enumerate(zip(outs, specs)):
        st, arg, nv, nb = _walk(values, s.dst_cap, s.offsets_cap)
        assert (out.status, out.arg, out.out_values, out.out_bytes) == (st, arg, nv, nb), (k, out.status, out.arg, out.out_values, out.out_bytes, st, arg, nv, nb)
        assert out.data.tobytes()[:nb] == b"".join(values)[:nb], k


def _walk(values, cap, ocap):
    """plain.go:76-94 with a destination of `cap` bytes and `ocap` offsets (the checker's rule, record by record)."""
    out = nv = 0
    for v in values:
        if nv + 2 > ocap:
            return 18, nv + 2, nv, out
        if out + len(v) > cap:
            return 16, out + len(v), nv, out
        out += len(v)
        nv += 1
    if nv + 1 > ocap:
        return 18, nv + 1, nv, out
    return 0, 0, nv, out


def test_tiled_decode_is_repeatable():
    # tiles that the quick look at zero bytes settles and tiles that need the walk, side by side in one batch,
    # several times over: the two ways of resolving a tile share LDS words
    P = pkg()
    rng = np.random.default_rng(22)
    cases = _pages(rng)
    specs = [spec(P.ENC.PLAIN, P.KIND.BYTE_ARRAY, _page(values), misalign=k % 16, dst_cap=len(_page(values)), offsets_cap=len(values) + 1)
             for k, (label, values) in enumerate(cases)]
    for trip in range(6):
        for (label, values), out in zip(cases, _decode(True, specs)):
            assert out.status == 0 and out.out_values == len(values), (trip, label)
            assert out.data.tobytes() == b"".join(values), (trip, label)


def test_one_batch_launched_again_and_again():
    # the tiles of a page find their places through state words in the batch's scratch that every launch has to
    # leave as it found them: the same batch object, launched four times into cleared destinations, one of its
    # pages damaged half way (its chain ends there, and the tiles behind it have nothing to copy)
    import torch
    P = pkg()
    rng = np.random.default_rng(23)
    values = [bytes(rng.integers(97, 123, size=int(k), dtype=np.uint8)) for k in rng.integers(0, 120, size=9000)]
    good = _page(values)
    bad = bytearray(good)
    at = sum(4 + len(v) for v in values[:5000])
    bad[at: at + 4] = (len(good)).to_bytes(4, "little")
    encs = [good, bytes(bad), good[: len(good) // 3 * 2 + 1]]
    wants = [O.plain_decode_byte_array(e) for e in encs]
    dev = torch.device("cuda:0")
    src = [torch.from_numpy(np.frombuffer(e, np.uint8).copy()).to(dev) for e in encs]
    dst = [torch.zeros(len(good) + 64, dtype=torch.uint8, device=dev) for _ in encs]
    off = [torch.zeros(len(values) + 8, dtype=torch.int32, device=dev) for _ in encs]
    pages = []
    for k, e in enumerate(encs):
        p = P.Page()
        p.src, p.src_len = src[k].data_ptr(), len(e)
        p.dst, p.dst_cap = dst[k].data_ptr(), len(good)
        p.offsets, p.offsets_cap = off[k].data_ptr(), len(values) + 1
        p.num_values = -1
        p.encoding, p.kind, p.bit_width = P.ENC.PLAIN, P.KIND.BYTE_ARRAY, 0
        pages.append(p)
    batch = P.Batch(_ctx(True), pages)
    stream = torch.cuda.current_stream().cuda_stream
    for trip in range(4):
        for d, o in zip(dst, off):
            d.zero_(); o.zero_()
        torch.cuda.synchronize()
        batch.launch(stream)
        res = batch.finish(stream)
        for k, want in enumerate(wants):
            assert (res[k].status, res[k].status_arg, res[k].out_values) == (want.status, want.arg, want.n_values), (trip, k)
            nb = int(want.values.size)
            assert dst[k][:nb].cpu().numpy().tobytes() == want.values.tobytes(), (trip, k)
            assert np.array_equal(off[k][: want.n_values].cpu().numpy().view(np.uint32), want.offsets[: want.n_values]), (trip, k)
