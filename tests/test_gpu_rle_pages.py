"""GPU: repetition / definition level pages (RLE hybrid, rle.go:295-350) through both of their
paths -- a workgroup per page that follows the run headers and expands the runs in one launch
(the default; with 256 and with 1024 threads to a page, as large and small batches get them: the
wider one follows the headers eight at a time), and the index + expand kernels every other RLE
page takes (PQD_RLE_PAGE=0) -- against the CPU checker: same values, statuses, status arguments and value counts.  The shapes
aim at the page kernel's seams: pages longer than its window, bit-packed runs longer than the
window, more runs inside a window than its table lists, varints it leaves to the retry kernel,
damaged pages, a destination that is too small, pages cut to num_values."""
import os

import numpy as np
import pytest

from oracle import pqoracle as O
from tests.gpu_util import assert_same, pkg, spec
from tests.test_gpu_rle import _levels_cases, _runs_of

pytestmark = pytest.mark.gpu

_ctxs = {}


def _ctx(mode):
    if mode not in _ctxs:
        want = {"PQD_RLE_PAGE": "2" if mode.startswith("page") else "0", "PQD_RLE_PAGE_THREADS": mode[4:] if mode.startswith("page") else None}
        old = {k: os.environ.get(k) for k in want}
        for k, v in want.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v
        try:
            _ctxs[mode] = pkg().Context(0)
        finally:
            for k, v in old.items():
                if v is None:
                    os.environ.pop(k, None)
                else:
                    os.environ[k] = v
    return _ctxs[mode]


def _decode(mode, specs, grow=True):
    from parquet_go_amd.runtime import decode_pages
    return decode_pages(_ctx(mode), specs, grow=grow)


def _uvarint(v):
    out = bytearray()
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


MODES = ("page256", "page1024", "kernels")


@pytest.mark.parametrize("mode", MODES)
def test_level_tables_and_mixed_runs(mode):
    P = pkg()
    cases = _levels_cases(np.random.default_rng(3))
    outs = _decode(mode, [spec(P.ENC.RLE, P.KIND.LEVELS, enc, bit_width=bw, misalign=k % 16) for k, (_, bw, enc) in enumerate(cases)])
    for (label, bw, enc), out in zip(cases, outs):
        assert_same(out, O.rle_decode_levels(enc, bw), np.uint8, f"{mode}-{label}")


def _seam_cases(rng):
    cases = []  # (label, bit width, bytes)
    for bw, short, long_ in ((1, 2, 40), (2, 1.2, 9), (3, 3, 60), (5, 8, 3000), (7, 1.5, 20), (8, 4, 100)):
        v = _runs_of(rng, 300_000, 1 << bw, short, long_).astype(np.uint8)
        cases.append((f"go-bw{bw}", bw, O.rle_encode_levels(v, bw)))
        cases.append((f"mr-bw{bw}", bw, O.rle_encode_levels_mr(v, bw)))
    for bw in (1, 3, 8):  # no runs at all: one bit-packed run of many windows
        v = rng.integers(0, 1 << bw, size=200_000, dtype=np.uint8)
        cases.append((f"one-bit-packed-run-bw{bw}", bw, O.rle_encode_levels(v, bw)))
    # runs of eight equal values, two bytes each: four thousand to a window
    cases.append(("short-runs", 2, b"".join(bytes([8 << 1, k & 3]) for k in range(30_000))))
    # bit-packed runs of exactly the window, and one byte more or less, between short runs
    for groups in (8192 // 3, 8192 // 3 + 1, 3 * 8192 // 3 - 1, 20_000):
        body = bytes(rng.integers(0, 256, size=3 * groups, dtype=np.uint8))
        cases.append((f"bit-packed-{groups}", 3, b"\x06\x05" * 5 + _uvarint((groups << 1) | 1) + body + b"\x08\x02" * 700))
    # empty runs, scattered
    parts = []
    for k in range(9000):
        parts.append(b"\x00" if k % 3 == 0 else (bytes([2 * (1 + k % 50)]) + bytes([k % 4])))
        if k % 500 == 499:
            parts.append(bytes([(8 << 1) | 1]) + bytes(rng.integers(0, 256, size=8 * 2, dtype=np.uint8)))
    cases.append(("empty-runs", 2, b"".join(parts)))
    # counts that need a varint of four or five bytes (left to the retry kernel), between ordinary runs
    cases.append(("long-varints", 1, b"\x06\x01" * 100 + _uvarint(3_000_000 << 1) + b"\x01" + b"\x08\x00" * 50 + _uvarint((300_000 << 1) | 1) + bytes(300_000)))
    # damaged pages
    good = O.rle_encode_levels_mr(_runs_of(rng, 200_000, 8, 2, 30).astype(np.uint8), 3)
    cases.append(("truncated", 3, good[: len(good) - 1]))
    cases.append(("truncated-early", 3, good[:9]))
    cases.append(("bad-varint", 3, good[:30_000] + b"\x80" * 11 + good[30_000:]))
    cases.append(("bit-packed-overruns", 3, good[:25_000] + b"\xff\xff\x03" + good[25_000:40_000]))
    cases.append(("no-value-byte", 3, good + b"\x10"))
    cases.append(("width-0", 0, b"\x10" * 40))
    cases.append(("width-9", 9, b"\x10\x01"))
    cases.append(("empty", 4, b""))
    return cases


@pytest.mark.parametrize("mode", MODES)
def test_seams(mode):
    P = pkg()
    cases = _seam_cases(np.random.default_rng(17))
    outs = _decode(mode, [spec(P.ENC.RLE, P.KIND.LEVELS, enc, bit_width=bw, misalign=(5 * k) % 16) for k, (_, bw, enc) in enumerate(cases)])
    for (label, bw, enc), out in zip(cases, outs):
        want = O.rle_decode_levels(enc, bw)
        assert_same(out, want, np.uint8, f"{mode}-{label}", values=want.status == 0)


@pytest.mark.parametrize("mode", MODES)
def test_too_many_values(mode):
    # the one limit the reference does not have: a page decodes to at most 2**31 - 1 values here
    P = pkg()
    huge = bytes([0xfe, 0xff, 0xff, 0xff, 0x07, 0x01])  # one run of 2**30 - 1 values
    small = bytes([0xfe, 0xff, 0x7f, 0x01])             # 2**20 - 1 values, a three-byte varint
    outs = _decode(mode, [spec(P.ENC.RLE, P.KIND.LEVELS, b"\x06\x01" * 9000 + huge * 3, bit_width=1, dst_cap=1 << 16),
                          spec(P.ENC.RLE, P.KIND.LEVELS, small * 2100, bit_width=1, dst_cap=1 << 16)], grow=False)
    assert outs[0].status == P.STATUS.RUN_TOO_LARGE
    assert outs[1].status == P.STATUS.RUN_TOO_LARGE


@pytest.mark.parametrize("mode", MODES)
def test_cut_and_capacity(mode):
    P = pkg()
    rng = np.random.default_rng(23)
    v = _runs_of(rng, 150_000, 4, 3, 500).astype(np.uint8)
    enc = O.rle_encode_levels(v, 2)
    full = O.rle_decode_levels(enc, 2).values
    specs, wants = [], []
    for nv in (0, 1, 7, 4096, 70_001, full.size - 1, full.size, full.size + 5):  # column.go:940-945: the caller keeps num_values levels
        specs.append(spec(P.ENC.RLE, P.KIND.LEVELS, enc, bit_width=2, num_values=nv, dst_cap=nv + 3))
        wants.append((0, min(nv, full.size)))
    for cap in (1, 5, 8191, 100_000):
        specs.append(spec(P.ENC.RLE, P.KIND.LEVELS, enc, bit_width=2, dst_cap=cap))
        wants.append((P.STATUS.DST_TOO_SMALL, cap))
    specs.append(spec(P.ENC.RLE, P.KIND.LEVELS, enc, bit_width=2, num_values=90_000, dst_cap=60_000))
    wants.append((P.STATUS.DST_TOO_SMALL, 60_000))
    outs = _decode(mode, specs, grow=False)
    for k, (out, (st, n)) in enumerate(zip(outs, wants)):
        assert out.status == st and out.out_values == n, (mode, k, out.status, out.arg, out.out_values)
        assert np.array_equal(out.data[:n], full[:n]), (mode, k)
        if st != 0:
            assert out.arg == (90_000 if k == len(wants) - 1 else full.size), (mode, k, out.arg)


@pytest.mark.parametrize("mode", ["page256", "page1024", "page"])
def test_many_pages_in_a_batch(mode):
    # a row group's worth of level pages: the page kernel's grid is the family's item list ("page": the
    # workgroup sized to the batch, as without any setting)
    P = pkg()
    rng = np.random.default_rng(29)
    specs, wants = [], []
    for k in range(300):
        bw = 1 + k % 4
        v = _runs_of(rng, int(rng.integers(1, 40_000)), 1 << bw, 2 + k % 5, 30 * (1 + k % 9)).astype(np.uint8)
        enc = O.rle_encode_levels(v, bw) if k % 2 else O.rle_encode_levels_mr(v, bw)
        specs.append(spec(P.ENC.RLE, P.KIND.LEVELS, enc, bit_width=bw, misalign=k % 16))
        wants.append(O.rle_decode_levels(enc, bw))
    for k, (out, want) in enumerate(zip(_decode(mode, specs), wants)):
        assert_same(out, want, np.uint8, f"page {k}")


def test_more_headers_in_a_window_than_a_round_lists():
    # runs of two values are two bytes each: a window of 4096 bytes holds 2048 headers, a round lists 256 or 1024
    # of them, and the wide workgroups' walk has to stop at a multiple of its eight-header steps
    P = pkg()
    rng = np.random.default_rng(31)
    for mode in ("page256", "page1024"):
        specs, wants = [], []
        for n_runs in (255, 256, 257, 1023, 1024, 1025, 1031, 5000):
            vals = rng.integers(0, 4, size=n_runs)
            vals[1:][vals[1:] == vals[:-1]] ^= 1  # neighbouring runs differ, or the checker's encoder would merge them
            counts = rng.integers(1, 8, size=n_runs)
            enc = b"".join(_uvarint(int(c) << 1) + bytes([int(v)]) for c, v in zip(counts, vals))
            specs.append(spec(P.ENC.RLE, P.KIND.LEVELS, enc, bit_width=2))
            wants.append(O.rle_decode_levels(enc, 2))
        for k, (out, want) in enumerate(zip(_decode(mode, specs), wants)):
            assert_same(out, want, np.uint8, f"{mode} case {k}")


@pytest.mark.parametrize("mode", MODES)
def test_destinations_at_every_alignment(mode):
    # the page kernel writes sixteen bytes per lane on 16-byte boundaries of the destination: the ragged
    # ends of every stretch, at every alignment, with canaries on both sides
    import torch
    from parquet_go_amd.runtime import Batch, DeviceBuffer, to_device
    P = pkg()
    rng = np.random.default_rng(31)
    pages, keep = [], []
    for k in range(48):
        bw = 1 + k % 8
        n = int(rng.integers(1, 9000)) if k % 3 else int(rng.integers(20_000, 70_000))
        v = _runs_of(rng, n, 1 << bw, 2 + k % 4, 15 * (1 + k % 7)).astype(np.uint8)
        enc = O.rle_encode_levels(v, bw) if k % 2 else O.rle_encode_levels_mr(v, bw)
        want = O.rle_decode_levels(enc, bw).values
        src = to_device(enc, misalign=(7 * k) % 16)
        buf = DeviceBuffer(want.size + 64, misalign=0, fill=0xA5)
        a = 16 + k % 16  # destination starts a bytes into the buffer
        p = P.Page()
        p.src, p.src_len, p.dst, p.dst_cap = src.ptr, len(enc), buf.ptr + a, want.size
        p.bit_width, p.num_values = bw, -1
        p.encoding, p.kind = P.ENC.RLE, P.KIND.LEVELS
        pages.append(p)
        keep.append((src, buf, a, want))
    b = Batch(_ctx(mode), pages)
    stream = torch.cuda.current_stream(0).cuda_stream
    b.launch(stream)
    res = b.finish(stream)
    for k, (src, buf, a, want) in enumerate(keep):
        raw = buf.numpy()
        assert res[k].status == 0 and res[k].out_values == want.size, (mode, k, res[k].status)
        assert np.array_equal(raw[a: a + want.size], want), (mode, k)
        assert (raw[:a] == 0xA5).all() and (raw[a + want.size:] == 0xA5).all(), (mode, k)


def test_long_pages_of_few_bytes_shared_by_several_workgroups():
    """Pages of a million levels and more in long runs (a few kilobytes each) in a batch of few pages: the page kernel
    takes them whatever their length, and several workgroups share a page's expansion.  Whole pages, a page cut to
    num_values, a destination that is too small, a damaged page (the retry kernel's), every destination alignment of
    the first few."""
    from tests.gpu_util import ctx
    P = pkg()
    from parquet_go_amd.runtime import decode_pages
    rng = np.random.default_rng(1203)
    cases = []
    for k, (bw, n, short, long_) in enumerate(((1, 1 << 20, 6, 6000), (3, 1_500_000, 9, 4000), (2, 1_048_577, 30, 9000), (1, 3 << 20, 500, 20000),
                                              (8, 700_001, 8, 5000))):
        v = _runs_of(rng, n, 1 << bw, short, long_).astype(np.uint8)
        enc = O.rle_encode_levels(v, bw) if k % 2 == 0 else O.rle_encode_levels_mr(v, bw)
        assert len(enc) <= 32768 or bw == 8, (k, len(enc))
        cases.append((f"long-bw{bw}", bw, enc, -1, n + 64))
    bw, enc = cases[0][1], cases[0][2]
    cases.append(("cut-to-num-values", bw, enc, 777_777, 800_000))
    cases.append(("destination-too-small", bw, enc, -1, 500_000))
    broken = bytearray(cases[1][2])
    broken[len(broken) // 2] ^= 0xff
    cases.append(("damaged", cases[1][1], bytes(broken), -1, 1_600_000))
    specs = []
    for k, (label, bw, enc, nv, cap) in enumerate(cases):
        kw = dict(bit_width=bw, misalign=k % 16)
        if nv >= 0:
            kw["num_values"] = nv
        if cap is not None:
            kw["dst_cap"] = cap
        specs.append(spec(P.ENC.RLE, P.KIND.LEVELS, enc, **kw))
    outs = decode_pages(ctx(), specs, grow=False)
    for (label, bw, enc, nv, cap), out in zip(cases, outs):
        want = O.rle_decode_levels(enc, bw)
        if want.status != 0:
            assert_same(out, want, np.uint8, label, values=False)
            continue
        full = want.values
        n = full.size if nv < 0 else min(nv, full.size)
        if cap is not None and n > cap:
            assert out.status == P.STATUS.DST_TOO_SMALL and out.arg == n and out.out_values == cap, (label, out.status, out.arg, out.out_values)
            n = cap
        else:
            assert out.status == 0 and out.out_values == n, (label, out.status, out.arg, out.out_values)
        assert np.array_equal(out.data[:n], full[:n]), label
