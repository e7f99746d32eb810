"""GPU parity at the benchmark's full sizes (BASELINE.json configs[1] and configs[0]): 2048 DELTA
int64 pages of 131072 values, and 64 RLE_DICTIONARY pages of 262144 values over a 64 Ki-entry
dictionary.  The distinct pages are compared value for value with what went into the encoder
(and, on a sample, with the CPU checker); every replica is compared with its original on the
device, so all 2 GiB of output are covered without moving them to the host."""
import numpy as np
import pytest

import bench
from oracle import pqoracle as O
from tests.gpu_util import ctx, pkg

pytestmark = pytest.mark.gpu


def test_delta_int64_256mi_values():
    import torch
    P = pkg()
    stream = torch.cuda.current_stream(0).cuda_stream
    unique, pages, per_page = 64, 2048, 131072
    wl = bench.build_delta_workload(P, ctx(), 0, 0, pages, per_page, unique)
    wl.dst.fill_(-1)
    wl.launch(stream)
    wl.check(stream)  # every page reports OK
    res = wl.batch.finish(stream)
    assert all(res[i].out_values == per_page and res[i].out_bytes == per_page * 8 for i in range(pages))
    assert all(res[i].consumed == len(wl.enc_pages[i % unique]) for i in range(pages))
    out = wl.dst.view(pages, per_page)
    for k in range(unique):
        assert np.array_equal(out[k].cpu().numpy(), wl.expect[k]), f"page {k}"
    for k in (0, 17, 63):  # the generator's pages are what the reference's writer would emit
        assert wl.enc_pages[k] == O.delta_encode_int64(wl.expect[k])
        assert np.array_equal(O.delta_decode_int64(wl.enc_pages[k]).values, wl.expect[k])
    first = out[:unique]
    for r in range(1, pages // unique):
        assert torch.equal(out[r * unique:(r + 1) * unique], first), f"replica {r}"
    # size-independent property: differences of the decoded timestamps are the generator's deltas
    d = (out[:, 1:] - out[:, :-1])
    assert int(d.min()) >= 1_000_000 and int(d.max()) < 1_000_000 + 4096


def test_rle_dictionary_int64_16mi_values():
    import torch
    P = pkg()
    stream = torch.cuda.current_stream(0).cuda_stream
    wl = bench.build_rle_dict_workload(P, ctx(), 0, 0, 64, 262144, 65536)
    wl.dst.fill_(-1)
    wl.launch(stream)
    wl.check(stream)
    out = wl.dst.view(64, 262144)
    dictionary = torch.from_numpy(wl.dictionary).to(out.device)
    for k in range(64):
        idx = torch.from_numpy(wl.expect_idx[k].astype(np.int64)).to(out.device)
        assert torch.equal(out[k], dictionary[idx]), f"page {k}"
    k = 5
    got_idx = O.rle_dict_decode_int32(wl.enc_pages[k]).values
    assert np.array_equal(got_idx, wl.expect_idx[k])
    assert np.array_equal(out[k].cpu().numpy(), O.dict_lookup(wl.dictionary, got_idx).values.view(np.int64))


def _batch(P, torch, specs):
    """specs: (src tensor, src offset, src_len, dst tensor, dst offset, dst_cap, offsets tensor | None, offsets entry, offsets_cap,
    encoding, kind, bit_width, num_values)."""
    pages = []
    for s, s_at, s_len, d, d_at, d_cap, o, o_at, o_cap, enc, kind, bw, nv in specs:
        p = P.Page()
        p.src, p.src_len = s.data_ptr() + s_at, s_len
        p.dst, p.dst_cap = d.data_ptr() + d_at, d_cap
        if o is not None:
            p.offsets, p.offsets_cap = o.data_ptr() + 4 * o_at, o_cap
        p.num_values = nv
        p.encoding, p.kind, p.bit_width = enc, kind, bw
        pages.append(p)
    return P.Batch(ctx(), pages)


def test_plain_byte_arrays_512_pages_of_a_megabyte():
    """PLAIN byte arrays, 512 pages of 1 MiB each (32 distinct ones, every tile-chaining path of the one-pass kernel at
    the size a reader's pages have): the distinct pages value for value against what was encoded, every replica against its
    original on the device, and size-independent properties over all of them -- offsets rise, the last offset is the
    page's bytes less its length fields, the bytes' sum is the values' sum."""
    import torch
    P = pkg()
    rng = np.random.default_rng(61)
    dev = torch.device("cuda:0")
    unique, pages = 32, 512
    encs, vals = [], []
    for k in range(unique):
        if k % 4 == 3:   # long binary values among short ones: records that span tiles, zero bytes inside values
            v = [bytes(rng.integers(0, 256, size=int(n), dtype=np.uint8)) for n in rng.choice([0, 3, 40, 900, 40000], size=2000, p=[0.2, 0.3, 0.4, 0.09, 0.01])]
        else:
            v = [b"https://example.org/%d/%08d" % (k, int(x)) for x in rng.integers(0, 10**8, size=26000)]
        enc = bytes(O.plain_encode_byte_array(v))[: 1 << 20]
        want = O.plain_decode_byte_array(enc)  # (the cut usually lands inside a record: the page then ends in an error, like any page cut short)
        encs.append(enc)
        vals.append(want)
    slot = (1 << 20) + 256
    host = np.zeros(unique * slot, np.uint8)
    for k, e in enumerate(encs):
        host[k * slot: k * slot + len(e)] = np.frombuffer(e, np.uint8)
    src = torch.from_numpy(host).to(dev).repeat(pages // unique)
    dst = torch.zeros(pages * slot, dtype=torch.uint8, device=dev)
    ocap = 1 << 18
    off = torch.zeros(pages * ocap, dtype=torch.int32, device=dev)
    batch = _batch(P, torch, [(src, i * slot, len(encs[i % unique]), dst, i * slot, slot, off, i * ocap, ocap, P.ENC.PLAIN, P.KIND.BYTE_ARRAY, 0, -1)
                              for i in range(pages)])
    stream = torch.cuda.current_stream(0).cuda_stream
    batch.launch(stream)
    res = batch.finish(stream)
    d2, o2 = dst.view(pages, slot), off.view(pages, ocap)
    for i in range(pages):
        w = vals[i % unique]
        assert (res[i].status, res[i].status_arg, res[i].out_values) == (w.status, w.arg, w.n_values), i
    for k in range(unique):
        w = vals[k]
        nb = int(w.values.size)
        assert d2[k, :nb].cpu().numpy().tobytes() == w.values.tobytes(), k
        assert np.array_equal(o2[k, : w.n_values].cpu().numpy().view(np.uint32), w.offsets[: w.n_values]), k
    for r in range(1, pages // unique):
        assert torch.equal(d2[r * unique:(r + 1) * unique], d2[:unique]) and torch.equal(o2[r * unique:(r + 1) * unique], o2[:unique]), r
    for k in range(unique):
        w = vals[k]
        o = o2[k, : w.n_values].to(torch.int64)
        assert bool((o[1:] >= o[:-1]).all()), k
        total = int(w.values.size)
        assert int(d2[k, :total].to(torch.int64).sum()) == int(w.values.astype(np.int64).sum()), k


def test_level_pages_512_of_128_ki_and_2048_of_32_ki():
    """Level pages at the two batch sizes that pick the two workgroup sizes of the page kernel (1024 threads up to 512
    pages, 256 beyond): distinct pages value for value against what was encoded, replicas on the device, and the
    count of levels equal to the maximum against numpy's."""
    import torch
    P = pkg()
    rng = np.random.default_rng(62)
    dev = torch.device("cuda:0")
    stream = torch.cuda.current_stream(0).cuda_stream
    for pages, n, bw in ((512, 131072, 3), (2048, 32768, 1)):
        unique = 32
        lv = []
        for k in range(unique):
            v = np.empty(n, np.uint8)
            at = 0
            while at < n:
                m = min(n - at, int(rng.geometric(1 / 300)) if rng.random() < 0.5 else int(rng.integers(1, 40)))
                v[at: at + m] = rng.integers(0, 1 << bw, size=(1 if rng.random() < 0.6 else m))
                at += m
            lv.append(v)
        encs = [bytes(O.rle_encode_levels(v, bw)) if k % 2 else bytes(O.rle_encode_levels_mr(v, bw)) for k, v in enumerate(lv)]
        slot = (max(map(len, encs)) + 255) // 256 * 256 + 256
        host = np.zeros(unique * slot, np.uint8)
        for k, e in enumerate(encs):
            host[k * slot: k * slot + len(e)] = np.frombuffer(e, np.uint8)
        src = torch.from_numpy(host).to(dev).repeat(pages // unique)
        dst = torch.full((pages * (n + 256),), 0xEE, dtype=torch.uint8, device=dev)
        batch = _batch(P, torch, [(src, i * slot, len(encs[i % unique]), dst, i * (n + 256), n, None, 0, 0, P.ENC.RLE, P.KIND.LEVELS, bw, n)  # (num_values: the page header's count, as the reader passes it)
                                  for i in range(pages)])
        batch.launch(stream)
        res = batch.finish(stream)
        assert all(res[i].status == 0 and res[i].out_values == n for i in range(pages)), (pages, n)
        out = dst.view(pages, n + 256)
        for k in range(unique):
            assert np.array_equal(out[k, :n].cpu().numpy(), lv[k]), (pages, k)
        assert bool((out[:, n:] == 0xEE).all())  # nothing written behind a page's levels
        for r in range(1, pages // unique):
            assert torch.equal(out[r * unique:(r + 1) * unique, :n], out[:unique, :n]), (pages, r)
        top = (1 << bw) - 1
        assert int((out[:, :n] == top).sum()) == (pages // unique) * sum(int((v == top).sum()) for v in lv)
