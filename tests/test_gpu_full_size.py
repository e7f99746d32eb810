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
