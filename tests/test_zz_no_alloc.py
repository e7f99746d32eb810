"""GPU: pqd_optional_expand_noalloc (the entry point whose workspace is a static device array,
for callers that cannot allocate) against pqd_optional_expand_device on the same inputs,
through the C ABI, bit exact, and both against the numpy restatement of page_optional.go:72-117
(oracle/levels_oracle.py); two calls in a row, the second smaller, so that what the first left in
the shared array would show."""
import ctypes as C

import numpy as np
import pytest

from oracle import levels_oracle as LO
from tests.gpu_util import ctx, pkg

pytestmark = pytest.mark.gpu


def _run(P, torch, noalloc, d, max_def, vals, n_values, width, offsets_mode):
    L = P.lib()
    dev = "cuda:0"
    n = d.size
    d_t = torch.from_numpy(d.copy()).to(dev) if n else torch.zeros(1, dtype=torch.uint8, device=dev)
    v_t = torch.from_numpy(vals.view(np.uint8).copy()).to(dev) if vals.size else torch.zeros(16, dtype=torch.uint8, device=dev)
    dst = torch.full(((n + (1 if offsets_mode else 0)) * width + 64,), 0xA5, dtype=torch.uint8, device=dev)
    val = torch.full((n + 64,), 0xA5, dtype=torch.uint8, device=dev)
    info = torch.zeros(2, dtype=torch.int64, device=dev)
    stream = C.c_void_p(torch.cuda.current_stream(0).cuda_stream)
    if noalloc:
        rc = L.pqd_optional_expand_noalloc(ctx().handle, d_t.data_ptr(), n, max_def, v_t.data_ptr(), n_values, width,
                                           1 if offsets_mode else 0, dst.data_ptr(), val.data_ptr(), info.data_ptr(), stream)
    else:
        ws_bytes = int(L.pqd_levels_workspace_size(n))
        ws = torch.full((ws_bytes // 8 + 2,), -1, dtype=torch.int64, device=dev)
        rc = L.pqd_optional_expand_device(ctx().handle, d_t.data_ptr(), n, max_def, v_t.data_ptr(), n_values, width,
                                          1 if offsets_mode else 0, dst.data_ptr(), val.data_ptr(), ws.data_ptr(), ws_bytes,
                                          info.data_ptr(), stream)
    assert rc == 0
    torch.cuda.synchronize()
    return dst.cpu().numpy(), val.cpu().numpy(), info.cpu().numpy().tolist()


def _case(rng, n, max_def, p_null, width, offsets_mode):
    d = np.where(rng.random(n) < p_null, rng.integers(0, max_def, size=n), max_def).astype(np.uint8)
    present = int((d == max_def).sum())
    if offsets_mode:
        vals = np.concatenate([[0], np.cumsum(rng.integers(0, 30, size=present))]).astype(np.uint32)
    else:
        vals = rng.integers(1, 256, size=present * width, dtype=np.uint8)
    return d, vals, present


def test_noalloc_matches_the_workspace_entry_point_twice_in_a_row():
    import torch
    P = pkg()
    rng = np.random.default_rng(77)
    # a large call, then smaller ones: the counts the large one left behind must not leak into them
    shapes = [(3_000_001, 1, 0.3, 8, False), (70_001, 3, 0.6, 8, False), (2049, 1, 0.5, 4, False), (1, 1, 0.0, 8, False),
              (0, 1, 0.0, 8, False), (500_000, 2, 0.4, 4, True), (4097, 2, 0.4, 4, True), (100_000, 1, 0.2, 12, False),
              (5000, 7, 0.9, 1, False), (6000, 1, 0.1, 3, False)]
    for n, max_def, p_null, width, offsets_mode in shapes:
        d, vals, present = _case(rng, n, max_def, p_null, width, offsets_mode)
        want = _run(P, torch, False, d, max_def, vals, present, width, offsets_mode)
        got = _run(P, torch, True, d, max_def, vals, present, width, offsets_mode)
        assert got[2] == want[2] == [present, 0], (n, width)
        assert np.array_equal(got[0], want[0]), (n, width, offsets_mode)
        assert np.array_equal(got[1], want[1]), (n, width, offsets_mode)
        # and what the reference's ReadValues leaves in the slots
        if offsets_mode:
            ref = LO.optional_offsets(d, max_def, vals)
            assert np.array_equal(got[0][: 4 * n].view(np.uint32), ref[:n]), (n, "offsets")
        else:
            slots, validity, cnt, eof = LO.optional_read_values(d, max_def, vals, width)
            assert cnt == present and not eof
            assert np.array_equal(got[0][: n * width], slots), (n, width)
            assert np.array_equal(got[1][:n], validity), (n, width)


def test_noalloc_runs_out_of_values_like_the_workspace_entry_point():
    import torch
    P = pkg()
    rng = np.random.default_rng(78)
    d, vals, present = _case(rng, 20_000, 1, 0.2, 8, False)
    short = vals[: (present - 100) * 8]
    want = _run(P, torch, False, d, 1, short, present - 100, 8, False)
    got = _run(P, torch, True, d, 1, short, present - 100, 8, False)
    assert got[2] == want[2] == [present, P.STATUS.UNEXPECTED_EOF]
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])


def test_noalloc_refuses_more_slots_than_its_array_covers():
    import torch
    P = pkg()
    L = P.lib()
    limit = ((1 << 20) // 8 - 1) * 2048
    one = torch.zeros(16, dtype=torch.uint8, device="cuda:0")
    info = torch.zeros(2, dtype=torch.int64, device="cuda:0")
    # refused on the host before anything is launched: the pointers are never followed
    rc = L.pqd_optional_expand_noalloc(ctx().handle, one.data_ptr(), limit + 1, 1, one.data_ptr(), 0, 8, 0, one.data_ptr(), None,
                                       info.data_ptr(), None)
    assert rc == P.STATUS.DST_TOO_SMALL
    rc = L.pqd_optional_expand_noalloc(ctx().handle, one.data_ptr(), 8, 1, one.data_ptr(), 0, 0, 0, one.data_ptr(), None,
                                       info.data_ptr(), None)
    assert rc == P.STATUS.INVALID_ARGUMENT
