"""GPU: one rank's share of a small wide table (the workload of `bench.py --config wide`:
column shapes cycling RLE_DICTIONARY int64, DELTA_BINARY_PACKED int64, an optional int32 column
with RLE definition levels, BYTE_STREAM_SPLIT double, DELTA_LENGTH_BYTE_ARRAY and
DELTA_BYTE_ARRAY strings), decoded as one batch and compared page by page with the CPU
checker's decode of the same page bytes.  Bit exact."""
import numpy as np
import pytest

import bench_wide as W
from oracle import pqoracle as O
from tests.gpu_util import ctx, pkg

pytestmark = pytest.mark.gpu


def _checker(P, part):
    """The checker's decode of one work item: (value bytes, offsets or None)."""
    E, K = P.ENC, P.KIND
    if part.encoding == E.RLE_DICTIONARY:
        idx = O.indexed_page_fixup(O.rle_dict_decode_int32(part.data).values, part.num_values)
        return O.dict_lookup(part.dictionary, idx).values, None
    if part.encoding == E.DELTA_BINARY_PACKED:
        return O.delta_decode_int64(part.data).values.view(np.uint8), None
    if part.encoding == E.RLE:
        return O.rle_decode_levels(part.data, part.bit_width).values[: part.num_values], None
    if part.encoding == E.BYTE_STREAM_SPLIT:
        return O.bss_decode(part.data, 8).values, None
    r = O.dlba_decode(part.data) if part.encoding == E.DELTA_LENGTH_BYTE_ARRAY else O.dba_decode(part.data)
    return r.values, r.offsets


@pytest.mark.parametrize("rank,world", [(0, 1), (1, 2)])
def test_share_of_a_small_wide_table_against_the_checker(rank, world):
    import torch
    P = pkg()
    wl = W.WideWorkload(P, ctx(), 0, rank, world, n_columns=7, n_row_groups=4, values_per_chunk=3 * 4096, page_values=4096, unique=3)
    assert wl.n_cells == 7 * (4 // world)
    stream = torch.cuda.current_stream(0).cuda_stream
    wl.launch(stream)
    res = wl.check(stream, sample=1)  # statuses, and every item against the generator's values
    seen = set()
    for i, (g, c, name, q, part, d_at, o_at) in enumerate(wl.parts):
        assert g % world == rank
        if (name, id(part)) in seen:  # the distinct pages once each
            continue
        seen.add((name, id(part)))
        want, want_offs = _checker(P, part)
        got = wl.dst[d_at: d_at + part.out_bytes].cpu().numpy()
        assert res[i].out_bytes == part.out_bytes == np.asarray(want).view(np.uint8).size, (name, q)
        assert got.tobytes() == np.ascontiguousarray(want).tobytes(), (name, q)
        if want_offs is not None:
            o = wl.off[o_at: o_at + part.offsets].cpu().numpy().view(np.uint32)
            assert np.array_equal(o, want_offs[: part.offsets]), (name, q)
    assert {n for n, _ in seen} == set(W.SHAPES)


def test_eight_row_groups_of_the_full_size_table():
    """What one of eight GPUs decodes in `bench.py --config wide`: 8 row groups x 64 columns x 1 Mi values per
    chunk, 4800 work items in one batch.  Every item's status and size, and every 13th item's bytes and
    offsets against the generator's values."""
    import torch
    P = pkg()
    wl = W.WideWorkload(P, ctx(), 0, 0, 8)
    assert wl.n_cells == 64 * 8 and wl.values == 64 * 8 * (1 << 20)
    stream = torch.cuda.current_stream(0).cuda_stream
    wl.launch(stream)
    res = wl.check(stream, sample=13)
    for i, (g, c, name, q, part, d_at, o_at) in enumerate(wl.parts):
        assert g % 8 == 0 and res[i].status == 0 and res[i].out_bytes == part.out_bytes, (i, name, q, res[i].status, res[i].out_bytes)
    # a second launch of the same prepared batch gives the same bytes (nothing of the first is left in the scratch)
    before = wl.dst[: 1 << 24].clone()
    wl.launch(stream)
    wl.check(stream, sample=499)
    assert torch.equal(before, wl.dst[: 1 << 24])
    del wl
    torch.cuda.empty_cache()


def test_bench_with_two_ranks_under_torch_distributed_run():
    """The N > 1 path of bench.py as the driver starts it, in a fresh child process: two ranks on the one GPU there is
    (PQD_BENCH_REHEARSE=1: gloo for the barrier and the reduction, RCCL refuses two ranks on one device), one JSON line."""
    import json
    import os
    import socket
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    env = dict(os.environ, PQD_BENCH_REHEARSE="1", MASTER_ADDR="127.0.0.1")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "bench.py", "--gpus", "2", "--steps", "3", "--warmup", "1", "--pages", "128", "--unique-pages", "16",
           "--no-cpu-baseline"]
    out = subprocess.run(cmd, cwd=root, env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d["n_gpus"] == 2 and d["steps"] == 3 and d["scaling"] == "weak" and d["value"] > 0
    assert d["cpu_baseline"] is None and d["roofline"]["frac"] > 0
    assert "2 GPU" in d["config"]["parallelism"]
