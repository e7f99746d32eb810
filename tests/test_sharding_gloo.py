"""The N > 1 path of the benchmark on CPU: two gloo ranks, the row-group deal, the
barrier-bracketed timed region and the MAX over ranks (no GPU needed)."""
import os
import socket
import sys
import time

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def _sharding():
    import importlib.util
    spec = importlib.util.spec_from_file_location("pqd_sharding", os.path.join(ROOT, "parquet-go_amd", "sharding.py"))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def test_row_group_deal_is_a_partition():
    S = _sharding()
    for world in (1, 2, 3, 4, 8):
        for n in (0, 1, 7, 64, 65):
            seen = []
            for r in range(world):
                mine = S.row_groups_of(r, world, n)
                assert all(S.owner_of(g, world) == r for g in mine)
                seen += mine
            assert sorted(seen) == list(range(n))
            sizes = [len(S.row_groups_of(r, world, n)) for r in range(world)]
            assert max(sizes) - min(sizes) <= 1
    with pytest.raises(ValueError):
        S.row_groups_of(2, 2, 10)


def _worker(rank, world, port, out_dir):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group(backend="gloo", rank=rank, world_size=world)
    S = _sharding()
    mine = S.row_groups_of(rank, world, 64)
    work = {"steps": 0}

    def step():  # rank 1 is the slow one
        work["steps"] += 1
        time.sleep(0.01 if rank == 0 else 0.03)

    # the page-dealing code of `bench.py --config wide`: this rank's column chunks of a 6 x 7 table
    import bench_wide as W
    cells = W.cells_of(rank, world, 6, 7)
    with open(os.path.join(out_dir, f"cells{rank}.txt"), "w") as f:
        f.write("\n".join(f"{g} {c} {name}" for g, c, name in cells))

    wall = S.timed_region(step, steps=5, warmup=2, sync=lambda: None, barrier=dist.barrier)
    slowest = S.max_over_ranks(wall, dist)
    rate = S.whole_job_rate(len(mine), world, slowest / 5)
    with open(os.path.join(out_dir, f"r{rank}.txt"), "w") as f:
        f.write(f"{work['steps']} {wall:.6f} {slowest:.6f} {rate:.3f} {len(mine)}")
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_gloo(tmp_path):
    import torch.multiprocessing as mp
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    # the ranks' shares of the wide table are disjoint and together are the table
    import bench_wide as W
    shares = [[tuple(line.split()) for line in open(tmp_path / f"cells{r}.txt").read().splitlines()] for r in range(2)]
    assert not set(shares[0]) & set(shares[1])
    table = [(str(g), str(c), name) for g, c, name in W.table_cells(6, 7)]
    assert sorted(shares[0] + shares[1]) == sorted(table) and len(table) == 42
    assert all(int(g) % 2 == r for r in range(2) for g, _, _ in shares[r])
    rows = [open(tmp_path / f"r{r}.txt").read().split() for r in range(2)]
    for r in rows:
        assert int(r[0]) == 7           # 2 warm-up + exactly 5 timed steps
        assert int(r[4]) == 32          # 64 row groups over 2 ranks
    slowest = [float(r[2]) for r in rows]
    assert slowest[0] == slowest[1]     # every rank reports the MAX over ranks
    assert slowest[0] >= 5 * 0.03 * 0.95
    # the barrier holds the fast rank until the slow one is done: its own wall time covers the slow rank's steps
    assert float(rows[0][1]) >= 5 * 0.03 * 0.9
    # whole-job rate = units of all ranks / slowest rank's step time
    assert abs(float(rows[0][3]) - 64 / (slowest[0] / 5)) / float(rows[0][3]) < 1e-3
