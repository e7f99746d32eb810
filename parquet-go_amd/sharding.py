"""Multi-GPU layout of the decode path: one process per GPU, row groups dealt round robin.

Pages are independent decode units (each carries its own RLE / DELTA headers,
encoding/delta/binary_packed.go:409-452), dictionaries belong to one column chunk
(file.go:1092,1372-1382) and row groups are disjoint (file.go:587,753), so the path
shards with no exchange step: row group g goes to rank g mod N, and no collective
runs on the data path.  The only collectives are the barrier and the MAX-reduce
that bracket a timed region.
"""
import time
from typing import Callable, List, Optional


def row_groups_of(rank: int, world: int, n_row_groups: int) -> List[int]:
    """Row groups decoded by `rank`: g with g mod world == rank."""
    if not (0 <= rank < world):
        raise ValueError(f"rank {rank} outside world of {world}")
    return list(range(rank, n_row_groups, world))


def owner_of(row_group: int, world: int) -> int:
    return row_group % world


def timed_region(step: Callable[[], None], steps: int, warmup: int, sync: Callable[[], None],
                 barrier: Optional[Callable[[], None]] = None) -> float:
    """`warmup` untimed steps, then exactly `steps` timed ones bracketed by barrier + device
    synchronisation on both sides.  Returns this rank's wall seconds for the timed steps."""
    for _ in range(warmup):
        step()
    sync()
    if barrier is not None:
        barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    sync()
    if barrier is not None:
        barrier()
    sync()
    return time.perf_counter() - t0


def max_over_ranks(value: float, dist=None, device: str = "cpu") -> float:
    """MAX of a per-rank scalar over all ranks (the slowest rank defines the step time)."""
    if dist is None:
        return float(value)
    import torch
    t = torch.tensor([value], dtype=torch.float64, device=device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def whole_job_rate(units_per_rank: float, world: int, seconds: float) -> float:
    """Units all ranks processed per second of the slowest rank (weak scaling: per-rank work is fixed)."""
    return units_per_rank * world / seconds
