"""Which unit a kernel's instruction counts would keep busy for how long, beside its duration:
python tools/issue_bounds.py <pmc file (SQ_INSTS_SALU, SQ_INSTS_VALU, SQ_BUSY_CYCLES ...)> <kernel-trace results.db>

A compute unit's four SIMDs share one scalar unit (one scalar instruction a cycle); a SIMD takes
four cycles for a vector instruction of 64 lanes.  Counts are whole-launch totals, so on 256
compute units at 2.1 GHz (what the trace's durations imply on this box)
    scalar us = SQ_INSTS_SALU / 256 / 2100,   vector us = SQ_INSTS_VALU * 4 / 1024 / 2100.
A kernel whose duration is close to one of them is bound by that unit, not by memory."""
import collections
import sqlite3
import sys

sys.path.insert(0, __file__.rsplit("/", 1)[0])
from pmc_summary import rows, short  # noqa: E402


def durations(db):
    c = sqlite3.connect(db)
    out = collections.defaultdict(list)
    names = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    view = "kernels" if "kernels" in names else None
    if view is None:
        return out
    for name, start, end in c.execute("select name, start, end from kernels"):
        out[short(name)].append((end - start) / 1e3)
    return out


def main(pmc, trace):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for name, counter, value in rows(pmc, ""):
        acc[short(name)][counter].append(value)
    dur = durations(trace) if trace else {}
    print(f"{'kernel':50s} {'us':>8s} {'scalar us':>10s} {'vector us':>10s}")
    for k, cs in acc.items():
        s = sum(cs["SQ_INSTS_SALU"]) / max(1, len(cs["SQ_INSTS_SALU"]))
        v = sum(cs["SQ_INSTS_VALU"]) / max(1, len(cs["SQ_INSTS_VALU"]))
        d = dur.get(k)
        d = sum(d) / len(d) if d else float("nan")
        print(f"{k[:50]:50s} {d:8.1f} {s / 256 / 2100:10.1f} {v * 4 / 1024 / 2100:10.1f}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else None)
