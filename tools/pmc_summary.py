"""Per-kernel means of the counters of a rocprofv3 --pmc run: either a counter_collection.csv or a
rocpd database (*_results.db).  python tools/pmc_summary.py <file> [kernel name substring]"""
import collections
import csv
import sqlite3
import sys


def rows(path, match):
    if path.endswith(".db"):
        c = sqlite3.connect(path)
        for name, counter, value in c.execute("select kernel_name, counter_name, value from counters_collection"):
            if match in name:
                yield name, counter, float(value)
    else:
        with open(path) as f:
            for r in csv.DictReader(f):
                if match in r["Kernel_Name"]:
                    yield r["Kernel_Name"], r["Counter_Name"], float(r["Counter_Value"])


def short(kernel):
    return kernel.replace("void ", "").replace("pqd::(anonymous namespace)::", "").split("(")[0]


def main(path, match):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for name, counter, value in rows(path, match):
        acc[short(name)][counter].append(value)
    for k, cs in acc.items():
        print(k)
        waves = sum(cs["SQ_WAVES"]) / len(cs["SQ_WAVES"]) if "SQ_WAVES" in cs else None
        for c, v in sorted(cs.items()):
            m = sum(v) / len(v)
            extra = f"  per-wave {m / waves:.1f}" if waves else ""
            print(f"  {c:30s} n={len(v):3d} mean={m:.5g}{extra}")


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
