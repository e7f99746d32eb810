"""Per-kernel means of the counters in a rocprofv3 counter_collection.csv."""
import csv, sys, collections

def main(path, match):
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    with open(path) as f:
        for r in csv.DictReader(f):
            if match in r["Kernel_Name"]:
                acc[r["Kernel_Name"][:60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
    for k, cs in acc.items():
        print(k)
        waves = None
        if "SQ_WAVES" in cs:
            waves = sum(cs["SQ_WAVES"]) / len(cs["SQ_WAVES"])
        for c, v in sorted(cs.items()):
            m = sum(v) / len(v)
            extra = f"  per-wave {m / waves:.1f}" if waves else ""
            print(f"  {c:28s} n={len(v):3d} mean={m:.4g}{extra}")

if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2] if len(sys.argv) > 2 else "")
