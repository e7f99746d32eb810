"""Copies what tools/gpu_round_check.sh left under gpurun_out/ into profiles/ (tracked):
the bench line, the rocprofv3 kernel statistics of the same command, and the HBM traffic of
the kernels per launch from the FETCH_SIZE / WRITE_SIZE counter passes (KiB; FETCH_SIZE
doubled: gfx950 tallies 128-byte read requests at 64 bytes).  Usage: python tools/refresh_profiles.py <tag for the file names>"""
import collections
import csv
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")
PROF = os.path.join(ROOT, "profiles")


def counter_means(name):
    path = os.path.join(OUT, f"pmc_{name}", "vm_counter_collection.csv")
    acc = collections.defaultdict(list)
    with open(path) as f:
        for r in csv.DictReader(f):
            if r["Counter_Name"] == name:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in acc.items()}


def short(kernel):
    k = kernel.replace("void ", "").replace("pqd::(anonymous namespace)::", "")
    return k.split("(")[0]


def main(tag):
    shutil.copy(os.path.join(OUT, "bench.json"), os.path.join(PROF, f"{tag}_bench_line.json"))
    shutil.copy(os.path.join(OUT, "prof_stats", "r01_kernel_stats.csv"), os.path.join(PROF, f"{tag}_kernel_stats_bench_default.csv"))
    fetch, write = counter_means("FETCH_SIZE"), counter_means("WRITE_SIZE")
    rows = []
    total = collections.defaultdict(float)
    for kernel in sorted(set(fetch) | set(write)):
        if "pqd::" not in kernel:
            continue
        name = short(kernel)
        f, w = fetch.get(kernel, 0.0), write.get(kernel, 0.0)
        rows.append((name, "FETCH_SIZE", f, "raw; gfx950 counts a wide streaming read at half its bytes: x2"))
        rows.append((name, "WRITE_SIZE", w, "exact for 16-byte-per-lane stores"))
        group = "delta" if name.startswith("delta") else ("rle_dict" if name.startswith("rle") else name)
        total[group] += (2 * f + w) * 1024
    with open(os.path.join(PROF, f"{tag}_pmc_hbm_traffic.csv"), "w") as f:
        f.write("kernel,counter,mean_per_launch_KiB,note\n")
        for r in rows:
            f.write(f"{r[0]},{r[1]},{r[2]:.1f},{r[3]}\n")
    traffic = {k: int(round(v)) for k, v in total.items()}
    traffic["source"] = ("rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, one pass each, of `python3 bench.py --steps 2 --warmup 1 "
                         f"--no-cpu-baseline --skip-check` (profiles/{tag}_pmc_hbm_traffic.csv); per launch; bytes = KiB x 1024; "
                         "FETCH_SIZE doubled (gfx950 tallies 128-byte read requests at 64 bytes)")
    with open(os.path.join(PROF, "traffic.json"), "w") as f:
        json.dump(traffic, f, indent=1)
    print(json.dumps(traffic, indent=1))


if __name__ == "__main__":
    if len(sys.argv) < 2:
        sys.exit("usage: python tools/refresh_profiles.py <tag for the file names>")
    main(sys.argv[1])
