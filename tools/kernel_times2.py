"""Per-kernel durations out of a rocprofv3 database (rocpd format), grouped by kernel and by both
grid dimensions, in launch order: python tools/kernel_times2.py gpurun_out/prof/x_results.db"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
q = (f"select s.display_name, d.grid_size_x, d.grid_size_y, count(*), avg(d.end-d.start)/1e3 from {kd} d join {sym} s "
     f"on d.kernel_id=s.id group by s.display_name, d.grid_size_x, d.grid_size_y order by min(d.start)")
print(f"{'kernel':46s} {'grid x':>8s} {'y':>5s} {'n':>4s} {'avg us':>9s}")
for name, gx, gy, n, avg in c.execute(q):
    if "pqd" in name:
        nm = name.replace("pqd::(anonymous namespace)::", "").split("(")[0]
        print(f"{nm[:46]:46s} {gx:8d} {gy:5d} {n:4d} {avg:9.1f}")
