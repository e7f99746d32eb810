"""Per-kernel durations out of a rocprofv3 database (rocpd format), grouped by kernel and grid
size: python tools/kernel_times.py gpurun_out/prof/x_results.db"""
import sqlite3
import sys

c = sqlite3.connect(sys.argv[1])
tabs = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
kd = [t for t in tabs if t.startswith("rocpd_kernel_dispatch")][0]
sym = [t for t in tabs if t.startswith("rocpd_info_kernel_symbol")][0]
q = (f"select s.display_name, d.grid_size_x, d.workgroup_size_x, count(*), avg(d.end-d.start)/1e3, min(d.end-d.start)/1e3 "
     f"from {kd} d join {sym} s on d.kernel_id=s.id group by s.display_name, d.grid_size_x order by min(d.start)")
print(f"{'kernel':50s} {'grid':>10s} {'wg':>5s} {'calls':>6s} {'avg us':>9s} {'min us':>9s}")
for name, grid, wg, n, avg, mn in c.execute(q):
    name = name.replace("pqd::(anonymous namespace)::", "").replace("void ", "").split("(")[0]
    print(f"{name[:50]:50s} {grid:10d} {wg:5d} {n:6d} {avg:9.1f} {mn:9.1f}")
