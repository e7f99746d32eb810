"""When the workgroups of rle_page_kernel start and end (a library built with -DPQD_PAGE_TIMING2 reports
the cycle counter at entry and at the result store): python tools/page_timing2.py [pages per batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import misc_rates as M
from parquet_go_amd.runtime import PageSpec, decode_pages
rng = np.random.default_rng(1)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
for n, bw in ((1 << 17, 3), (1 << 15, 3)):
    pages = [M.rle_levels_page(rng, n, bw)[0] for _ in range(4)]
    specs = [PageSpec(M.pkg.ENC.RLE, M.pkg.KIND.LEVELS, pages[k % 4], bit_width=bw, dst_cap=n) for k in range(reps)]
    for trial in range(2):
        outs = decode_pages(M.ctx, specs, grow=False)
        t0 = np.array([o.arg for o in outs], dtype=np.int64); t1 = np.array([o.consumed for o in outs], dtype=np.int64)
        base = t0.min()
        d = t1 - t0
        print(n, bw, "pages", reps, "span", int(t1.max() - base), "entry pct 50/90/100", [int(x) for x in np.percentile(t0 - base, [50, 90, 100])],
              "duration pct 10/50/90/100", [int(x) for x in np.percentile(d, [10, 50, 90, 100])])
