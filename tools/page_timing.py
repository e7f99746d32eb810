"""Cycle counts of the phases of rle_page_kernel (a library built with -DPQD_PAGE_TIMING reports them in
place of the status argument and the consumed bytes): python tools/page_timing.py [pages per batch]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import numpy as np
import misc_rates as M
from parquet_go_amd.runtime import PageSpec, decode_pages
rng = np.random.default_rng(1)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 4
for n, bw in ((1 << 20, 1), (1 << 17, 3), (1 << 15, 3)):
    pages = [M.rle_levels_page(rng, n, bw)[0] for _ in range(4)]
    specs = [PageSpec(M.pkg.ENC.RLE, M.pkg.KIND.LEVELS, pages[k % 4], bit_width=bw, dst_cap=n) for k in range(reps)]
    for trial in range(3):
        outs = decode_pages(M.ctx, specs, grow=False)
        c = np.array([o.arg >> 32 for o in outs]); w = np.array([o.arg & 0xffffffff for o in outs]); l = np.array([o.consumed >> 32 for o in outs]); e = np.array([o.consumed & 0xffffffff for o in outs])
        print(n, bw, "pages", reps, "trial", trial, "bytes", len(pages[0]), "status", outs[0].status, "mean ticks: load+A", int(l.mean()), "chase", int(w.mean()), "C", int(c.mean()), "expand", int(e.mean()))
