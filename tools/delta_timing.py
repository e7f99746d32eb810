"""When and where the workgroups of delta_fast_kernel run (a library built with -DPQD_DELTA_TIMING reports the cycle
counter at entry and at the end, and the hardware ids, in place of the status argument, the consumed bytes and the
value count): python tools/delta_timing.py [pages]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import misc_rates as M
import bench
pkg = M.pkg
rng = np.random.default_rng(5)
n_pages = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
n = 131072
t0 = 1_600_000_000_000_000_000
vals = [t0 + np.cumsum(1_000_000 + rng.integers(0, 4096, size=n, dtype=np.int64)) for _ in range(64)]
pages_bytes = [bench.delta_page_int64(v) for v in vals]
uniq = len(pages_bytes)
slot = max((len(p) + 255) // 256 * 256 for p in pages_bytes) + 256
host = np.zeros(uniq * slot, np.uint8)
for k, p in enumerate(pages_bytes):
    host[k * slot: k * slot + len(p)] = np.frombuffer(p, np.uint8)
src = torch.from_numpy(host).to(M.DEV).repeat(n_pages // uniq)
oslot = 8 * n + 256
dst = torch.empty(n_pages * oslot, dtype=torch.uint8, device=M.DEV)
pages = []
for i in range(n_pages):
    p = pkg.Page()
    p.src, p.src_len = src.data_ptr() + i * slot, len(pages_bytes[i % uniq])
    p.dst, p.dst_cap = dst.data_ptr() + i * oslot, oslot - 256
    p.num_values = -1
    p.encoding, p.kind, p.bit_width = pkg.ENC.DELTA_BINARY_PACKED, pkg.KIND.INT64, 0
    pages.append(p)
batch = pkg.Batch(M.ctx, pages)
for trial in range(4):
    batch.launch(M.stream)
    res = batch.finish(M.stream)
    assert all(res[i].status == 0 for i in range(n_pages))
    te = np.array([res[i].status_arg for i in range(n_pages)], dtype=np.int64)
    tx = np.array([res[i].consumed for i in range(n_pages)], dtype=np.int64)
    hw = np.array([res[i].out_values for i in range(n_pages)], dtype=np.int64)
    xcc = hw >> 32
    cu = (hw >> 8) & 0xf
    sh = (hw >> 12) & 1
    se = (hw >> 13) & 7
    base = te.min()
    d = tx - te
    print("trial", trial, "span", int(tx.max() - base), "entry pct 50/90/100", [int(x) for x in np.percentile(te - base, [50, 90, 100])],
          "lifetime pct 0/10/50/90/100", [int(x) for x in np.percentile(d, [0, 10, 50, 90, 100])],
          "end pct 0/10/50/90/100", [int(x) for x in np.percentile(tx - base, [0, 10, 50, 90, 100])])
    if trial == 3:
        for x in range(8):
            m = xcc == x
            print("  xcc", x, "pages", int(m.sum()), "end median", int(np.median(tx[m] - base)), "max", int((tx[m] - base).max()))
        key = (xcc * 8 + se) * 32 + sh * 16 + cu
        ends = {}
        for k, e in zip(key, tx - base):
            ends.setdefault(int(k), []).append(int(e))
        counts = np.array([len(v) for v in ends.values()])
        print("  compute units seen", len(ends), "workgroups per unit min/median/max", counts.min(), int(np.median(counts)), counts.max())
        last = np.array([max(v) for v in ends.values()]); first = np.array([min(v) for v in ends.values()])
        print("  per unit: last end pct 0/50/100", [int(x) for x in np.percentile(last, [0, 50, 100])], "first end pct 0/50/100", [int(x) for x in np.percentile(first, [0, 50, 100])])
        groups = {}
        for i, k in enumerate(key):
            groups.setdefault(int(k), []).append(i)
        shown = 0
        rank_life = [[] for _ in range(8)]
        for k, idx in groups.items():
            idx = sorted(idx, key=lambda i: te[i])
            for r, i in enumerate(idx[:8]):
                rank_life[r].append(int(d[i]))
            if shown < 6:
                print("  unit", k, "pages in entry order", idx, "entry", [int(te[i] - te[idx[0]]) for i in idx], "lifetime", [int(d[i]) for i in idx])
                shown += 1
        print("  median lifetime by entry order within the unit", [int(np.median(v)) for v in rank_life if v])
        by_count = {}
        for v in ends.values():
            by_count.setdefault(len(v), []).append(max(v))
        for c in sorted(by_count):
            print("  units with", c, "workgroups:", len(by_count[c]), "median last end", int(np.median(by_count[c])))
