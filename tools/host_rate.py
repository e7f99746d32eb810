"""PCIe-inclusive rate of the per-method host wrapper (host src -> host dst), for the record."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
import bench

pkg = entry.load_package()
from parquet_go_amd import encoding as E
rng = np.random.default_rng(1)
n = 16 << 20
ts = 1_600_000_000_000_000_000 + np.cumsum(rng.integers(1_000_000, 1_004_096, size=n, dtype=np.int64))
page = bench.delta_page_int64(ts)
codec = E.DeltaBinaryPacked()
dst = np.empty(n, np.int64)
out = codec.DecodeInt64(dst, page)
assert np.array_equal(out, ts)
best = 1e9
for _ in range(5):
    t0 = time.perf_counter()
    codec.DecodeInt64(dst, page)
    best = min(best, time.perf_counter() - t0)
print(f"host wrapper, one DELTA int64 page of {n} values ({len(page)/1e6:.1f} MB in, {dst.nbytes/1e6:.1f} MB out): "
      f"{best*1e3:.2f} ms = {dst.nbytes/best/1e9:.2f} GB/s decoded, pageable host memory, hipMalloc/hipFree per call")
