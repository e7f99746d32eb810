"""RLE_DICTIONARY int64 pages (64 x 262144 values, 16-bit indices, 64 Ki-entry dictionary) with the
indices confined to the first H dictionary entries: how much of the kernel time is the gather's reach."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import __graft_entry__ as entry
import bench

pkg = entry.load_package()
from parquet_go_amd import sharding as S
ctx = pkg.Context(0)
stream = torch.cuda.current_stream(0).cuda_stream
for hot in (65536, 16384, 4096, 1024, 64):
    rng = bench.splitmix64(hot)
    dictionary = rng.integers(-2**62, 2**62, size=65536, dtype=np.int64)
    enc = []
    for k in range(64):
        idx = rng.integers(0, hot, size=262144, dtype=np.int64).astype(np.int32)
        idx[0] = 65535
        enc.append(bench.rle_dict_page(idx))
    wl = bench.Workload(pkg, ctx, 0, enc, 1, pkg.KIND.INT64, pkg.ENC.RLE_DICTIONARY, 262144, dictionary)
    wl.launch(stream); wl.check(stream)
    wall, kms = bench.time_steps(S, torch, wl, stream, 20, 3, None)
    print(f"indices within the first {hot:6d} entries ({hot*8//1024:4d} KiB): kernels {np.mean(kms)*1e3:.1f} us, "
          f"{wl.algorithmic_bytes/np.mean(kms)/1e6:.0f} GB/s algorithmic")
    del wl
