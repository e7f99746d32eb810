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

# The same job as a reader with its own staging would run it: the headline batch (2048 DELTA pages)
# in eight slices, pinned host buffers, copy in -> decode -> copy out on three streams so that the
# slices overlap.  Encoded bytes go up, decoded bytes come back; both directions counted in the time.
import torch  # noqa: E402


ctx = pkg.Context(0)
dev = "cuda:0"
vals = [1_600_000_000_000_000_000 + np.cumsum(rng.integers(1_000_000, 1_004_096, size=131072, dtype=np.int64)) for _ in range(16)]
pages = [bench.delta_page_int64(v) for v in vals]
slot = max((len(p) + 255) // 256 * 256 for p in pages)
SLICES, PER = 8, 256
h_in = torch.empty(SLICES * PER * slot, dtype=torch.uint8).pin_memory()
for k in range(SLICES * PER):
    p = pages[k % len(pages)]
    h_in[k * slot: k * slot + len(p)] = torch.from_numpy(np.frombuffer(p, np.uint8).copy())
h_out = torch.empty(SLICES * PER * 131072, dtype=torch.int64).pin_memory()
d_in = [torch.empty(PER * slot, dtype=torch.uint8, device=dev) for _ in range(SLICES)]
d_out = [torch.empty(PER * 131072, dtype=torch.int64, device=dev) for _ in range(SLICES)]
batches = []
for s in range(SLICES):
    ps = []
    for k in range(PER):
        p = pkg.Page()
        p.src, p.src_len = d_in[s].data_ptr() + k * slot, len(pages[(s * PER + k) % len(pages)])
        p.dst, p.dst_cap = d_out[s].data_ptr() + k * 131072 * 8, 131072 * 8
        p.num_values, p.encoding, p.kind = -1, pkg.ENC.DELTA_BINARY_PACKED, pkg.KIND.INT64
        ps.append(p)
    batches.append(pkg.Batch(ctx, ps))
streams = [torch.cuda.Stream(device=dev) for _ in range(3)]


def job():
    for s in range(SLICES):
        st = streams[s % len(streams)]
        with torch.cuda.stream(st):
            d_in[s].copy_(h_in[s * PER * slot: (s + 1) * PER * slot], non_blocking=True)
            batches[s].launch(st.cuda_stream)
            h_out[s * PER * 131072: (s + 1) * PER * 131072].copy_(d_out[s], non_blocking=True)
    torch.cuda.synchronize()


job()
for k in (0, SLICES * PER - 1):
    assert np.array_equal(h_out[k * 131072: (k + 1) * 131072].numpy(), vals[k % len(vals)])
best = 1e9
for _ in range(5):
    t0 = time.perf_counter()
    job()
    best = min(best, time.perf_counter() - t0)
print(f"pinned staging, {SLICES} slices of {PER} DELTA int64 pages on {len(streams)} streams ({h_in.numel()/1e6:.0f} MB up, "
      f"{h_out.numel()*8/1e6:.0f} MB down): {best*1e3:.2f} ms = {h_out.numel()*8/best/1e9:.1f} GB/s decoded, copies included")
