"""Host-side cost of a batch: pqd_batch_create (classification, one device allocation, descriptor
upload), pqd_batch_finish (synchronise, results back) and pqd_batch_destroy, for the headline
batch and for a row-group-sized mixed batch.  The decode itself is not timed here."""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import misc_rates as M  # noqa: E402
import bench  # noqa: E402

torch, pkg = M.torch, M.pkg


def pages_delta(n_pages):
    rng = np.random.default_rng(1)
    v = 1_600_000_000_000_000_000 + np.cumsum(rng.integers(1_000_000, 1_004_096, size=131072, dtype=np.int64))
    enc = bench.delta_page_int64(v)
    slot = (len(enc) + 255) // 256 * 256
    src = torch.from_numpy(np.frombuffer(enc + b"\0" * (slot - len(enc)), np.uint8).copy()).to(M.DEV).repeat(n_pages)
    dst = torch.empty(n_pages * 131072, dtype=torch.int64, device=M.DEV)
    out = []
    for i in range(n_pages):
        p = pkg.Page()
        p.src, p.src_len = src.data_ptr() + i * slot, len(enc)
        p.dst, p.dst_cap = dst.data_ptr() + i * 131072 * 8, 131072 * 8
        p.num_values, p.encoding, p.kind = -1, pkg.ENC.DELTA_BINARY_PACKED, pkg.KIND.INT64
        out.append(p)
    return out, (src, dst)


def measure(name, pages):
    arr = (pkg.Page * len(pages))(*pages)
    import ctypes as C
    L = pkg.lib()
    best = [1e9, 1e9, 1e9, 1e9]
    for _ in range(10):
        h = C.c_void_p()
        t0 = time.perf_counter()
        assert L.pqd_batch_create(M.ctx.handle, arr, len(pages), C.byref(h)) == 0
        t1 = time.perf_counter()
        assert L.pqd_batch_launch(h, C.c_void_p(M.stream)) == 0
        t2 = time.perf_counter()
        assert L.pqd_batch_finish(h, C.c_void_p(M.stream), arr, len(pages)) == 0
        t3 = time.perf_counter()
        L.pqd_batch_destroy(h)
        t4 = time.perf_counter()
        for k, d in enumerate((t1 - t0, t2 - t1, t3 - t2, t4 - t3)):
            best[k] = min(best[k], d)
    print(f"{name:28s} {len(pages):5d} pages  create {best[0]*1e6:8.1f} us  launch (enqueue) {best[1]*1e6:7.1f} us  "
          f"finish (decode + results) {best[2]*1e6:8.1f} us  destroy {best[3]*1e6:7.1f} us", flush=True)


def main():
    for n in (2048, 128, 8):
        pages, keep = pages_delta(n)
        measure(f"DELTA int64 pages", pages)


if __name__ == "__main__":
    main()
