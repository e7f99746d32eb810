"""PCIe-inclusive rates of the per-method host entry points (host src -> host dst), for the record:
one call per page as a cgo binding of encoding.Encoding makes them, at the reference's page size
(1 MiB of decoded values) from 1 and from 16 host threads, then one very large page, then the same
bytes through a reader's own pinned staging and the batched device API."""
import ctypes as C
import os, sys, threading, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry
import bench

pkg = entry.load_package()
from parquet_go_amd import encoding as E
from oracle import pqoracle as O
rng = np.random.default_rng(1)
L = pkg.lib()
ctx = pkg.Context(0)
H = ctx.handle
VP = C.c_void_p


def _ptr(a):
    return VP(a.ctypes.data)


def per_call(name, call_of, out_bytes, threads, seconds=2.0):
    """call_of(t) -> a zero-argument function making ONE library call with thread t's own buffers."""
    calls = [call_of(t) for t in range(threads)]
    for c in calls:
        for _ in range(3):
            assert c() == 0
    counts = [0] * threads
    start = threading.Barrier(threads + 1)
    stop = [False]

    def work(t):
        start.wait()
        c, k = calls[t], 0
        while not stop[0]:
            c()
            k += 1
        counts[t] = k

    ts = [threading.Thread(target=work, args=(t,)) for t in range(threads)]
    for t in ts:
        t.start()
    start.wait()
    t0 = time.perf_counter()
    time.sleep(seconds)
    stop[0] = True
    for t in ts:
        t.join()
    dt = time.perf_counter() - t0
    n = sum(counts)
    print(f"  {name:34s} {threads:2d} thread(s): {n / dt:9.0f} calls/s, {dt / n * threads * 1e6:8.1f} us per call, "
          f"{n * out_bytes / dt / 1e9:6.2f} GB/s decoded", flush=True)


PAGE = 131072  # 1 MiB of int64 values: the reference's default page size
ts_vals = 1_600_000_000_000_000_000 + np.cumsum(rng.integers(1_000_000, 1_004_096, size=PAGE, dtype=np.int64))
delta_page = np.frombuffer(bench.delta_page_int64(ts_vals), np.uint8).copy()
plain_page = ts_vals.view(np.uint8).copy()
idx = rng.integers(0, 65536, size=PAGE * 2, dtype=np.int32)
rle_page = np.frombuffer(bytes(O.rle_dict_encode_int32(idx)), np.uint8).copy()
strs = [b"https://www.example.com/catalog/items/%08d" % int(v) for v in rng.integers(0, 10**8, size=PAGE // 6)]
dlba_page = np.frombuffer(bytes(O.dlba_encode(strs)), np.uint8).copy()
str_bytes = sum(map(len, strs))
small_vals = ts_vals[:1024].copy()
small_page = np.frombuffer(bench.delta_page_int64(small_vals), np.uint8).copy()


def fixed(fn, code, page, n_out, dtype, *pre):
    def make(t):
        dst = np.empty(n_out, dtype)
        nb, arg = C.c_uint64(0), C.c_int64(0)
        args = (H, code) + pre + (_ptr(page), C.c_uint64(page.size), _ptr(dst), C.c_uint64(dst.nbytes), C.byref(nb), C.byref(arg))
        return lambda keep=(dst, page): fn(*args)
    return make


def byte_array(code, page, n_bytes, n_vals):
    def make(t):
        dst = np.empty(n_bytes, np.uint8)
        off = np.empty(n_vals + 1, np.uint32)
        nb, no, arg = C.c_uint64(0), C.c_uint64(0), C.c_int64(0)
        args = (H, code, _ptr(page), C.c_uint64(page.size), _ptr(dst), C.c_uint64(dst.size), _ptr(off), C.c_uint64(off.size),
                C.byref(nb), C.byref(no), C.byref(arg))
        return lambda keep=(dst, off, page): L.pqd_decode_byte_array(*args)
    return make


def estimate(code, page):
    def make(t):
        args = (H, code, _ptr(page), C.c_uint64(page.size))
        want = str_bytes
        return lambda: 0 if L.pqd_estimate_decode_byte_array_size(*args) == want else 1
    return make


check = np.empty(PAGE, np.int64)
assert np.array_equal(E.DeltaBinaryPacked().DecodeInt64(check, delta_page), ts_vals)
print("per-call host entry points, pageable caller memory, pages of 1 MiB decoded (staging, stream and arena from the context's pool):")
for threads in (1, 16):
    per_call("DELTA_BINARY_PACKED int64 1 MiB", fixed(L.pqd_decode_int64, pkg.ENC.DELTA_BINARY_PACKED, delta_page, PAGE, np.int64), PAGE * 8, threads)
    per_call("PLAIN int64 1 MiB", fixed(L.pqd_decode_int64, pkg.ENC.PLAIN, plain_page, PAGE, np.int64), PAGE * 8, threads)
    per_call("RLE_DICTIONARY indexes 1 MiB", fixed(L.pqd_decode_int32, pkg.ENC.RLE_DICTIONARY, rle_page, PAGE * 2, np.int32, 0), PAGE * 8, threads)
    per_call("DELTA_LENGTH_BYTE_ARRAY ~1 MiB", byte_array(pkg.ENC.DELTA_LENGTH_BYTE_ARRAY, dlba_page, str_bytes, len(strs)), str_bytes, threads)
    per_call("estimate DELTA_LENGTH_BYTE_ARRAY", estimate(pkg.ENC.DELTA_LENGTH_BYTE_ARRAY, dlba_page), str_bytes, threads)
    per_call("DELTA_BINARY_PACKED int64 8 KiB", fixed(L.pqd_decode_int64, pkg.ENC.DELTA_BINARY_PACKED, small_page, 1024, np.int64), 8192, threads)

# The same calls with the caller's buffers in page-locked memory the library knows (pqd_host_alloc, or pqd_host_register
# where the driver locks user memory): a reader's buffer pool.  Every thread's destination and the page bytes come out
# of one slab.
print("the same with the caller's buffers in page-locked memory of the library (pqd_host_alloc): copies go from and to it directly")
SLAB_T = 16
SLAB_BYTES = SLAB_T * (PAGE * 8 + 4096) + 4 * (1 << 20)
_slab_at = C.c_void_p()
assert L.pqd_host_alloc(H, C.c_uint64(SLAB_BYTES), C.byref(_slab_at)) == 0
slab = np.frombuffer((C.c_uint8 * SLAB_BYTES).from_address(_slab_at.value), np.uint8)
_at = [0]


def from_slab(nbytes, dtype=np.uint8):
    a = (_at[0] + 255) & ~255
    _at[0] = a + nbytes
    assert _at[0] <= slab.nbytes
    return slab[a: a + nbytes].view(dtype)


def reg_copy(page):
    out = from_slab(page.size)
    out[:] = page
    return out


def fixed_reg(fn, code, page, n_out, dtype, *pre):
    dsts = [from_slab(n_out * np.dtype(dtype).itemsize, dtype) for _ in range(SLAB_T)]

    def make(t):
        dst = dsts[t]
        nb, arg = C.c_uint64(0), C.c_int64(0)
        args = (H, code) + pre + (_ptr(page), C.c_uint64(page.size), _ptr(dst), C.c_uint64(dst.nbytes), C.byref(nb), C.byref(arg))
        return lambda keep=(dst, page): fn(*args)
    return make


r_delta, r_plain, r_rle = reg_copy(delta_page), reg_copy(plain_page), reg_copy(rle_page)
_at_keep = _at[0]
for threads in (1, 16):
    _at[0] = _at_keep
    per_call("DELTA_BINARY_PACKED int64 1 MiB", fixed_reg(L.pqd_decode_int64, pkg.ENC.DELTA_BINARY_PACKED, r_delta, PAGE, np.int64), PAGE * 8, threads)
    _at[0] = _at_keep
    per_call("PLAIN int64 1 MiB", fixed_reg(L.pqd_decode_int64, pkg.ENC.PLAIN, r_plain, PAGE, np.int64), PAGE * 8, threads)
    _at[0] = _at_keep
    per_call("RLE_DICTIONARY indexes 1 MiB", fixed_reg(L.pqd_decode_int32, pkg.ENC.RLE_DICTIONARY, r_rle, PAGE * 2, np.int32, 0), PAGE * 8, threads)
_at[0] = _at_keep
one = fixed_reg(L.pqd_decode_int64, pkg.ENC.DELTA_BINARY_PACKED, r_delta, PAGE, np.int64)(0)
assert one() == 0
del one, r_delta, r_plain, r_rle, slab
assert L.pqd_host_free(H, _slab_at) == 0

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
print(f"host entry point, one DELTA int64 page of {n} values ({len(page)/1e6:.1f} MB in, {dst.nbytes/1e6:.1f} MB out): "
      f"{best*1e3:.2f} ms = {dst.nbytes/best/1e9:.2f} GB/s decoded, pageable caller memory")

# The same job as a reader with its own staging would run it: the headline batch (2048 DELTA pages)
# in eight slices, pinned host buffers, copy in -> decode -> copy out on three streams so that the
# slices overlap.  Encoded bytes go up, decoded bytes come back; both directions counted in the time.
import torch  # noqa: E402


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
