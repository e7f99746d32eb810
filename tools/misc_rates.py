"""Device-resident rates of the encodings beside the headline: RLE levels, BIT_PACKED levels,
BYTE_STREAM_SPLIT, PLAIN, DELTA_LENGTH_BYTE_ARRAY, DELTA_BYTE_ARRAY, PLAIN byte arrays.

Pages are generated with numpy here (the formats are simple enough to emit directly), placed in
HBM once, decoded as one batch per encoding and timed with events on the launch stream.
Prints one line per encoding: time per launch, decoded GB/s, algorithmic GB/s (read + written)
and its fraction of the 8 TB/s HBM peak.  Results are checked against the generator's values.
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402
import __graft_entry__ as entry  # noqa: E402
import bench  # noqa: E402

pkg = entry.load_package()
from parquet_go_amd import sharding as S  # noqa: E402

ctx = pkg.Context(0)
DEV = "cuda:0"
stream = torch.cuda.current_stream(0).cuda_stream


def uvarint(v):
    return bench._uvarint(v)


def rle_levels_page(rng, n, bw):
    """parquet-mr style hybrid: runs of arbitrary length, half short (mean 4, bit-packed in
    groups of 8), half long (mean 4096, run-length)."""
    out = bytearray()
    vals = np.empty(n, np.uint8)
    i = 0
    while i < n:
        if rng.random() < 0.5:
            cnt = min(n - i, 8 * int(rng.integers(1, 64)))
            cnt -= cnt % 8
            if cnt == 0:
                cnt = n - i  # tail shorter than a group: emit as runs of one value below
                v = int(rng.integers(0, 1 << bw))
                out += uvarint(cnt << 1) + bytes([v])
                vals[i:i + cnt] = v
                i += cnt
                continue
            v = rng.integers(0, 1 << bw, size=cnt, dtype=np.uint8)
            bits = np.unpackbits(v.reshape(-1, 1), axis=1, bitorder="little")[:, :bw].reshape(-1)
            out += uvarint(((cnt // 8) << 1) | 1) + np.packbits(bits, bitorder="little").tobytes()
            vals[i:i + cnt] = v
            i += cnt
        else:
            cnt = min(n - i, int(rng.geometric(1 / 4096)))
            v = int(rng.integers(0, 1 << bw))
            out += uvarint(cnt << 1) + bytes([v])
            vals[i:i + cnt] = v
            i += cnt
    return bytes(out), vals


def bitpacked_levels_page(rng, n, bw):
    v = rng.integers(0, 1 << bw, size=n, dtype=np.uint8)
    bits = np.unpackbits(v.reshape(-1, 1), axis=1, bitorder="big")[:, 8 - bw:].reshape(-1)
    return np.packbits(bits, bitorder="big").tobytes(), v


def strings(rng, n):
    heads = [b"https://www.example.com/catalog/items/", b"https://cdn.example.org/static/img/", b"http://a.io/",
             b"https://api.example.net/v2/accounts/transactions/history/", b"ftp://files.example.com/pub/"]
    return sorted(heads[int(h)] + b"%08d" % int(t) for h, t in zip(rng.integers(0, 5, size=n), rng.integers(0, 10**8, size=n)))


def dlba_page(vals):
    lens = np.array([len(v) for v in vals], np.int64)
    return bench.delta_page_int64(lens) + b"".join(vals)


def dba_page(vals):
    prefix = np.zeros(len(vals), np.int64)
    suffix = []
    last = b""
    for k, v in enumerate(vals):
        m = min(len(v), len(last))
        p = 0
        while p < m and v[p] == last[p]:
            p += 1
        prefix[k] = p
        suffix.append(v[p:])
        last = v
    return bench.delta_page_int64(prefix) + bench.delta_page_int64(np.array([len(s) for s in suffix], np.int64)) + b"".join(suffix)


def plain_ba_page(vals):
    return b"".join(len(v).to_bytes(4, "little") + v for v in vals)


ONLY = os.environ.get("PQD_RATES_ONLY", "")  # run only the lines whose name contains this


def run(name, pages_bytes, replicate, encoding, kind, bit_width, out_bytes_per_page, offsets_per_page, expect, steps=20,
        dictionary=None, num_values=-1, dictionary_offsets=None):
    if ONLY and ONLY not in name:
        return
    dict_t = torch.from_numpy(dictionary.copy()).to(DEV) if dictionary is not None else None
    dict_off_t = torch.from_numpy(dictionary_offsets.copy()).to(DEV) if dictionary_offsets is not None else None
    uniq = len(pages_bytes)
    slot = max((len(p) + 255) // 256 * 256 for p in pages_bytes) + 256
    n_pages = uniq * replicate
    host = np.zeros(uniq * slot, np.uint8)
    for k, p in enumerate(pages_bytes):
        host[k * slot: k * slot + len(p)] = np.frombuffer(p, np.uint8)
    src = torch.from_numpy(host).to(DEV).repeat(replicate)
    oslot = (out_bytes_per_page + 255) // 256 * 256 + 256
    dst = torch.empty(n_pages * oslot, dtype=torch.uint8, device=DEV)
    off = torch.empty(n_pages * (offsets_per_page + 8), dtype=torch.int32, device=DEV) if offsets_per_page else None
    pages = []
    for i in range(n_pages):
        p = pkg.Page()
        p.src, p.src_len = src.data_ptr() + i * slot, len(pages_bytes[i % uniq])
        p.dst, p.dst_cap = dst.data_ptr() + i * oslot, oslot - 256
        if off is not None:
            p.offsets, p.offsets_cap = off.data_ptr() + i * (offsets_per_page + 8) * 4, offsets_per_page + 8
        p.num_values = num_values
        p.encoding, p.kind, p.bit_width = encoding, kind, bit_width
        if dict_t is not None:
            p.dict, p.dict_len = dict_t.data_ptr(), dictionary.size
        if dict_off_t is not None:
            p.dict_offsets, p.dict_len = dict_off_t.data_ptr(), dictionary_offsets.size - 1
        pages.append(p)
    batch = pkg.Batch(ctx, pages)
    batch.launch(stream)
    res = batch.finish(stream)
    bad = [(i, res[i].status, res[i].status_arg) for i in range(n_pages) if res[i].status != 0]
    assert not bad, (name, bad[:3])
    for i in (0, n_pages - 1) if not os.environ.get("PQD_RATES_NOCHECK") else ():
        got = dst[i * oslot: i * oslot + res[i].out_bytes].cpu().numpy()
        want = expect[i % uniq]
        assert got[: want.size].tobytes() == want.tobytes(), f"{name}: page {i} decoded wrong"

    class W:
        def launch(self, s):
            batch.launch(s)
    wall, kms = bench.time_steps(S, torch, W(), stream, steps, 3, None)
    t = float(np.mean(kms)) / 1e3
    enc_b = sum(len(pages_bytes[i % uniq]) for i in range(n_pages)) + (dictionary.nbytes if dictionary is not None else 0)
    dec_b = sum(int(res[i].out_bytes) + (4 * (int(res[i].out_values) + 1) if off is not None else 0) for i in range(n_pages))
    print(f"{name:34s} {n_pages:5d} pages  {t*1e6:9.1f} us  decoded {dec_b/t/1e9:8.1f} GB/s  algorithmic {(enc_b+dec_b)/t/1e9:8.1f} GB/s  "
          f"frac {(enc_b+dec_b)/t/8e12:.3f}", flush=True)
    batch.close()
    return t


def main():
    rng = np.random.default_rng(0xC3)
    E, K = pkg.ENC, pkg.KIND
    n = 262144
    for bw in (1, 3):
        pg = [rle_levels_page(rng, 4 * n, bw) for _ in range(8)]
        run(f"RLE levels bw{bw} (1 Mi per page)", [p for p, _ in pg], 8, E.RLE, K.LEVELS, bw, 4 * n, 0, [v for _, v in pg])
    for bw in (1, 3):  # the levels of a 1 MiB page of int64 values
        pg = [rle_levels_page(rng, 131072, bw) for _ in range(32)]
        run(f"RLE levels bw{bw} (128 Ki per page)", [p for p, _ in pg], 16, E.RLE, K.LEVELS, bw, 131072, 0, [v for _, v in pg])
        for rep in (1, 2, 4, 8, 20, 24, 32, 64) if os.environ.get("PQD_RATES_PAGE_COUNTS") else ():
            run(f"RLE levels bw{bw} (128 Ki per page) batch of {32 * rep}", [p for p, _ in pg], rep, E.RLE, K.LEVELS, bw, 131072, 0, [v for _, v in pg])
    for bw in (1, 3):
        pg = [rle_levels_page(rng, 32768, bw) for _ in range(64)]
        run(f"RLE levels bw{bw} (32 Ki per page)", [p for p, _ in pg], 32, E.RLE, K.LEVELS, bw, 32768, 0, [v for _, v in pg])
    pg = [bitpacked_levels_page(rng, 4 * n, 3) for _ in range(8)]
    run("BIT_PACKED levels bw3 (1 Mi)", [p for p, _ in pg], 8, E.BIT_PACKED, K.LEVELS, 3, 4 * n + 8, 0, [v for _, v in pg])
    f64 = [rng.standard_normal(n) for _ in range(8)]
    run("BYTE_STREAM_SPLIT double (256 Ki)", [np.ascontiguousarray(v.view(np.uint8).reshape(-1, 8).T).tobytes() for v in f64], 32,
        E.BYTE_STREAM_SPLIT, K.DOUBLE, 0, 8 * n, 0, [v.view(np.uint8) for v in f64])
    f32 = [rng.standard_normal(2 * n).astype(np.float32) for _ in range(8)]
    run("BYTE_STREAM_SPLIT float (512 Ki)", [np.ascontiguousarray(v.view(np.uint8).reshape(-1, 4).T).tobytes() for v in f32], 32,
        E.BYTE_STREAM_SPLIT, K.FLOAT, 0, 8 * n, 0, [v.view(np.uint8) for v in f32])
    i64 = [rng.integers(-2**62, 2**62, size=n, dtype=np.int64) for _ in range(8)]
    run("PLAIN int64 (256 Ki)", [v.tobytes() for v in i64], 32, E.PLAIN, K.INT64, 0, 8 * n, 0, [v.view(np.uint8) for v in i64])
    svals = [strings(rng, 32768) for _ in range(4)]
    joined = [np.frombuffer(b"".join(v), np.uint8) for v in svals]
    nb = max(j.size for j in joined)
    run("DELTA_LENGTH_BYTE_ARRAY urls (32 Ki)", [dlba_page(v) for v in svals], 64, E.DELTA_LENGTH_BYTE_ARRAY, K.BYTE_ARRAY, 0, nb, 32768, joined)
    run("DELTA_BYTE_ARRAY urls (32 Ki)", [dba_page(v) for v in svals], 64, E.DELTA_BYTE_ARRAY, K.BYTE_ARRAY, 0, nb, 32768, joined, steps=5)
    # FIXED_LEN_BYTE_ARRAY through DELTA_BYTE_ARRAY: sorted 16-byte keys (big-endian counters with random gaps)
    keys = [[int(x).to_bytes(16, "big") for x in np.cumsum(rng.integers(1, 1 << 40, size=65536, dtype=np.int64).astype(object)) + (1 << 100)] for _ in range(4)]
    kj = [np.frombuffer(b"".join(v), np.uint8) for v in keys]
    run("DELTA_BYTE_ARRAY fixed 16 (64 Ki)", [dba_page(v) for v in keys], 64, E.DELTA_BYTE_ARRAY, K.FIXED_LEN_BYTE_ARRAY, 16, 16 * 65536, 0, kj, steps=5)
    run("PLAIN byte arrays urls (32 Ki)", [plain_ba_page(v) for v in svals], 64, E.PLAIN, K.BYTE_ARRAY, 0, nb, 32768, joined, steps=5)
    # the same kind of strings through a dictionary of 4096 of them: indexes of 12 bits, one bit-packed run per page
    words = sorted(set(strings(rng, 6000)))[:4096]
    dict_bytes = np.frombuffer(b"".join(words), np.uint8)
    dict_offsets = np.concatenate([[0], np.cumsum([len(w) for w in words])]).astype(np.uint32)
    idx = [rng.integers(0, 4096, size=32768) for _ in range(4)]
    def index_page(ix):
        bits = ((ix[:, None] >> np.arange(12)) & 1).astype(np.uint8).reshape(-1)
        return bytes([12]) + uvarint(((ix.size // 8) << 1) | 1) + np.packbits(bits, bitorder="little").tobytes()
    looked = [np.frombuffer(b"".join(words[int(i)] for i in ix), np.uint8) for ix in idx]
    run("RLE_DICTIONARY strings (32 Ki)", [index_page(ix) for ix in idx], 64, E.RLE_DICTIONARY, K.BYTE_ARRAY, 0,
        max(l.size for l in looked), 32768, looked, steps=5, dictionary=dict_bytes, dictionary_offsets=dict_offsets, num_values=32768)
    # and through a dictionary of 1024 (47 KB: it fits LDS), 10-bit indexes
    words1k = words[::4]
    d1k = np.frombuffer(b"".join(words1k), np.uint8)
    o1k = np.concatenate([[0], np.cumsum([len(w) for w in words1k])]).astype(np.uint32)
    idx1k = [rng.integers(0, 1024, size=32768) for _ in range(4)]
    def index_page10(ix):
        bits = ((ix[:, None] >> np.arange(10)) & 1).astype(np.uint8).reshape(-1)
        return bytes([10]) + uvarint(((ix.size // 8) << 1) | 1) + np.packbits(bits, bitorder="little").tobytes()
    looked1k = [np.frombuffer(b"".join(words1k[int(i)] for i in ix), np.uint8) for ix in idx1k]
    run("RLE_DICTIONARY strings, 1 Ki entries (32 Ki)", [index_page10(ix) for ix in idx1k], 64, E.RLE_DICTIONARY, K.BYTE_ARRAY, 0,
        max(l.size for l in looked1k), 32768, looked1k, steps=5, dictionary=d1k, dictionary_offsets=o1k, num_values=32768)
    # the same strings cut into many small pages: the byte-array walkers run one wavefront per page
    small = [strings(rng, 4096) for _ in range(16)]
    sj = [np.frombuffer(b"".join(v), np.uint8) for v in small]
    snb = max(j.size for j in sj)
    run("DELTA_LENGTH_BYTE_ARRAY urls (4 Ki)", [dlba_page(v) for v in small], 128, E.DELTA_LENGTH_BYTE_ARRAY, K.BYTE_ARRAY, 0, snb, 4096, sj)
    run("DELTA_BYTE_ARRAY urls (4 Ki)", [dba_page(v) for v in small], 128, E.DELTA_BYTE_ARRAY, K.BYTE_ARRAY, 0, snb, 4096, sj, steps=5)
    run("PLAIN byte arrays urls (4 Ki)", [plain_ba_page(v) for v in small], 128, E.PLAIN, K.BYTE_ARRAY, 0, snb, 4096, sj, steps=5)
    # and into very many very small ones
    tiny = [strings(rng, 512) for _ in range(64)]
    tj = [np.frombuffer(b"".join(v), np.uint8) for v in tiny]
    tnb = max(j.size for j in tj)
    run("PLAIN byte arrays urls (512)", [plain_ba_page(v) for v in tiny], 256, E.PLAIN, K.BYTE_ARRAY, 0, tnb, 512, tj, steps=5)
    for per in (128, 32) if os.environ.get("PQD_RATES_PAGE_COUNTS") else ():  # pages of 6 KB and 1.5 KB
        t2 = [strings(rng, per) for _ in range(64)]
        t2j = [np.frombuffer(b"".join(v), np.uint8) for v in t2]
        run(f"PLAIN byte arrays urls ({per})", [plain_ba_page(v) for v in t2], 512, E.PLAIN, K.BYTE_ARRAY, 0, max(j.size for j in t2j), per, t2j, steps=5)


if __name__ == "__main__":
    main()
