"""The wide-table workload of the benchmark (BASELINE.json configs[4]): 64 columns x 64 row
groups of mixed encodings, 1 Mi values per column chunk in pages of 131072 values, decoded with
row group g on rank g mod N (parquet-go_amd/sharding.py) and no collective on the data path.

Column c has shape c mod 6:
  0  RLE_DICTIONARY int64, 64 Ki-entry dictionary, uniform indexes (index width 16)     configs[0]
  1  DELTA_BINARY_PACKED int64 timestamps, blocks of 128 / 4 miniblocks                 configs[1]
  2  optional int32 column: RLE definition levels (width 1, as parquet-go's writer
     encodes them) + RLE_DICTIONARY int32 values of the present slots (index width 12)  configs[2]
  3  BYTE_STREAM_SPLIT float64                                                          configs[3]
  4  DELTA_LENGTH_BYTE_ARRAY URL-like strings                                           configs[3]
  5  DELTA_BYTE_ARRAY, the same strings (sorted: long shared prefixes)                  configs[3]

Every page of the table is a physically distinct copy in HBM (a few distinct seeded pages per
shape are generated on the host and tiled), so the encoded input is really read from memory.
`python bench.py --config wide` runs it; tests/test_gpu_wide.py checks a small table page by
page against the CPU checker, tests/test_sharding_gloo.py the deal of the pages over ranks.
"""
import numpy as np

import bench

PAGE_VALUES = 131072
SHAPES = ("rle_dict_int64", "delta_int64", "optional_int32", "bss_double", "dlba_strings", "dba_strings")


# ----------------------------------------------------------------------------- page generators

def pack_lsb(values, bw):
    """values (n,) unsigned, n a multiple of 8 -> bytes, fields of bw bits, LSB first."""
    if bw == 0:
        return b""
    bits = np.unpackbits(values.astype("<u4").view(np.uint8).reshape(-1, 4), axis=1, bitorder="little")[:, :bw]
    return np.packbits(bits.reshape(-1), bitorder="little").tobytes()


def bitpacked_index_page(idx, bw):
    """RLE_DICTIONARY page of indexes without repeats: width byte + one bit-packed run (rle/dictionary.go:23-28)."""
    n = (idx.size + 7) // 8 * 8
    v = np.zeros(n, np.int64)
    v[: idx.size] = idx
    return bytes([bw]) + bytes(bench._uvarint(((n // 8) << 1) | 1)) + pack_lsb(v, bw)


def level_values(rng, n):
    """Definition levels of an optional column: runs half of mean 4, half of mean 4096; about 30 % null."""
    out = np.empty(n, np.uint8)
    i = 0
    while i < n:
        k = 4096
        lens = np.where(rng.random(k) < 0.5, rng.geometric(1 / 4, size=k), rng.geometric(1 / 4096, size=k))
        vals = (rng.random(k) < 0.7).astype(np.uint8)
        rep = np.repeat(vals, lens)
        m = min(rep.size, n - i)
        out[i:i + m] = rep[:m]
        i += m
    return out


def levels_page_go(vals, bw):
    """encodeBytes of the reference writer (rle.go:141-201) for a multiple of eight values."""
    assert vals.size % 8 == 0 and bw > 0
    words = vals.view("<u8")
    bcast = (words & np.uint64(0xff)) * np.uint64(0x0101010101010101)
    W = words.size
    uniform = words == bcast
    change = np.nonzero(words[1:] != words[:-1])[0] + 1
    stop = np.nonzero(words[1:] == bcast[:-1])[0] + 1
    out = bytearray()
    i = 0
    while i < W:
        if uniform[i]:
            k = np.searchsorted(change, i + 1)
            j = int(change[k]) if k < change.size else W
            out += bench._uvarint((8 * (j - i)) << 1) + bytes([int(vals[8 * i])])
        else:
            k = np.searchsorted(stop, i + 1)
            j = int(stop[k]) if k < stop.size else W
            out += bench._uvarint(((j - i) << 1) | 1) + pack_lsb(vals[8 * i: 8 * j], bw)
        i = j
    return bytes(out)


def strings(rng, n):
    heads = [b"https://www.example.com/catalog/items/", b"https://cdn.example.org/static/img/", b"http://a.io/",
             b"https://api.example.net/v2/accounts/transactions/history/", b"ftp://files.example.com/pub/"]
    return sorted(heads[int(h)] + b"%08d" % int(t) for h, t in zip(rng.integers(0, 5, size=n), rng.integers(0, 10**8, size=n)))


def dlba_page(vals):
    return bench.delta_page_int64(np.array([len(v) for v in vals], np.int64)) + b"".join(vals)


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


class Part:
    """One work item of a page: encoded bytes, how to decode them, what must come out."""

    def __init__(self, data, encoding, kind, bit_width, num_values, out_bytes, expect, offsets=0, dictionary=None, expect_offsets=None):
        self.data, self.encoding, self.kind, self.bit_width, self.num_values = data, encoding, kind, bit_width, num_values
        self.out_bytes, self.expect, self.offsets, self.dictionary, self.expect_offsets = out_bytes, expect, offsets, dictionary, expect_offsets


def shape_pages(pkg, shape, seed, page_values=PAGE_VALUES, unique=4):
    """`unique` distinct pages of a shape; a page is a list of Parts (levels + values for the optional column).
    Returns (pages, dictionary or None)."""
    E, K = pkg.ENC, pkg.KIND
    rng = np.random.default_rng(0xC5 * 1000 + seed)
    n = page_values
    out, dictionary = [], None
    if shape == "rle_dict_int64":
        dictionary = rng.integers(-2**62, 2**62, size=65536, dtype=np.int64)
        for _ in range(unique):
            idx = rng.integers(0, 65536, size=n, dtype=np.int64)
            out.append([Part(bitpacked_index_page(idx, 16), E.RLE_DICTIONARY, K.INT64, 0, n, 8 * n, dictionary[idx].view(np.uint8), dictionary=dictionary)])
    elif shape == "delta_int64":
        t0 = 1_600_000_000_000_000_000
        for _ in range(unique):
            ts = t0 + np.cumsum(rng.integers(1_000_000, 1_004_096, size=n, dtype=np.int64))
            t0 = int(ts[-1])
            out.append([Part(bench.delta_page_int64(ts), E.DELTA_BINARY_PACKED, K.INT64, 0, n, 8 * n, ts.view(np.uint8))])
    elif shape == "optional_int32":
        dictionary = rng.integers(-2**31, 2**31 - 1, size=4096, dtype=np.int64).astype(np.int32)
        for _ in range(unique):
            lv = level_values(rng, n)
            present = int(lv.sum())
            idx = rng.integers(0, 4096, size=present, dtype=np.int64)
            out.append([Part(levels_page_go(lv, 1), E.RLE, K.LEVELS, 1, n, n, lv),
                        Part(bitpacked_index_page(idx, 12), E.RLE_DICTIONARY, K.INT32, 0, present, 4 * present, dictionary[idx].view(np.uint8),
                             dictionary=dictionary)])
    elif shape == "bss_double":
        for _ in range(unique):
            v = rng.standard_normal(n)
            out.append([Part(np.ascontiguousarray(v.view(np.uint8).reshape(-1, 8).T).tobytes(), E.BYTE_STREAM_SPLIT, K.DOUBLE, 0, -1, 8 * n, v.view(np.uint8))])
    else:
        for _ in range(unique):
            vals = strings(rng, n)
            joined = np.frombuffer(b"".join(vals), np.uint8)
            offs = np.concatenate([[0], np.cumsum([len(v) for v in vals])]).astype(np.uint32)
            enc = dlba_page(vals) if shape == "dlba_strings" else dba_page(vals)
            code = E.DELTA_LENGTH_BYTE_ARRAY if shape == "dlba_strings" else E.DELTA_BYTE_ARRAY
            out.append([Part(enc, code, K.BYTE_ARRAY, 0, -1, joined.size, joined, offsets=n + 1, expect_offsets=offs)])
    return out, dictionary


# ----------------------------------------------------------------------------- the table and its deal

def table_cells(n_columns, n_row_groups):
    """Every column chunk of the table: (row group, column, shape)."""
    return [(g, c, SHAPES[c % len(SHAPES)]) for g in range(n_row_groups) for c in range(n_columns)]


def cells_of(rank, world, n_columns, n_row_groups):
    """The column chunks rank `rank` decodes: those of row groups g with g mod world == rank."""
    import __graft_entry__ as entry
    entry.load_package()
    from parquet_go_amd import sharding as S
    mine = set(S.row_groups_of(rank, world, n_row_groups))
    return [cell for cell in table_cells(n_columns, n_row_groups) if cell[0] in mine]


class WideWorkload:
    """This rank's share of the table, resident in HBM, as one prepared batch."""

    def __init__(self, pkg, ctx, device, rank, world, n_columns=64, n_row_groups=64, values_per_chunk=1 << 20, page_values=PAGE_VALUES, unique=4):
        import torch
        self.torch, self.pkg = torch, pkg
        dev = f"cuda:{device}"
        cells = cells_of(rank, world, n_columns, n_row_groups)
        pages_per_chunk = max(1, values_per_chunk // page_values)
        shapes = {}
        for s, name in enumerate(SHAPES):
            pages, dictionary = shape_pages(pkg, name, s, page_values, unique)
            shapes[name] = {"pages": pages, "dictionary": dictionary,
                            "dict_t": None if dictionary is None else torch.from_numpy(dictionary.copy()).to(dev)}
        # sizes first: one region for the encoded bytes, one for the output, one for offsets
        src_bytes = dst_bytes = off_entries = 0
        plan = []  # (cell, page number in chunk, unique page, part number)
        for (g, c, name) in cells:
            for q in range(pages_per_chunk):
                u = (g + c + q) % unique
                for k, part in enumerate(shapes[name]["pages"][u]):
                    plan.append((g, c, name, q, u, k, src_bytes, dst_bytes, off_entries))
                    src_bytes += (len(part.data) + 255) // 256 * 256
                    dst_bytes += (part.out_bytes + 255) // 256 * 256
                    off_entries += (part.offsets + 63) // 64 * 64
        self.src = torch.empty(src_bytes + 256, dtype=torch.uint8, device=dev)
        self.dst = torch.empty(dst_bytes + 256, dtype=torch.uint8, device=dev)
        self.off = torch.empty(off_entries + 64, dtype=torch.int32, device=dev)
        # the distinct pages once on the device, then copied to every place they occupy
        staged = {}
        for name, sh in shapes.items():
            for u, page in enumerate(sh["pages"]):
                for k, part in enumerate(page):
                    staged[(name, u, k)] = torch.from_numpy(np.frombuffer(part.data, np.uint8).copy()).to(dev)
        items, self.parts = [], []
        self.encoded_bytes = self.decoded_bytes = self.values = 0
        for (g, c, name, q, u, k, s_at, d_at, o_at) in plan:
            part = shapes[name]["pages"][u][k]
            t = staged[(name, u, k)]
            self.src[s_at: s_at + t.numel()].copy_(t)
            p = pkg.Page()
            p.src, p.src_len = self.src.data_ptr() + s_at, len(part.data)
            p.dst, p.dst_cap = self.dst.data_ptr() + d_at, (part.out_bytes + 255) // 256 * 256
            if part.offsets:
                p.offsets, p.offsets_cap = self.off.data_ptr() + 4 * o_at, part.offsets
            p.num_values = part.num_values
            p.encoding, p.kind, p.bit_width = part.encoding, part.kind, part.bit_width
            if part.dictionary is not None:
                p.dict, p.dict_len = shapes[name]["dict_t"].data_ptr(), part.dictionary.size
            items.append(p)
            self.parts.append((g, c, name, q, part, d_at, o_at))
            self.encoded_bytes += len(part.data)
            self.decoded_bytes += part.out_bytes + 4 * part.offsets
            self.values += page_values if k == len(shapes[name]["pages"][u]) - 1 else 0
        self.dict_bytes = sum(sh["dictionary"].nbytes for sh in shapes.values() if sh["dictionary"] is not None)
        self.n_items = len(items)
        self.n_cells = len(cells)
        self.shapes = shapes
        self.batch = pkg.Batch(ctx, items)

    @property
    def algorithmic_bytes(self):
        return self.encoded_bytes + self.decoded_bytes + self.dict_bytes

    def launch(self, stream):
        self.batch.launch(stream)

    def check(self, stream, sample=97):
        """Statuses of every item, and the bytes of every `sample`-th one against the generator's values."""
        res = self.batch.finish(stream)
        bad = [(i, res[i].status, res[i].status_arg) for i in range(self.n_items) if res[i].status != 0]
        if bad:
            raise RuntimeError(f"decode failed on {len(bad)} items, first {bad[0]}")
        for i in list(range(0, self.n_items, sample)) + [self.n_items - 1]:
            g, c, name, q, part, d_at, o_at = self.parts[i]
            got = self.dst[d_at: d_at + part.out_bytes].cpu().numpy()
            if got.tobytes() != np.ascontiguousarray(part.expect).tobytes():
                raise RuntimeError(f"row group {g} column {c} ({name}) page {q}: decoded wrong")
            if part.expect_offsets is not None:
                o = self.off[o_at: o_at + part.offsets].cpu().numpy().view(np.uint32)
                if not np.array_equal(o, part.expect_offsets):
                    raise RuntimeError(f"row group {g} column {c} ({name}) page {q}: offsets wrong")
        return res
