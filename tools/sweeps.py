"""Variants of the measured configurations, device-resident, one batch each:

  DELTA_BINARY_PACKED int64   constant delta (width 0), full-range deltas (width 64), widths in between; int32
  RLE_DICTIONARY int64        index widths 8 / 12 / 16 / 20 (dictionary of 2**width entries), Zipf indices in sorted runs
  RLE int32 (index unpack)    one bit-packed run per page, widths 1..20
  RLE levels                  the same level values encoded the way parquet-go's writer does (rle.go:141-201:
                              run-length only for 8-aligned groups of eight equal values, the rest bit-packed)
                              and with runs of arbitrary length (parquet-mr style)

Pages come from numpy generators written here (the formats are simple enough to emit directly);
results are checked against the generators' values.  Prints one line per variant like
tools/misc_rates.py does."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import misc_rates as M  # noqa: E402
import bench  # noqa: E402

pkg = M.pkg
E, K = pkg.ENC, pkg.KIND
uvarint = bench._uvarint


def pack_lsb(values, bw):
    """values (n,) unsigned, n a multiple of 8 -> bytes, fields of bw bits, LSB first."""
    if bw == 0:
        return b""
    v = values.astype("<u4")
    bits = np.unpackbits(v.view(np.uint8).reshape(-1, 4), axis=1, bitorder="little")[:, :bw]
    return np.packbits(bits.reshape(-1), bitorder="little").tobytes()


# ------------------------------------------------------------------ levels --

def level_values(rng, n, bw):
    """Run lengths: half the runs of mean 4, half of mean 4096; values uniform in the width."""
    out = np.empty(n, np.uint8)
    i = 0
    while i < n:
        k = 4096
        short = rng.random(k) < 0.5
        lens = np.where(short, rng.geometric(1 / 4, size=k), rng.geometric(1 / 4096, size=k))
        vals = rng.integers(0, 1 << bw, size=k, dtype=np.uint8)
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
    change = np.nonzero(words[1:] != words[:-1])[0] + 1           # j with words[j] != words[j-1]
    stop = np.nonzero(words[1:] == bcast[:-1])[0] + 1             # j with words[j] == broadcast(words[j-1])
    out = bytearray()
    i = 0
    while i < W:
        if uniform[i]:
            k = np.searchsorted(change, i + 1)
            j = int(change[k]) if k < change.size else W
            out += uvarint((8 * (j - i)) << 1) + bytes([int(vals[8 * i])])
        else:
            k = np.searchsorted(stop, i + 1)
            j = int(stop[k]) if k < stop.size else W
            out += uvarint(((j - i) << 1) | 1) + pack_lsb(vals[8 * i: 8 * j], bw)
        i = j
    return bytes(out)


def levels_page_mr(vals, bw):
    """Runs of arbitrary length: a repeat of eight or more that starts on a group boundary of the
    current bit-packed run becomes a run-length run, everything else goes into bit-packed groups."""
    n = vals.size
    starts = np.concatenate([[0], np.nonzero(vals[1:] != vals[:-1])[0] + 1, [n]])
    run_end = np.empty(n, np.int64)  # end of the maximal equal run holding each position
    run_end[:] = np.repeat(starts[1:], np.diff(starts))
    out = bytearray()
    i = 0
    while i < n:
        if run_end[i] - i >= 8:
            out += uvarint(int(run_end[i] - i) << 1) + bytes([int(vals[i])])
            i = int(run_end[i])
            continue
        j = i
        while j < n and run_end[j] - j < 8:
            j += 8
        j = min(j, n)
        chunk = vals[i:j]
        groups = (chunk.size + 7) // 8
        if chunk.size % 8:
            chunk = np.concatenate([chunk, np.zeros(groups * 8 - chunk.size, np.uint8)])
        out += uvarint((groups << 1) | 1) + pack_lsb(chunk, bw)
        i = j
    return bytes(out)


def levels(rng):
    n = 131072
    for bw in (1, 2, 3):
        vals = [level_values(rng, n, bw) for _ in range(16)]
        for style, enc in (("parquet-go", levels_page_go), ("parquet-mr", levels_page_mr)):
            pages = [enc(v, bw) for v in vals]
            per_value = sum(len(p) for p in pages) / (len(pages) * n)
            M.run(f"RLE levels bw{bw} {style} ({per_value:.3f} B/value)", pages, 32, E.RLE, K.LEVELS, bw, n, 0, vals, num_values=n)


# ----------------------------------------------------------------- indexes --

def index_unpack(rng):
    n = 262144
    for bw in range(1, 21):
        vals = [rng.integers(0, 1 << bw, size=n, dtype=np.int64).astype(np.int32) for _ in range(4)]
        pages = [uvarint(((n // 8) << 1) | 1) + pack_lsb(v, bw) for v in vals]
        M.run(f"RLE int32 unpack bw{bw:2d}", pages, 16, E.RLE, K.INT32, bw, 4 * n, 0, [v.view(np.uint8) for v in vals], num_values=n)


def rle_int32_page(idx, bw):
    """Writer-style hybrid of int32 indexes: run-length for 8-aligned groups of eight equal values."""
    groups = idx.reshape(-1, 8)
    uniform = (groups == groups[:, :1]).all(axis=1)
    out = bytearray()
    g, G = 0, groups.shape[0]
    vb = (bw + 7) // 8
    while g < G:
        h = g + 1
        if uniform[g]:
            while h < G and uniform[h] and groups[h, 0] == groups[g, 0]:
                h += 1
            out += uvarint((8 * (h - g)) << 1) + int(groups[g, 0]).to_bytes(vb, "little")
        else:
            while h < G and not uniform[h]:
                h += 1
            out += uvarint(((h - g) << 1) | 1) + pack_lsb(groups[g:h].reshape(-1), bw)
        g = h
    return bytes(out)


def dictionary(rng):
    n = 262144
    for bw in (8, 10, 12, 14, 15, 16, 17, 20):
        d = rng.integers(-2**62, 2**62, size=1 << bw, dtype=np.int64)
        idx = [rng.integers(0, 1 << bw, size=n, dtype=np.int64).astype(np.int32) for _ in range(8)]
        pages = [bytes([bw]) + uvarint(((n // 8) << 1) | 1) + pack_lsb(v, bw) for v in idx]
        M.run(f"RLE_DICTIONARY int64 uniform bw{bw:2d}", pages, 8, E.RLE_DICTIONARY, K.INT64, 0, 8 * n, 0,
              [d[v].view(np.uint8) for v in idx], dictionary=d, num_values=n)
    # Zipf(1.1) indexes, sorted inside stretches of 4096 values: long repeats of the hot entries
    d = rng.integers(-2**62, 2**62, size=65536, dtype=np.int64)
    idx = []
    for _ in range(8):
        z = np.minimum(rng.zipf(1.1, size=n) - 1, 65535).astype(np.int32)
        idx.append(np.sort(z.reshape(-1, 4096), axis=1).reshape(-1))
    pages = []
    for v in idx:
        bw = int(v.max()).bit_length()
        pages.append(bytes([bw]) + rle_int32_page(v, bw))
    runs = sum(len(p) for p in pages) / (8 * n)
    M.run(f"RLE_DICTIONARY int64 Zipf runs ({runs:.2f} B/value)", pages, 8, E.RLE_DICTIONARY, K.INT64, 0, 8 * n, 0,
          [d[v].view(np.uint8) for v in idx], dictionary=d, num_values=n)


# ------------------------------------------------------------------- delta --

DELTA_REPLICATE = int(os.environ.get("PQD_SWEEP_DELTA_REPLICATE", "64"))  # x 4 distinct pages; 512 fills the chip like the headline


def delta(rng):
    n = 131072
    t0 = 1_600_000_000_000_000_000
    shapes = [("constant delta (width 0)", lambda: t0 + 1_000_000 * np.arange(n, dtype=np.int64))]
    for bw in (1, 4, 8, 12, 16, 24, 32, 48):
        shapes.append((f"width {bw}", lambda bw=bw: t0 + np.cumsum(rng.integers(0, 1 << bw, size=n, dtype=np.int64))))
    shapes.append(("full-range values (width 64)", lambda: rng.integers(-2**63, 2**63 - 1, size=n, dtype=np.int64)))
    for name, gen in shapes:
        vals = [gen() for _ in range(4)]
        with np.errstate(over="ignore"):
            pages = [bench.delta_page_int64(v) for v in vals]
        M.run(f"DELTA int64 {name}", pages, DELTA_REPLICATE, E.DELTA_BINARY_PACKED, K.INT64, 0, 8 * n, 0, [v.view(np.uint8) for v in vals])


def main():
    rng = np.random.default_rng(0xC3)
    which = sys.argv[1:] or ["levels", "unpack", "dictionary", "delta"]
    if "levels" in which:
        levels(rng)
    if "unpack" in which:
        index_unpack(rng)
    if "dictionary" in which:
        dictionary(rng)
    if "delta" in which:
        delta(rng)


if __name__ == "__main__":
    main()
