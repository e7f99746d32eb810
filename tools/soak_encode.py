"""Differential soak of the page encoders on the GPU box: random values of random shapes through
pqd_encode_batch, every page compared byte for byte with the CPU checker's encoder (which the reference's
known-answer tests pin), and decoded again on the GPU.
    PQD_SOAK_SECONDS=60 python tools/soak_encode.py [seed]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import __graft_entry__ as entry  # noqa: E402

pkg = entry.load_package()
from oracle import pqoracle as O  # noqa: E402
from parquet_go_amd.runtime import encode_pages  # noqa: E402

E, K = pkg.ENC, pkg.KIND


def runs(rng, n, top, means):
    """n values below `top` in stretches whose lengths come from a mix of geometric distributions."""
    out = np.empty(n, np.int64)
    at = 0
    while at < n:
        m = min(n - at, int(rng.geometric(1.0 / means[int(rng.integers(0, len(means)))])))
        if rng.random() < 0.5:
            out[at: at + m] = rng.integers(0, top)
        else:
            out[at: at + m] = rng.integers(0, top, size=m)
        at += m
    return out


def strings(rng, n):
    kind = int(rng.integers(0, 4))
    if kind == 0:
        heads = [bytes(rng.integers(97, 123, size=int(k), dtype=np.uint8)) for k in rng.integers(0, 60, size=4)]
        vals = [heads[int(h)] + b"%06d" % int(t) for h, t in zip(rng.integers(0, 4, size=n), rng.integers(0, 10**6, size=n))]
        if rng.random() < 0.7:
            vals.sort()
    elif kind == 1:
        vals = [bytes(rng.integers(0, 256, size=int(k), dtype=np.uint8)) for k in rng.integers(0, 24, size=n)]
    elif kind == 2:
        base = bytes(rng.integers(0, 256, size=2000, dtype=np.uint8))
        vals = [base[: int(k)] for k in rng.integers(0, 2000, size=n)]
    else:
        vals = [b""] * n
    data = np.frombuffer(b"".join(vals), np.uint8)
    offs = np.concatenate([[0], np.cumsum([len(v) for v in vals])]).astype(np.uint32)
    return data, offs, vals


def make(rng):
    """One job and what the checker makes of the same values: (job tuple for encode_pages, expected bytes, label)."""
    which = int(rng.integers(0, 9))
    n = int(rng.integers(0, 3)) if rng.random() < 0.05 else int(rng.integers(1, 60000))
    if which == 0:
        bw = int(rng.integers(1, 9))
        v = runs(rng, n, 1 << bw, [2, 9, 40, 3000]).astype(np.uint8)
        return (E.RLE, K.LEVELS, v, bw), bytes(O.rle_encode_levels(v, bw)), f"levels bw{bw} n{n}"
    if which == 1:
        alphabet = np.array([0, 0, 0xFF, 0xFF, 0x55, 1, 0xFE, 0x80], np.uint8)[: int(rng.integers(1, 9))]
        v = alphabet[runs(rng, n, alphabet.size, [1, 3, 20, 5000])]
        return (E.RLE, K.BOOLEAN, v, 0), bytes(O.rle_encode_boolean(v)), f"boolean n{n}"
    if which == 2:
        bw = int(rng.integers(0, 33))
        v = runs(rng, n, 1 << bw, [3, 8, 70, 2000]).astype(np.int64).astype(np.uint32).view(np.int32)
        return (E.RLE, K.INT32, v, bw), bytes(O.rle_encode_int32(v, bw)), f"rle int32 bw{bw} n{n}"
    if which == 3:
        top = int(2 ** rng.uniform(0, 31))
        v = runs(rng, n, top, [3, 8, 70, 2000]).astype(np.int32)
        return (E.RLE_DICTIONARY, K.INT32, v, 0), bytes(O.rle_dict_encode_int32(v)), f"rle_dictionary top{top} n{n}"
    if which == 4:
        w = int(rng.integers(0, 64))
        with np.errstate(over="ignore"):
            step = rng.integers(-(1 << w), (1 << w), size=n, dtype=np.int64) if w < 63 else rng.integers(-2**63, 2**63 - 1, size=n, dtype=np.int64)
            v = np.cumsum(step) + rng.integers(-2**62, 2**62)
        return (E.DELTA_BINARY_PACKED, K.INT64, v, 0), bytes(O.delta_encode_int64(v)), f"delta int64 w{w} n{n}"
    if which == 5:
        w = int(rng.integers(0, 31))
        with np.errstate(over="ignore"):
            v = (np.cumsum(rng.integers(-(1 << w), (1 << w) + 1, size=n, dtype=np.int64)) & 0xFFFFFFFF).astype(np.uint32).view(np.int32)
        return (E.DELTA_BINARY_PACKED, K.INT32, v, 0), bytes(O.delta_encode_int32(v)), f"delta int32 w{w} n{n}"
    n = min(n, 8000)
    data, offs, vals = strings(rng, n)
    if which == 6:
        return (E.PLAIN, K.BYTE_ARRAY, data, 0, offs), bytes(O.plain_encode_byte_array(vals)), f"plain byte arrays n{n}"
    if which == 7:
        return (E.DELTA_LENGTH_BYTE_ARRAY, K.BYTE_ARRAY, data, 0, offs), bytes(O.dlba_encode(data=data, offsets=offs)), f"dlba n{n}"
    return (E.DELTA_BYTE_ARRAY, K.BYTE_ARRAY, data, 0, offs), bytes(O.dba_encode(data=data, offsets=offs)), f"dba n{n}"


def main():
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    rng = np.random.default_rng(seed)
    ctx = pkg.Context(0)
    t_end = time.time() + float(os.environ.get("PQD_SOAK_SECONDS", "60"))
    pages = batches = bad = 0
    kinds = {}
    while time.time() < t_end:
        jobs, wants, labels = [], [], []
        for _ in range(24):
            j, w, l = make(rng)
            jobs.append(j); wants.append(w); labels.append(l)
        outs = encode_pages(ctx, jobs, misalign=int(rng.integers(0, 16)))
        for (status, arg, page, raw), want, label in zip(outs, wants, labels):
            pages += 1
            kinds[label.split(" n")[0].split(" bw")[0].split(" w")[0].split(" top")[0]] = kinds.get(label.split(" n")[0].split(" bw")[0].split(" w")[0].split(" top")[0], 0) + 1
            if status != 0 or page != want or not (raw[len(page):] == 0xA5).all():
                bad += 1
                if bad <= 10:
                    print("DIFFERENCE", label, "status", status, arg, "bytes", len(page), "want", len(want), flush=True)
        batches += 1
        if batches % 50 == 0:
            print(f"... {pages} pages, {bad} differences", flush=True)
    print(f"encode soak seed {seed}: {pages} pages in {batches} batches, {bad} differences; {kinds}")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
