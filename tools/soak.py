"""Differential soak of the page decoders: random pages of every encoding, whole and damaged
(bytes changed, cut short, garbage appended), decoded on the GPU through the C ABI and by the CPU
checker in oracle/.  Every difference in status, status argument or decoded output is printed;
the exit code is the number of differences (capped).  Run on a GPU box:

    PQD_SOAK_SECONDS=60 python tools/soak.py [seed]

With PQD_SOAK_HOST=1 the same pages go one at a time through the host entry points
(pqd_decode_int64 and its like, by way of the codec objects of parquet-go_amd/encoding.py) instead
of a batch of device pointers: the calls a cgo binding of encoding.Encoding makes.
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

from oracle import pqoracle as O  # noqa: E402
from tests.gpu_util import ctx, pkg  # noqa: E402

LIMIT = 64 << 20  # larger outputs than this are not materialised, here or by the checker


def _bounded_retry(call, dtype, cap_bytes):
    for _ in range(48):
        dst = np.zeros(max(cap_bytes, 8), dtype=np.uint8)
        res = call(dst, cap_bytes)
        if res.status == 16 and cap_bytes < res.arg <= LIMIT:
            cap_bytes = max(int(res.arg), 2 * cap_bytes)
            continue
        return res, dst[: res.n_bytes].view(dtype) if res.n_bytes % np.dtype(dtype).itemsize == 0 else dst[: res.n_bytes]
    raise RuntimeError("checker retry loop did not converge")


def _bounded_byte_array(fn, src, cap, ocap):
    s = O._src(src)
    cap = cap or max(64, 4 * s.size)
    ocap = ocap or 64
    for _ in range(48):
        dst = np.zeros(cap, dtype=np.uint8)
        off = np.zeros(ocap, dtype=np.uint32)
        res = fn(O._p(dst), cap, O._p(off), ocap, O._p(s), s.size)
        if res.status == 16 and cap < res.arg <= LIMIT:
            cap = max(int(res.arg), 2 * cap)
            continue
        if res.status == 18 and ocap < res.arg <= LIMIT // 4:
            ocap = max(int(res.arg), 2 * ocap)
            continue
        no = res.n_values + 1 if res.status == 0 else res.n_values
        return O._mk(res, dst[: res.n_bytes], off[:no])
    raise RuntimeError("checker retry loop did not converge")


O._retry = _bounded_retry
O._byte_array = _bounded_byte_array


def strings(rng, n):
    style = rng.integers(0, 4)
    if style == 0:
        return [bytes(rng.integers(0, 256, size=int(k), dtype=np.uint8)) for k in rng.integers(0, 22, size=n)]
    if style == 1:
        heads = [b"https://www.example.com/catalog/items/", b"http://a.io/", b"ftp://files.example.com/pub/"]
        return sorted(heads[int(h)] + b"%06d" % int(t) for h, t in zip(rng.integers(0, 3, size=n), rng.integers(0, 10**6, size=n)))
    if style == 2:
        return sorted(bytes(rng.integers(97, 100, size=int(k), dtype=np.uint8)) for k in rng.integers(0, 9, size=n))
    return [bytes(rng.integers(0, 256, size=int(k), dtype=np.uint8)) for k in rng.integers(0, 700, size=max(n // 40, 1))]


def deltas(rng, n, bits):
    # deltas whose width changes from stretch to stretch, as the miniblocks of real columns do
    out = np.zeros(n, np.int64)
    i = 0
    while i < n:
        m = int(rng.integers(1, 700))
        w = int(rng.integers(0, bits))
        lo = int(rng.integers(-(1 << w), 1)) if w else 0
        hi = lo + (1 << w)
        out[i: i + m] = rng.integers(lo, hi, size=min(m, n - i), dtype=np.int64)
        i += m
    return out


def runs(rng, n, top):
    out = np.zeros(n, np.int64)
    i = 0
    while i < n:
        m = int(rng.integers(1, 9)) if rng.integers(0, 2) else int(rng.integers(1, 3000))
        if rng.integers(0, 2):
            out[i: i + m] = rng.integers(0, top + 1)
        else:
            out[i: i + m] = rng.integers(0, top + 1, size=min(m, n - i))
        i += m
    return out


def make_page(rng, P, only=None, n_max=400000):
    """(label, spec arguments, checker call) of one random page."""
    n = int(rng.choice([0, 1, 2, 7, 64, 127, 128, 129, 1000, 5000, 20000, 60000, 400000]))
    n = min(n, n_max)
    n = int(rng.integers(0, n + 1)) if rng.integers(0, 2) else n
    which = int(rng.integers(0, 18)) if only is None else int(rng.choice(only))
    if which >= 16:
        # one width for the whole page, the odd narrower miniblock aside: the shape the tile-parallel DELTA kernels take
        w = int(rng.integers(0, 34))
        d = rng.integers(0, 1 << w, size=n, dtype=np.int64) + int(rng.integers(-70, 70))
        if which == 16:
            return f"delta64-steady-w{w}", dict(encoding=P.ENC.DELTA_BINARY_PACKED, kind=P.KIND.INT64), \
                O.delta_encode_int64(np.cumsum(d)), O.delta_decode_int64, np.int64
        d = rng.integers(0, 1 << min(w, 12), size=n, dtype=np.int64) + int(rng.integers(-70, 70))
        return f"delta32-steady-w{min(w, 12)}", dict(encoding=P.ENC.DELTA_BINARY_PACKED, kind=P.KIND.INT32), \
            O.delta_encode_int32(np.cumsum(d).astype(np.int32)), O.delta_decode_int32, np.int32
    if which >= 14:
        # dictionary indexes looked up as they are unpacked (newIndexedPage + Dictionary.Lookup): the stream may be
        # shorter or longer than the page's value count, and an index may lie outside the dictionary
        dn = int(rng.choice([1, 2, 10, 300, 5000, 70000]))
        dt, kind = [(np.int64, P.KIND.INT64), (np.int32, P.KIND.INT32), (np.float64, P.KIND.DOUBLE)][int(rng.integers(0, 3))]
        d = rng.integers(-2**31, 2**31 - 1, size=dn).astype(dt)
        top = dn - 1 if rng.integers(0, 8) else dn + 3
        v = runs(rng, n, top).astype(np.int32)
        nv = int(rng.choice([-1, n, n, max(n - 5, 0), n + 9]))

        def check_fused(src, d=d, nv=nv):
            r = O.rle_dict_decode_int32(src)
            if r.status != 0:
                return r
            return O.dict_lookup(d, O.indexed_page_fixup(r.values, nv) if nv >= 0 else r.values)
        return f"dict{dn}-{np.dtype(dt).name}-nv{nv}", dict(encoding=P.ENC.RLE_DICTIONARY, kind=kind, dictionary=d, num_values=nv), \
            O.rle_dict_encode_int32(v), check_fused, np.uint8
    if which == 0:
        v = np.cumsum(deltas(rng, n, 40))
        enc = O.delta_encode_int64(v)
        return "delta64", dict(encoding=P.ENC.DELTA_BINARY_PACKED, kind=P.KIND.INT64), enc, O.delta_decode_int64, np.int64
    if which == 1:
        v = np.cumsum(deltas(rng, n, 20)).astype(np.int32)
        enc = O.delta_encode_int32(v)
        return "delta32", dict(encoding=P.ENC.DELTA_BINARY_PACKED, kind=P.KIND.INT32), enc, O.delta_decode_int32, np.int32
    if which == 2:
        v = rng.integers(-2**62, 2**62, size=n, dtype=np.int64)
        enc = O.delta_encode_int64(v)
        return "delta64-wide", dict(encoding=P.ENC.DELTA_BINARY_PACKED, kind=P.KIND.INT64), enc, O.delta_decode_int64, np.int64
    if which in (3, 4):
        bw = int(rng.integers(0, 9))
        v = runs(rng, n, (1 << bw) - 1).astype(np.uint8)
        try:
            enc = (O.rle_encode_levels if which == 3 else O.rle_encode_levels_mr)(v, bw)
        except ValueError:  # the arbitrary-run-length encoder declines some inputs
            enc = O.rle_encode_levels(v, bw)
        return f"rle-levels-bw{bw}", dict(encoding=P.ENC.RLE, kind=P.KIND.LEVELS, bit_width=bw), enc, \
            (lambda s, bw=bw: O.rle_decode_levels(s, bw)), np.uint8
    if which == 5:
        bw = int(rng.integers(0, 33))
        v = runs(rng, n, (1 << bw) - 1 if bw < 32 else 2**31 - 1).astype(np.uint32).view(np.int32)
        enc = O.rle_encode_int32(v, bw)
        return f"rle-int32-bw{bw}", dict(encoding=P.ENC.RLE, kind=P.KIND.INT32, bit_width=bw), enc, \
            (lambda s, bw=bw: O.rle_decode_int32(s, bw)), np.int32
    if which == 6:
        v = runs(rng, n, int(rng.integers(0, 70000))).astype(np.int32)
        enc = O.rle_dict_encode_int32(v)
        return "rle-dict-indexes", dict(encoding=P.ENC.RLE_DICTIONARY, kind=P.KIND.INT32), enc, O.rle_dict_decode_int32, np.int32
    if which == 7:
        bits = np.packbits(runs(rng, n, 1).astype(np.uint8), bitorder="little")
        enc = O.rle_encode_boolean(bits)
        return "rle-boolean", dict(encoding=P.ENC.RLE, kind=P.KIND.BOOLEAN, bit_width=1), enc, O.rle_decode_boolean, np.uint8
    if which == 8:
        bw = int(rng.integers(0, 9))
        v = rng.integers(0, 1 << bw, size=n, dtype=np.uint8) if bw else np.zeros(n, np.uint8)
        enc = O.bitpacked_encode_levels(v, bw)
        return f"bitpacked-bw{bw}", dict(encoding=P.ENC.BIT_PACKED, kind=P.KIND.LEVELS, bit_width=bw), enc, \
            (lambda s, bw=bw: O.bitpacked_decode_levels(s, bw)), np.uint8
    if which == 9:
        size, kind = [(4, P.KIND.FLOAT), (8, P.KIND.DOUBLE), (4, P.KIND.INT32), (8, P.KIND.INT64)][int(rng.integers(0, 4))]
        raw = rng.integers(0, 256, size=n * size, dtype=np.uint8).tobytes()
        enc = O.bss_encode(raw, size)
        return f"bss{size}", dict(encoding=P.ENC.BYTE_STREAM_SPLIT, kind=kind), enc, (lambda s, size=size: O.bss_decode(s, size)), np.uint8
    if which == 10:
        size, kind = [(4, P.KIND.INT32), (8, P.KIND.INT64), (12, P.KIND.INT96), (8, P.KIND.DOUBLE)][int(rng.integers(0, 4))]
        enc = rng.integers(0, 256, size=n * size, dtype=np.uint8).tobytes()
        return f"plain{size}", dict(encoding=P.ENC.PLAIN, kind=kind), enc, (lambda s, size=size: O.plain_decode_fixed(s, size)), np.uint8
    vals = strings(rng, min(n, 20000))
    if which == 11:
        return "plain-byte-array", dict(encoding=P.ENC.PLAIN, kind=P.KIND.BYTE_ARRAY), O.plain_encode_byte_array(vals), \
            O.plain_decode_byte_array, None
    if which == 12:
        return "dlba", dict(encoding=P.ENC.DELTA_LENGTH_BYTE_ARRAY, kind=P.KIND.BYTE_ARRAY), O.dlba_encode(vals), O.dlba_decode, None
    return "dba", dict(encoding=P.ENC.DELTA_BYTE_ARRAY, kind=P.KIND.BYTE_ARRAY), O.dba_encode(vals), O.dba_decode, None


def damage(rng, enc):
    how = int(rng.integers(0, 6))
    b = bytearray(enc)
    if how <= 1 or len(b) == 0:
        return "whole", bytes(b)
    if how == 2:
        return "cut", bytes(b[: int(rng.integers(0, len(b)))])
    if how == 3:  # a byte near the front (headers) or anywhere
        k = int(rng.integers(0, min(len(b), 12))) if rng.integers(0, 2) else int(rng.integers(0, len(b)))
        b[k] = int(rng.integers(0, 256))
        return f"byte{k}", bytes(b)
    if how == 4:
        for _ in range(int(rng.integers(2, 6))):
            b[int(rng.integers(0, len(b)))] ^= 1 << int(rng.integers(0, 8))
        return "bits", bytes(b)
    return "tail", bytes(b) + bytes(rng.integers(0, 256, size=int(rng.integers(1, 40)), dtype=np.uint8))


def main():
    P = pkg()
    from parquet_go_amd.runtime import PageSpec, decode_pages
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    budget = float(os.environ.get("PQD_SOAK_SECONDS", "60"))
    rng = np.random.default_rng(seed)
    t_end = time.time() + budget
    pages = diffs = skipped = rounds = 0
    by_status = {}
    while time.time() < t_end and diffs < 40:
        rounds += 1
        items = []
        # every fourth batch: 300 DELTA pages, above the page count up to which the tile-parallel kernels are tried
        many = rounds % 4 == 0
        for k in range(300 if many else 48):
            label, args, enc, check, dtype = make_page(rng, P, (0, 1, 2, 16, 17), 20000) if many else make_page(rng, P)
            how, src = damage(rng, enc)
            items.append((f"{label}/{how}/n{len(src)}", args, src, check, dtype))
        specs = [PageSpec(src=src, misalign=int(rng.integers(0, 16)), **args) for _, args, src, _, _ in items]
        outs = decode_pages(ctx(), specs, grow=False)
        for i, ((label, args, src, check, dtype), out) in enumerate(zip(items, outs)):
            pages += 1
            if out.status in (P.STATUS.DST_TOO_SMALL, P.STATUS.OFFSETS_TOO_SMALL):
                if out.arg > (LIMIT if out.status == P.STATUS.DST_TOO_SMALL else LIMIT // 4):
                    skipped += 1
                    continue
                out = decode_pages(ctx(), [specs[i]], grow=True)[0]
                if out.status in (P.STATUS.DST_TOO_SMALL, P.STATUS.OFFSETS_TOO_SMALL):
                    skipped += 1
                    continue
            want = check(src)
            if want.status in (P.STATUS.DST_TOO_SMALL, P.STATUS.OFFSETS_TOO_SMALL):
                # the checker stopped at an output above LIMIT that a run header announces; the GPU indexes the
                # whole page first and may name a data error further in (a Go caller, growing dst, ends there too)
                skipped += 1
                continue
            by_status[want.status] = by_status.get(want.status, 0) + 1
            bad = None
            if out.status != want.status:
                bad = f"status {out.status} (arg {out.arg}) != {want.status} (arg {want.arg})"
            elif want.status != 0 and out.arg != want.arg:
                bad = f"status {out.status}: arg {out.arg} != {want.arg}"
            elif want.status == 0:
                if dtype is None:
                    if out.out_values != want.n_values or out.data.tobytes() != want.values.tobytes() or \
                            not np.array_equal(out.offsets[: want.n_values + 1], want.offsets) or out.consumed != want.consumed:
                        bad = f"byte arrays differ ({out.out_values} / {want.n_values} values, consumed {out.consumed} / {want.consumed})"
                else:
                    got = out.data.tobytes()
                    exp = np.ascontiguousarray(want.values).tobytes()
                    if got != exp:
                        bad = f"values differ ({len(got)} / {len(exp)} bytes)"
            if bad:
                diffs += 1
                name = f"gpurun_out/soak_{seed}_{diffs}.bin"
                os.makedirs("gpurun_out", exist_ok=True)
                with open(name, "wb") as f:
                    f.write(src)
                print(f"DIFF {label} { {k: v for k, v in args.items() if k != 'dictionary'} }: {bad} -> {name}", flush=True)
        if rounds % 20 == 0:
            print(f"... {pages} pages, {diffs} differences, {skipped} too large to compare", flush=True)
    print(f"soak seed {seed}: {pages} pages in {rounds} batches, {diffs} differences, {skipped} too large to compare; "
          f"checker statuses {dict(sorted(by_status.items()))}")
    return min(diffs, 100)


class _TooBig(Exception):
    pass


class _BoundedNumpy:
    """numpy for the codec objects, with destinations above LIMIT refused: a damaged page may ask for terabytes."""

    def __getattr__(self, name):
        return getattr(np, name)

    @staticmethod
    def empty(shape, dtype=float):
        if int(np.prod(shape)) * np.dtype(dtype).itemsize > LIMIT:
            raise _TooBig()
        return np.empty(shape, dtype)


def main_host():
    P = pkg()
    from parquet_go_amd import encoding as E
    E.np = _BoundedNumpy()
    K = P.KIND
    method = {K.LEVELS: "DecodeLevels", K.BOOLEAN: "DecodeBoolean", K.INT32: "DecodeInt32", K.INT64: "DecodeInt64",
              K.INT96: "DecodeInt96", K.FLOAT: "DecodeFloat", K.DOUBLE: "DecodeDouble"}
    seed = int(sys.argv[1]) if len(sys.argv) > 1 else 1
    rng = np.random.default_rng(seed)
    t_end = time.time() + float(os.environ.get("PQD_SOAK_SECONDS", "60"))
    pages = diffs = skipped = 0
    by_status = {}
    while time.time() < t_end and diffs < 40:
        label, args, enc, check, dtype = make_page(rng, P, tuple(range(14)) + (16, 17), 60000)
        how, src = damage(rng, enc)
        want = check(src)
        if want.status in (P.STATUS.DST_TOO_SMALL, P.STATUS.OFFSETS_TOO_SMALL):
            skipped += 1
            continue
        codec = E.LookupEncoding(args["encoding"])
        codec.BitWidth = args.get("bit_width", 0)
        status = arg = 0
        values = offsets = None
        try:
            if args["kind"] == K.BYTE_ARRAY:
                values, offsets = codec.DecodeByteArray(None, src, None)
            else:
                values = getattr(codec, method[args["kind"]])(None, src)
        except _TooBig:
            skipped += 1
            continue
        except E.EncodingError as e:
            status, arg = e.status, e.arg
        pages += 1
        by_status[want.status] = by_status.get(want.status, 0) + 1
        bad = None
        if status != want.status:
            bad = f"status {status} (arg {arg}) != {want.status} (arg {want.arg})"
        elif status != 0 and arg != want.arg:
            bad = f"status {status}: arg {arg} != {want.arg}"
        elif status == 0:
            if np.ascontiguousarray(values).tobytes() != np.ascontiguousarray(want.values).tobytes():
                bad = f"values differ ({values.nbytes} / {want.values.nbytes} bytes)"
            elif offsets is not None and not np.array_equal(offsets, want.offsets):
                bad = f"offsets differ ({offsets.size} / {want.offsets.size})"
        if bad:
            diffs += 1
            name = f"gpurun_out/soak_host_{seed}_{diffs}.bin"
            os.makedirs("gpurun_out", exist_ok=True)
            with open(name, "wb") as f:
                f.write(src)
            print(f"DIFF {label}/{how}/n{len(src)}: {bad} -> {name}", flush=True)
        if pages % 1000 == 0:
            print(f"... {pages} pages, {diffs} differences, {skipped} too large to compare", flush=True)
    print(f"host soak seed {seed}: {pages} pages, one call each, {diffs} differences, {skipped} too large to compare; "
          f"checker statuses {dict(sorted(by_status.items()))}")
    return min(diffs, 100)


if __name__ == "__main__":
    sys.exit(main_host() if os.environ.get("PQD_SOAK_HOST") == "1" else main())
