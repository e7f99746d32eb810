"""A mixed job on one GPU: row groups of column chunks whose encodings cycle through the four
measured shapes -- RLE_DICTIONARY int64 (64 Ki-entry dictionary), DELTA_BINARY_PACKED int64
timestamps, RLE definition levels, BYTE_STREAM_SPLIT doubles -- 1 Mi values per chunk in pages
of 131072 values.  Decoded once with a batch per row group (what a reader that goes row group
by row group submits; on one stream, and dealt to eight streams) and once with all row groups
in one batch.  Row group g belongs to rank
g mod N in a multi-GPU run; this tool measures one rank's share.

    python tools/mixed_job.py [row_groups] [columns]      (defaults 16, 16)
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import misc_rates as M  # noqa: E402
import sweeps as W  # noqa: E402
import bench  # noqa: E402

torch = M.torch
pkg = M.pkg
E, K = pkg.ENC, pkg.KIND
PAGE = 131072
PAGES_PER_CHUNK = 8


def shapes(rng):
    """Per shape: encoded pages (distinct ones, reused by every chunk), page parameters, expected bytes."""
    out = []
    d = rng.integers(-2**62, 2**62, size=65536, dtype=np.int64)
    idx = [rng.integers(0, 65536, size=PAGE, dtype=np.int64).astype(np.int32) for _ in range(4)]
    out.append(dict(name="RLE_DICTIONARY int64", enc=E.RLE_DICTIONARY, kind=K.INT64, bw=0, elem=8, dictionary=d,
                    pages=[bytes([16]) + bench._uvarint(((PAGE // 8) << 1) | 1) + W.pack_lsb(v, 16) for v in idx],
                    expect=[d[v].view(np.uint8) for v in idx]))
    t0 = 1_600_000_000_000_000_000
    ts = [t0 + np.cumsum(rng.integers(1_000_000, 1_004_096, size=PAGE, dtype=np.int64)) for _ in range(4)]
    out.append(dict(name="DELTA_BINARY_PACKED int64", enc=E.DELTA_BINARY_PACKED, kind=K.INT64, bw=0, elem=8, dictionary=None,
                    pages=[bench.delta_page_int64(v) for v in ts], expect=[v.view(np.uint8) for v in ts]))
    lv = [W.level_values(rng, PAGE, 1) for _ in range(4)]
    out.append(dict(name="RLE levels", enc=E.RLE, kind=K.LEVELS, bw=1, elem=1, dictionary=None,
                    pages=[W.levels_page_go(v, 1) for v in lv], expect=lv))
    f64 = [rng.standard_normal(PAGE) for _ in range(4)]
    out.append(dict(name="BYTE_STREAM_SPLIT double", enc=E.BYTE_STREAM_SPLIT, kind=K.DOUBLE, bw=0, elem=8, dictionary=None,
                    pages=[np.ascontiguousarray(v.view(np.uint8).reshape(-1, 8).T).tobytes() for v in f64],
                    expect=[v.view(np.uint8) for v in f64]))
    return out


def main():
    row_groups = int(sys.argv[1]) if len(sys.argv) > 1 else 16
    columns = int(sys.argv[2]) if len(sys.argv) > 2 else 16
    rng = np.random.default_rng(0xC5)
    sh = shapes(rng)
    # device copies of the distinct encoded pages and dictionaries
    for s in sh:
        s["slot"] = max((len(p) + 255) // 256 * 256 for p in s["pages"])
        host = np.zeros(len(s["pages"]) * s["slot"], np.uint8)
        for k, p in enumerate(s["pages"]):
            host[k * s["slot"]: k * s["slot"] + len(p)] = np.frombuffer(p, np.uint8)
        s["src"] = torch.from_numpy(host).to(M.DEV)
        s["dict_t"] = torch.from_numpy(s["dictionary"].copy()).to(M.DEV) if s["dictionary"] is not None else None
    out_bytes = sum(sh[c % 4]["elem"] for c in range(columns)) * PAGE * PAGES_PER_CHUNK * row_groups
    dst = torch.empty(out_bytes + 4096, dtype=torch.uint8, device=M.DEV)
    groups, at, enc_b = [], 0, 0
    for g in range(row_groups):
        pages = []
        for c in range(columns):
            s = sh[c % 4]
            for q in range(PAGES_PER_CHUNK):
                u = (g + c + q) % len(s["pages"])
                p = pkg.Page()
                p.src, p.src_len = s["src"].data_ptr() + u * s["slot"], len(s["pages"][u])
                p.dst, p.dst_cap = dst.data_ptr() + at, PAGE * s["elem"]
                p.num_values = PAGE
                p.encoding, p.kind, p.bit_width = s["enc"], s["kind"], s["bw"]
                if s["dict_t"] is not None:
                    p.dict, p.dict_len = s["dict_t"].data_ptr(), s["dictionary"].size
                pages.append((p, s, u, at))
                at += PAGE * s["elem"]
                enc_b += len(s["pages"][u])
        groups.append(pages)
    stream = M.stream

    def check(batch, pages):
        batch.launch(stream)
        res = batch.finish(stream)
        bad = [(i, res[i].status) for i in range(len(pages)) if res[i].status != 0]
        assert not bad, bad[:3]
        for i in (0, len(pages) // 2, len(pages) - 1):
            p, s, u, off = pages[i]
            want = s["expect"][u]
            got = dst[off: off + want.size].cpu().numpy()
            assert got.tobytes() == want.tobytes(), (s["name"], i)

    per_group = [pkg.Batch(M.ctx, [p for p, *_ in pages]) for pages in groups]
    for b, pages in zip(per_group, groups):
        check(b, pages)
    one = pkg.Batch(M.ctx, [p for pages in groups for p, *_ in pages])
    check(one, [x for pages in groups for x in pages])

    class PerGroup:
        def launch(self, s):
            for b in per_group:
                b.launch(s)

    class One:
        def launch(self, s):
            one.launch(s)

    side = [torch.cuda.Stream(device=M.DEV) for _ in range(8)]

    class PerGroupStreams:
        """The same batches, dealt round robin to eight streams that fork from the caller's and join it again."""
        def launch(self, s):
            main = torch.cuda.ExternalStream(s, device=M.DEV)
            fork = torch.cuda.Event()
            fork.record(main)
            for st in side[: len(per_group)]:
                st.wait_event(fork)
            for i, b in enumerate(per_group):
                b.launch(side[i % len(side)].cuda_stream)
            for st in side[: len(per_group)]:
                done = torch.cuda.Event()
                done.record(st)
                main.wait_event(done)

    n_pages = row_groups * columns * PAGES_PER_CHUNK
    print(f"mixed job: {row_groups} row groups x {columns} column chunks x {PAGES_PER_CHUNK} pages of {PAGE} values "
          f"({n_pages} pages, {out_bytes / 1e9:.2f} GB decoded, {enc_b / 1e9:.2f} GB encoded)")
    for name, wl in (("a batch per row group", PerGroup()), ("a batch per row group, 8 streams", PerGroupStreams()),
                     ("all row groups in one batch", One())):
        wall, kms = bench.time_steps(M.S, torch, wl, stream, 20, 3, None)
        t = float(np.mean(kms)) / 1e3
        print(f"  {name:30s} {t * 1e6:9.1f} us  decoded {out_bytes / t / 1e9:8.1f} GB/s  algorithmic {(out_bytes + enc_b) / t / 1e9:8.1f} GB/s  "
              f"frac {(out_bytes + enc_b) / t / 8e12:.3f}", flush=True)


if __name__ == "__main__":
    main()
