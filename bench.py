#!/usr/bin/env python3
"""Benchmark of the MI355X page-value decode path.

Metric (BASELINE.json): device-resident decoded GB/s per GPU & node, RLE-dict + DELTA int64 pages.
A step at N=1 decodes both configurations the metric names, pages and output resident in HBM:
BASELINE.json configs[1], DELTA_BINARY_PACKED int64 timestamps, 256 Mi values in 2048 pages of
131072, blocks of 128 / 4 miniblocks, deltas 1e6 + U[0,4096) ns; and configs[0], RLE_DICTIONARY
int64, 16 Mi values, 64 Ki-entry dictionary, 64 pages of 262144.  `value` is the decoded bytes
of both over the step; `roofline` is that of the dominant kernel (DELTA, 94 % of the bytes) and
carries the RLE_DICTIONARY kernel's under "other_kernels".  --config wide runs the wide table
of mixed encodings (configs[4], bench_wide.py).

  python bench.py --gpus 1 --steps 20 --warmup 3
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 \
         --master-port P bench.py --gpus N --steps K --warmup W

With N > 1 every rank decodes its own row groups (row group g -> GPU g mod N, no collective on
the data path); per-GPU work is fixed, so scaling is weak.  Rank 0 prints one JSON line.
"""
import argparse
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0  # MI355X HBM3E peak (spec)


# --------------------------------------------------------------------------- input generation
# Page bytes are produced here with numpy (the layout parquet-go's writer emits:
# encoding/delta/binary_packed.go:123-181, encoding/rle/dictionary.go:23-28); the checker under
# oracle/ is used only by the cpu_baseline leg below.

def _uvarint(v: int) -> bytes:
    out = bytearray()
    while v >= 0x80:
        out.append((v & 0x7F) | 0x80)
        v >>= 7
    out.append(v)
    return bytes(out)


def _zigzag(v: int) -> int:
    return (v << 1) ^ (v >> 63) if v >= 0 else (((-v) << 1) - 1)


def _pack_rows(vals: np.ndarray, w: int) -> np.ndarray:
    """vals: (M, 32) uint64, all < 2**w -> (M, 4*w) bytes, fields LSB first."""
    if w == 0:
        return np.zeros((vals.shape[0], 0), np.uint8)
    bits = np.unpackbits(vals.astype("<u8").view(np.uint8).reshape(vals.shape[0], 32, 8), axis=2, bitorder="little")
    bits = bits[:, :, :w].reshape(vals.shape[0], 32 * w)
    return np.packbits(bits, axis=1, bitorder="little")


def delta_page_int64(values: np.ndarray) -> bytes:
    """DELTA_BINARY_PACKED page, block 128 / 4 miniblocks, as the reference writer lays it out."""
    n = values.size
    head = _uvarint(128) + _uvarint(4) + _uvarint(n) + _uvarint(_zigzag(int(values[0])) if n else 0)
    if n < 2:
        return head
    d = np.diff(values.astype(np.int64))
    nblocks = (d.size + 127) // 128
    pad = nblocks * 128 - d.size
    # the writer deltas over a zero padded block and takes the minimum over all 128 slots
    if pad:
        tail = np.zeros(pad, np.int64)
        tail[0] = -int(values[-1])
        d = np.concatenate([d, tail])
    blocks = d.reshape(nblocks, 128)
    mins = blocks.min(axis=1)
    rel = (blocks - mins[:, None]).astype(np.uint64)
    if pad:
        rel[-1, 128 - pad:] = 0
    mb = rel.reshape(nblocks * 4, 32)
    mx = mb.max(axis=1)
    widths = np.zeros(mb.shape[0], np.uint8)
    nz = mx > 0
    widths[nz] = np.floor(np.log2(mx[nz].astype(np.float64))).astype(np.uint8) + 1
    # float log2 can be off by one at powers of two of large values: fix up exactly
    for k in np.nonzero(nz)[0]:
        widths[k] = int(mx[k]).bit_length()
    packed = [None] * mb.shape[0]
    for w in np.unique(widths):
        sel = np.nonzero(widths == w)[0]
        rows = _pack_rows(mb[sel], int(w))
        for r, k in zip(rows, sel):
            packed[k] = r.tobytes()
    out = bytearray(head)
    for b in range(nblocks):
        out += _uvarint(_zigzag(int(mins[b])))
        out += widths[4 * b: 4 * b + 4].tobytes()
        for m in range(4):
            out += packed[4 * b + m]
    return bytes(out)


def rle_dict_page(indexes: np.ndarray) -> bytes:
    """RLE_DICTIONARY index page of uniformly random indexes: width byte + one bit-packed run
    (random 16-bit indexes never give the writer eight equal values in a row to run-length)."""
    bw = int(indexes.max()).bit_length()
    assert indexes.size % 8 == 0 and bw == 16
    return bytes([bw]) + _uvarint(((indexes.size // 8) << 1) | 1) + indexes.astype("<u2").tobytes()


def splitmix64(seed: int):
    return np.random.Generator(np.random.PCG64(seed))


# --------------------------------------------------------------------------- workloads

class Workload:
    """Pages resident in HBM + a prepared pqd batch."""

    def __init__(self, pkg, ctx, device, enc_pages, replicate, kind, encoding, values_per_page, dictionary=None):
        import torch
        self.torch = torch
        self.pkg = pkg
        dev = f"cuda:{device}"
        uniq = len(enc_pages)
        # one contiguous HBM region of page slots, 256-byte aligned starts
        slot = max((len(p) + 255) // 256 * 256 for p in enc_pages)
        self.n_pages = uniq * replicate
        self.src = torch.zeros(self.n_pages * slot + 256, dtype=torch.uint8, device=dev)
        host = np.zeros(uniq * slot, np.uint8)
        for k, p in enumerate(enc_pages):
            host[k * slot: k * slot + len(p)] = np.frombuffer(p, np.uint8)
        first = torch.from_numpy(host).to(dev)
        for r in range(replicate):
            self.src[r * uniq * slot: (r + 1) * uniq * slot].copy_(first)
        del first
        elem = 8
        self.values = self.n_pages * values_per_page
        self.dst = torch.empty(self.values * elem // 8, dtype=torch.int64, device=dev)
        self.dict_t = None
        if dictionary is not None:
            self.dict_t = torch.from_numpy(dictionary.copy()).to(dev)
        pages = []
        enc_bytes = 0
        for i in range(self.n_pages):
            p = pkg.Page()
            plen = len(enc_pages[i % uniq])
            p.src, p.src_len = self.src.data_ptr() + i * slot, plen
            p.dst, p.dst_cap = self.dst.data_ptr() + i * values_per_page * elem, values_per_page * elem
            p.num_values = values_per_page
            p.encoding, p.kind = encoding, kind
            if dictionary is not None:
                p.dict, p.dict_len = self.dict_t.data_ptr(), dictionary.size
            pages.append(p)
            enc_bytes += plen
        self.encoded_bytes = enc_bytes
        self.decoded_bytes = self.values * elem
        self.dict_bytes = 0 if dictionary is None else dictionary.nbytes
        self.batch = pkg.Batch(ctx, pages)
        self.values_per_page = values_per_page

    @property
    def algorithmic_bytes(self):
        return self.encoded_bytes + self.decoded_bytes + self.dict_bytes

    def launch(self, stream):
        self.batch.launch(stream)

    def check(self, stream):
        res = self.batch.finish(stream)
        bad = [(i, res[i].status) for i in range(self.n_pages) if res[i].status != 0]
        if bad:
            raise RuntimeError(f"decode failed on {len(bad)} pages, first {bad[0]}")


def build_delta_workload(pkg, ctx, device, seed, n_pages, values_per_page, unique):
    rng = splitmix64(0xC2 + seed)
    enc, expect = [], []
    t0 = 1_600_000_000_000_000_000
    for k in range(unique):
        deltas = rng.integers(1_000_000, 1_000_000 + 4096, size=values_per_page, dtype=np.int64)
        ts = t0 + np.cumsum(deltas)
        t0 = int(ts[-1])
        enc.append(delta_page_int64(ts))
        expect.append(ts)
    wl = Workload(pkg, ctx, device, enc, n_pages // unique, pkg.KIND.INT64, pkg.ENC.DELTA_BINARY_PACKED, values_per_page)
    wl.expect, wl.enc_pages = expect, enc
    return wl


def build_rle_dict_workload(pkg, ctx, device, seed, n_pages, values_per_page, dict_size):
    rng = splitmix64(0xC1 + seed)
    dictionary = rng.integers(-2**62, 2**62, size=dict_size, dtype=np.int64)
    enc, expect = [], []
    for k in range(n_pages):
        idx = rng.integers(0, dict_size, size=values_per_page, dtype=np.int64).astype(np.int32)
        enc.append(rle_dict_page(idx))
        expect.append(idx)
    wl = Workload(pkg, ctx, device, enc, 1, pkg.KIND.INT64, pkg.ENC.RLE_DICTIONARY, values_per_page, dictionary)
    wl.expect_idx, wl.dictionary, wl.enc_pages = expect, dictionary, enc
    return wl


def time_steps(S, torch, wl, stream, steps, warmup, dist=None):
    """Wall seconds of exactly `steps` launches (barrier + synchronize on both sides) and the
    per-launch kernel time from HIP events on the stream the kernels are launched on."""
    ev = []
    warm = {"left": warmup}

    def step():
        if warm["left"] > 0:
            warm["left"] -= 1
            wl.launch(stream)
            return
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        wl.launch(stream)
        b.record()
        ev.append((a, b))

    wall = S.timed_region(step, steps, warmup, torch.cuda.synchronize, dist.barrier if dist is not None else None)
    kernel_ms = [a.elapsed_time(b) for a, b in ev]
    assert len(kernel_ms) == steps
    return wall, kernel_ms


def cpu_baseline(wl, c1, budget_s=3.0):
    """CPU decoders of oracle/ on a bounded sample of the same pages, on one host core and on the box's
    share of host cores (a page per task, a thread per core; the C decoders run outside the interpreter
    lock).  Two ports are timed: "port-opt" (oracle/pq_port_opt.c: width-specialised word-at-a-time
    unpack, scalar running sum, organised like the reference's amd64 kernels) is the baseline reported;
    "port" (oracle/pq_oracle.c, the bit-by-bit restatement the parity tests check against) stands beside it."""
    import threading
    from oracle import pqoracle as O
    L = O.lib()
    cores = max(1, min(16, os.cpu_count() or 1))  # a one-GPU box is given 16 host cores

    def run(n_threads, n_pages, one_page):
        """one_page(j, scratch) decodes page j and returns the bytes it produced."""
        out = [0] * n_threads
        t0 = time.perf_counter()

        def worker(k):
            scratch = {}
            done = 0
            while time.perf_counter() < t0 + budget_s:
                for j in (range(k, n_pages, n_threads) if n_threads <= n_pages else (k % n_pages,)):
                    done += one_page(j, scratch)
            out[k] = done

        ts = [threading.Thread(target=worker, args=(k,)) for k in range(n_threads)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        return sum(out) / (time.perf_counter() - t0) / 1e9

    srcs = [np.frombuffer(p, np.uint8) for p in wl.enc_pages[: min(len(wl.enc_pages), 16)]]

    def delta_with(fn):
        def one(j, scratch):
            dst = scratch.setdefault("dst", np.zeros(wl.values_per_page, np.int64))
            r = fn(dst.ctypes.data, dst.nbytes, srcs[j].ctypes.data, srcs[j].size)
            assert r.status == 0
            return dst.nbytes
        return one

    dsrcs = [np.frombuffer(p, np.uint8) for p in c1.enc_pages[: min(len(c1.enc_pages), 16)]]
    dictionary = np.ascontiguousarray(c1.dictionary)

    def dict_opt(j, scratch):
        dst = scratch.setdefault("dst", np.zeros(c1.values_per_page, np.int64))
        r = L.pqopt_rle_dict_lookup_int64(dst.ctypes.data, c1.values_per_page, dsrcs[j].ctypes.data, dsrcs[j].size, dictionary.ctypes.data, dictionary.size)
        assert r.status == 0
        return dst.nbytes

    def dict_port(j, scratch):
        dst = scratch.setdefault("dst", np.zeros(c1.values_per_page, np.int64))
        idx = scratch.setdefault("idx", np.zeros(c1.values_per_page, np.int32))
        r = L.pqo_rle_dict_decode_int32(idx.ctypes.data, idx.nbytes, dsrcs[j].ctypes.data, dsrcs[j].size)
        r2 = L.pqo_dict_lookup(dst.ctypes.data, 8, dictionary.ctypes.data, dictionary.size, 8, idx.ctypes.data, c1.values_per_page)
        assert r.status == 0 and r2.status == 0
        return dst.nbytes

    def both(n_pages, one_page):
        one = run(1, n_pages, one_page)
        return round(one, 3), round(run(cores, n_pages, one_page) if cores > 1 else one, 3)

    opt_one, opt_many = both(len(srcs), delta_with(L.pqopt_delta_decode_int64))
    port_one, port_many = both(len(srcs), delta_with(L.pqo_delta_decode_int64))
    dopt_one, dopt_many = both(len(dsrcs), dict_opt)
    dport_one, dport_many = both(len(dsrcs), dict_port)
    return {"value": opt_many, "unit": "GB/s", "cores": cores, "kind": "port-opt", "one_core": opt_one,
            "port": {"value": port_many, "one_core": port_one, "kind": "port"},
            "rle_dictionary_int64": {"value": dopt_many, "one_core": dopt_one, "kind": "port-opt", "port": {"value": dport_many, "one_core": dport_one, "kind": "port"}},
            "sample": f"DELTA_BINARY_PACKED int64 (the figure in `value`): {len(srcs)} of the workload's pages ({wl.values_per_page} values each) decoded over "
                      f"and over for {budget_s:.0f} s per leg, on one host core and then on {cores} threads with a page per task, decoded GB/s; "
                      f"rle_dictionary_int64: the same with {len(dsrcs)} of the RLE_DICTIONARY pages ({c1.values_per_page} values each, indexes decoded "
                      f"and looked up in the 64 Ki-entry dictionary).  port-opt = oracle/pq_port_opt.c, a tuned C port (unpackers specialised per "
                      f"width, a miniblock of 32 values per call, scalar running sum, -O3 -march=x86-64-v3); port = oracle/pq_oracle.c, the "
                      f"bit-by-bit restatement the parity tests use.  The reference itself is Go and is not installed here; its amd64 kernels "
                      f"decode this family at about 6 GB/s per core (docs/delta-byte-array-prefix.md:235-238 of the reference)"}


def _timed_pair(S, torch, first, second, stream, steps, warmup, dist):
    """Like time_steps for a step of two batches launched back to back: wall seconds of `steps` steps and,
    per step, the event-timed milliseconds of each batch's kernels."""
    ev = []
    warm = {"left": warmup}

    def step():
        if warm["left"] > 0:
            warm["left"] -= 1
            first.launch(stream)
            second.launch(stream)
            return
        a, m, b = (torch.cuda.Event(enable_timing=True) for _ in range(3))
        a.record()
        first.launch(stream)
        m.record()
        second.launch(stream)
        b.record()
        ev.append((a, m, b))

    wall = S.timed_region(step, steps, warmup, torch.cuda.synchronize, dist.barrier if dist is not None else None)
    return wall, [a.elapsed_time(m) for a, m, b in ev], [m.elapsed_time(b) for a, m, b in ev]


def _cold_launches(torch, wl, stream, device, n):
    """Event-timed launches of one batch, each behind 2 GB of stores to another buffer and an idle device."""
    big = torch.empty(256 << 20, dtype=torch.int64, device=f"cuda:{device}")
    ms = []
    for k in range(n):
        big.fill_(k)
        torch.cuda.synchronize()
        time.sleep(0.002)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        wl.launch(stream)
        b.record()
        torch.cuda.synchronize()
        ms.append(a.elapsed_time(b))
    del big
    return ms


def _roofline(algorithmic_bytes, kernel_s, kernel, traffic):
    achieved = algorithmic_bytes / kernel_s / 1e9
    return {"bound": "hbm", "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": round(achieved / HBM_PEAK_GBS, 4),
            "traffic": traffic, "traffic_source": None if traffic is None else "profiles/traffic.json: rocprofv3 --pmc FETCH_SIZE x 2 + WRITE_SIZE per launch, "
            "collected by tools/gpu_pmc_bench.sh and committed; not measured in this run", "kernel": kernel, "kernel_ms": round(kernel_s * 1e3, 4), "algorithmic_bytes": algorithmic_bytes}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--config", choices=("headline", "wide"), default="headline",
                    help="headline: RLE_DICTIONARY + DELTA_BINARY_PACKED int64 pages (BASELINE.json configs[0] and [1]); "
                         "wide: 64 columns x 64 row groups of mixed encodings sharded by row group (configs[4])")
    ap.add_argument("--pages", type=int, default=2048)
    ap.add_argument("--values-per-page", type=int, default=131072)
    ap.add_argument("--unique-pages", type=int, default=64, help="distinct seeded pages generated on the host; tiled in HBM to --pages")
    ap.add_argument("--wide-columns", type=int, default=64)
    ap.add_argument("--wide-row-groups", type=int, default=64)
    ap.add_argument("--wide-values-per-chunk", type=int, default=1 << 20)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--skip-check", action="store_true")
    args = ap.parse_args()

    import torch
    import __graft_entry__ as entry
    pkg = entry.load_package()
    from parquet_go_amd import sharding as S

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    dist = None
    # One rank per GPU over RCCL.  PQD_BENCH_REHEARSE=1 is for a box with fewer GPUs than ranks:
    # ranks share the GPUs there are and synchronise over gloo (RCCL refuses two ranks on one GPU),
    # which exercises the N > 1 code path; its numbers mean nothing.
    rehearse = os.environ.get("PQD_BENCH_REHEARSE") == "1"
    device = local_rank % torch.cuda.device_count() if (world > 1 and rehearse) else (local_rank if world > 1 else 0)
    if world > 1:
        import torch.distributed as dist_mod
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if rehearse:
            dist_mod.init_process_group(backend="gloo")
        else:
            dist_mod.init_process_group(backend="nccl", device_id=torch.device(f"cuda:{device}"))
        dist = dist_mod
    torch.cuda.set_device(device)
    ctx = pkg.Context(device)
    stream = torch.cuda.current_stream(device).cuda_stream
    red_dev = "cpu" if rehearse else f"cuda:{device}"

    if args.config == "wide":
        result = run_wide(args, pkg, S, torch, ctx, device, stream, rank, world, dist, red_dev)
    else:
        result = run_headline(args, pkg, S, torch, ctx, device, stream, rank, world, dist, red_dev)

    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(result))


def run_headline(args, pkg, S, torch, ctx, device, stream, rank, world, dist, red_dev):
    """A step decodes this rank's RLE_DICTIONARY pages (configs[0]) and its DELTA_BINARY_PACKED pages
    (configs[1]), one prepared batch each, launched back to back on one stream."""
    unique = min(args.unique_pages, args.pages)
    while args.pages % unique:
        unique -= 1
    # Row groups of `unique` pages each, dealt round robin over the ranks; every rank holds
    # args.pages // unique of them (weak scaling), seeded by the first row group it owns.
    n_row_groups = args.pages // unique
    mine = S.row_groups_of(rank, world, world * n_row_groups)
    assert len(mine) == n_row_groups
    wl = build_delta_workload(pkg, ctx, device, mine[0], args.pages, args.values_per_page, unique)
    c1 = build_rle_dict_workload(pkg, ctx, device, mine[0], 64, 262144, 65536)

    # correctness of what is being timed: every distinct page and its last replica
    wl.launch(stream)
    c1.launch(stream)
    wl.check(stream)
    c1.check(stream)
    if not args.skip_check:
        out = wl.dst.view(wl.n_pages, wl.values_per_page)
        for k in range(unique):
            for slot in (k, wl.n_pages - unique + k):
                if not np.array_equal(out[slot].cpu().numpy(), wl.expect[k]):
                    raise RuntimeError(f"DELTA_BINARY_PACKED page {slot} decoded wrong")
        out = c1.dst.view(c1.n_pages, c1.values_per_page)
        for k in range(c1.n_pages):
            if not np.array_equal(out[k].cpu().numpy(), c1.dictionary[c1.expect_idx[k]]):
                raise RuntimeError(f"RLE_DICTIONARY page {k} decoded wrong")

    wall, ms_dict, ms_delta = _timed_pair(S, torch, c1, wl, stream, args.steps, args.warmup, dist)
    # The RLE_DICTIONARY batch by itself as well, two ways.  Launched over and over, its whole working set
    # (168 MB) sits in the 256 MB last-level cache and neither its indexes nor its output touch HBM: that is
    # the most favourable figure.  Inside the step every launch follows 2 GB of DELTA output, so the
    # indexes come from HBM and the output displaces dirty lines; "cold" reproduces that by itself --
    # 2 GB written to another buffer and the device left idle before each launch -- to tell a cold
    # cache (it is that) from a memory system still draining the stores before (it is not).
    _, ms_dict_alone = time_steps(S, torch, c1, stream, args.steps, args.warmup, None)
    ms_dict_cold = _cold_launches(torch, c1, stream, device, min(args.steps, 10))
    wall_max = S.max_over_ranks(wall, dist, red_dev)
    step_s = wall_max / args.steps
    decoded = wl.decoded_bytes + c1.decoded_bytes
    value = S.whole_job_rate(decoded / 1e9, world, step_s)
    t_delta, t_dict = float(np.mean(ms_delta)) / 1e3, float(np.mean(ms_dict)) / 1e3
    roof = _roofline(wl.algorithmic_bytes, t_delta, "delta_fast_kernel<uint64_t>", _traffic("delta"))
    # the step's other kernel, with a roofline of its own: it decodes 6 % of the step's bytes, so the figure
    # above says little about it
    roof["other_kernels"] = [_roofline(c1.algorithmic_bytes, t_dict, "rle_dict_lds_kernel<8, true> (page index fused)", _traffic("rle_dict"))]
    roof["step"] = {"algorithmic_bytes": wl.algorithmic_bytes + c1.algorithmic_bytes, "kernels_ms": round((t_delta + t_dict) * 1e3, 4),
                    "frac": round((wl.algorithmic_bytes + c1.algorithmic_bytes) / (t_delta + t_dict) / 1e9 / HBM_PEAK_GBS, 4)}
    result = {
        "metric": "device-resident decoded GB/s (RLE_DICTIONARY + DELTA_BINARY_PACKED int64 pages)",
        "value": round(value, 1),
        "unit": "GB/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(step_s * 1e3, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "int64",
        "data": f"synthetic: {unique} distinct seeded timestamp pages tiled in HBM to {wl.n_pages} page slots per GPU, deltas 1e6 + U[0,4096) ns "
                f"(miniblock width 12); {c1.n_pages} seeded pages of uniform indexes into a 64 Ki-entry int64 dictionary; uncompressed",
        "config": {"workload": f"per GPU and step: DELTA_BINARY_PACKED int64, {wl.values} values in {wl.n_pages} pages of {wl.values_per_page}, "
                               f"blocks of 128 / 4 miniblocks (BASELINE.json configs[1]) + RLE_DICTIONARY int64, {c1.values} values in "
                               f"{c1.n_pages} pages of {c1.values_per_page}, 64 Ki-entry (512 KB) dictionary, index width 16 (configs[0])",
                   "parallelism": f"row groups sharded over {world} GPU(s), no data-path collective"},
        "roofline": roof,
        "parts": {"delta_binary_packed_int64": {"decoded_GBps": round(wl.decoded_bytes / t_delta / 1e9, 1), "kernels_ms": round(t_delta * 1e3, 4),
                                                "frac": roof["frac"]},
                  "rle_dictionary_int64": {"decoded_GBps": round(c1.decoded_bytes / t_dict / 1e9, 1), "kernels_ms": round(t_dict * 1e3, 4),
                                           "frac": roof["other_kernels"][0]["frac"],
                                           "launched_by_itself": _alone(c1, ms_dict_alone, "launched over and over: working set resident in the last-level cache"),
                                           "launched_by_itself_cold": _alone(c1, ms_dict_cold, "each launch behind 2 GB of stores to another buffer and 2 ms of idle device")}},
    }
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        result["cpu_baseline"] = cpu_baseline(wl, c1)
    else:
        result["cpu_baseline"] = None
    return result


def run_wide(args, pkg, S, torch, ctx, device, stream, rank, world, dist, red_dev):
    """A step decodes this rank's row groups of the wide table in one batch; the table is fixed, so
    more GPUs means fewer row groups each (strong scaling)."""
    import bench_wide as W
    wl = W.WideWorkload(pkg, ctx, device, rank, world, args.wide_columns, args.wide_row_groups, args.wide_values_per_chunk)
    wl.launch(stream)
    wl.check(stream, sample=1 if args.wide_row_groups * args.wide_columns <= 64 else 97)
    wall, kernel_ms = time_steps(S, torch, wl, stream, args.steps, args.warmup, dist)
    wall_max = S.max_over_ranks(wall, dist, red_dev)
    step_s = wall_max / args.steps
    totals = torch.tensor([wl.decoded_bytes, wl.algorithmic_bytes, wl.n_items], dtype=torch.float64, device=red_dev)
    if dist is not None:
        dist.all_reduce(totals, op=dist.ReduceOp.SUM)
    decoded_all, algorithmic_all, items_all = (float(x) for x in totals.tolist())
    t = float(np.mean(kernel_ms)) / 1e3
    return {
        "metric": "device-resident decoded GB/s (wide table of mixed encodings, row groups sharded over the GPUs)",
        "value": round(decoded_all / step_s / 1e9, 1),
        "unit": "GB/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(step_s * 1e3, 4),
        "higher_is_better": True,
        "scaling": "strong",
        "vs_baseline": None,
        "dtype": "u8",
        "data": "synthetic: four distinct seeded pages per column shape, tiled in HBM to every page of the table (distinct copies), uncompressed",
        "config": {"workload": f"{args.wide_columns} columns x {args.wide_row_groups} row groups, {args.wide_values_per_chunk} values per column chunk in "
                               f"pages of {W.PAGE_VALUES}; column shapes cycle {', '.join(W.SHAPES)} (BASELINE.json configs[4]); "
                               f"{int(items_all)} work items in all, one batch per GPU",
                   "parallelism": f"row group g on GPU g mod {world}, no data-path collective"},
        # all kernels of this rank's batch together: the job is a mix, no one kernel dominates
        "roofline": _roofline(wl.algorithmic_bytes, t, "all kernels of the batch (rank 0)", None),
        "cpu_baseline": None,
    }


def _alone(wl, ms, how):
    t = float(np.mean(ms)) / 1e3
    return {"kernels_ms": round(t * 1e3, 4), "decoded_GBps": round(wl.decoded_bytes / t / 1e9, 1),
            "frac": round(wl.algorithmic_bytes / t / 1e9 / HBM_PEAK_GBS, 4), "how": how}


def _traffic(which):
    """HBM bytes per launch from the committed rocprofv3 --pmc summary, if one exists."""
    path = os.path.join(ROOT, "profiles", "traffic.json")
    try:
        return json.load(open(path)).get(which)
    except Exception:
        return None


if __name__ == "__main__":
    main()
