"""Device plumbing: torch owns HBM buffers and streams, libpqd does the work.

`Batch` is a thin wrapper over pqd_batch_{create,launch,finish}: pages already
in HBM go in, decoded values stay in HBM.  `decode_pages` is the convenience
used by tests: host bytes in, numpy out, through the same batched entry points.
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import List, Optional, Sequence

import numpy as np

from ._lib import ENC, KIND, STATUS, Context, Page, PqdError, lib


def _torch():
    import torch
    if not torch.cuda.is_available():
        raise RuntimeError("no GPU visible: the decode path has no CPU fallback")
    return torch


class DeviceBuffer:
    """A byte buffer in HBM (a torch uint8 tensor) with a few bytes of slack in front
    so tests can place a page at any alignment."""

    def __init__(self, nbytes: int, device: int = 0, misalign: int = 0, fill: Optional[int] = None):
        torch = _torch()
        self.nbytes = int(nbytes)
        self.misalign = int(misalign)
        n = self.nbytes + self.misalign + 64
        self.t = torch.empty(n, dtype=torch.uint8, device=f"cuda:{device}")
        if fill is not None:
            self.t.fill_(fill)

    @property
    def ptr(self) -> int:
        return self.t.data_ptr() + self.misalign

    def view(self):
        return self.t[self.misalign: self.misalign + self.nbytes]

    def numpy(self) -> np.ndarray:
        return self.view().cpu().numpy()


def to_device(data, device: int = 0, misalign: int = 0) -> DeviceBuffer:
    torch = _torch()
    a = np.frombuffer(bytes(data), dtype=np.uint8) if not isinstance(data, np.ndarray) else np.ascontiguousarray(data).view(np.uint8).reshape(-1)
    buf = DeviceBuffer(a.size, device, misalign)
    if a.size:
        buf.view().copy_(torch.from_numpy(a.copy()))
    return buf


class Batch:
    """pqd_batch over an array of `Page` descriptors holding device pointers."""

    def __init__(self, ctx: Context, pages: Sequence[Page]):
        self.ctx = ctx
        self.n = len(pages)
        self.pages = (Page * max(self.n, 1))(*pages)
        h = C.c_void_p()
        rc = lib().pqd_batch_create(ctx.handle, self.pages, self.n, C.byref(h))
        if rc != STATUS.OK:
            raise PqdError(rc, 0, "pqd_batch_create")
        self.handle = h

    def launch(self, stream: int = 0):
        rc = lib().pqd_batch_launch(self.handle, C.c_void_p(stream))
        if rc != STATUS.OK:
            raise PqdError(rc, 0, "pqd_batch_launch")

    def finish(self, stream: int = 0):
        rc = lib().pqd_batch_finish(self.handle, C.c_void_p(stream), self.pages, self.n)
        if rc != STATUS.OK:
            raise PqdError(rc, 0, "pqd_batch_finish")
        return self.pages

    def close(self):
        if self.handle:
            lib().pqd_batch_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


@dataclass
class PageSpec:
    """A page to decode, host side: what a test hands to `decode_pages`."""
    encoding: int
    kind: int
    src: bytes
    bit_width: int = 0
    num_values: int = -1
    dst_cap: int = 0              # bytes; 0 = let decode_pages pick a generous bound
    offsets_cap: int = 0          # entries
    dictionary: Optional[np.ndarray] = None  # dictionary values (fixed width; bytes of a BYTE_ARRAY dictionary; packed bits of a BOOLEAN one)
    dictionary_offsets: Optional[np.ndarray] = None  # BYTE_ARRAY dictionary: entries + 1 uint32 offsets
    dictionary_len: int = -1      # entries, when len(dictionary) does not say (byte and bit dictionaries)
    dst_stride: int = 0
    misalign: int = 0             # byte misalignment of src in device memory


@dataclass
class PageOut:
    status: int
    arg: int
    out_values: int
    out_bytes: int
    consumed: int
    data: np.ndarray              # dst bytes [0, out_bytes)
    offsets: Optional[np.ndarray]
    raw: Optional[np.ndarray] = None  # whole dst buffer (with canaries)


_CANARY = 0xA5


def decode_pages(ctx: Context, specs: List[PageSpec], device: int = 0, grow: bool = True) -> List[PageOut]:
    """Uploads the pages, runs ONE batch, downloads results.  DST_TOO_SMALL items
    are retried once with the size the library asked for, the way a Go caller
    grows its slice."""
    torch = _torch()
    stream = torch.cuda.current_stream(device).cuda_stream
    srcs, dsts, offs, dicts, pages = [], [], [], [], []
    for s in specs:
        src = to_device(s.src, device, s.misalign)
        cap = s.dst_cap or max(4096, 80 * len(s.src) + 4096)
        dst = DeviceBuffer(cap + 64, device, fill=_CANARY)
        ocap = s.offsets_cap or (len(s.src) + 8 if s.kind == KIND.BYTE_ARRAY else 0)
        off = DeviceBuffer(ocap * 4 + 64, device, fill=_CANARY) if ocap else None
        d = to_device(s.dictionary, device) if s.dictionary is not None else None
        p = Page()
        p.src, p.src_len = src.ptr, len(s.src)
        p.dst, p.dst_cap = dst.ptr, cap
        p.offsets, p.offsets_cap = (off.ptr if off else None), ocap
        p.dict, p.dict_len = (d.ptr if d is not None else None), (len(s.dictionary) if s.dictionary is not None else 0)
        if d is not None and p.dict_len == 0:
            p.dict = d.ptr
        if s.dictionary_len >= 0:
            p.dict_len = s.dictionary_len
        if s.dictionary_offsets is not None:
            do = to_device(np.ascontiguousarray(s.dictionary_offsets, dtype=np.uint32), device)
            dicts.append(do)
            p.dict_offsets, p.dict_len = do.ptr, len(s.dictionary_offsets) - 1
        p.num_values = s.num_values
        p.encoding, p.kind, p.bit_width, p.dst_stride = s.encoding, s.kind, s.bit_width, s.dst_stride
        srcs.append(src); dsts.append(dst); offs.append(off); dicts.append(d); pages.append(p)
    b = Batch(ctx, pages)
    b.launch(stream)
    res = b.finish(stream)
    outs = []
    for i, s in enumerate(specs):
        r = res[i]
        raw = dsts[i].numpy()
        cap = pages[i].dst_cap
        # nothing may be written past the capacity handed in
        assert (raw[cap:] == _CANARY).all(), f"page {i}: write past dst_cap"
        o = offs[i].numpy().view(np.uint32) if offs[i] is not None else None
        if o is not None:
            ocap = pages[i].offsets_cap
            assert (o[ocap:] == 0xA5A5A5A5).all(), f"page {i}: write past offsets_cap"
        outs.append(PageOut(r.status, r.status_arg, r.out_values, r.out_bytes, r.consumed,
                            raw[: min(r.out_bytes, cap)].copy(), None if o is None else o.copy(), raw))
    b.close()
    if grow:
        for i, s in enumerate(specs):
            if outs[i].status == STATUS.DST_TOO_SMALL and outs[i].arg > (s.dst_cap or 0):
                s2 = PageSpec(**{**s.__dict__, "dst_cap": int(outs[i].arg)})
                outs[i] = decode_pages(ctx, [s2], device, grow=False)[0]
            elif outs[i].status == STATUS.OFFSETS_TOO_SMALL:
                s2 = PageSpec(**{**s.__dict__, "offsets_cap": int(outs[i].arg)})
                outs[i] = decode_pages(ctx, [s2], device, grow=True)[0]
    return outs


def encode_pages(ctx: Context, jobs, device: int = 0, dst_slack: int = 0, misalign: int = 0, src_misalign: int = 0):
    """Encodes pages in one batch through pqd_encode_batch.  `jobs` is a list of
    (encoding, kind, values, bit_width) with `values` a numpy array; returns a list of
    (status, status_arg, page bytes).  `dst_slack` < 0 makes every destination that many
    bytes too small (to see the size the library asks for); `misalign` shifts the
    destinations, `src_misalign` the values."""
    from ._lib import EncodePage
    L = lib()
    torch = _torch()
    pages, keep = [], []
    for job in jobs:
        encoding, kind, values, bit_width = job[:4]
        offsets = job[4] if len(job) > 4 else None  # BYTE_ARRAY pages: n + 1 offsets into `values` (bytes)
        v = np.ascontiguousarray(values)
        src = to_device(v, device, src_misalign)
        p = EncodePage()
        off = None
        if kind == KIND.BYTE_ARRAY or (kind == KIND.FIXED_LEN_BYTE_ARRAY and encoding == 7):
            v = v.view(np.uint8).reshape(-1)
            if offsets is not None:
                o = np.ascontiguousarray(offsets, dtype=np.uint32)
                n = max(o.size - 1, 0)
                off = to_device(o if o.size else np.zeros(1, np.uint32), device)
                p.offsets = off.ptr
            else:
                n = v.size // bit_width if kind == KIND.FIXED_LEN_BYTE_ARRAY and bit_width > 0 else 0
            p.src_bytes = v.size
            cap = max(0, int(L.pqd_encode_bound_byte_array(encoding, n, v.size)) + dst_slack)
        else:
            n = v.size // bit_width if kind == KIND.FIXED_LEN_BYTE_ARRAY else v.size
            cap = max(0, int(L.pqd_encode_bound(encoding, kind, bit_width, n)) + dst_slack)
        dst = DeviceBuffer(cap, device, misalign, fill=0xA5)
        p.src, p.n_values, p.dst, p.dst_cap = src.ptr, n, dst.ptr, cap
        p.encoding, p.kind, p.bit_width = encoding, kind, bit_width
        pages.append(p)
        keep.append((src, dst, off))
    arr = (EncodePage * max(len(pages), 1))(*pages)
    ws_bytes = int(L.pqd_encode_workspace_size(arr, len(pages)))
    ws = torch.empty(ws_bytes + 256, dtype=torch.uint8, device=f"cuda:{device}")
    stream = torch.cuda.current_stream(device).cuda_stream
    rc = L.pqd_encode_batch(ctx.handle, arr, len(pages), ws.data_ptr(), ws.numel(), stream)
    if rc != STATUS.OK:
        raise PqdError(rc, 0, "pqd_encode_batch")
    out = []
    for i, (src, dst, _) in enumerate(keep):
        raw = dst.numpy()
        out.append((arr[i].status, arr[i].status_arg, raw[: arr[i].out_bytes].tobytes(), raw))
    return out


def dict_bounds(ctx: Context, dictionary: np.ndarray, indexes, unsigned: bool = False, device: int = 0):
    """Dictionary.Bounds on the device (pqd_dict_bounds_device): (status, arg, min, max) of
    dictionary[indexes] for int32 / int64 / float32 / float64 (or uint32 / uint64) dictionaries."""
    torch = _torch()
    L = lib()
    d = np.ascontiguousarray(dictionary)
    kind = {"i4": KIND.INT32, "u4": KIND.INT32, "i8": KIND.INT64, "u8": KIND.INT64, "f4": KIND.FLOAT, "f8": KIND.DOUBLE}[d.dtype.str[1:]]
    idx = np.ascontiguousarray(indexes, dtype=np.int32)
    d_dev, i_dev = to_device(d, device), to_device(idx, device)
    out = DeviceBuffer(2 * d.dtype.itemsize, device, fill=0xA5)
    status = DeviceBuffer(16, device, fill=0xA5)
    ws = torch.empty(64 << 10, dtype=torch.uint8, device=f"cuda:{device}")
    stream = torch.cuda.current_stream(device).cuda_stream
    rc = L.pqd_dict_bounds_device(ctx.handle, kind, int(unsigned or d.dtype.kind == "u"), d_dev.ptr, d.size, i_dev.ptr, idx.size,
                                  out.ptr, ws.data_ptr(), ws.numel(), status.ptr, stream)
    if rc != STATUS.OK:
        raise PqdError(rc, 0, "pqd_dict_bounds_device")
    torch.cuda.synchronize(device)
    st = status.numpy().view(np.int64)
    lo, hi = out.numpy().view(d.dtype)
    return int(st[0]), int(st[1]), lo, hi
