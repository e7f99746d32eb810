"""Host-side mirror of what parquet-go does with decoded levels, over the device-resident
entry points of include/pqd.h: countLevelsEqual / countLevelsNotEqual (level.go:11-17),
optionalPageValues.ReadValues (page_optional.go:72-117) and the row search of
repeatedPage.Slice (page_repeated.go:73-98).  Inputs are device buffers (or anything
`to_device` accepts); results stay on the device until asked for."""
import ctypes as C

import numpy as np

from ._lib import Context, PqdError, STATUS, lib
from .runtime import DeviceBuffer, to_device, _torch


def _dev(x, device):
    return x if isinstance(x, DeviceBuffer) else to_device(x, device)


def _workspace(n, device):
    return DeviceBuffer(int(lib().pqd_levels_workspace_size(n)) + 8, device, misalign=0)


def _aligned_ptr(buf):
    return (buf.t.data_ptr() + 7) & ~7


def count_levels_equal(ctx: Context, levels, value: int, stream: int = 0) -> int:
    """countLevelsEqual (level.go:11-13)."""
    torch = _torch()
    d = _dev(levels, ctx.device)
    ws = _workspace(d.nbytes, ctx.device)
    out = torch.zeros(1, dtype=torch.int64, device=f"cuda:{ctx.device}")
    rc = lib().pqd_levels_count_equal_device(ctx.handle, d.ptr, d.nbytes, value, _aligned_ptr(ws), ws.nbytes, out.data_ptr(), stream)
    if rc != STATUS.OK:
        raise PqdError(rc, 0, "pqd_levels_count_equal_device")
    torch.cuda.synchronize()
    return int(out.item())


def count_levels_not_equal(ctx: Context, levels, value: int, stream: int = 0) -> int:
    """countLevelsNotEqual (level.go:15-17)."""
    d = _dev(levels, ctx.device)
    return d.nbytes - count_levels_equal(ctx, d, value, stream)


class Expanded:
    """Result of `optional_expand`: slot-aligned values, validity bytes, the number of slots
    with a value and the status (OK or UNEXPECTED_EOF)."""

    def __init__(self, values, validity, present, status):
        self.values, self.validity, self.present, self.status = values, validity, present, status


def optional_expand(ctx: Context, def_levels, max_definition_level: int, values, n_values: int, width: int,
                    offsets_mode: bool = False, stream: int = 0, misalign: int = 0) -> Expanded:
    torch = _torch()
    d = _dev(def_levels, ctx.device)
    v = _dev(values, ctx.device)
    n = d.nbytes
    dst = DeviceBuffer((n + (1 if offsets_mode else 0)) * width, ctx.device, misalign=misalign, fill=0xA5)
    val = DeviceBuffer(n, ctx.device, misalign=misalign, fill=0xA5)
    ws = _workspace(n, ctx.device)
    info = torch.zeros(2, dtype=torch.int64, device=f"cuda:{ctx.device}")
    rc = lib().pqd_optional_expand_device(ctx.handle, d.ptr, n, max_definition_level, v.ptr, n_values, width, 1 if offsets_mode else 0,
                                          dst.ptr, val.ptr, _aligned_ptr(ws), ws.nbytes, info.data_ptr(), stream)
    if rc != STATUS.OK:
        raise PqdError(rc, 0, "pqd_optional_expand_device")
    torch.cuda.synchronize()
    present, status = (int(x) for x in info.cpu().numpy())
    return Expanded(dst.numpy(), val.numpy(), present, status)


def repeated_row_starts(ctx: Context, rep_levels, row_cap: int = -1, stream: int = 0):
    """(row_starts, n_rows, status): row k starts at the k-th repetition level equal to 0."""
    torch = _torch()
    d = _dev(rep_levels, ctx.device)
    n = d.nbytes
    cap = n + 1 if row_cap < 0 else row_cap
    rows = torch.full((max(cap, 1),), -1, dtype=torch.int64, device=f"cuda:{ctx.device}")
    ws = _workspace(n, ctx.device)
    info = torch.zeros(2, dtype=torch.int64, device=f"cuda:{ctx.device}")
    rc = lib().pqd_repeated_row_starts_device(ctx.handle, d.ptr, n, rows.data_ptr(), cap, _aligned_ptr(ws), ws.nbytes, info.data_ptr(), stream)
    if rc != STATUS.OK:
        raise PqdError(rc, 0, "pqd_repeated_row_starts_device")
    torch.cuda.synchronize()
    n_rows, status = (int(x) for x in info.cpu().numpy())
    return rows.cpu().numpy()[:cap], n_rows, status
