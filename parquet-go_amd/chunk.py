"""Host-side mirror of the reader's page loop over one column chunk: FilePages.ReadPage
(file.go:1193-1259) finds the page headers, Column.decodeDataPageV1 / V2 / decodeDictionary
(column.go:733-843, 963-993) cut the pages into levels and values, the codecs decode them and
the levels place the values (page_optional.go:72-117).  Headers and parts come from the C ABI
(pqd_chunk_scan_pages, pqd_chunk_plan_pages); every part of every page of the chunk is decoded
by one batch on the device (two when the chunk has a dictionary page: the dictionary first)."""
import ctypes as C
from dataclasses import dataclass, field
from typing import List, Optional

import numpy as np

from ._lib import ColumnInfo, Context, KIND, ENC, Page, PageHeader, PagePart, PqdError, STATUS, lib
from .runtime import Batch, DeviceBuffer, to_device

PAGE_DATA, PAGE_INDEX, PAGE_DICTIONARY, PAGE_DATA_V2 = 0, 1, 2, 3
PART_REP, PART_DEF, PART_VALUES, PART_DICT = 0, 1, 2, 3

KIND_OF = {"BOOLEAN": KIND.BOOLEAN, "INT32": KIND.INT32, "INT64": KIND.INT64, "INT96": KIND.INT96, "FLOAT": KIND.FLOAT,
           "DOUBLE": KIND.DOUBLE, "BYTE_ARRAY": KIND.BYTE_ARRAY, "FIXED_LEN_BYTE_ARRAY": KIND.FIXED_LEN_BYTE_ARRAY}
WIDTH_OF = {KIND.INT32: 4, KIND.INT64: 8, KIND.INT96: 12, KIND.FLOAT: 4, KIND.DOUBLE: 8}


def scan_pages(chunk: bytes, num_values: int = -1) -> List[PageHeader]:
    """pqd_chunk_scan_pages: the page headers of a column chunk.  Raises PqdError on a corrupt chunk."""
    buf = (C.c_uint8 * max(len(chunk), 1)).from_buffer_copy(bytes(chunk) or b"\0")
    n, arg = C.c_uint64(0), C.c_int64(0)
    cap = 64
    while True:
        out = (PageHeader * cap)()
        rc = lib().pqd_chunk_scan_pages(C.addressof(buf), len(chunk), num_values, out, cap, C.byref(n), C.byref(arg))
        if rc == STATUS.DST_TOO_SMALL:
            cap = int(n.value)
            continue
        if rc != STATUS.OK:
            raise PqdError(rc, arg.value, "pqd_chunk_scan_pages")
        return [out[i] for i in range(n.value)]


def plan_pages(headers: List[PageHeader], chunk: bytes, chunk_dev: int, kind: int, type_length: int, max_def: int, max_rep: int):
    """pqd_chunk_plan_pages: (items, parts) for the pages of an uncompressed chunk."""
    buf = (C.c_uint8 * max(len(chunk), 1)).from_buffer_copy(bytes(chunk) or b"\0")
    hs = (PageHeader * max(len(headers), 1))(*headers)
    col = ColumnInfo(kind, type_length, max_def, max_rep)
    cap = 3 * len(headers) + 1
    items, parts = (Page * cap)(), (PagePart * cap)()
    n, arg = C.c_uint64(0), C.c_int64(0)
    rc = lib().pqd_chunk_plan_pages(hs, len(headers), C.addressof(buf), len(chunk), chunk_dev, C.byref(col), items, parts, cap,
                                    C.byref(n), C.byref(arg))
    if rc != STATUS.OK:
        raise PqdError(rc, arg.value, "pqd_chunk_plan_pages")
    return [items[i] for i in range(n.value)], [parts[i] for i in range(n.value)]


@dataclass
class Column:
    """A decoded column chunk, on the host: levels of every slot, the values that are present
    back to back (`values`, with `offsets` for byte arrays; dictionary indexes in `indexes`
    only when a dictionary-encoded page came without a dictionary page), and the dictionary if the
    chunk has one."""
    num_values: int
    definition_levels: Optional[np.ndarray]
    repetition_levels: Optional[np.ndarray]
    values: Optional[np.ndarray] = None
    offsets: Optional[np.ndarray] = None
    indexes: Optional[np.ndarray] = None
    dictionary: Optional[np.ndarray] = None
    dictionary_offsets: Optional[np.ndarray] = None
    headers: List[PageHeader] = field(default_factory=list)


def _run(ctx, pages, bufs_of, stream=0):
    """One batch; items that report DST_TOO_SMALL / OFFSETS_TOO_SMALL get the room they ask for and a second batch."""
    for attempt in range(3):
        b = Batch(ctx, pages)
        b.launch(stream)
        res = b.finish(stream)
        out = [Page.from_buffer_copy(res[i]) for i in range(len(pages))]
        b.close()
        again = False
        for i, r in enumerate(out):
            if r.status == STATUS.DST_TOO_SMALL and r.status_arg > 0:
                bufs_of[i]["dst"] = DeviceBuffer(int(r.status_arg) + 64, ctx.device)
                pages[i].dst, pages[i].dst_cap = bufs_of[i]["dst"].ptr, int(r.status_arg) + 64
                again = True
            elif r.status == STATUS.OFFSETS_TOO_SMALL and r.status_arg > 0:
                bufs_of[i]["offsets"] = DeviceBuffer(4 * int(r.status_arg) + 64, ctx.device)
                pages[i].offsets, pages[i].offsets_cap = bufs_of[i]["offsets"].ptr, int(r.status_arg) + 8
                again = True
        if not again:
            return out
    return out


def decode_chunk(ctx: Context, chunk: bytes, physical_type: str, type_length: int, max_def: int, max_rep: int,
                 num_values: int = -1) -> Column:
    kind = KIND_OF[physical_type]
    headers = scan_pages(chunk, num_values)
    dev = to_device(chunk, ctx.device)
    items, parts = plan_pages(headers, chunk, dev.ptr, kind, type_length, max_def, max_rep)
    byte_array = kind == KIND.BYTE_ARRAY
    width = type_length if kind == KIND.FIXED_LEN_BYTE_ARRAY else WIDTH_OF.get(kind, 1)

    def room(it, part):
        h = headers[part.page]
        n = h.num_values + 8  # a bit-packed run of dictionary indexes ends on a multiple of eight
        bufs = {}
        if part.role in (PART_REP, PART_DEF):
            bufs["dst"] = DeviceBuffer(n + 64, ctx.device)
        elif byte_array:
            # (a dictionary-encoded page may need more than this: the batch says how much and is run again)
            bufs["dst"] = DeviceBuffer(max(4 * h.uncompressed_page_size, 4096), ctx.device)
            bufs["offsets"] = DeviceBuffer(4 * (n + 8), ctx.device)
            it.offsets, it.offsets_cap = bufs["offsets"].ptr, n + 1
        elif it.encoding == ENC.RLE_DICTIONARY:
            bufs["dst"] = DeviceBuffer(n * max(width, 4) + 64, ctx.device)
        else:
            bufs["dst"] = DeviceBuffer(max(n * width, h.uncompressed_page_size) + 64, ctx.device)
        it.dst, it.dst_cap = bufs["dst"].ptr, bufs["dst"].nbytes - 32
        return bufs

    col = Column(0, None, None, headers=headers)
    # ---- the dictionary page first: data pages look their values up in it ----
    dict_buf = dict_off = None
    dict_n = 0
    for it, part in zip(items, parts):
        if part.role != PART_DICT:
            continue
        bufs = [room(it, part)]
        r = _run(ctx, [it], bufs)[0]
        if r.status != STATUS.OK:
            raise PqdError(r.status, r.status_arg, "dictionary page")
        dict_buf, dict_n = bufs[0]["dst"], int(r.out_values)
        col.dictionary = dict_buf.numpy()[: r.out_bytes].copy()
        if byte_array:
            dict_off = bufs[0]["offsets"]
            col.dictionary_offsets = dict_off.numpy()[: 4 * (dict_n + 1)].view(np.uint32).copy()
        if kind == KIND.BOOLEAN:
            dict_n = 8 * int(r.out_bytes)  # PLAIN booleans are bits: the entry count is the header's (checked by the look-up itself)
    # ---- every level and value part of every data page in one batch ----
    todo = [(it, part) for it, part in zip(items, parts) if part.role != PART_DICT]
    bufs = []
    for it, part in todo:
        bufs.append(room(it, part))
        if part.role == PART_VALUES and it.encoding == ENC.RLE_DICTIONARY and dict_buf is not None:
            it.dict, it.dict_len = dict_buf.ptr, dict_n  # Dictionary.Lookup fused into the index decode
            if byte_array:
                it.dict_offsets = dict_off.ptr
        elif part.role == PART_VALUES and it.encoding == ENC.RLE_DICTIONARY:
            it.kind = KIND.INT32  # the indexes themselves, as indexedPageType.Decode yields them (dictionary.go:217-219)
    res = _run(ctx, [it for it, _ in todo], bufs) if todo else []
    reps, defs, vals, offs, idx = [], [], [], [], []
    base = 0
    present_of = {}  # page -> slots with a value, from its definition levels (column.go:772)
    for (it, part), r, b in zip(todo, res, bufs):
        h = headers[part.page]
        if r.status != STATUS.OK:
            raise PqdError(r.status, r.status_arg, f"page {part.page} part {part.role}")
        if part.role in (PART_REP, PART_DEF):
            if r.out_values < h.num_values:  # column.go:940-942
                raise PqdError(STATUS.UNEXPECTED_EOF, r.out_values, "levels")
            lv = b["dst"].numpy()[: h.num_values].copy()
            (reps if part.role == PART_REP else defs).append(lv)
            if part.role == PART_DEF:
                present_of[part.page] = int(np.count_nonzero(lv == max_def))
    for (it, part), r, b in zip(todo, res, bufs):
        if part.role != PART_VALUES:
            continue
        h = headers[part.page]
        n = present_of.get(part.page, h.num_values)
        col.num_values += h.num_values
        if int(r.out_values) < n and kind != KIND.BOOLEAN:
            raise PqdError(STATUS.UNEXPECTED_EOF, r.out_values, f"page {part.page}: {r.out_values} values for {n} slots")
        if byte_array and (it.encoding != ENC.RLE_DICTIONARY or it.dict):
            o = b["offsets"].numpy()[: 4 * (n + 1)].view(np.uint32).astype(np.int64)
            vals.append(b["dst"].numpy()[: int(o[-1])].copy())
            offs.append(o[:-1] + base)
            base += int(o[-1])
        elif it.encoding == ENC.RLE_DICTIONARY and not it.dict:
            idx.append(b["dst"].numpy()[: 4 * n].view(np.int32).copy())
        elif kind == KIND.BOOLEAN and it.dict:
            vals.append(b["dst"].numpy()[:n].copy())  # looked up in a dictionary: a byte per value
        elif kind == KIND.BOOLEAN:
            vals.append(np.unpackbits(b["dst"].numpy()[: int(r.out_bytes)], bitorder="little")[:n].copy())
        else:
            vals.append(b["dst"].numpy()[: n * width].copy())
    col.definition_levels = np.concatenate(defs) if defs else None
    col.repetition_levels = np.concatenate(reps) if reps else None
    if vals:
        col.values = np.concatenate(vals)
    if offs:
        col.offsets = np.concatenate(offs + [np.array([base], np.int64)]).astype(np.uint32)
    if idx:
        col.indexes = np.concatenate(idx)
    return col
