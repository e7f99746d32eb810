"""Host-side mirror of OpenFile (file.go:65-250) as far as a page decoder needs it: the footer
found from the file's last bytes and parsed by the C ABI (pqd_file_footer_range,
pqd_file_parse_footer) into leaf columns with their levels, row groups, and the byte range of
every column chunk (file.go:1119-1128).  `File.read_column_chunk` hands an uncompressed chunk
to `chunk.decode_chunk`, so a file goes from bytes to decoded columns through the library alone."""
import ctypes as C
from dataclasses import dataclass
from typing import List

import numpy as np

from ._lib import Context, ENC, FileChunk, FileColumn, FileInfo, FileRowGroup, KIND, PqdError, STATUS, lib
from . import chunk as _chunk

CODEC_NAMES = ["UNCOMPRESSED", "SNAPPY", "GZIP", "LZO", "BROTLI", "LZ4", "ZSTD", "LZ4_RAW"]
KIND_NAMES = {v: k for k, v in _chunk.KIND_OF.items()}


@dataclass
class LeafColumn:
    """Column of column.go for a leaf: Path, Type, MaxDefinitionLevel, MaxRepetitionLevel, Depth, Index."""
    index: int
    path: List[str]
    kind: int
    physical_type: str
    type_length: int
    repetition: int
    converted_type: int
    max_definition_level: int
    max_repetition_level: int
    depth: int
    schema_index: int


@dataclass
class ChunkRange:
    """ColumnChunk.meta_data of one row group and column; the pages are file[offset : offset + total_compressed_size]."""
    row_group: int
    column: int
    num_values: int
    offset: int
    total_compressed_size: int
    total_uncompressed_size: int
    data_page_offset: int
    dictionary_page_offset: int
    kind: int
    codec: int
    encodings: List[int]
    has_metadata: bool
    is_encrypted: bool


@dataclass
class RowGroup:
    num_rows: int
    total_byte_size: int
    file_offset: int
    total_compressed_size: int


def footer_range(tail: bytes, file_size: int):
    """(offset, length) of the thrift FileMetaData, from the last >= 8 bytes of the file."""
    buf = (C.c_uint8 * max(len(tail), 1)).from_buffer_copy(bytes(tail) or b"\0")
    off, n = C.c_uint64(0), C.c_uint64(0)
    rc = lib().pqd_file_footer_range(C.addressof(buf), len(tail), file_size, C.byref(off), C.byref(n))
    if rc != STATUS.OK:
        raise PqdError(rc, 0, "pqd_file_footer_range")
    return int(off.value), int(n.value)


def parse_footer(footer: bytes):
    """(info, columns, row_groups, chunks): the tables of pqd_file_parse_footer as Python objects;
    chunks[row_group][column]."""
    buf = (C.c_uint8 * max(len(footer), 1)).from_buffer_copy(bytes(footer) or b"\0")
    info, arg = FileInfo(), C.c_int64(0)
    L = lib()
    rc = L.pqd_file_parse_footer(C.addressof(buf), len(footer), C.byref(info), None, 0, None, 0, None, 0, None, 0, C.byref(arg))
    if rc not in (STATUS.OK, STATUS.DST_TOO_SMALL):
        raise PqdError(rc, arg.value, "pqd_file_parse_footer")
    nc, ng, nk, nb = int(info.n_columns), int(info.n_row_groups), int(info.n_chunks), int(info.names_bytes)
    cols, rgs, chs = (FileColumn * max(nc, 1))(), (FileRowGroup * max(ng, 1))(), (FileChunk * max(nk, 1))()
    names = C.create_string_buffer(max(nb, 1))
    rc = L.pqd_file_parse_footer(C.addressof(buf), len(footer), C.byref(info), cols, nc, rgs, ng, chs, nk, C.addressof(names), nb,
                                 C.byref(arg))
    if rc != STATUS.OK:
        raise PqdError(rc, arg.value, "pqd_file_parse_footer")
    raw = names.raw
    columns = []
    for i in range(nc):
        c = cols[i]
        path = raw[c.name_offset:c.name_offset + c.name_len].decode("utf-8", "replace")
        columns.append(LeafColumn(i, path.split("\x00") if path else [], c.kind, KIND_NAMES.get(c.kind, str(c.kind)), c.type_length,
                                  c.repetition, c.converted_type, c.max_definition_level, c.max_repetition_level, c.depth,
                                  c.schema_index))
    row_groups = [RowGroup(rgs[g].num_rows, rgs[g].total_byte_size, rgs[g].file_offset, rgs[g].total_compressed_size)
                  for g in range(ng)]
    chunks = []
    for g in range(ng):
        row = []
        for i in range(nc):
            k = chs[g * nc + i]
            row.append(ChunkRange(g, i, k.num_values, k.offset, k.total_compressed_size, k.total_uncompressed_size,
                                  k.data_page_offset, k.dictionary_page_offset, k.kind, k.codec,
                                  [e for e in range(32) if k.encodings >> e & 1], bool(k.has_metadata), bool(k.is_encrypted)))
        chunks.append(row)
    return info, columns, row_groups, chunks


class File:
    """A parquet file held in memory (bytes, a memoryview or a numpy memory map of uint8)."""

    def __init__(self, data):
        self.data = data if isinstance(data, np.ndarray) else np.frombuffer(data, dtype=np.uint8)
        size = self.data.size
        if size >= 4 and bytes(self.data[:4]) not in (b"PAR1", b"PARE"):  # file.go:77-87
            raise PqdError(STATUS.INVALID_ARGUMENT, 0, "magic header")
        off, n = footer_range(bytes(self.data[max(size - 8, 0):]), size)
        self.info, self.columns, self.row_groups, self.chunks = parse_footer(bytes(self.data[off:off + n]))
        self.num_rows = int(self.info.num_rows)
        self.version = int(self.info.version)

    @classmethod
    def open(cls, path: str) -> "File":
        return cls(np.memmap(path, dtype=np.uint8, mode="r"))

    def chunk_bytes(self, row_group: int, column: int) -> bytes:
        k = self.chunks[row_group][column]
        if k.offset < 0 or k.total_compressed_size < 0 or k.offset + k.total_compressed_size > self.data.size:
            raise PqdError(STATUS.UNEXPECTED_EOF, k.offset, "column chunk range")
        return bytes(self.data[k.offset:k.offset + k.total_compressed_size])

    def read_column_chunk(self, ctx: Context, row_group: int, column: int) -> "_chunk.Column":
        """The decoded chunk.  Compressed chunks are the caller's to decompress page by page first."""
        k, col = self.chunks[row_group][column], self.columns[column]
        if k.codec != 0 or k.is_encrypted or not k.has_metadata:
            raise PqdError(STATUS.NOT_SUPPORTED, k.codec, "column chunk is compressed or encrypted")
        return _chunk.decode_chunk(ctx, self.chunk_bytes(row_group, column), col.physical_type, col.type_length,
                                   col.max_definition_level, col.max_repetition_level, k.num_values)


__all__ = ["File", "LeafColumn", "ChunkRange", "RowGroup", "footer_range", "parse_footer", "CODEC_NAMES", "ENC", "KIND"]
