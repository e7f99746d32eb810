#!/usr/bin/env python3
"""Build tests/golden/pages.npz + pages.json from the reference's parquet test files.

Run in the authoring container only (needs /root/reference/testdata and
pyarrow); the outputs are committed so nothing here runs on the GPU box.

For every selected column chunk the script walks the pages itself (a small
thrift-compact PageHeader reader, field ids as in the reference's
format/parquet.go:752-821,910-953), undoes block compression with pyarrow's
codecs (decompression is out of the decode path's scope), slices each page into
repetition-level bytes, definition-level bytes and value bytes the way
column.go:733-843 does, and stores them next to the values pyarrow -- an
independent implementation -- decodes from the same file.  The fixtures are
data: page payload bytes and expected values.
"""
import json
import os
import struct
import sys
import zlib

import numpy as np
import pyarrow as pa
import pyarrow.parquet as pq

TESTDATA = "/root/reference/testdata"
OUT_DIR = os.path.dirname(os.path.abspath(__file__))

# file -> (column indexes or None for all, max data pages kept per column)
SELECTION = {
    "delta_binary_packed.parquet": (None, 8),
    "delta_byte_array.parquet": (None, 8),
    "delta_encoding_required_column.parquet": (None, 8),
    "delta_encoding_optional_column.parquet": (None, 8),
    "delta_length_byte_array.parquet": (None, 8),
    "alltypes_dictionary.parquet": (None, 8),
    "alltypes_tiny_pages.parquet": ([0, 1, 2, 4, 5, 7, 9], 6),
    "datapage_v2.snappy.parquet": ([0, 1, 2, 3], 8),
    "rle_boolean_encoding.parquet": (None, 8),
    "nested_lists.snappy.parquet": ([1], 8),
}


# ---- thrift compact protocol (just enough for PageHeader) ------------------
class Reader:
    def __init__(self, buf, pos=0):
        self.b, self.p = buf, pos

    def byte(self):
        v = self.b[self.p]
        self.p += 1
        return v

    def uvarint(self):
        x = s = 0
        while True:
            c = self.byte()
            x |= (c & 0x7F) << s
            if c < 0x80:
                return x
            s += 7

    def zigzag(self):
        u = self.uvarint()
        return (u >> 1) ^ -(u & 1)

    def skip(self, t):
        if t in (1, 2):
            return
        if t == 3:
            self.p += 1
        elif t in (4, 5, 6):
            self.uvarint()
        elif t == 7:
            self.p += 8
        elif t == 8:
            n = self.uvarint()
            self.p += n
        elif t in (9, 10):
            h = self.byte()
            n, et = h >> 4, h & 0xF
            if n == 15:
                n = self.uvarint()
            for _ in range(n):
                self.skip(et)
        elif t == 12:
            self.struct()
        else:
            raise ValueError(f"thrift type {t}")

    def struct(self):
        """Returns {field id: value}; nested structs are dicts, bools are bools."""
        out, last = {}, 0
        while True:
            h = self.byte()
            if h == 0:
                return out
            t, d = h & 0xF, h >> 4
            fid = last + d if d else self.zigzag()
            last = fid
            if t == 1:
                out[fid] = True
            elif t == 2:
                out[fid] = False
            elif t in (4, 5, 6):
                out[fid] = self.zigzag()
            elif t == 3:
                out[fid] = self.byte()
            elif t == 12:
                out[fid] = self.struct()
            else:
                self.skip(t)


def decompress(codec, data, size):
    if codec == "UNCOMPRESSED":
        return bytes(data)
    if codec == "GZIP":
        return zlib.decompress(bytes(data), 31)
    return pa.decompress(bytes(data), decompressed_size=size, codec=codec.lower()).to_pybytes()


def bit_len(v):
    return int(v).bit_length()


def expected_dense(col, physical):
    """Non-null values of an arrow column as (numpy bytes, offsets or None)."""
    col = col.combine_chunks() if isinstance(col, pa.ChunkedArray) else col
    valid = np.ones(len(col), bool) if col.null_count == 0 else ~np.asarray(col.is_null())
    dense = col.drop_null()
    t = dense.type
    if pa.types.is_binary(t) or pa.types.is_string(t) or pa.types.is_large_binary(t):
        vals = [v.as_py() for v in dense]
        vals = [v.encode() if isinstance(v, str) else v for v in vals]
        offs = np.zeros(len(vals) + 1, np.uint32)
        offs[1:] = np.cumsum([len(v) for v in vals])
        return np.frombuffer(b"".join(vals), np.uint8), offs, valid
    if pa.types.is_boolean(t):
        return np.asarray(dense.to_numpy(zero_copy_only=False), np.uint8), None, valid
    if pa.types.is_timestamp(t) or pa.types.is_date(t) or pa.types.is_decimal(t) or pa.types.is_fixed_size_binary(t):
        return None, None, valid
    out = np.ascontiguousarray(dense.to_numpy(zero_copy_only=False))
    cast = {"INT32": np.int32, "INT64": np.int64, "FLOAT": np.float32, "DOUBLE": np.float64}.get(physical)
    if cast is not None and out.dtype != cast:
        out = out.astype(cast)  # logical INT8/INT16 columns are stored as INT32
    return out, None, valid


def main():
    arrays, manifest = {}, []
    for fname, (cols, max_pages) in SELECTION.items():
        path = os.path.join(TESTDATA, fname)
        raw = open(path, "rb").read()
        pf = pq.ParquetFile(path)
        md, schema = pf.metadata, pf.schema
        table = pf.read(use_threads=False)
        for rg_i in range(md.num_row_groups):
            rg = md.row_group(rg_i)
            row_base = sum(md.row_group(k).num_rows for k in range(rg_i))
            for ci in range(rg.num_columns):
                if cols is not None and ci not in cols:
                    continue
                cm = rg.column(ci)
                sc = schema.column(ci)
                max_def, max_rep = sc.max_definition_level, sc.max_repetition_level
                flat = max_rep == 0 and "." not in cm.path_in_schema
                start = cm.dictionary_page_offset if cm.dictionary_page_offset else cm.data_page_offset
                start = min(start, cm.data_page_offset)
                end = start + cm.total_compressed_size
                exp_vals = exp_offs = valid = None
                if flat:
                    exp_vals, exp_offs, valid = expected_dense(table.column(ci).slice(row_base, rg.num_rows), cm.physical_type)
                pos, page_i, data_pages, slot = start, 0, 0, 0
                vcur = 0  # dense values consumed so far
                while pos < end and data_pages < max_pages:
                    rd = Reader(raw, pos)
                    h = rd.struct()
                    ptype, usize, csize = h[1], h[2], h[3]
                    body = raw[rd.p: rd.p + csize]
                    pos = rd.p + csize
                    rec = dict(file=fname, row_group=rg_i, column=ci, path=cm.path_in_schema,
                               physical_type=cm.physical_type, type_length=sc.length if sc.length else 0,
                               max_def=max_def, max_rep=max_rep, page=page_i, codec=cm.compression)
                    key = f"{fname}:{rg_i}:{ci}:{page_i}"
                    page_i += 1
                    if ptype == 2:  # dictionary page
                        dh = h[7]
                        rec.update(kind="dict", num_values=dh[1], encoding=dh[2])
                        arrays[key + ":values"] = np.frombuffer(decompress(cm.compression, body, usize), np.uint8)
                        manifest.append(rec)
                        continue
                    if ptype == 0:  # data page v1: levels and values compressed together
                        dh = h[5]
                        nv = dh[1]
                        data = decompress(cm.compression, body, usize)
                        rep = dfl = b""
                        p = 0
                        if max_rep > 0:
                            (ln,) = struct.unpack_from("<I", data, p)
                            rep = data[p + 4: p + 4 + ln]
                            p += 4 + ln
                        if max_def > 0:
                            (ln,) = struct.unpack_from("<I", data, p)
                            dfl = data[p + 4: p + 4 + ln]
                            p += 4 + ln
                        vals = data[p:]
                        rec.update(kind="v1", num_values=nv, encoding=dh[2], def_encoding=dh[3], rep_encoding=dh[4])
                    elif ptype == 3:  # data page v2: levels uncompressed in front
                        dh = h[8]
                        nv, nnull = dh[1], dh[2]
                        dl, rl = dh[5], dh[6]
                        rep, dfl = body[:rl], body[rl: rl + dl]
                        rest = body[rl + dl:]
                        compressed = dh.get(7, True)
                        vals = decompress(cm.compression, rest, usize - rl - dl) if compressed else bytes(rest)
                        rec.update(kind="v2", num_values=nv, num_nulls=nnull, encoding=dh[4], def_encoding=3, rep_encoding=3)
                    else:
                        continue
                    data_pages += 1
                    rec["def_bit_width"] = bit_len(max_def)
                    rec["rep_bit_width"] = bit_len(max_rep)
                    arrays[key + ":rep"] = np.frombuffer(bytes(rep), np.uint8)
                    arrays[key + ":def"] = np.frombuffer(bytes(dfl), np.uint8)
                    arrays[key + ":values"] = np.frombuffer(bytes(vals), np.uint8)
                    if flat and valid is not None:
                        # a flat column has one slot per row
                        v = valid[slot: slot + nv]
                        nn = int(v.sum())
                        if max_def > 0:
                            arrays[key + ":expected_def"] = (v.astype(np.uint8) * max_def)
                        rec["num_non_null"] = nn
                        if exp_vals is not None:
                            if exp_offs is not None:
                                o = exp_offs[vcur: vcur + nn + 1]
                                arrays[key + ":expected_offsets"] = (o - o[0]).astype(np.uint32)
                                arrays[key + ":expected_values"] = exp_vals[o[0]: o[-1]]
                            else:
                                arrays[key + ":expected_values"] = exp_vals[vcur: vcur + nn]
                        vcur += nn
                        slot += nv
                    manifest.append(rec)
    np.savez_compressed(os.path.join(OUT_DIR, "pages.npz"), **arrays)
    with open(os.path.join(OUT_DIR, "pages.json"), "w") as f:
        json.dump(manifest, f, indent=0)
    print(f"{len(manifest)} pages, {len(arrays)} arrays,",
          os.path.getsize(os.path.join(OUT_DIR, "pages.npz")), "bytes")


if __name__ == "__main__":
    sys.exit(main())
