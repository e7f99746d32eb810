#!/usr/bin/env python3
"""Build tests/golden/chunks.npz + chunks.json: whole column chunks of the reference's parquet
test files, byte for byte, next to (a) the page headers a small thrift-compact reader finds in
them (the reader of make_fixtures.py, field ids as in the reference's format/parquet.go) and
(b) the column pyarrow -- an independent implementation -- reads from the same file: validity
per row, non-null values, byte-array offsets.

Run in the authoring container only (needs /root/reference/testdata and pyarrow); the outputs
are committed.  The fixtures are data: file bytes and expected values.
"""
import json
import os
import sys

import numpy as np
import pyarrow.parquet as pq

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_fixtures import Reader, TESTDATA, OUT_DIR, expected_dense  # noqa: E402

# file -> column indexes (None: all)
SELECTION = {
    "delta_binary_packed.parquet": None,
    "delta_byte_array.parquet": None,
    "delta_encoding_required_column.parquet": None,
    "delta_encoding_optional_column.parquet": None,
    "alltypes_dictionary.parquet": None,
    "alltypes_plain.parquet": None,
    "alltypes_tiny_pages.parquet": [0, 1, 2, 4, 5, 7, 9],
    "alltypes_tiny_pages_plain.parquet": [3, 6],
    "datapage_v2.snappy.parquet": None,       # compressed: page headers only
    "nested_lists.snappy.parquet": None,      # compressed, repeated: page headers only
    "nullable.impala.parquet": [0],
}


PAGE_FIELDS = ["type", "uncompressed_page_size", "compressed_page_size", "has_crc", "crc", "num_values", "encoding",
               "definition_level_encoding", "repetition_level_encoding", "num_nulls", "num_rows", "definition_levels_byte_length",
               "repetition_levels_byte_length", "is_compressed", "is_sorted", "header_offset", "data_offset"]


def main():
    arrays, manifest = {}, []
    for fname, cols in SELECTION.items():
        path = os.path.join(TESTDATA, fname)
        raw = open(path, "rb").read()
        pf = pq.ParquetFile(path)
        md, schema = pf.metadata, pf.schema
        table = pf.read(use_threads=False) if not fname.startswith("nested") and not fname.startswith("nullable") else None
        for rg_i in range(md.num_row_groups):
            rg = md.row_group(rg_i)
            row_base = sum(md.row_group(k).num_rows for k in range(rg_i))
            for ci in range(rg.num_columns):
                if cols is not None and ci not in cols:
                    continue
                cm, sc = rg.column(ci), schema.column(ci)
                start = cm.dictionary_page_offset if cm.dictionary_page_offset else cm.data_page_offset
                start = min(start, cm.data_page_offset)
                end = start + cm.total_compressed_size
                chunk = raw[start:end]
                key = f"{fname}:{rg_i}:{ci}"
                rec = dict(file=fname, row_group=rg_i, column=ci, path=cm.path_in_schema, physical_type=cm.physical_type,
                           type_length=sc.length if sc.length else 0, max_def=sc.max_definition_level,
                           max_rep=sc.max_repetition_level, num_values=cm.num_values, codec=cm.compression, pages=[])
                pos = 0
                seen = 0
                while pos < len(chunk) and seen < cm.num_values:
                    rd = Reader(chunk, pos)
                    h = rd.struct()
                    page = dict(type=h[1], uncompressed_page_size=h[2], compressed_page_size=h[3], has_crc=int(4 in h),
                                crc=h.get(4, 0), header_offset=pos, data_offset=rd.p, num_nulls=-1, num_rows=-1,
                                definition_levels_byte_length=0, repetition_levels_byte_length=0, is_compressed=1, is_sorted=0,
                                num_values=0, encoding=0, definition_level_encoding=0, repetition_level_encoding=0)
                    if h[1] == 0:
                        d = h[5]
                        page.update(num_values=d[1], encoding=d[2], definition_level_encoding=d[3], repetition_level_encoding=d[4])
                        seen += d[1]
                    elif h[1] == 2:
                        d = h[7]
                        page.update(num_values=d[1], encoding=d[2], is_sorted=int(d.get(3, False)))
                    elif h[1] == 3:
                        d = h[8]
                        page.update(num_values=d[1], num_nulls=d[2], num_rows=d[3], encoding=d[4], definition_levels_byte_length=d[5],
                                    repetition_levels_byte_length=d[6], is_compressed=int(d.get(7, True)),
                                    definition_level_encoding=3, repetition_level_encoding=3)
                        seen += d[1]
                    rec["pages"].append(page)
                    pos = rd.p + h[3]
                arrays[key + ":chunk"] = np.frombuffer(chunk, np.uint8)
                # the page table as rows of PAGE_FIELDS
                arrays[key + ":pages"] = np.array([[pg[f] for f in PAGE_FIELDS] for pg in rec.pop("pages")], np.int64).reshape(-1, len(PAGE_FIELDS))
                rec["n_pages"] = int(arrays[key + ":pages"].shape[0])
                flat = sc.max_repetition_level == 0 and "." not in cm.path_in_schema
                rec["decodable"] = bool(flat and cm.compression == "UNCOMPRESSED" and table is not None)
                if rec["decodable"]:
                    vals, offs, valid = expected_dense(table.column(ci).slice(row_base, rg.num_rows), cm.physical_type)
                    rec["has_values"] = vals is not None
                    arrays[key + ":valid"] = valid.astype(np.uint8)
                    if vals is not None:
                        arrays[key + ":values"] = np.ascontiguousarray(vals).view(np.uint8).reshape(-1)
                    if offs is not None:
                        arrays[key + ":offsets"] = offs
                manifest.append(rec)
    np.savez_compressed(os.path.join(OUT_DIR, "chunks.npz"), **arrays)
    with open(os.path.join(OUT_DIR, "chunks.json"), "w") as f:
        json.dump(dict(page_fields=PAGE_FIELDS, chunks=manifest), f, indent=0)
    print(len(manifest), "chunks,", sum(r["n_pages"] for r in manifest), "pages,",
          os.path.getsize(os.path.join(OUT_DIR, "chunks.npz")), "bytes")


if __name__ == "__main__":
    sys.exit(main())
