#!/usr/bin/env python3
"""Build tests/golden/footers.npz + footers.json: the footer (thrift FileMetaData) of every
parquet file among the reference's test data that pyarrow -- an independent implementation --
opens, next to what pyarrow reports of it: leaf columns (path, physical type, length, maximum
levels), row groups, and per column chunk the value count, page offsets, sizes, codec and
encodings.

Run in the authoring container only (needs /root/reference/testdata and pyarrow); the outputs
are committed.  The fixtures are data: file bytes and expected values.
"""
import json
import os
import sys

import numpy as np
import pyarrow.parquet as pq

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_fixtures import TESTDATA, OUT_DIR  # noqa: E402


def main():
    arrays, manifest, skipped = {}, [], []
    for root, _, files in sorted(os.walk(TESTDATA)):
        for fname in sorted(files):
            if not fname.endswith(".parquet"):
                continue
            path = os.path.join(root, fname)
            rel = os.path.relpath(path, TESTDATA)
            raw = open(path, "rb").read()
            try:
                pf = pq.ParquetFile(path)
            except Exception as e:  # encrypted or deliberately broken files
                skipped.append((rel, type(e).__name__))
                continue
            md, schema = pf.metadata, pf.schema
            size = int.from_bytes(raw[-8:-4], "little")
            if len(raw) - 8 - size < 4 or size > (1 << 20):
                skipped.append((rel, "footer size"))
                continue
            arrays[rel + ":footer"] = np.frombuffer(raw[len(raw) - 8 - size:len(raw) - 8], np.uint8)
            arrays[rel + ":tail"] = np.frombuffer(raw[-8:], np.uint8)
            cols = []
            for i in range(md.num_columns):
                sc = schema.column(i)
                cols.append(dict(path=sc.path, physical_type=sc.physical_type, type_length=sc.length if sc.length and sc.length > 0 else 0,
                                 max_def=sc.max_definition_level, max_rep=sc.max_repetition_level))
            rgs = []
            for g in range(md.num_row_groups):
                rg = md.row_group(g)
                chunks = []
                for i in range(rg.num_columns):
                    cm = rg.column(i)
                    chunks.append(dict(num_values=cm.num_values, data_page_offset=cm.data_page_offset,
                                       dictionary_page_offset=cm.dictionary_page_offset if cm.has_dictionary_page else None,
                                       total_compressed_size=cm.total_compressed_size,
                                       total_uncompressed_size=cm.total_uncompressed_size, codec=cm.compression,
                                       encodings=sorted(set(cm.encodings)), physical_type=cm.physical_type))
                rgs.append(dict(num_rows=rg.num_rows, total_byte_size=rg.total_byte_size, chunks=chunks))
            manifest.append(dict(file=rel, file_size=len(raw), footer_len=size, num_rows=md.num_rows, version=md.format_version,
                                 columns=cols, row_groups=rgs))
    np.savez_compressed(os.path.join(OUT_DIR, "footers.npz"), **arrays)
    with open(os.path.join(OUT_DIR, "footers.json"), "w") as f:
        json.dump(dict(files=manifest, skipped=skipped), f, indent=0)
    print(len(manifest), "footers,", sum(len(m["row_groups"]) * len(m["columns"]) for m in manifest), "chunks,",
          os.path.getsize(os.path.join(OUT_DIR, "footers.npz")), "bytes; skipped", skipped)


if __name__ == "__main__":
    sys.exit(main())
