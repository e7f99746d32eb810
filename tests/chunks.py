"""Loader for tests/golden/chunks.{json,npz}: whole column chunks of the reference's parquet
test files, their page tables and the columns pyarrow reads from them (see
tests/golden/make_chunks.py)."""
import json
import os

import numpy as np

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_cache = {}


def load():
    if not _cache:
        m = json.load(open(os.path.join(_DIR, "chunks.json")))
        _cache["fields"], _cache["chunks"] = m["page_fields"], m["chunks"]
        _cache["arrays"] = dict(np.load(os.path.join(_DIR, "chunks.npz")))
    return _cache["fields"], _cache["chunks"], _cache["arrays"]


def chunks():
    fields, recs, arrays = load()
    for rec in recs:
        key = f"{rec['file']}:{rec['row_group']}:{rec['column']}"
        yield rec, {k.split(":")[-1]: v for k, v in arrays.items() if k.startswith(key + ":")}


def chunk_id(rec):
    return f"{rec['file'].split('.')[0]}-rg{rec['row_group']}-c{rec['column']}"
