"""Loader for tests/golden/pages.{json,npz}: page payloads cut from the reference's
parquet test files plus the values pyarrow decodes from them (see
tests/golden/make_fixtures.py)."""
import json
import os

import numpy as np

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
_cache = {}

WIDTH = {"INT32": 4, "INT64": 8, "FLOAT": 4, "DOUBLE": 8, "INT96": 12, "BOOLEAN": 1}
NP = {"INT32": np.int32, "INT64": np.int64, "FLOAT": np.float32, "DOUBLE": np.float64}


def load():
    if not _cache:
        _cache["manifest"] = json.load(open(os.path.join(_DIR, "pages.json")))
        _cache["arrays"] = dict(np.load(os.path.join(_DIR, "pages.npz")))
    return _cache["manifest"], _cache["arrays"]


def key(rec):
    return f"{rec['file']}:{rec['row_group']}:{rec['column']}:{rec['page']}"


def pages(kind=None, encoding=None, physical=None):
    manifest, arrays = load()
    for rec in manifest:
        if kind is not None and rec["kind"] not in kind:
            continue
        if encoding is not None and rec["encoding"] not in encoding:
            continue
        if physical is not None and rec["physical_type"] not in physical:
            continue
        yield rec, {k.split(":")[-1]: v for k, v in arrays.items() if k.startswith(key(rec) + ":")}


def dictionary_of(rec):
    """The dictionary page of the column chunk a data page belongs to, or None."""
    manifest, arrays = load()
    for d in manifest:
        if d["kind"] == "dict" and (d["file"], d["row_group"], d["column"]) == (rec["file"], rec["row_group"], rec["column"]):
            return d, arrays[key(d) + ":values"]
    return None, None


def page_id(rec):
    return f"{rec['file'].split('.')[0]}-c{rec['column']}-p{rec['page']}"
