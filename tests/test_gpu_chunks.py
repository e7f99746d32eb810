"""Whole column chunks of the reference's parquet test files, from file bytes to columns: page
headers and parts from the host entry points, every level and value part decoded on the GPU,
compared with what pyarrow reads from the same files (tests/golden/make_chunks.py)."""
import numpy as np
import pytest

from tests import chunks as K
from tests.gpu_util import ctx, pkg

pytestmark = pytest.mark.gpu


def _decodable():
    for rec, arr in K.chunks():
        if rec["decodable"]:
            yield rec, arr


def test_chunks_decode_to_what_pyarrow_reads():
    P = pkg()
    seen = set()
    checked = values_checked = dict_ba = 0
    for rec, arr in _decodable():
        col = P.chunk.decode_chunk(ctx(), arr["chunk"].tobytes(), rec["physical_type"], rec["type_length"], rec["max_def"],
                                   rec["max_rep"], rec["num_values"])
        what = K.chunk_id(rec)
        valid = arr["valid"]
        assert col.num_values == rec["num_values"] == valid.size, what
        if rec["max_def"] > 0:
            assert np.array_equal(col.definition_levels, valid * rec["max_def"]), what
        else:
            assert col.definition_levels is None and valid.all()
        seen.update(int(h.encoding) for h in col.headers)
        checked += 1
        if not rec["has_values"]:
            continue
        want = arr["values"]
        assert col.indexes is None, what  # dictionary pages of every type are looked up on the device
        if rec["physical_type"] == "BYTE_ARRAY":
            got, got_offsets = col.values.tobytes() if col.values is not None else b"", col.offsets
            dict_ba += col.dictionary_offsets is not None
            assert got == want.tobytes(), what
            assert np.array_equal(got_offsets, arr["offsets"]), what
        else:
            got = col.values if col.values is not None else np.zeros(0, np.uint8)
            assert got.tobytes() == want.tobytes(), what
        values_checked += 1
    assert checked > 100 and values_checked > 90
    assert dict_ba >= 3  # alltypes_dictionary / alltypes_tiny_pages string columns among them
    # PLAIN, PLAIN_DICTIONARY, DELTA_BINARY_PACKED, DELTA_BYTE_ARRAY pages all came by
    assert {0, 2, 5, 7} <= seen


def test_optional_chunk_expands_on_the_device():
    # a chunk with nulls: values placed into their slots by the device kernels
    P = pkg()
    done = 0
    for rec, arr in _decodable():
        if rec["max_def"] == 0 or arr["valid"].all() or not rec["has_values"] or rec["physical_type"] not in ("INT32", "INT64", "DOUBLE", "FLOAT"):
            continue
        col = P.chunk.decode_chunk(ctx(), arr["chunk"].tobytes(), rec["physical_type"], rec["type_length"], rec["max_def"],
                                   rec["max_rep"], rec["num_values"])
        w = P.chunk.WIDTH_OF[P.chunk.KIND_OF[rec["physical_type"]]]
        n_present = col.values.size // w
        out = P.levels.optional_expand(ctx(), col.definition_levels, rec["max_def"], col.values, n_present, w)
        assert out.status == 0 and out.present == int(arr["valid"].sum())
        assert np.array_equal(out.validity, arr["valid"])
        slots = out.values.reshape(-1, w)
        assert slots[arr["valid"] == 1].tobytes() == arr["values"].tobytes(), K.chunk_id(rec)
        assert not slots[arr["valid"] == 0].any()
        assert P.levels.count_levels_not_equal(ctx(), col.definition_levels, rec["max_def"]) == int((arr["valid"] == 0).sum())
        done += 1
    assert done >= 3
