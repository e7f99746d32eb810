"""The CPU checker decodes page payloads cut from the reference's own parquet test
files (testdata/*.parquet, the files TestOpenFile reads: file_test.go:23-63) to the
values pyarrow, an independent implementation, reads from the same files."""
import numpy as np
import pytest

from oracle import pqoracle as O
from tests import fixtures as F

DATA = ("v1", "v2")


def _ids(items):
    return [F.page_id(r) for r, _ in items]


_DELTA = list(F.pages(kind=DATA, encoding=(5,)))
_DBA = list(F.pages(kind=DATA, encoding=(7,)))
_DLBA = list(F.pages(kind=DATA, encoding=(6,)))
_DICT = list(F.pages(kind=DATA, encoding=(2, 8)))
_LEVELS = [(r, a) for r, a in F.pages(kind=DATA) if r["max_def"] > 0 and "expected_def" in a]
_BOOL_RLE = list(F.pages(kind=DATA, encoding=(3,), physical=("BOOLEAN",)))
_PLAIN = list(F.pages(kind=DATA, encoding=(0,)))


@pytest.mark.parametrize("rec,a", _DELTA, ids=_ids(_DELTA))
def test_delta_binary_packed_pages(rec, a):
    dec = O.delta_decode_int64 if rec["physical_type"] == "INT64" else O.delta_decode_int32
    r = dec(a["values"])
    assert r.status == 0
    assert r.n_values == rec["num_non_null"]
    assert np.array_equal(r.values, a["expected_values"])


@pytest.mark.parametrize("rec,a", _DBA, ids=_ids(_DBA))
def test_delta_byte_array_pages(rec, a):
    r = O.dba_decode(a["values"])
    assert r.status == 0 and r.n_values == rec["num_non_null"]
    assert np.array_equal(r.values, a["expected_values"])
    assert np.array_equal(r.offsets, a["expected_offsets"])
    assert O.dba_estimate(a["values"]) >= r.n_bytes


@pytest.mark.parametrize("rec,a", _DLBA, ids=_ids(_DLBA))
def test_delta_length_byte_array_pages(rec, a):
    r = O.dlba_decode(a["values"])
    assert r.status == 0 and r.n_values == rec["num_non_null"]
    assert np.array_equal(r.values, a["expected_values"])
    assert np.array_equal(r.offsets, a["expected_offsets"])
    assert O.dlba_estimate(a["values"]) == r.n_bytes


@pytest.mark.parametrize("rec,a", _LEVELS, ids=_ids(_LEVELS))
def test_definition_levels(rec, a):
    # column.go:931-948: decode, then cut to the page's value count
    assert rec["def_encoding"] == 3
    r = O.rle_decode_levels(a["def"], rec["def_bit_width"])
    assert r.status == 0 and r.n_values >= rec["num_values"]
    assert np.array_equal(r.values[: rec["num_values"]], a["expected_def"])


@pytest.mark.parametrize("rec,a", _DICT, ids=_ids(_DICT))
def test_dictionary_pages(rec, a):
    # column.go:845-857 + dictionary.go:136-162,230-243: RLE_DICTIONARY indexes, fix-up, lookup
    drec, dbytes = F.dictionary_of(rec)
    assert drec is not None
    if "expected_values" not in a:
        pytest.skip("no expected values for this physical type")
    idx = O.rle_dict_decode_int32(a["values"])
    assert idx.status == 0
    nn = rec["num_non_null"]
    indexes = O.indexed_page_fixup(idx.values, nn)
    pt = rec["physical_type"]
    if pt == "BYTE_ARRAY":
        d = O.plain_decode_byte_array(dbytes)
        assert d.status == 0 and d.n_values == drec["num_values"]
        got = b"".join(d.values[d.offsets[i]: d.offsets[i + 1]].tobytes() for i in indexes)
        assert got == a["expected_values"].tobytes()
        lens = [int(d.offsets[i + 1] - d.offsets[i]) for i in indexes]
        assert np.array_equal(np.diff(a["expected_offsets"].astype(np.int64)), lens)
    else:
        d = O.plain_decode_fixed(dbytes, F.WIDTH[pt])
        assert d.status == 0 and d.n_values == drec["num_values"]
        dictionary = d.values.view(F.NP[pt])
        r = O.dict_lookup(dictionary, indexes)
        assert r.status == 0
        assert r.values.tobytes() == np.ascontiguousarray(a["expected_values"]).tobytes()


@pytest.mark.parametrize("rec,a", _BOOL_RLE, ids=_ids(_BOOL_RLE))
def test_rle_boolean_pages(rec, a):
    r = O.rle_decode_boolean(a["values"])
    assert r.status == 0
    bits = np.unpackbits(r.values, bitorder="little")[: rec["num_non_null"]]
    assert np.array_equal(bits, a["expected_values"])


@pytest.mark.parametrize("rec,a", _PLAIN, ids=_ids(_PLAIN))
def test_plain_pages(rec, a):
    if "expected_values" not in a:
        pytest.skip("no expected values for this physical type")
    pt = rec["physical_type"]
    if pt == "BOOLEAN":
        r = O.plain_decode_fixed(a["values"], 1)
        bits = np.unpackbits(r.values, bitorder="little")[: rec["num_non_null"]]
        assert r.status == 0 and np.array_equal(bits, a["expected_values"])
    elif pt == "BYTE_ARRAY":
        r = O.plain_decode_byte_array(a["values"])
        assert r.status == 0 and np.array_equal(r.values, a["expected_values"])
    else:
        r = O.plain_decode_fixed(a["values"], F.WIDTH[pt])
        assert r.status == 0 and r.values.tobytes() == np.ascontiguousarray(a["expected_values"]).tobytes()
