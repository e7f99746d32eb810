"""The CPU checker for level post-processing against the reference's own files: the null
counts the page headers and pyarrow report for the pages of testdata/*.parquet, and the
loops of level.go / page_optional.go / page_repeated.go restated value by value."""
import numpy as np

from oracle import levels_oracle as L
from tests import fixtures as F


def _with_levels():
    for rec, arr in F.pages(kind=("v1", "v2")):
        if "expected_def" in arr and arr["expected_def"].size:
            yield rec, arr


def test_null_counts_of_reference_pages():
    seen = nulls = 0
    for rec, arr in _with_levels():
        d = arr["expected_def"]
        assert d.size == rec["num_values"], F.page_id(rec)
        not_max = L.count_levels_not_equal(d, rec["max_def"])
        assert d.size - not_max == rec["num_non_null"], F.page_id(rec)   # column.go:772 / page_optional.go:32-34
        if rec["kind"] == "v2":
            assert not_max == rec["num_nulls"], F.page_id(rec)           # DataPageHeaderV2.num_nulls
        seen += 1
        nulls += not_max
    assert seen > 50 and nulls > 0


def _read_values_loop(def_levels, max_def, values):
    # page_optional.go:72-117, one Value at a time
    out, j = [], 0
    for d in def_levels:
        if d != max_def:
            out.append(None)
        else:
            if j >= len(values):
                return out, True
            out.append(values[j])
            j += 1
    return out, False


def test_optional_read_values_matches_the_loop():
    rng = np.random.default_rng(5)
    for n, max_def, p_null in ((0, 1, 0.5), (1, 1, 0.0), (257, 1, 0.3), (5000, 3, 0.5), (4096, 2, 1.0), (4097, 1, 0.0)):
        d = np.where(rng.random(n) < p_null, rng.integers(0, max_def, size=n), max_def).astype(np.uint8)
        present = int((d == max_def).sum())
        for n_values in (present, max(0, present - 3)):
            vals = rng.integers(1, 2**62, size=n_values, dtype=np.int64)
            want, eof = _read_values_loop(d.tolist(), max_def, vals.tolist())
            slots, validity, cnt, got_eof = L.optional_read_values(d, max_def, vals, 8)
            slots = slots.view(np.int64)
            assert cnt == present and got_eof == (present > n_values) and got_eof == eof
            for i, w in enumerate(want):
                assert validity[i] == (d[i] == max_def)
                assert slots[i] == (0 if w is None else w)
            for i in range(len(want), n):
                assert slots[i] == 0


def test_optional_values_of_reference_pages():
    # the non-null values pyarrow reads, spread over the slots by the file's definition levels
    checked = 0
    for rec, arr in _with_levels():
        if rec["physical_type"] not in F.WIDTH or rec["physical_type"] == "BOOLEAN" or rec["max_rep"] != 0 or "expected_values" not in arr:
            continue
        w = F.WIDTH[rec["physical_type"]]
        vals = arr["expected_values"].view(np.uint8).reshape(-1)
        if vals.size != rec["num_non_null"] * w:
            continue
        slots, validity, present, eof = L.optional_read_values(arr["expected_def"], rec["max_def"], vals, w)
        assert present == rec["num_non_null"] and not eof
        assert slots.reshape(-1, w)[validity == 1].tobytes() == vals.tobytes()
        assert not slots.reshape(-1, w)[validity == 0].any()
        checked += 1
    assert checked > 20


def test_row_starts():
    rep = np.array([0, 1, 1, 0, 0, 2, 1, 0], np.uint8)
    assert L.repeated_row_starts(rep).tolist() == [0, 3, 4, 7, 8]
    assert L.count_levels_equal(rep, 0) == 4                                # repeatedPage.NumRows
    assert L.repeated_row_starts(np.zeros(0, np.uint8)).tolist() == [0]
    assert L.optional_offsets(np.array([1, 0, 1, 1, 0], np.uint8), 1, np.array([0, 3, 3, 10], np.uint32)).tolist() == [0, 3, 3, 3, 10, 10]
