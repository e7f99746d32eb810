"""GPU parity of the level post-processing entry points (null counts, optional-column
expansion, row starts) against the CPU checker, through the C ABI.  Bit exact."""
import numpy as np
import pytest

from oracle import levels_oracle as L
from tests import fixtures as F
from tests.gpu_util import ctx, pkg

pytestmark = pytest.mark.gpu

SIZES = (0, 1, 2, 63, 64, 65, 127, 128, 129, 2047, 2048, 2049, 4097, 70001, 1 << 20)


def _levels(rng, n, max_def, p_null, runs):
    if runs:  # long stretches of nulls and of values
        out = np.empty(n, np.uint8)
        at = 0
        while at < n:
            k = int(rng.geometric(1.0 / runs))
            out[at: at + k] = max_def if rng.random() >= p_null else rng.integers(0, max_def)
            at += k
        return out
    return np.where(rng.random(n) < p_null, rng.integers(0, max_def, size=n), max_def).astype(np.uint8)


def test_count_levels():
    P = pkg()
    rng = np.random.default_rng(31)
    for n in SIZES:
        for max_def in (1, 3, 7):
            d = _levels(rng, n, max_def, 0.3, 0)
            for mis in (0, 1, 5):
                buf = P.to_device(d, misalign=mis)
                assert P.levels.count_levels_equal(ctx(), buf, max_def) == L.count_levels_equal(d, max_def), (n, max_def, mis)
            assert P.levels.count_levels_not_equal(ctx(), d, max_def) == L.count_levels_not_equal(d, max_def)
            assert P.levels.count_levels_equal(ctx(), d, 0) == L.count_levels_equal(d, 0)


@pytest.mark.parametrize("width", [1, 4, 8, 12, 3])
def test_optional_expand(width):
    P = pkg()
    rng = np.random.default_rng(32 + width)
    for n in SIZES:
        for max_def, p_null, runs in ((1, 0.3, 0), (3, 0.9, 0), (1, 0.0, 0), (2, 1.0, 0), (1, 0.5, 300)):
            d = _levels(rng, n, max_def, p_null, runs)
            present = int((d == max_def).sum())
            vals = rng.integers(1, 256, size=present * width, dtype=np.uint8)
            want, validity, cnt, eof = L.optional_read_values(d, max_def, vals, width)
            out = P.levels.optional_expand(ctx(), d, max_def, vals, present, width, misalign=(n % 3) * width)
            assert out.present == cnt and out.status == 0, (n, max_def, p_null)
            assert out.values.tobytes() == want.tobytes(), (n, width, max_def, p_null, runs)
            assert out.validity.tobytes() == validity.tobytes()


def test_optional_expand_runs_out_of_values():
    # the page under the levels is shorter than the levels say: io.EOF in page_optional.go:95-99
    P = pkg()
    rng = np.random.default_rng(33)
    d = _levels(rng, 10000, 1, 0.2, 0)
    present = int((d == 1).sum())
    vals = rng.integers(1, 2**62, size=present - 100, dtype=np.int64)
    want, validity, cnt, eof = L.optional_read_values(d, 1, vals, 8)
    out = P.levels.optional_expand(ctx(), d, 1, vals, present - 100, 8)
    assert eof and out.status == P.STATUS.UNEXPECTED_EOF and out.present == present
    assert out.values.tobytes() == want.tobytes() and out.validity.tobytes() == validity.tobytes()


def test_optional_offsets():
    P = pkg()
    rng = np.random.default_rng(34)
    for n in (0, 1, 129, 5000, 70001):
        d = _levels(rng, n, 2, 0.4, 0)
        present = int((d == 2).sum())
        offs = np.concatenate([[0], np.cumsum(rng.integers(0, 30, size=present))]).astype(np.uint32)
        out = P.levels.optional_expand(ctx(), d, 2, offs, present, 4, offsets_mode=True)
        assert out.status == 0 and out.present == present
        assert np.array_equal(out.values.view(np.uint32), L.optional_offsets(d, 2, offs)), n


def test_reference_pages():
    # definition levels and non-null values of the reference's own test files
    P = pkg()
    checked = 0
    for rec, arr in F.pages(kind=("v1", "v2")):
        if "expected_def" not in arr or "expected_values" not in arr or rec["physical_type"] not in ("INT32", "INT64", "FLOAT", "DOUBLE") or rec["max_rep"] != 0:
            continue
        w = F.WIDTH[rec["physical_type"]]
        vals = arr["expected_values"].view(np.uint8).reshape(-1)
        if vals.size != rec["num_non_null"] * w:
            continue
        d = arr["expected_def"]
        assert d.size - P.levels.count_levels_not_equal(ctx(), d, rec["max_def"]) == rec["num_non_null"]
        want, validity, cnt, eof = L.optional_read_values(d, rec["max_def"], vals, w)
        out = P.levels.optional_expand(ctx(), d, rec["max_def"], vals, rec["num_non_null"], w)
        assert out.status == 0 and out.present == rec["num_non_null"]
        assert out.values.tobytes() == want.tobytes() and out.validity.tobytes() == validity.tobytes(), F.page_id(rec)
        checked += 1
    assert checked > 20


def test_row_starts():
    P = pkg()
    rng = np.random.default_rng(35)
    for n in SIZES:
        for p_zero in (0.0, 0.2, 1.0):
            r = np.where(rng.random(n) < p_zero, 0, rng.integers(1, 3, size=n)).astype(np.uint8)
            want = L.repeated_row_starts(r)
            rows, n_rows, status = P.levels.repeated_row_starts(ctx(), r)
            assert status == 0 and n_rows == want.size - 1 == L.count_levels_equal(r, 0)
            assert np.array_equal(rows[: n_rows + 1], want), (n, p_zero)
    r = np.zeros(1000, np.uint8)
    rows, n_rows, status = P.levels.repeated_row_starts(ctx(), r, row_cap=10)
    assert n_rows == 1000 and status == P.STATUS.DST_TOO_SMALL and rows.tolist() == list(range(10))


def test_row_starts_misaligned_and_short_capacity():
    P = pkg()
    rng = np.random.default_rng(36)
    for n in (15, 16, 17, 1023, 1025, 2049, 70001):
        r = np.where(rng.random(n) < 0.3, 0, 1).astype(np.uint8)
        want = L.repeated_row_starts(r)
        for mis in (1, 2, 3, 7):
            rows, n_rows, status = P.levels.repeated_row_starts(ctx(), P.to_device(r, misalign=mis))
            assert status == 0 and np.array_equal(rows[: n_rows + 1], want), (n, mis)
        cap = max(1, (want.size - 1) // 2)
        rows, n_rows, status = P.levels.repeated_row_starts(ctx(), r, row_cap=cap)
        assert n_rows == want.size - 1
        assert np.array_equal(rows[:cap], want[:cap]), n


@pytest.mark.parametrize("n", [(1 << 22) + 1, 9_000_001, (1 << 25) + 4099])
def test_long_columns(n):
    # beyond 2048 stretches of 2048 levels several workgroups share the running sums of the counts
    P = pkg()
    rng = np.random.default_rng(37)
    d = _levels(rng, n, 1, 0.3, 0)
    d[n // 3: n // 3 + 200000] = 0  # stretches without a value
    d[n // 2: n // 2 + 200000] = 1  # and full ones
    present = int((d == 1).sum())
    assert P.levels.count_levels_equal(ctx(), d, 1) == present
    vals = rng.integers(1, 2**31, size=present, dtype=np.int32)
    want, validity, cnt, eof = L.optional_read_values(d, 1, vals, 4)
    out = P.levels.optional_expand(ctx(), d, 1, vals, present, 4)
    assert out.status == 0 and out.present == cnt
    assert out.values.tobytes() == want.tobytes() and out.validity.tobytes() == validity.tobytes()
    want = L.repeated_row_starts(d)
    rows, n_rows, status = P.levels.repeated_row_starts(ctx(), d)
    assert status == 0 and n_rows == want.size - 1 and np.array_equal(rows[: n_rows + 1], want)
