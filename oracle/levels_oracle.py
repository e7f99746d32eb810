"""CPU checker for the level post-processing entry points -- test infrastructure, never part of
the product path.  Plain restatements of the reference's loops:

  count_levels_equal / count_levels_not_equal    level.go:11-17
  optional_read_values                            page_optional.go:72-117
  repeated_row_starts                             page_repeated.go:73-98

These functions have no dependency outside /root/reference's own Go files; they are pinned by
the level and value fixtures cut from the reference's testdata (tests/golden/pages.npz: the
definition levels and the non-null values pyarrow reads from the same files).
"""
import numpy as np


def count_levels_equal(levels, value):
    """level.go:11-13 (bytealg.Count)."""
    return int(np.count_nonzero(np.asarray(levels, np.uint8) == value))


def count_levels_not_equal(levels, value):
    """level.go:15-17."""
    return int(np.asarray(levels).size) - count_levels_equal(levels, value)


def optional_read_values(def_levels, max_definition_level, values, width):
    """page_optional.go:72-117 over a whole page.  Returns (slots, validity, present, eof):
    slot i holds the next value when def_levels[i] == max_definition_level (validity 1) and
    zeros otherwise; `eof` is set when the values run out before the levels do
    (page_optional.go:95-99), the remaining slots then stay zero."""
    d = np.asarray(def_levels, np.uint8)
    v = np.frombuffer(bytes(values), np.uint8) if not isinstance(values, np.ndarray) else values.view(np.uint8).reshape(-1)
    n_values = v.size // width
    v = v[: n_values * width].reshape(n_values, width)
    out = np.zeros((d.size, width), np.uint8)
    validity = (d == max_definition_level).astype(np.uint8)
    present = int(validity.sum())
    take = min(present, n_values)
    where = np.flatnonzero(validity)
    out[where[:take]] = v[:take]
    return out.reshape(-1), validity, present, present > n_values


def optional_offsets(def_levels, max_definition_level, offsets):
    """The byte-array form: slot i spans offsets[r] .. offsets[r + present(i)], r = values
    before slot i, so null slots are empty."""
    d = np.asarray(def_levels, np.uint8)
    o = np.asarray(offsets, np.uint32)
    n_values = o.size - 1
    present = (d == max_definition_level)
    rank = np.concatenate([[0], np.cumsum(present)])
    return o[np.minimum(rank, n_values)].astype(np.uint32)


def repeated_row_starts(rep_levels):
    """page_repeated.go:73-98: the index of every repetition level equal to 0, then len."""
    r = np.asarray(rep_levels, np.uint8)
    return np.concatenate([np.flatnonzero(r == 0), [r.size]]).astype(np.int64)
