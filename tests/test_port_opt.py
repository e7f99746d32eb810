"""The tuned CPU port behind bench.py's cpu_baseline (oracle/pq_port_opt.c) against the plain
restatement it stands beside (oracle/pq_oracle.c, pinned by tests/test_oracle_kat.py and
tests/test_oracle_fixtures.py): same values, statuses and bytes consumed, on whole pages of every
width and block geometry and on damaged ones.  CPU only."""
import numpy as np
import pytest

from oracle import pqoracle as O
from tests import refdata


def _same(a, b, what):
    assert (a.status, a.arg) == (b.status, b.arg), what
    if a.status == 0:
        assert a.n_values == b.n_values and a.consumed == b.consumed, what
        assert np.array_equal(a.values, b.values), what


@pytest.mark.parametrize("bw", list(range(0, 65)))
def test_delta_int64_every_width(bw):
    rng = np.random.default_rng(100 + bw)
    for n in (1, 2, 31, 32, 33, 127, 128, 129, 130, 1000, 4097):
        mask = (1 << bw) - 1 if bw < 64 else -1
        d = rng.integers(-2**63, 2**63 - 1, size=n, dtype=np.int64) & np.int64(mask if mask < 2**63 else -1)
        v = np.cumsum(d.astype(np.uint64)).astype(np.int64)
        for bs, mb in ((128, 4), (256, 2), (128, 1), (1024, 32)):
            enc = O.delta_encode_int64(v, bs, mb)
            _same(O.opt_delta_decode_int64(enc), O.delta_decode_int64(enc), f"bw {bw} n {n} block {bs}/{mb}")
            # garbage behind the page is not consumed
            _same(O.opt_delta_decode_int64(enc + b"\x01\x02\x03"), O.delta_decode_int64(enc + b"\x01\x02\x03"), "trailing bytes")


def test_delta_int64_reference_tables_and_damage():
    rng = np.random.default_rng(7)
    for v in refdata.INT64_TESTS:
        enc = O.delta_encode_int64(np.array(v, np.int64))
        _same(O.opt_delta_decode_int64(enc), O.delta_decode_int64(enc), "reference table")
    v = np.cumsum(rng.integers(-5000, 5000, size=3000, dtype=np.int64))
    enc = bytearray(O.delta_encode_int64(v))
    for cut in list(range(0, 40)) + list(range(40, len(enc), 37)):
        _same(O.opt_delta_decode_int64(bytes(enc[:cut])), O.delta_decode_int64(bytes(enc[:cut])), f"cut {cut}")
    for k in range(400):
        bad = bytearray(enc)
        for _ in range(1 + k % 3):
            bad[int(rng.integers(0, len(bad)))] = int(rng.integers(0, 256))
        _same(O.opt_delta_decode_int64(bytes(bad)), O.delta_decode_int64(bytes(bad)), f"damage {k}")


def _plain_lookup(enc, num_values, dictionary):
    r = O.rle_dict_decode_int32(enc)
    if r.status != 0:
        return r
    idx = O.indexed_page_fixup(r.values, num_values)
    out = O.dict_lookup(dictionary, idx)
    out.values = out.values.view(np.int64) if out.status == 0 else out.values
    return out


@pytest.mark.parametrize("bw", list(range(0, 33)))
def test_rle_dictionary_int64_every_width(bw):
    rng = np.random.default_rng(200 + bw)
    dn = min(1 << min(bw, 20), 1 << 20) if bw else 1
    dictionary = rng.integers(-2**62, 2**62, size=max(dn, 1), dtype=np.int64)
    for n in (1, 7, 8, 9, 64, 1000, 5000):
        hi = min(dn, 1 << bw) if bw else 1
        idx = rng.integers(0, hi, size=n, dtype=np.int64).astype(np.int32)
        if bw and n > 8:
            idx[-1] = hi - 1 if bw <= 20 else idx[-1]
        runs = idx.copy()
        runs[n // 4: n // 2] = runs[n // 4] if n >= 4 else runs[0]  # a repeated run in the middle
        for vals in (idx, runs):
            enc = O.rle_dict_encode_int32(vals)
            for nv in (n, max(1, n - 3), n + 11):
                a, b = O.opt_rle_dict_lookup_int64(enc, nv, dictionary), _plain_lookup(enc, nv, dictionary)
                assert (a.status, a.arg) == (b.status, b.arg), (bw, n, nv)
                if a.status == 0:
                    assert np.array_equal(a.values, b.values), (bw, n, nv)


def test_rle_dictionary_int64_errors_follow_the_restatement():
    rng = np.random.default_rng(9)
    dictionary = rng.integers(-2**62, 2**62, size=100, dtype=np.int64)
    idx = rng.integers(0, 100, size=2000, dtype=np.int32)
    idx[700] = 100  # outside the dictionary
    enc = O.rle_dict_encode_int32(idx)
    a, b = O.opt_rle_dict_lookup_int64(enc, 2000, dictionary), _plain_lookup(enc, 2000, dictionary)
    assert (a.status, a.arg) == (b.status, b.arg) == (15, 700)
    good = bytearray(O.rle_dict_encode_int32(rng.integers(0, 100, size=2000, dtype=np.int32)))
    for k in range(300):
        bad = bytearray(good)
        bad[int(rng.integers(0, len(bad)))] = int(rng.integers(0, 256))
        cut = int(rng.integers(1, len(bad) + 1)) if k % 2 else len(bad)
        a, b = O.opt_rle_dict_lookup_int64(bytes(bad[:cut]), 2000, dictionary), _plain_lookup(bytes(bad[:cut]), 2000, dictionary)
        assert (a.status, a.arg) == (b.status, b.arg), k
        if a.status == 0:
            assert np.array_equal(a.values, b.values), k
    a = O.opt_rle_dict_lookup_int64(b"", 5, dictionary)
    assert a.status == 0 and np.array_equal(a.values, np.full(5, dictionary[0]))
