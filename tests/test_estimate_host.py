"""EstimateDecodeByteArraySize (encoding/delta/length_byte_array.go:65-70, byte_array.go:176-183,
plain.go:106-108) is answered on the host from the page's length streams, so it runs here without
a GPU.  It is checked against the CPU checker on well-formed pages and on damaged ones, where the
reference ignores the decode error and sums what its buffer holds."""
import ctypes

import numpy as np

import __graft_entry__ as entry
from oracle import pqoracle as O


def _estimate(code, raw: bytes) -> int:
    L = entry.load_package().lib()
    buf = np.frombuffer(raw, np.uint8) if raw else np.zeros(0, np.uint8)
    ptr = ctypes.c_void_p(buf.ctypes.data) if buf.size else None
    return int(L.pqd_estimate_decode_byte_array_size(None, code, ptr, buf.size))


def _strings(rng, n, lo, hi):
    return [bytes(rng.integers(97, 123, size=int(rng.integers(lo, hi)), dtype=np.uint8)) for _ in range(n)]


def _pack(values):
    data = np.frombuffer(b"".join(values), np.uint8)
    offs = np.zeros(len(values) + 1, np.uint32)
    np.cumsum([len(v) for v in values], out=offs[1:])
    return data, offs


def test_well_formed_pages():
    entry.build()
    P = entry.load_package()
    rng = np.random.default_rng(1)
    for n, lo, hi in ((0, 0, 1), (1, 0, 9), (2, 5, 6), (129, 0, 40), (1000, 0, 3), (5000, 10, 300), (4097, 64, 65)):
        values = _strings(rng, n, lo, hi)
        if n > 10:  # shared prefixes for DELTA_BYTE_ARRAY
            values = sorted(values)
        dlba, dba = O.dlba_encode(values), O.dba_encode(values)
        assert _estimate(P.ENC.DELTA_LENGTH_BYTE_ARRAY, bytes(dlba)) == O.dlba_estimate(dlba) == sum(map(len, values))
        assert _estimate(P.ENC.DELTA_BYTE_ARRAY, bytes(dba)) == O.dba_estimate(dba) >= sum(map(len, values))
        plain = O.plain_encode_byte_array(values) if hasattr(O, "plain_encode_byte_array") else b"\0" * 17
        assert _estimate(P.ENC.PLAIN, bytes(plain)) == len(bytes(plain))
    assert _estimate(P.ENC.RLE, b"abc") == 0


def test_damaged_pages_sum_what_the_decoder_leaves():
    P = entry.load_package()
    rng = np.random.default_rng(2)
    values = sorted(_strings(rng, 700, 0, 90))
    for enc, est, code in ((O.dlba_encode, O.dlba_estimate, P.ENC.DELTA_LENGTH_BYTE_ARRAY),
                           (O.dba_encode, O.dba_estimate, P.ENC.DELTA_BYTE_ARRAY)):
        page = bytes(enc(values))
        n_len = len(page) - (sum(map(len, values)) if enc is O.dlba_encode else 0)
        # cut anywhere inside the length streams: header, block header, widths, packed deltas
        for cut in list(range(0, 40)) + [int(c) for c in rng.integers(40, n_len, size=60)]:
            assert _estimate(code, page[:cut]) == est(np.frombuffer(page[:cut], np.uint8)), (code, cut)
        # flipped bytes: bad widths (> 32), huge counts, block sizes that are not multiples of 128 ...
        for _ in range(400):
            raw = bytearray(page[: n_len])
            for at in rng.integers(0, min(len(raw), 60), size=int(rng.integers(1, 4))):
                raw[at] = int(rng.integers(0, 256))
            raw = bytes(raw)
            assert _estimate(code, raw) == est(np.frombuffer(raw, np.uint8)), (code, raw[:60].hex())


def test_random_bytes():
    P = entry.load_package()
    rng = np.random.default_rng(3)
    for k in range(600):
        # a plausible header in front so the walk gets past it now and then
        head = bytes([128, 1, 4, int(rng.integers(1, 120)), int(rng.integers(0, 256))]) if k % 2 else b""
        raw = head + bytes(rng.integers(0, 256, size=int(rng.integers(0, 300)), dtype=np.uint8))
        buf = np.frombuffer(raw, np.uint8)
        assert _estimate(P.ENC.DELTA_LENGTH_BYTE_ARRAY, raw) == O.dlba_estimate(buf), raw.hex()
        assert _estimate(P.ENC.DELTA_BYTE_ARRAY, raw) == O.dba_estimate(buf), raw.hex()
