"""pqd_chunk_scan_pages / pqd_chunk_plan_pages (host code of the C ABI, no GPU needed) against
the page tables of the reference's parquet test files and against the page parts
tests/golden/make_fixtures.py cut from the same files."""
import ctypes as C

import numpy as np
import pytest

import __graft_entry__ as entry
from tests import chunks as K
from tests import fixtures as F

P = entry.load_package()


def test_scan_finds_every_page_header():
    fields, recs, arrays = K.load()
    pages = 0
    for rec, arr in K.chunks():
        got = P.chunk.scan_pages(arr["chunk"].tobytes(), rec["num_values"])
        want = arr["pages"]
        assert len(got) == want.shape[0] == rec["n_pages"], K.chunk_id(rec)
        for h, row in zip(got, want):
            for name, v in zip(fields, row.tolist()):
                assert getattr(h, name) == v, (K.chunk_id(rec), name)
        pages += len(got)
    assert pages > 3000
    kinds = {int(t) for _, arr in K.chunks() for t in arr["pages"][:, 0]}
    assert kinds == {0, 2, 3}  # data pages v1 and v2, dictionary pages


def test_scan_without_a_value_count_walks_to_the_end():
    for rec, arr in K.chunks():
        assert len(P.chunk.scan_pages(arr["chunk"].tobytes(), -1)) == rec["n_pages"], K.chunk_id(rec)


def test_scan_rejects_corrupt_chunks():
    rec, arr = next((r, a) for r, a in K.chunks() if r["n_pages"] >= 2)
    chunk = arr["chunk"].tobytes()
    second = int(arr["pages"][1, K.load()[0].index("header_offset")])
    with pytest.raises(P.PqdError) as e:   # cut inside the second header
        P.chunk.scan_pages(chunk[: second + 3], -1)
    assert e.value.status == P.STATUS.UNEXPECTED_EOF
    with pytest.raises(P.PqdError) as e:   # cut inside the second page
        P.chunk.scan_pages(chunk[:-1], -1)
    assert e.value.status == P.STATUS.UNEXPECTED_EOF
    # a negative compressed_page_size (zigzag varint 1 = -1) is ErrCorrupted in the reference (file.go:1414-1445)
    bad = bytes([0x15, 0x00, 0x15, 0x02, 0x15, 0x01, 0x00])
    with pytest.raises(P.PqdError) as e:
        P.chunk.scan_pages(bad, -1)
    assert e.value.status == P.STATUS.INVALID_ARGUMENT
    assert P.chunk.scan_pages(b"", -1) == []


def test_scan_bounds_the_nesting_of_skipped_collections():
    # an unknown field of type LIST whose elements are lists of lists ...: one header byte per level.
    # The reader gives up past a few dozen levels instead of recursing once per byte.
    import zlib  # noqa: F401  (the crc test below uses it)
    for elem in (0x19, 0x1A):  # list of one list / one set
        bad = bytes([0x15, 0x00]) + bytes([0xF9 if elem == 0x19 else 0xFA]) + bytes([elem]) * 300_000
        with pytest.raises(P.PqdError) as e:
            P.chunk.scan_pages(bad, -1)
        assert e.value.status == P.STATUS.UNEXPECTED_EOF
    # maps: field of type MAP, one entry, key and value both maps
    bad = bytes([0x15, 0x00, 0xFB]) + bytes([0x01, 0xBB]) * 150_000
    with pytest.raises(P.PqdError) as e:
        P.chunk.scan_pages(bad, -1)
    assert e.value.status == P.STATUS.UNEXPECTED_EOF
    # a modest nesting is still skipped: list<list<list<i32>>> with one element each, then STOP
    ok = bytes([0x15, 0x00, 0x15, 0x00, 0x15, 0x00, 0xF9, 0x19, 0x19, 0x15, 0x02, 0x00])
    assert len(P.chunk.scan_pages(ok, -1)) == 1


def _with_crc(header_fields: bytes, page: bytes, crc: int) -> bytes:
    """PageHeader{type DATA_PAGE, sizes = len(page), crc, data_page_header{n 1, PLAIN, RLE, RLE}} + page."""
    def zz(v):
        v = (v << 1) ^ (v >> 31) if v >= 0 else ((-v) << 1) - 1
        out = bytearray()
        while v >= 0x80:
            out.append((v & 0x7F) | 0x80); v >>= 7
        out.append(v)
        return bytes(out)
    if crc >= 2**31:
        crc -= 2**32
    return (bytes([0x15]) + zz(0) + bytes([0x15]) + zz(len(page)) + bytes([0x15]) + zz(len(page)) + bytes([0x15]) + zz(crc) +
            bytes([0x1C, 0x15]) + zz(1) + bytes([0x15, 0x00, 0x15, 0x06, 0x15, 0x06, 0x00, 0x00]) + page)


def test_scan_checks_page_crc32():
    # FilePages.readPage (file.go:1455-1475): a CRC other than zero must match crc32.ChecksumIEEE of the page bytes
    import zlib
    rng = np.random.default_rng(5)
    for n in (0, 1, 7, 8, 9, 1000, 4099):
        page = rng.integers(0, 256, size=n, dtype=np.uint8).tobytes()
        crc = zlib.crc32(page)
        good = _with_crc(b"", page, crc)
        hs = P.chunk.scan_pages(good + good, -1)
        assert len(hs) == 2 and hs[0].has_crc == 1 and (hs[0].crc & 0xFFFFFFFF) == crc
        if crc == 0 or n == 0:
            continue
        bad = bytearray(good + good)
        bad[-1] ^= 0x40  # damage the second page
        with pytest.raises(P.PqdError) as e:
            P.chunk.scan_pages(bytes(bad), -1)
        assert e.value.status == P.STATUS.CHECKSUM and e.value.arg == len(good)
    # a zero CRC is "not set" to the reference: no check
    assert len(P.chunk.scan_pages(_with_crc(b"", b"abc", 0), -1)) == 1
    # the reference's own files that carry checksums pass
    with_crc = 0
    for rec, arr in K.chunks():
        hs = P.chunk.scan_pages(arr["chunk"].tobytes(), rec["num_values"])
        with_crc += sum(1 for h in hs if h.has_crc and h.crc != 0)
    print("pages with a checksum in the reference files:", with_crc)


def test_plan_cuts_pages_as_the_fixture_script_does():
    # the parts of the first pages of every chunk, as make_fixtures.py (following column.go:733-843) sliced them
    manifest, parr = F.load()
    by_page = {(r["file"], r["row_group"], r["column"], r["page"]): r for r in manifest}
    checked = 0
    for rec, arr in K.chunks():
        if rec["codec"] != "UNCOMPRESSED":
            continue
        chunk = arr["chunk"].tobytes()
        headers = P.chunk.scan_pages(chunk, rec["num_values"])
        kind = P.chunk.KIND_OF[rec["physical_type"]]
        items, parts = P.chunk.plan_pages(headers, chunk, 0x1000, kind, rec["type_length"], rec["max_def"], rec["max_rep"])
        for it, part in zip(items, parts):
            fx = by_page.get((rec["file"], rec["row_group"], rec["column"], int(part.page)))
            if fx is None:
                continue
            key = F.key(fx)
            assert (it.src or 0) - 0x1000 == part.src_offset
            body = chunk[part.src_offset: part.src_offset + it.src_len]
            if part.role == P.chunk.PART_DICT:
                assert fx["kind"] == "dict" and body == parr[key + ":values"].tobytes()
                assert it.encoding == (0 if fx["encoding"] == 2 else fx["encoding"]) and it.num_values == fx["num_values"]
            elif part.role == P.chunk.PART_VALUES:
                assert body == parr[key + ":values"].tobytes(), key
                assert it.encoding == (8 if fx["encoding"] == 2 else fx["encoding"]) and it.kind == kind
                if fx["kind"] == "v2":
                    assert it.num_values == fx["num_values"] - fx["num_nulls"]
            else:
                name = ":def" if part.role == P.chunk.PART_DEF else ":rep"
                assert body == parr[key + name].tobytes(), key
                assert it.kind == P.KIND.LEVELS and it.num_values == fx["num_values"]
                assert it.bit_width == (fx["def_bit_width"] if part.role == P.chunk.PART_DEF else fx["rep_bit_width"])
                assert it.encoding == (fx["def_encoding"] if part.role == P.chunk.PART_DEF else fx["rep_encoding"])
            checked += 1
    assert checked > 300
