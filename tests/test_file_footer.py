"""pqd_file_footer_range / pqd_file_parse_footer (host code of the C ABI, no GPU needed) against
what pyarrow reports of the footers of the reference's parquet test files
(tests/golden/make_footers.py), against files written here, and on damaged footers."""
import io
import json
import os

import numpy as np
import pytest

import __graft_entry__ as entry

P = entry.load_package()
GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
ENCODING_NAMES = {"PLAIN": 0, "PLAIN_DICTIONARY": 2, "RLE": 3, "BIT_PACKED": 4, "DELTA_BINARY_PACKED": 5,
                  "DELTA_LENGTH_BYTE_ARRAY": 6, "DELTA_BYTE_ARRAY": 7, "RLE_DICTIONARY": 8, "BYTE_STREAM_SPLIT": 9}


def _load():
    with open(os.path.join(GOLDEN, "footers.json")) as f:
        man = json.load(f)
    return man["files"], np.load(os.path.join(GOLDEN, "footers.npz"))


def _check(rec, info, columns, row_groups, chunks):
    what = rec["file"]
    assert info.num_rows == rec["num_rows"], what
    assert len(columns) == len(rec["columns"]), what
    for c, want in zip(columns, rec["columns"]):
        assert ".".join(c.path) == want["path"], what
        assert c.physical_type == want["physical_type"], (what, want["path"])
        assert c.max_definition_level == want["max_def"] and c.max_repetition_level == want["max_rep"], (what, want["path"])
        if want["physical_type"] == "FIXED_LEN_BYTE_ARRAY":
            assert c.type_length == want["type_length"], (what, want["path"])
    assert len(row_groups) == len(rec["row_groups"]), what
    for g, (rg, want_rg) in enumerate(zip(row_groups, rec["row_groups"])):
        assert rg.num_rows == want_rg["num_rows"] and rg.total_byte_size == want_rg["total_byte_size"], (what, g)
        for k, want in zip(chunks[g], want_rg["chunks"]):
            where = (what, g, k.column)
            assert k.has_metadata and not k.is_encrypted, where
            assert k.num_values == want["num_values"], where
            assert k.data_page_offset == want["data_page_offset"], where
            assert k.total_compressed_size == want["total_compressed_size"], where
            assert k.total_uncompressed_size == want["total_uncompressed_size"], where
            # (pyarrow says LZ4 for the LZ4_RAW codec)
            assert P.file.CODEC_NAMES[k.codec] == want["codec"] or (k.codec == 7 and want["codec"] == "LZ4"), where
            assert P.file.KIND_NAMES[k.kind] == want["physical_type"], where
            assert sorted(k.encodings) == sorted({ENCODING_NAMES[e] for e in want["encodings"]}), where
            if want["dictionary_page_offset"] is not None:
                assert k.dictionary_page_offset == want["dictionary_page_offset"], where
            first = k.dictionary_page_offset if k.dictionary_page_offset else k.data_page_offset
            assert k.offset == first, where


def test_footers_of_the_reference_test_files():
    files, arrays = _load()
    assert len(files) >= 50
    n_chunks = nested = 0
    for rec in files:
        footer, tail = arrays[rec["file"] + ":footer"].tobytes(), arrays[rec["file"] + ":tail"].tobytes()
        assert P.file.footer_range(tail, rec["file_size"]) == (rec["file_size"] - 8 - rec["footer_len"], rec["footer_len"])
        info, columns, row_groups, chunks = P.file.parse_footer(footer)
        _check(rec, info, columns, row_groups, chunks)
        assert info.footer_bytes_read == len(footer), rec["file"]
        n_chunks += len(columns) * len(row_groups)
        nested += any(c.max_repetition_level > 1 for c in columns)
    assert n_chunks >= 700 and nested >= 2


def _pyarrow_file(**kw):
    pa = pytest.importorskip("pyarrow")
    import pyarrow.parquet as pq
    n = 5000
    rng = np.random.default_rng(5)
    table = pa.table({
        "id": pa.array(np.arange(n, dtype=np.int64)),
        "x": pa.array(rng.standard_normal(n), mask=rng.random(n) < 0.1),
        "s": pa.array([f"v{i % 37}" for i in range(n)]),
        "tags": pa.array([[f"t{j}" for j in range(i % 4)] if i % 11 else None for i in range(n)], pa.list_(pa.string())),
        "pt": pa.array([{"a": i, "b": [{"c": float(i), "d": None if i % 3 else i}] * (i % 3)} for i in range(n)]),
        "fx": pa.array([bytes([i % 256] * 5) for i in range(n)], pa.binary(5)),
    })
    buf = io.BytesIO()
    pq.write_table(table, buf, row_group_size=1500, **kw)
    raw = buf.getvalue()
    return raw, pq.ParquetFile(io.BytesIO(raw)), table


@pytest.mark.parametrize("kw", [dict(compression="NONE"), dict(compression="NONE", use_dictionary=False, data_page_version="2.0"),
                                dict(compression="snappy", column_encoding=None)])
def test_footer_of_a_file_written_here(kw):
    kw = {k: v for k, v in kw.items() if v is not None}
    raw, pf, _ = _pyarrow_file(**kw)
    f = P.file.File(raw)
    md, schema = pf.metadata, pf.schema
    assert f.num_rows == md.num_rows and len(f.row_groups) == md.num_row_groups == 4 and len(f.columns) == md.num_columns
    for i, c in enumerate(f.columns):
        sc = schema.column(i)
        assert ".".join(c.path) == sc.path and c.physical_type == sc.physical_type
        assert (c.max_definition_level, c.max_repetition_level) == (sc.max_definition_level, sc.max_repetition_level)
    for g in range(md.num_row_groups):
        for i in range(md.num_columns):
            cm, k = md.row_group(g).column(i), f.chunks[g][i]
            assert (k.num_values, k.data_page_offset, k.total_compressed_size) == (cm.num_values, cm.data_page_offset, cm.total_compressed_size)
            assert P.file.CODEC_NAMES[k.codec] == cm.compression
            # the range holds exactly the chunk's pages
            headers = P.chunk.scan_pages(f.chunk_bytes(g, i), k.num_values)
            assert sum(h.num_values for h in headers if h.type in (0, 3)) == k.num_values
            last = headers[-1]
            assert last.data_offset + last.compressed_page_size == k.total_compressed_size


def test_field_names_that_contain_a_dot_keep_their_components():
    """Column.Path is a []string (column.go:378-383): "a.b" inside "s" and "b" inside "s.a" are different columns."""
    pa = pytest.importorskip("pyarrow")
    import pyarrow.parquet as pq
    table = pa.table({
        "s": pa.array([{"a.b": i, "a": {"b": i + 1}} for i in range(10)]),
        "s.a": pa.array([{"b": float(i)} for i in range(10)]),
        "plain.dotted": pa.array(range(10), pa.int32()),
    })
    buf = io.BytesIO()
    pq.write_table(table, buf, compression="NONE")
    f = P.file.File(buf.getvalue())
    assert [c.path for c in f.columns] == [["s", "a.b"], ["s", "a", "b"], ["s.a", "b"], ["plain.dotted"]]


def test_damaged_footers_fail_without_harm():
    files, arrays = _load()
    rec = max(files, key=lambda r: r["footer_len"] if r["footer_len"] < 20000 else 0)
    footer = arrays[rec["file"] + ":footer"].tobytes()
    tail = arrays[rec["file"] + ":tail"].tobytes()
    # every truncation: an error or (when only optional trailing fields went) fewer tables, never a crash
    bad = 0
    for cut in range(0, len(footer), max(1, len(footer) // 400)):
        try:
            P.file.parse_footer(footer[:cut])
        except P.PqdError as e:
            assert e.status in (P.STATUS.UNEXPECTED_EOF, P.STATUS.INVALID_ARGUMENT)
            bad += 1
    assert bad > 300
    rng = np.random.default_rng(11)
    for _ in range(300):
        b = bytearray(footer)
        for p in rng.integers(0, len(b), size=int(rng.integers(1, 6))):
            b[p] = int(rng.integers(0, 256))
        try:
            P.file.parse_footer(bytes(b))
        except P.PqdError:
            pass
    for _ in range(200):
        try:
            P.file.parse_footer(rng.integers(0, 256, size=int(rng.integers(0, 400)), dtype=np.uint8).tobytes())
        except P.PqdError:
            pass
    # lists nested without end, a group that claims more children than the schema has
    with pytest.raises(P.PqdError):
        P.file.parse_footer(b"\x19" * 5000)
    with pytest.raises(P.PqdError) as e:
        #           version=1      schema: list of 2 structs   root{name "r", num_children 7}   leaf{type 1, name "a"}    stop
        P.file.parse_footer(bytes([0x15, 2, 0x19, 0x2c, 0x48, 1, ord("r"), 0x15, 14, 0, 0x15, 2, 0x38, 1, ord("a"), 0, 0]))
    assert e.value.status == P.STATUS.INVALID_ARGUMENT
    # magic and length
    with pytest.raises(P.PqdError) as e:
        P.file.footer_range(tail[:4] + b"PARE", rec["file_size"])
    assert e.value.status == P.STATUS.NOT_SUPPORTED
    with pytest.raises(P.PqdError) as e:
        P.file.footer_range(tail[:4] + b"PAR2", rec["file_size"])
    assert e.value.status == P.STATUS.INVALID_ARGUMENT
    with pytest.raises(P.PqdError) as e:
        P.file.footer_range(tail, rec["footer_len"] + 7)
    assert e.value.status == P.STATUS.UNEXPECTED_EOF
    with pytest.raises(P.PqdError) as e:
        P.file.footer_range(tail[:7], 7)
    assert e.value.status == P.STATUS.UNEXPECTED_EOF


def test_row_group_must_hold_one_chunk_per_leaf():
    # version, schema [root{name, num_children 1}, leaf{type INT32, name}], num_rows, row_groups [ {columns: [] , ...} ]
    footer = bytes([0x15, 2, 0x19, 0x2c, 0x48, 1, ord("r"), 0x15, 2, 0, 0x15, 2, 0x38, 1, ord("a"), 0,
                    0x16, 0, 0x19, 0x1c, 0x19, 0x0c, 0x16, 0, 0x16, 0, 0, 0])
    with pytest.raises(P.PqdError) as e:
        P.file.parse_footer(footer)
    assert e.value.status == P.STATUS.INVALID_ARGUMENT and e.value.arg == 0
