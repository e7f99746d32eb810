"""From the bytes of a parquet file to decoded columns with nothing but the library: footer
(pqd_file_parse_footer), page headers and parts (pqd_chunk_*), values and levels on the GPU.
The file is written here by pyarrow, uncompressed, and compared with the table it was written from."""
import io

import numpy as np
import pytest

from tests.gpu_util import ctx, pkg

pytestmark = pytest.mark.gpu


def _file(**kw):
    pa = pytest.importorskip("pyarrow")
    import pyarrow.parquet as pq
    n = 40000
    rng = np.random.default_rng(17)
    table = pa.table({
        "id": pa.array(np.arange(n, dtype=np.int64) * 3 + 7),
        "x": pa.array(rng.standard_normal(n), mask=rng.random(n) < 0.15),
        "k": pa.array(rng.integers(0, 50, n).astype(np.int32)),
        "s": pa.array([f"value-{i % 61}" for i in range(n)]),
        "u": pa.array([f"{i:x}" * (i % 5) for i in range(n)]),
        "fx": pa.array([bytes([i % 251] * 6) for i in range(n)], pa.binary(6)),
        "tags": pa.array([[j for j in range(i % 4)] if i % 7 else None for i in range(n)], pa.list_(pa.int32())),
    })
    buf = io.BytesIO()
    pq.write_table(table, buf, row_group_size=15000, compression="NONE", **kw)
    return buf.getvalue(), table


@pytest.mark.parametrize("kw", [dict(), dict(use_dictionary=False, data_page_version="2.0"),
                                dict(use_dictionary=False, column_encoding={"id": "DELTA_BINARY_PACKED", "k": "DELTA_BINARY_PACKED",
                                                                            "x": "BYTE_STREAM_SPLIT", "s": "DELTA_BYTE_ARRAY",
                                                                            "u": "DELTA_LENGTH_BYTE_ARRAY"})])
def test_file_to_columns(kw):
    P = pkg()
    raw, table = _file(**kw)
    f = P.file.File(raw)
    assert f.num_rows == table.num_rows and len(f.row_groups) == 3
    names = [".".join(c.path) for c in f.columns]
    assert names[:6] == ["id", "x", "k", "s", "u", "fx"] and names[6].startswith("tags.")
    base = 0
    for g, rg in enumerate(f.row_groups):
        part = table.slice(base, rg.num_rows)
        base += rg.num_rows
        for i, name in enumerate(names[:6]):
            col = f.read_column_chunk(ctx(), g, i)
            arr = part.column(name).combine_chunks()
            valid = np.asarray(arr.is_valid())
            if f.columns[i].max_definition_level:
                assert np.array_equal(col.definition_levels, valid.astype(np.uint8)), (g, name)
            else:
                assert col.definition_levels is None
            assert col.indexes is None
            present = arr.drop_null()
            if name in ("s", "u"):
                want = [v.as_py().encode() for v in present]
                off = col.offsets
                assert off.size == len(want) + 1, (g, name)
                data = col.values.tobytes() if col.values is not None else b""
                assert data == b"".join(want), (g, name)
                assert np.array_equal(np.diff(off.astype(np.int64)), [len(w) for w in want]), (g, name)
            elif name == "fx":
                assert col.values.tobytes() == b"".join(v.as_py() for v in present), (g, name)
            else:
                assert col.values.tobytes() == present.to_numpy(zero_copy_only=False).tobytes(), (g, name)
        # the repeated column: levels against the list structure
        col = f.read_column_chunk(ctx(), g, 6)
        lists = part.column("tags").to_pylist()
        want_def, want_rep = [], []
        for v in lists:
            if v is None:
                want_def.append(0); want_rep.append(0)
            elif not v:
                want_def.append(1); want_rep.append(0)
            else:
                for j in range(len(v)):
                    want_def.append(f.columns[6].max_definition_level); want_rep.append(1 if j else 0)
        assert np.array_equal(col.definition_levels, want_def) and np.array_equal(col.repetition_levels, want_rep), g
        flat = [x for v in lists if v for x in v]
        assert col.values.view(np.int32).tolist() == flat, g


def test_compressed_chunks_are_declined():
    P = pkg()
    pa = pytest.importorskip("pyarrow")
    import pyarrow.parquet as pq
    buf = io.BytesIO()
    pq.write_table(pa.table({"a": pa.array(np.arange(100, dtype=np.int64))}), buf, compression="snappy")
    f = P.file.File(buf.getvalue())
    with pytest.raises(P.PqdError) as e:
        f.read_column_chunk(ctx(), 0, 0)
    assert e.value.status == P.STATUS.NOT_SUPPORTED
