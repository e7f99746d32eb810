"""GPU: dictionary look-ups for BYTE_ARRAY, FIXED_LEN_BYTE_ARRAY (any width) and BOOLEAN
dictionaries -- fused with RLE_DICTIONARY pages and through pqd_dict_lookup_bytes_device --
against the reference's definition (dictionary_purego.go:40-52: the value the index names;
dictionary_boolean.go:93-98) written out in numpy, with the checker's index decode and
newIndexedPage fix-up in front.  Bit exact."""
import ctypes as C

import numpy as np
import pytest

from oracle import pqoracle as O
from tests.gpu_util import ctx, decode, pkg, spec

pytestmark = pytest.mark.gpu


def _ba_dict(rng, n, max_len):
    lens = rng.integers(0, max_len + 1, size=n)
    offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32)
    return rng.integers(0, 256, size=int(offs[-1]), dtype=np.uint8), offs


def _ba_lookup(data, offs, idx):
    o = offs.astype(np.int64)
    lens = (o[1:] - o[:-1])[idx]
    out_offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32)
    out = np.concatenate([data[o[i]: o[i + 1]] for i in idx] + [np.zeros(0, np.uint8)])
    return out, out_offs


def _runs(rng, n, dn):
    out = np.empty(n, np.int32)
    at = 0
    while at < n:
        k = int(rng.geometric(1 / 20))
        out[at: at + k] = rng.integers(0, dn)
        at += k
    return out


def test_byte_array_dictionary_pages():
    P = pkg()
    rng = np.random.default_rng(41)
    cases = []
    for dn, max_len in ((1, 5), (7, 0), (50, 12), (3000, 40), (20, 300), (70000, 9)):
        data, offs = _ba_dict(rng, dn, max_len)
        for n, nv in ((1, 1), (9, 9), (1000, 1000), (5000, 4097), (300, 1000), (20000, 20000)):
            idx = rng.integers(0, dn, size=n, dtype=np.int32) if n % 2 else _runs(rng, n, dn)
            cases.append((data, offs, idx, nv))
    specs = [spec(P.ENC.RLE_DICTIONARY, P.KIND.BYTE_ARRAY, O.rle_dict_encode_int32(idx), num_values=nv, dictionary=data if data.size else np.zeros(1, np.uint8),
                  dictionary_offsets=offs, dst_cap=max(64, nv * 310), offsets_cap=nv + 1, misalign=k % 16)
             for k, (data, offs, idx, nv) in enumerate(cases)]
    outs = decode(specs, grow=False)
    for (data, offs, idx, nv), out in zip(cases, outs):
        want_idx = O.indexed_page_fixup(O.rle_dict_decode_int32(O.rle_dict_encode_int32(idx)).values, nv)
        want, want_offs = _ba_lookup(data, offs, want_idx)
        assert out.status == 0 and out.out_values == nv and out.out_bytes == want.size
        assert np.array_equal(out.offsets[: nv + 1], want_offs)
        assert out.data.tobytes() == want.tobytes()


def test_byte_array_dictionary_errors_and_capacity():
    P = pkg()
    rng = np.random.default_rng(42)
    data, offs = _ba_dict(rng, 100, 20)
    idx = rng.integers(0, 100, size=3000, dtype=np.int32)
    enc = O.rle_dict_encode_int32(idx)
    want, want_offs = _ba_lookup(data, offs, idx)
    # room for the offsets, not for the bytes: the size to come back with
    out = decode([spec(P.ENC.RLE_DICTIONARY, P.KIND.BYTE_ARRAY, enc, num_values=3000, dictionary=data, dictionary_offsets=offs,
                       dst_cap=want.size - 1, offsets_cap=3001)], grow=False)[0]
    assert (out.status, out.arg) == (P.STATUS.DST_TOO_SMALL, want.size)
    out = decode([spec(P.ENC.RLE_DICTIONARY, P.KIND.BYTE_ARRAY, enc, dictionary=data, dictionary_offsets=offs,
                       dst_cap=want.size, offsets_cap=100)], grow=False)[0]
    assert (out.status, out.arg) == (P.STATUS.OFFSETS_TOO_SMALL, 3001)
    # grown the way a Go caller grows its slices
    out = decode([spec(P.ENC.RLE_DICTIONARY, P.KIND.BYTE_ARRAY, enc, dictionary=data, dictionary_offsets=offs, dst_cap=8, offsets_cap=100)])[0]
    assert out.status == 0 and out.data.tobytes() == want.tobytes() and np.array_equal(out.offsets[:3001], want_offs)
    # an index outside the dictionary: the reference panics "index out of bounds" (errors.go:75-83)
    bad = idx.copy()
    bad[[1234, 2500]] = [100, 7000]
    out = decode([spec(P.ENC.RLE_DICTIONARY, P.KIND.BYTE_ARRAY, O.rle_dict_encode_int32(bad), num_values=3000, dictionary=data,
                       dictionary_offsets=offs, dst_cap=10**6, offsets_cap=3001)], grow=False)[0]
    assert (out.status, out.arg) == (P.STATUS.INDEX_OUT_OF_BOUNDS, 1234)
    # a corrupt index stream keeps its own error
    out = decode([spec(P.ENC.RLE_DICTIONARY, P.KIND.BYTE_ARRAY, bytes([40, 3, 1]), num_values=8, dictionary=data, dictionary_offsets=offs,
                       dst_cap=1000, offsets_cap=9)], grow=False)[0]
    assert out.status == P.STATUS.INVALID_BIT_WIDTH


def test_byte_array_lookup_past_4_gib_is_an_error_not_a_wrapped_size():
    """Few indexes times one long dictionary entry come to more bytes than 32-bit offsets reach.  The
    size has to be summed in 64 bits: wrapped, it would come back as a capacity the caller can meet,
    and the copy would then write far past the buffer."""
    import torch
    P = pkg()
    L = P.lib()
    entry = 1 << 20
    data = np.full(entry + 3, 0x5A, np.uint8)
    offs = np.array([0, entry, entry + 3], np.uint32)
    stream = C.c_void_p(torch.cuda.current_stream(0).cuda_stream)
    d_t, o_t = torch.from_numpy(data).cuda(), torch.from_numpy(offs.view(np.int32)).cuda()
    st = torch.zeros(3, dtype=torch.int64, device="cuda")
    for n, want in ((5000, (P.STATUS.LENGTH_OUT_OF_BOUNDS, 5000 * entry)),      # 4.88 GiB
                    (4097, (P.STATUS.LENGTH_OUT_OF_BOUNDS, 4097 * entry)),      # wraps to 1 MiB
                    (3000, (P.STATUS.DST_TOO_SMALL, 3000 * entry))):            # 2.9 GiB: representable, just too large for dst
        idx = np.zeros(n, np.int32)
        idx[-1] = 1  # (the sum is not a multiple of the entry size)
        want = (want[0], want[1] - entry + 3)
        # fused with the page's index decode
        out = decode([spec(P.ENC.RLE_DICTIONARY, P.KIND.BYTE_ARRAY, O.rle_dict_encode_int32(idx), num_values=n, dictionary=data,
                           dictionary_offsets=offs, dst_cap=4096, offsets_cap=n + 1)], grow=False)[0]
        assert (out.status, out.arg) == want
        assert (out.raw == 0xA5).all(), "nothing is copied for a page that ends in an error"
        # stand-alone, on device pointers, with a canary behind a small destination
        i_t = torch.from_numpy(idx).cuda()
        dst = torch.full((4096 + 4096,), 0xA5, dtype=torch.uint8, device="cuda")
        out_offs = torch.zeros(n + 1, dtype=torch.int32, device="cuda")
        rc = L.pqd_dict_lookup_bytes_device(ctx().handle, P.KIND.BYTE_ARRAY, d_t.data_ptr(), o_t.data_ptr(), 2, 0, i_t.data_ptr(), n,
                                            dst.data_ptr(), 4096, out_offs.data_ptr(), n + 1, st.data_ptr(), stream)
        torch.cuda.synchronize()
        assert rc == 0 and tuple(st.tolist()[:2]) == want
        assert (dst.cpu().numpy() == 0xA5).all()


@pytest.mark.parametrize("width", [1, 2, 3, 5, 16, 17, 100])
def test_fixed_len_byte_array_dictionary_pages(width):
    P = pkg()
    rng = np.random.default_rng(43 + width)
    for dn, n, nv in ((1, 10, 10), (33, 5000, 5000), (1000, 7, 30), (4000, 30000, 29999)):
        d = rng.integers(0, 256, size=dn * width, dtype=np.uint8)
        idx = rng.integers(0, dn, size=n, dtype=np.int32)
        out = decode([spec(P.ENC.RLE_DICTIONARY, P.KIND.FIXED_LEN_BYTE_ARRAY, O.rle_dict_encode_int32(idx), bit_width=width, num_values=nv,
                           dictionary=d, dictionary_len=dn, dst_cap=nv * width, misalign=width % 16)], grow=False)[0]
        want_idx = O.indexed_page_fixup(O.rle_dict_decode_int32(O.rle_dict_encode_int32(idx)).values, nv)
        want = O.dict_lookup(d.view(np.dtype(f"V{width}")), want_idx)
        assert out.status == 0 and out.out_values == nv
        assert out.data.tobytes() == want.values.tobytes()
    out = decode([spec(P.ENC.RLE_DICTIONARY, P.KIND.FIXED_LEN_BYTE_ARRAY, O.rle_dict_encode_int32(idx), bit_width=width, num_values=100,
                       dictionary=d, dictionary_len=dn, dst_cap=100 * width - 1)], grow=False)[0]
    assert (out.status, out.arg) == (P.STATUS.DST_TOO_SMALL, 100 * width)


def test_boolean_dictionary_pages():
    P = pkg()
    rng = np.random.default_rng(44)
    for dn in (1, 2, 9, 70):
        bits = rng.integers(0, 2, size=dn, dtype=np.uint8)
        packed = np.packbits(bits, bitorder="little")
        idx = rng.integers(0, dn, size=5000, dtype=np.int32)
        out = decode([spec(P.ENC.RLE_DICTIONARY, P.KIND.BOOLEAN, O.rle_dict_encode_int32(idx), num_values=5000, dictionary=packed,
                           dictionary_len=dn, dst_cap=5000)], grow=False)[0]
        assert out.status == 0 and np.array_equal(out.data, bits[idx])


def test_lookup_entry_point_on_device_pointers():
    import torch
    P = pkg()
    L = P.lib()
    rng = np.random.default_rng(45)
    stream = C.c_void_p(torch.cuda.current_stream(0).cuda_stream)
    data, offs = _ba_dict(rng, 500, 30)
    idx = rng.integers(0, 500, size=40000, dtype=np.int32)
    want, want_offs = _ba_lookup(data, offs, idx)
    d_t, o_t, i_t = torch.from_numpy(data).cuda(), torch.from_numpy(offs.view(np.int32)).cuda(), torch.from_numpy(idx).cuda()
    dst = torch.full((want.size + 64,), 0xA5, dtype=torch.uint8, device="cuda")
    out_offs = torch.zeros(idx.size + 1, dtype=torch.int32, device="cuda")
    st = torch.zeros(3, dtype=torch.int64, device="cuda")
    rc = L.pqd_dict_lookup_bytes_device(ctx().handle, P.KIND.BYTE_ARRAY, d_t.data_ptr(), o_t.data_ptr(), 500, 0, i_t.data_ptr(), idx.size,
                                        dst.data_ptr(), want.size, out_offs.data_ptr(), idx.size + 1, st.data_ptr(), stream)
    torch.cuda.synchronize()
    assert rc == 0 and st.tolist() == [0, 0, want.size]
    got = dst.cpu().numpy()
    assert got[: want.size].tobytes() == want.tobytes() and (got[want.size:] == 0xA5).all()
    assert np.array_equal(out_offs.cpu().numpy().view(np.uint32), want_offs)
    # FIXED_LEN_BYTE_ARRAY, width 6, and an index outside the dictionary
    d6 = rng.integers(0, 256, size=300 * 6, dtype=np.uint8)
    idx6 = rng.integers(0, 300, size=1001, dtype=np.int32)
    d_t, i_t = torch.from_numpy(d6).cuda(), torch.from_numpy(idx6).cuda()
    dst = torch.zeros(1001 * 6, dtype=torch.uint8, device="cuda")
    rc = L.pqd_dict_lookup_bytes_device(ctx().handle, P.KIND.FIXED_LEN_BYTE_ARRAY, d_t.data_ptr(), None, 300, 6, i_t.data_ptr(), 1001,
                                        dst.data_ptr(), 1001 * 6, None, 0, st.data_ptr(), stream)
    torch.cuda.synchronize()
    assert rc == 0 and st.tolist() == [0, 0, 1001 * 6]
    assert np.array_equal(dst.cpu().numpy().reshape(-1, 6), d6.reshape(-1, 6)[idx6])
    idx6[77] = 300
    i_t = torch.from_numpy(idx6).cuda()
    rc = L.pqd_dict_lookup_bytes_device(ctx().handle, P.KIND.FIXED_LEN_BYTE_ARRAY, d_t.data_ptr(), None, 300, 6, i_t.data_ptr(), 1001,
                                        dst.data_ptr(), 1001 * 6, None, 0, st.data_ptr(), stream)
    torch.cuda.synchronize()
    assert rc == 0 and st.tolist()[:2] == [P.STATUS.INDEX_OUT_OF_BOUNDS, 77]
    assert L.pqd_dict_lookup_bytes_device(ctx().handle, P.KIND.INT64, d_t.data_ptr(), None, 300, 8, i_t.data_ptr(), 1, dst.data_ptr(), 8,
                                          None, 0, st.data_ptr(), stream) == P.STATUS.NOT_SUPPORTED


@pytest.mark.parametrize("dict_misalign", [0, 1, 7, 15])
def test_small_dictionaries_at_every_alignment(dict_misalign):
    """Look-ups of many values in small dictionaries through the stand-alone entry point: dictionaries and destinations
    at every alignment, one or two values of tens of kilobytes, values long enough that 64 of them pass the
    wavefront's image, empty values."""
    import torch
    P = pkg()
    L = P.lib()
    rng = np.random.default_rng(46 + dict_misalign)
    stream = C.c_void_p(torch.cuda.current_stream(0).cuda_stream)
    shapes = [(4096, 11), (4097, 11), (1500, 60), (1600, 61), (3, 0), (40, 900), (1, 48 << 10), (2, 24 << 10), (300, 150)]
    for dn, max_len in shapes:
        if dn <= 2:
            lens = np.full(dn, max_len)
            offs = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint32)
            data = rng.integers(0, 256, size=int(offs[-1]), dtype=np.uint8)
        else:
            data, offs = _ba_dict(rng, dn, max_len)
        n = 4096 if dn <= 2 else 30001
        idx = rng.integers(0, dn, size=n, dtype=np.int32)
        want, want_offs = _ba_lookup(data, offs, idx)
        raw = torch.zeros(data.size + 64, dtype=torch.uint8, device="cuda")
        raw[dict_misalign: dict_misalign + data.size] = torch.from_numpy(data).cuda()
        o_t, i_t = torch.from_numpy(offs.view(np.int32)).cuda(), torch.from_numpy(idx).cuda()
        a = (dict_misalign * 5) % 16
        dst = torch.full((want.size + 64 + a,), 0xA5, dtype=torch.uint8, device="cuda")
        out_offs = torch.zeros(n + 1, dtype=torch.int32, device="cuda")
        st = torch.zeros(3, dtype=torch.int64, device="cuda")
        rc = L.pqd_dict_lookup_bytes_device(ctx().handle, P.KIND.BYTE_ARRAY, raw.data_ptr() + dict_misalign, o_t.data_ptr(), dn, 0, i_t.data_ptr(), n,
                                            dst.data_ptr() + a, want.size, out_offs.data_ptr(), n + 1, st.data_ptr(), stream)
        torch.cuda.synchronize()
        assert rc == 0 and st.tolist() == [0, 0, want.size], (dn, max_len)
        got = dst.cpu().numpy()
        assert got[a: a + want.size].tobytes() == want.tobytes(), (dn, max_len)
        assert (got[:a] == 0xA5).all() and (got[a + want.size:] == 0xA5).all(), (dn, max_len)
        assert np.array_equal(out_offs.cpu().numpy().view(np.uint32), want_offs), (dn, max_len)
