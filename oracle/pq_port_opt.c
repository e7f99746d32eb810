/*
 * pq_port_opt.c -- a TUNED CPU port of the two decoders the benchmark's metric names, for the
 * cpu_baseline leg of bench.py ("kind": "port-opt").
 *
 * TEST INFRASTRUCTURE ONLY, like the rest of oracle/: nothing in the product library may call,
 * link or load it.  pq_oracle.c restates the reference value by value (a bit field at a time)
 * and runs an order of magnitude below the reference's amd64 kernels; this file decodes the
 * same bytes the way those kernels are organised, so that the CPU figure printed beside the
 * GPU's is an honest one:
 *
 *   DELTA_BINARY_PACKED int64 (encoding/delta/binary_packed.go:346-407): a miniblock of 32
 *     values per call of an unpacker specialised for its width -- word-at-a-time shifts with
 *     compile-time offsets, what bitpack.Unpack does (call site binary_packed.go:376-399) --
 *     then the scalar running sum of decodeBlockInt64 (binary_packed_amd64.s:902-920).
 *   RLE_DICTIONARY int64 (encoding/rle/dictionary.go:30-37, rle.go:352-419, and
 *     dictionaryLookup64, dictionary_amd64.s:646-711): bit-packed groups of eight indexes by a
 *     width-specialised unpacker, repeated runs as fills, then a bounds-checked gather.
 *
 * Only well-formed pages take the fast paths.  Anything else (a header that does not
 * validate, input that ends early, a width above the value width, an index outside the
 * dictionary) is handed to the plain restatement, so statuses and results are the checker's
 * in every case (tests/test_port_opt.py compares the two on whole and damaged pages).
 */
#include <stdlib.h>
#include <string.h>

#include "pq_oracle.h"

#define PAD 16 /* bytes an unpacker may read past the fields it extracts */

static inline uint64_t load64(const uint8_t* p) { uint64_t v; memcpy(&v, p, 8); return v; }

/* 32 fields of W bits from p, least significant bit first (the layout TestEncodeMiniBlockInt64
 * spells out, encoding/delta/binary_packed_test.go:293-322).  May read up to 4 W + 9 bytes. */
#define UNPACK32(W)                                                                   \
    static void unpack32_##W(const uint8_t* p, uint64_t* out) {                        \
        _Pragma("GCC unroll 32") for (unsigned i = 0; i < 32; i++) {                   \
            const unsigned bit = i * (W), s = bit & 7;                                 \
            uint64_t v = load64(p + (bit >> 3)) >> s;                                  \
            if ((W) + 7 > 64 && s) v |= (uint64_t)p[(bit >> 3) + 8] << (64 - s);       \
            out[i] = (W) >= 64 ? v : v & ((1ull << ((W) & 63)) - 1);                   \
        }                                                                              \
    }
UNPACK32(1) UNPACK32(2) UNPACK32(3) UNPACK32(4) UNPACK32(5) UNPACK32(6) UNPACK32(7) UNPACK32(8)
UNPACK32(9) UNPACK32(10) UNPACK32(11) UNPACK32(12) UNPACK32(13) UNPACK32(14) UNPACK32(15) UNPACK32(16)
UNPACK32(17) UNPACK32(18) UNPACK32(19) UNPACK32(20) UNPACK32(21) UNPACK32(22) UNPACK32(23) UNPACK32(24)
UNPACK32(25) UNPACK32(26) UNPACK32(27) UNPACK32(28) UNPACK32(29) UNPACK32(30) UNPACK32(31) UNPACK32(32)
UNPACK32(33) UNPACK32(34) UNPACK32(35) UNPACK32(36) UNPACK32(37) UNPACK32(38) UNPACK32(39) UNPACK32(40)
UNPACK32(41) UNPACK32(42) UNPACK32(43) UNPACK32(44) UNPACK32(45) UNPACK32(46) UNPACK32(47) UNPACK32(48)
UNPACK32(49) UNPACK32(50) UNPACK32(51) UNPACK32(52) UNPACK32(53) UNPACK32(54) UNPACK32(55) UNPACK32(56)
UNPACK32(57) UNPACK32(58) UNPACK32(59) UNPACK32(60) UNPACK32(61) UNPACK32(62) UNPACK32(63) UNPACK32(64)
static void unpack32_0(const uint8_t* p, uint64_t* out) { (void)p; memset(out, 0, 32 * 8); }
typedef void (*unpack32_fn)(const uint8_t*, uint64_t*);
static const unpack32_fn UNPACK32_TAB[65] = {
    unpack32_0,  unpack32_1,  unpack32_2,  unpack32_3,  unpack32_4,  unpack32_5,  unpack32_6,  unpack32_7,  unpack32_8,  unpack32_9,
    unpack32_10, unpack32_11, unpack32_12, unpack32_13, unpack32_14, unpack32_15, unpack32_16, unpack32_17, unpack32_18, unpack32_19,
    unpack32_20, unpack32_21, unpack32_22, unpack32_23, unpack32_24, unpack32_25, unpack32_26, unpack32_27, unpack32_28, unpack32_29,
    unpack32_30, unpack32_31, unpack32_32, unpack32_33, unpack32_34, unpack32_35, unpack32_36, unpack32_37, unpack32_38, unpack32_39,
    unpack32_40, unpack32_41, unpack32_42, unpack32_43, unpack32_44, unpack32_45, unpack32_46, unpack32_47, unpack32_48, unpack32_49,
    unpack32_50, unpack32_51, unpack32_52, unpack32_53, unpack32_54, unpack32_55, unpack32_56, unpack32_57, unpack32_58, unpack32_59,
    unpack32_60, unpack32_61, unpack32_62, unpack32_63, unpack32_64};

static int uvarint(const uint8_t* p, uint64_t n, uint64_t* out) { /* encoding/binary.Uvarint: bytes read, 0 short, -1 overflow */
    uint64_t x = 0;
    unsigned s = 0;
    for (uint64_t i = 0; i < n && i < 10; i++) {
        const uint8_t b = p[i];
        if (b < 0x80) {
            if (i == 9 && b > 1) return -1;
            *out = x | (uint64_t)b << s;
            return (int)i + 1;
        }
        x |= (uint64_t)(b & 0x7f) << s;
        s += 7;
    }
    return n >= 10 ? -1 : 0;
}

/* 1: the page was decoded; 0: not a page the fast path takes (nothing of dst is to be trusted). */
static int delta_int64_fast(int64_t* dst, uint64_t cap_bytes, const uint8_t* src, uint64_t n, pqo_result* r) {
    uint64_t i = 0, u[4];
    for (int f = 0; f < 4; f++) {
        const int k = uvarint(src + i, n - i, &u[f]);
        if (k <= 0) return 0;
        i += (uint64_t)k;
    }
    const uint64_t block_size = u[0], num_mb = u[1], total = u[2];
    const int64_t first = (int64_t)(u[3] >> 1) ^ -(int64_t)(u[3] & 1);
    if (num_mb == 0 || num_mb > 512 || block_size == 0 || block_size % 128 || block_size > 65536 || (block_size / num_mb) % 32 ||
        block_size % num_mb || total == 0 || total > 0x7fffffffull || total * 8 > cap_bytes)
        return 0;
    const uint64_t groups = block_size / num_mb / 32; /* unpacker calls per miniblock */
    dst[0] = first;
    uint64_t w = 1, remaining = total - 1, last = (uint64_t)first;
    uint64_t tmp[32];
    uint8_t tail[8 * 64 + 2 * PAD];
    while (remaining > 0) {
        uint64_t zz;
        const int k = uvarint(src + i, n - i, &zz);
        if (k <= 0) return 0;
        i += (uint64_t)k;
        const uint64_t min_delta = (zz >> 1) ^ (0 - (zz & 1));
        if (n - i < num_mb) return 0;
        const uint8_t* widths = src + i;
        i += num_mb;
        for (uint64_t m = 0; m < num_mb && remaining > 0; m++) {
            const unsigned bw = widths[m];
            if (bw > 64) return 0;
            for (uint64_t g = 0; g < groups; g++) {
                const uint64_t bytes = 4ull * bw;
                if (n - i < bytes) return 0; /* a miniblock cut short: zero padded by the reference, left to the restatement */
                const uint8_t* p = src + i;
                if (n - i < bytes + PAD) { /* the unpacker's over-read would leave the input */
                    memset(tail, 0, sizeof tail);
                    memcpy(tail, p, bytes);
                    p = tail;
                }
                i += bytes; /* the whole miniblock is consumed even when the page needs only part of it */
                if (remaining == 0) continue;
                UNPACK32_TAB[bw](p, tmp);
                const uint64_t nv = remaining < 32 ? remaining : 32;
                int64_t* o = dst + w;
                if (nv == 32) {
                    for (unsigned v = 0; v < 32; v++) { last += tmp[v] + min_delta; o[v] = (int64_t)last; }
                } else {
                    for (uint64_t v = 0; v < nv; v++) { last += tmp[v] + min_delta; o[v] = (int64_t)last; }
                }
                w += nv;
                remaining -= nv;
            }
        }
    }
    r->status = PQD_OK;
    r->arg = 0;
    r->n_values = total;
    r->n_bytes = total * 8;
    r->consumed = i;
    return 1;
}

pqo_result pqopt_delta_decode_int64(int64_t* dst, uint64_t cap_bytes, const uint8_t* src, uint64_t n) {
    pqo_result r;
    if (delta_int64_fast(dst, cap_bytes, src, n, &r)) return r;
    return pqo_delta_decode_int64(dst, cap_bytes, src, n);
}

/* Eight fields of W bits (W <= 32) from p: one group of a bit-packed run (rle.go:376-396). */
#define UNPACK8(W)                                                                    \
    static void unpack8_##W(const uint8_t* p, uint32_t* out) {                         \
        _Pragma("GCC unroll 8") for (unsigned i = 0; i < 8; i++) {                     \
            const unsigned bit = i * (W);                                              \
            const uint64_t v = load64(p + (bit >> 3)) >> (bit & 7);                    \
            out[i] = (W) >= 32 ? (uint32_t)v : (uint32_t)v & ((1u << ((W) & 31)) - 1); \
        }                                                                              \
    }
UNPACK8(1) UNPACK8(2) UNPACK8(3) UNPACK8(4) UNPACK8(5) UNPACK8(6) UNPACK8(7) UNPACK8(8)
UNPACK8(9) UNPACK8(10) UNPACK8(11) UNPACK8(12) UNPACK8(13) UNPACK8(14) UNPACK8(15) UNPACK8(16)
UNPACK8(17) UNPACK8(18) UNPACK8(19) UNPACK8(20) UNPACK8(21) UNPACK8(22) UNPACK8(23) UNPACK8(24)
UNPACK8(25) UNPACK8(26) UNPACK8(27) UNPACK8(28) UNPACK8(29) UNPACK8(30) UNPACK8(31) UNPACK8(32)
static void unpack8_0(const uint8_t* p, uint32_t* out) { (void)p; memset(out, 0, 32); }
typedef void (*unpack8_fn)(const uint8_t*, uint32_t*);
static const unpack8_fn UNPACK8_TAB[33] = {
    unpack8_0,  unpack8_1,  unpack8_2,  unpack8_3,  unpack8_4,  unpack8_5,  unpack8_6,  unpack8_7,  unpack8_8,  unpack8_9,  unpack8_10,
    unpack8_11, unpack8_12, unpack8_13, unpack8_14, unpack8_15, unpack8_16, unpack8_17, unpack8_18, unpack8_19, unpack8_20, unpack8_21,
    unpack8_22, unpack8_23, unpack8_24, unpack8_25, unpack8_26, unpack8_27, unpack8_28, unpack8_29, unpack8_30, unpack8_31, unpack8_32};

#define IDX_CHUNK 1024 /* indexes unpacked before they are looked up: they stay in the first-level cache */

static inline int gather(int64_t* dst, const uint32_t* idx, uint64_t k, const int64_t* dict, uint64_t dict_len) {
    uint32_t over = 0;
    for (uint64_t j = 0; j < k; j++) over |= (uint32_t)(idx[j] >= dict_len);
    if (over) return 0;
    for (uint64_t j = 0; j < k; j++) dst[j] = dict[idx[j]];
    return 1;
}

/* The values of an RLE_DICTIONARY page: indexes decoded, cut or zero-extended to num_values
 * (newIndexedPage, dictionary.go:136-162) and looked up.  1: done; 0: not a page for the fast path. */
static int rle_dict_int64_fast(int64_t* dst, uint64_t num_values, const uint8_t* src, uint64_t n, const int64_t* dict, uint64_t dict_len,
                               pqo_result* r) {
    if (n == 0) return 0;
    const unsigned bw = src[0];
    if (bw > 32) return 0;
    uint64_t i = 1, w = 0;
    uint32_t idx[IDX_CHUNK + 8];
    while (i < n && w < num_values) {
        uint64_t h;
        const int k = uvarint(src + i, n - i, &h);
        if (k <= 0) return 0;
        i += (uint64_t)k;
        uint64_t count = h >> 1;
        if (count == 0) continue;
        if (h & 1) {
            if (count > 0x7fffffffull / 8) return 0;
            uint64_t left = count; /* groups of eight */
            if (n - i < left * bw) return 0;
            while (left > 0 && w < num_values) {
                uint64_t g = left < IDX_CHUNK / 8 ? left : IDX_CHUNK / 8;
                if (n - i < g * bw + PAD) { /* the last groups of the input: one at a time out of a padded copy */
                    uint8_t tail[32 + PAD];
                    memset(tail, 0, sizeof tail);
                    memcpy(tail, src + i, bw);
                    UNPACK8_TAB[bw](tail, idx);
                    g = 1;
                } else {
                    for (uint64_t q = 0; q < g; q++) UNPACK8_TAB[bw](src + i + q * bw, idx + q * 8);
                }
                uint64_t k2 = g * 8;
                if (k2 > num_values - w) k2 = num_values - w;
                if (!gather(dst + w, idx, k2, dict, dict_len)) return 0;
                w += k2;
                i += g * bw;
                left -= g;
            }
            i += left * bw;
        } else {
            if (count > 0x7fffffffull) return 0;
            const unsigned vb = (bw + 7) / 8;
            if (n - i < vb) return 0;
            uint32_t v = 0;
            memcpy(&v, src + i, vb);
            i += vb;
            if (v >= dict_len) return 0;
            const int64_t x = dict[v];
            uint64_t k2 = count < num_values - w ? count : num_values - w;
            for (uint64_t j = 0; j < k2; j++) dst[w + j] = x;
            w += k2;
        }
    }
    if (i < n) return 0; /* what follows the values asked for may still be malformed: the restatement decides */
    if (w < num_values) { /* zero-extended indexes */
        if (dict_len == 0) return 0;
        for (; w < num_values; w++) dst[w] = dict[0];
    }
    r->status = PQD_OK;
    r->arg = 0;
    r->n_values = num_values;
    r->n_bytes = num_values * 8;
    r->consumed = n;
    return 1;
}

pqo_result pqopt_rle_dict_lookup_int64(int64_t* dst, uint64_t num_values, const uint8_t* src, uint64_t n, const int64_t* dict,
                                       uint64_t dict_len) {
    pqo_result r;
    if (rle_dict_int64_fast(dst, num_values, src, n, dict, dict_len, &r)) return r;
    /* the plain way: DecodeInt32 into a buffer grown like a Go slice, newIndexedPage, Lookup */
    uint64_t room = 64 * n + 64;
    if (room < num_values * 4) room = num_values * 4;
    int32_t* buf = NULL;
    for (int turn = 0; turn < 48; turn++) {
        free(buf);
        buf = (int32_t*)calloc(room, 1);
        if (buf == NULL) { r.status = PQD_ERR_DEVICE; r.arg = 0; r.n_values = r.n_bytes = r.consumed = 0; return r; }
        r = pqo_rle_dict_decode_int32(buf, room, src, n);
        if (r.status != PQD_ERR_DST_TOO_SMALL || (uint64_t)r.arg <= room) break;
        room = (uint64_t)r.arg > 2 * room ? (uint64_t)r.arg : 2 * room;
    }
    if (r.status == PQD_OK) {
        pqo_indexed_page_fixup(buf, r.n_values, num_values);
        r = pqo_dict_lookup((uint8_t*)dst, 8, (const uint8_t*)dict, dict_len, 8, buf, num_values);
    }
    free(buf);
    return r;
}
