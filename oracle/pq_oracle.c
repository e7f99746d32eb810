/*
 * pq_oracle.c -- scalar CPU restatement of parquet-go's page-value decoders.
 *
 * TEST INFRASTRUCTURE ONLY: the checker the GPU path is compared against.
 * See pq_oracle.h.  Every function cites the reference lines it restates
 * (paths relative to the parquet-go source tree).
 *
 * Deliberate, documented differences from the reference (all on malformed
 * input where the reference panics or reads out of bounds):
 *  - rle.decodeInt32 slices a bit-packed run without a length check
 *    (encoding/rle/rle.go:385); here a short run is UNEXPECTED_EOF.
 *  - plain.DecodeByteArray compares a length against the whole input, not the
 *    remainder (encoding/plain/plain.go:84); here an overrunning value is
 *    UNEXPECTED_EOF.
 *  - DELTA miniblock widths above the value width (of miniblocks that hold
 *    values) are handed by the reference
 *    to the external bitpack.Unpack, whose behaviour there is not pinned by
 *    anything in the reference; here they are INVALID_BIT_WIDTH.
 *  - PLAIN FIXED_LEN_BYTE_ARRAY with size 0 divides by zero in the reference
 *    (encoding/plain/plain.go:100); here INVALID_ARGUMENT.
 */
#include "pq_oracle.h"

#include <stdlib.h>
#include <string.h>

#define MAX_INT32 2147483647LL
#define MIN_INT32 (-2147483647LL - 1)
#define MAX_FLBA_SIZE 32767 /* encoding.MaxFixedLenByteArraySize, encoding/encoding.go:12-14 */

static pqo_result ok_result(void) {
    pqo_result r;
    memset(&r, 0, sizeof r);
    return r;
}

static pqo_result fail(pqo_result r, int32_t status, int64_t arg) {
    r.status = status;
    r.arg = arg;
    return r;
}

/* Go's encoding/binary.Uvarint: returns bytes read, 0 when the buffer ends
 * inside the varint, -(i+1) on overflow (more than 10 bytes or a 10th byte
 * above 1). */
static int go_uvarint(const uint8_t* p, uint64_t n, uint64_t* out) {
    uint64_t x = 0;
    unsigned s = 0;
    for (uint64_t i = 0; i < n; i++) {
        uint8_t b = p[i];
        if (i == 10) return -(int)(i + 1);
        if (b < 0x80) {
            if (i == 9 && b > 1) return -(int)(i + 1);
            *out = x | ((uint64_t)b << s);
            return (int)(i + 1);
        }
        x |= (uint64_t)(b & 0x7f) << s;
        s += 7;
    }
    *out = 0;
    return 0;
}

/* Go's encoding/binary.Varint (zigzag). */
static int go_varint(const uint8_t* p, uint64_t n, int64_t* out) {
    uint64_t ux;
    int k = go_uvarint(p, n, &ux);
    int64_t x = (int64_t)(ux >> 1);
    if (ux & 1) x = ~x;
    *out = x;
    return k;
}

static uint64_t put_uvarint(uint8_t* p, uint64_t v) {
    uint64_t i = 0;
    while (v >= 0x80) {
        p[i++] = (uint8_t)v | 0x80;
        v >>= 7;
    }
    p[i++] = (uint8_t)v;
    return i;
}

static uint64_t put_varint(uint8_t* p, int64_t v) {
    uint64_t ux = (uint64_t)v << 1;
    if (v < 0) ux = ~ux;
    return put_uvarint(p, ux);
}

/* LSB-first field extraction: value i of width w occupies bits [i*w,(i+1)*w)
 * of the little-endian byte stream (bitpack.Unpack; layout spelled out by
 * encoding/delta/binary_packed_test.go:113-143,293-322).  Bytes at or past
 * `avail` read as zero. */
static uint64_t unpack_field_bits(const uint8_t* data, uint64_t avail, uint64_t index, uint32_t w) {
    uint64_t v = 0;
    uint64_t bit = index * (uint64_t)w;
    for (uint32_t b = 0; b < w; b++, bit++) {
        uint64_t byte = bit >> 3;
        if (byte < avail && ((data[byte] >> (bit & 7)) & 1)) v |= 1ull << b;
    }
    return v;
}

static uint64_t unpack_field(const uint8_t* data, uint64_t avail, uint64_t index, uint32_t w) {
    uint64_t bit = index * (uint64_t)w;
    if (w == 0) return 0;
    if ((bit >> 3) + 16 <= avail) { /* word path: two little-endian loads */
        uint64_t lo, hi;
        memcpy(&lo, data + (bit >> 3), 8);
        memcpy(&hi, data + (bit >> 3) + 8, 8);
        unsigned s = (unsigned)(bit & 7);
        uint64_t v = lo >> s;
        if (s) v |= hi << (64 - s);
        return w == 64 ? v : (v & ((1ull << w) - 1));
    }
    return unpack_field_bits(data, avail, index, w);
}

/* 0 when the word path and the bit-by-bit definition agree on every field of
 * a pseudo-random buffer for every width. */
int pqo_selftest_unpack(void) {
    uint8_t buf[1024];
    uint64_t x = 0x9E3779B97F4A7C15ull;
    for (unsigned i = 0; i < sizeof buf; i++) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; buf[i] = (uint8_t)x; }
    for (uint32_t w = 0; w <= 64; w++)
        for (uint64_t k = 0; w == 0 ? k < 1 : (k + 1) * w <= 8 * sizeof buf; k++)
            if (unpack_field(buf, sizeof buf, k, w) != unpack_field_bits(buf, sizeof buf, k, w)) return (int)w + 1;
    return 0;
}

static void pack_field(uint8_t* data, uint64_t index, uint32_t w, uint64_t v) {
    uint64_t bit = index * (uint64_t)w;
    for (uint32_t b = 0; b < w; b++, bit++) {
        if ((v >> b) & 1) data[bit >> 3] |= (uint8_t)(1u << (bit & 7));
    }
}

/* ===================================================================== */
/* RLE / bit-packed hybrid                                                */
/* ===================================================================== */

/* encoding/rle/rle.go:295-350 (decodeBytes) + :550-576 (decodeBytesBitpackDefault). */
pqo_result pqo_rle_decode_levels(uint8_t* dst, uint64_t cap, const uint8_t* src, uint64_t n, uint32_t bit_width) {
    pqo_result r = ok_result();
    if (bit_width > 8) return fail(r, PQD_ERR_INVALID_BIT_WIDTH, bit_width);
    uint64_t out = 0;
    for (uint64_t i = 0; i < n;) {
        uint64_t u;
        int k = go_uvarint(src + i, n - i, &u);
        r.n_values = r.n_bytes = out;
        if (k == 0) return fail(r, PQD_ERR_UNEXPECTED_EOF, PQD_EOF_RUN_HEADER);
        if (k < 0) return fail(r, PQD_ERR_VARINT_OVERFLOW, (int64_t)i);
        i += (uint64_t)k;
        uint64_t count = u >> 1;
        int bitpacked = (int)(u & 1);
        if (count == 0) continue;
        if (count > (uint64_t)MAX_INT32) return fail(r, PQD_ERR_RUN_TOO_LARGE, (int64_t)count);
        if (bitpacked) {
            count *= 8;
            uint64_t nbytes = (count * bit_width + 7) / 8;
            if (nbytes > n - i) return fail(r, PQD_ERR_UNEXPECTED_EOF, PQD_EOF_BITPACKED_RUN);
            if (out + count > cap) return fail(r, PQD_ERR_DST_TOO_SMALL, (int64_t)(out + count));
            uint64_t mask = (1ull << bit_width) - 1;
            for (uint64_t g = 0; g < count / 8; g++) {
                uint8_t bits[8] = {0};
                memcpy(bits, src + i + g * bit_width, bit_width);
                uint64_t word = 0;
                for (int b = 7; b >= 0; b--) word = (word << 8) | bits[b];
                for (unsigned v = 0; v < 8; v++) dst[out + g * 8 + v] = (uint8_t)((word >> (v * bit_width)) & mask);
            }
            out += count;
            i += nbytes;
        } else {
            if (bit_width != 0 && i + 1 > n) return fail(r, PQD_ERR_UNEXPECTED_EOF, PQD_EOF_RLE_VALUE);
            uint8_t word = 0;
            if (bit_width != 0) word = src[i++];
            if (out + count > cap) return fail(r, PQD_ERR_DST_TOO_SMALL, (int64_t)(out + count));
            memset(dst + out, word, count);
            out += count;
        }
    }
    r.n_values = r.n_bytes = out;
    return r;
}

/* encoding/rle/rle.go:68-82 (DecodeBoolean) + :251-293 (decodeBits). */
pqo_result pqo_rle_decode_boolean(uint8_t* dst, uint64_t cap, const uint8_t* src, uint64_t n) {
    pqo_result r = ok_result();
    if (n == 4) return r;
    if (n < 4) return fail(r, PQD_ERR_UNEXPECTED_EOF, PQD_EOF_LENGTH_PREFIX);
    uint64_t len = (uint64_t)src[0] | ((uint64_t)src[1] << 8) | ((uint64_t)src[2] << 16) | ((uint64_t)src[3] << 24);
    src += 4;
    n -= 4;
    if (len > n) return fail(r, PQD_ERR_UNEXPECTED_EOF, PQD_EOF_LENGTH_PREFIX);
    n = len;
    uint64_t out = 0;
    for (uint64_t i = 0; i < n;) {
        uint64_t u;
        int k = go_uvarint(src + i, n - i, &u);
        r.n_bytes = out;
        r.n_values = out * 8;
        if (k == 0) return fail(r, PQD_ERR_UNEXPECTED_EOF, PQD_EOF_RUN_HEADER);
        if (k < 0) return fail(r, PQD_ERR_VARINT_OVERFLOW, (int64_t)i);
        i += (uint64_t)k;
        uint64_t count = u >> 1;
        int bitpacked = (int)(u & 1);
        if (count == 0) continue;
        if (count > (uint64_t)MAX_INT32) return fail(r, PQD_ERR_RUN_TOO_LARGE, (int64_t)count);
        if (bitpacked) {
            /* the header counts BYTES here (rle.go:269-278) */
            if (count > n - i) return fail(r, PQD_ERR_UNEXPECTED_EOF, PQD_EOF_BITPACKED_RUN);
            if (out + count > cap) return fail(r, PQD_ERR_DST_TOO_SMALL, (int64_t)(out + count));
            memcpy(dst + out, src + i, count);
            out += count;
            i += count;
        } else {
            uint8_t word = 0;
            if (i < n) word = src[i++];
            uint64_t length = (count + 7) / 8;
            if (out + length > cap) return fail(r, PQD_ERR_DST_TOO_SMALL, (int64_t)(out + length));
            memset(dst + out, word, length);
            out += length;
        }
    }
    r.n_bytes = out;
    r.n_values = out * 8;
    return r;
}

/* encoding/rle/rle.go:352-419 (decodeInt32). */
pqo_result pqo_rle_decode_int32(int32_t* dst, uint64_t cap_bytes, const uint8_t* src, uint64_t n, uint32_t bit_width) {
    pqo_result r = ok_result();
    if (bit_width > 32) return fail(r, PQD_ERR_INVALID_BIT_WIDTH, bit_width);
    uint64_t cap = cap_bytes / 4;
    uint64_t out = 0;
    for (uint64_t i = 0; i < n;) {
        uint64_t u;
        int k = go_uvarint(src + i, n - i, &u);
        r.n_values = out;
        r.n_bytes = out * 4;
        if (k == 0) return fail(r, PQD_ERR_UNEXPECTED_EOF, PQD_EOF_RUN_HEADER);
        if (k < 0) return fail(r, PQD_ERR_VARINT_OVERFLOW, (int64_t)i);
        i += (uint64_t)k;
        uint64_t count = u >> 1;
        int bitpacked = (int)(u & 1);
        if (count == 0) continue;
        if (count > (uint64_t)MAX_INT32) return fail(r, PQD_ERR_RUN_TOO_LARGE, (int64_t)count);
        if (bitpacked) {
            uint64_t length = count * bit_width;
            if (length > n - i) return fail(r, PQD_ERR_UNEXPECTED_EOF, PQD_EOF_BITPACKED_RUN);
            uint64_t nv = 8 * count;
            if (out + nv > cap) return fail(r, PQD_ERR_DST_TOO_SMALL, (int64_t)((out + nv) * 4));
            for (uint64_t v = 0; v < nv; v++) dst[out + v] = (int32_t)(uint32_t)unpack_field(src + i, length, v, bit_width);
            out += nv;
            i += length;
        } else {
            uint64_t vb = (bit_width + 7) / 8;
            if (vb > n - i) return fail(r, PQD_ERR_UNEXPECTED_EOF, PQD_EOF_RLE_VALUE);
            uint32_t value = 0;
            for (uint64_t b = 0; b < vb; b++) value |= (uint32_t)src[i + b] << (8 * b);
            if (out + count > cap) return fail(r, PQD_ERR_DST_TOO_SMALL, (int64_t)((out + count) * 4));
            for (uint64_t v = 0; v < count; v++) dst[out + v] = (int32_t)value;
            out += count;
            i += vb;
        }
    }
    r.n_values = out;
    r.n_bytes = out * 4;
    return r;
}

/* encoding/rle/dictionary.go:30-37. */
pqo_result pqo_rle_dict_decode_int32(int32_t* dst, uint64_t cap_bytes, const uint8_t* src, uint64_t n) {
    if (n == 0) return ok_result();
    return pqo_rle_decode_int32(dst, cap_bytes, src + 1, n - 1, src[0]);
}

/* Go shifts of a byte by a uint amount: amounts >= 8 give 0. */
static uint8_t shl8(uint8_t v, uint64_t s) { return s >= 8 ? 0 : (uint8_t)(v << s); }
static uint8_t shr8(uint8_t v, uint64_t s) { return s >= 8 ? 0 : (uint8_t)(v >> s); }

/* encoding/bitpacked/bitpacked.go:75-119 (decodeLevels), MSB-first. */
pqo_result pqo_bitpacked_decode_levels(uint8_t* dst, uint64_t cap, const uint8_t* src, uint64_t n, uint32_t bit_width) {
    pqo_result r = ok_result();
    if (bit_width == 0 || n == 0) {
        if (cap < 1) return fail(r, PQD_ERR_DST_TOO_SMALL, 1);
        dst[0] = 0;
        r.n_values = r.n_bytes = 1;
        return r;
    }
    uint64_t num_bits = 8 * n;
    uint64_t num_values = num_bits / bit_width;
    if (num_bits % bit_width) num_values++;
    if (num_values > cap) return fail(r, PQD_ERR_DST_TOO_SMALL, (int64_t)num_values);
    uint8_t mask = (uint8_t)(shl8(1, bit_width) - 1);
    uint64_t bit_offset = 0;
    for (uint64_t k = 0; k < num_values; k++) {
        uint64_t i = bit_offset / 8, j = bit_offset % 8, available = 8 - j;
        uint8_t v;
        if (bit_width <= available) {
            v = shr8(src[i], available - bit_width) & mask;
        } else {
            uint64_t remaining = bit_width - available;
            uint8_t top = src[i] & (uint8_t)(shl8(1, available) - 1);
            uint8_t bottom = 0;
            /* 8-remaining is an unsigned subtraction in the reference: a
             * remaining above 8 wraps to a huge shift, i.e. 0 */
            if (i + 1 < n) bottom = remaining > 8 ? 0 : shr8(src[i + 1], 8 - remaining);
            v = shl8(top, remaining) | bottom;
        }
        dst[k] = v;
        bit_offset += bit_width;
    }
    r.n_values = r.n_bytes = num_values;
    return r;
}

/* ===================================================================== */
/* DELTA_BINARY_PACKED                                                    */
/* ===================================================================== */

/* encoding/delta/binary_packed.go:279-344 (int32), :346-407 (int64),
 * :409-452 (header), :454-466 (block header),
 * encoding/delta/binary_packed_purego.go:51-67 (prefix sum). */
static pqo_result delta_decode(void* dstv, uint64_t cap_bytes, const uint8_t* src, uint64_t n, int is64) {
    pqo_result r = ok_result();
    const uint32_t elem = is64 ? 8 : 4;
    const uint32_t max_bw = is64 ? 64 : 32;
    uint64_t i = 0, u[3];
    for (int f = 0; f < 3; f++) {
        int k = go_uvarint(src + i, n - i, &u[f]);
        if (k == 0) return fail(r, PQD_ERR_UNEXPECTED_EOF, PQD_EOF_DELTA_HEADER);
        if (k < 0) return fail(r, PQD_ERR_VARINT_OVERFLOW, (int64_t)i);
        i += (uint64_t)k;
    }
    int64_t first;
    {
        int k = go_varint(src + i, n - i, &first);
        if (k == 0) return fail(r, PQD_ERR_UNEXPECTED_EOF, PQD_EOF_DELTA_HEADER);
        if (k < 0) return fail(r, PQD_ERR_VARINT_OVERFLOW, (int64_t)i);
        i += (uint64_t)k;
    }
    r.consumed = i;
    int64_t block_size = (int64_t)u[0], num_mb = (int64_t)u[1], total = (int64_t)u[2];
    if (num_mb == 0) return fail(r, PQD_ERR_BAD_DELTA_HEADER, PQD_DELTA_BAD_NUM_MINIBLOCKS);
    if (block_size <= 0 || (block_size % 128) != 0) return fail(r, PQD_ERR_BAD_DELTA_HEADER, PQD_DELTA_BAD_BLOCK_SIZE);
    if (block_size > 65536) return fail(r, PQD_ERR_BAD_DELTA_HEADER, PQD_DELTA_BLOCK_TOO_LARGE);
    if (num_mb <= 0 || ((block_size / num_mb) % 32) != 0) return fail(r, PQD_ERR_BAD_DELTA_HEADER, PQD_DELTA_BAD_MINIBLOCK_SIZE);
    if (total < 0) return fail(r, PQD_ERR_BAD_DELTA_HEADER, PQD_DELTA_NEGATIVE_TOTAL);
    if (total > MAX_INT32) return fail(r, PQD_ERR_BAD_DELTA_HEADER, PQD_DELTA_TOO_MANY_VALUES);
    if (total == 0) return r;
    if (!is64 && (first < MIN_INT32 || first > MAX_INT32)) return fail(r, PQD_ERR_FIRST_VALUE_RANGE, first);
    if ((uint64_t)total * elem > cap_bytes) return fail(r, PQD_ERR_DST_TOO_SMALL, (int64_t)((uint64_t)total * elem));

    int64_t* o64 = (int64_t*)dstv;
    int32_t* o32 = (int32_t*)dstv;
    memset(dstv, 0, (uint64_t)total * elem); /* delta.resize zero-fills, delta.go:61-72 */
    if (is64) o64[0] = first; else o32[0] = (int32_t)first;
    uint64_t w = 1, remaining = (uint64_t)total - 1;
    uint64_t last = (uint64_t)first;
    const uint64_t vpm = (uint64_t)(block_size / num_mb);
    r.n_values = (uint64_t)total;
    r.n_bytes = (uint64_t)total * elem;

    while (remaining > 0 && i < n) {
        int64_t min_delta;
        int k = go_varint(src + i, n - i, &min_delta);
        r.consumed = i;
        if (k == 0) return fail(r, PQD_ERR_UNEXPECTED_EOF, PQD_EOF_DELTA_MIN_DELTA);
        if (k < 0) return fail(r, PQD_ERR_VARINT_OVERFLOW, (int64_t)i);
        i += (uint64_t)k;
        uint64_t nbw = (uint64_t)num_mb;
        if (n - i < nbw) nbw = n - i;
        const uint8_t* widths = src + i;
        i += nbw;
        uint64_t block_off = w;
        for (uint64_t m = 0; m < nbw; m++) {
            uint32_t bw = widths[m];
            uint64_t nv = vpm < remaining ? vpm : remaining;
            if (bw != 0) {
                /* a miniblock of no values (more miniblocks than values per block) unpacks nothing: its width goes unexamined */
                if (bw > max_bw && nv > 0) { r.consumed = i; return fail(r, PQD_ERR_INVALID_BIT_WIDTH, bw); }
                uint64_t mbsize = vpm * bw / 8;
                uint64_t avail = mbsize <= n - i ? mbsize : n - i;
                for (uint64_t v = 0; v < nv; v++) {
                    uint64_t f = unpack_field(src + i, avail, v, bw); /* short data is zero padded */
                    if (is64) o64[w + v] = (int64_t)f; else o32[w + v] = (int32_t)(uint32_t)f;
                }
                i += avail;
            }
            w += nv;
            remaining -= nv;
            if (remaining == 0) break;
        }
        /* x[i] += minDelta; x[i] += x[i-1], wrapping */
        if (is64) {
            for (uint64_t v = block_off; v < w; v++) {
                last = (uint64_t)o64[v] + (uint64_t)min_delta + last;
                o64[v] = (int64_t)last;
            }
        } else {
            uint32_t l32 = (uint32_t)last;
            for (uint64_t v = block_off; v < w; v++) {
                l32 = (uint32_t)o32[v] + (uint32_t)(int32_t)min_delta + l32;
                o32[v] = (int32_t)l32;
            }
            last = l32;
        }
    }
    r.consumed = i;
    if (remaining > 0) return fail(r, PQD_ERR_MISSING_VALUES, (int64_t)remaining);
    return r;
}

pqo_result pqo_delta_decode_int32(int32_t* dst, uint64_t cap_bytes, const uint8_t* src, uint64_t n) {
    return delta_decode(dst, cap_bytes, src, n, 0);
}

pqo_result pqo_delta_decode_int64(int64_t* dst, uint64_t cap_bytes, const uint8_t* src, uint64_t n) {
    return delta_decode(dst, cap_bytes, src, n, 1);
}

/* int32Buffer.decode (encoding/delta/delta.go:22-26): decode a length stream
 * into a fresh buffer; *out_n is the slice length the reference keeps, which
 * is the header's total even when values are missing. */
static pqo_result decode_lengths(const uint8_t* src, uint64_t n, int32_t** out, uint64_t* out_n) {
    pqo_result probe = delta_decode(NULL, 0, src, n, 0);
    *out = NULL;
    *out_n = 0;
    if (probe.status != PQD_ERR_DST_TOO_SMALL) return probe; /* header error or zero values */
    uint64_t bytes = (uint64_t)probe.arg;
    int32_t* buf = (int32_t*)malloc(bytes ? bytes : 4);
    pqo_result r = delta_decode(buf, bytes, src, n, 0);
    *out = buf;
    *out_n = r.n_values;
    return r;
}

/* encoding/delta/length_byte_array.go:37-63 + length_byte_array_purego.go:11-24. */
pqo_result pqo_dlba_decode(uint8_t* dst, uint64_t cap, uint32_t* offsets, uint64_t offsets_cap,
                           const uint8_t* src, uint64_t n) {
    int32_t* lengths;
    uint64_t nl;
    pqo_result r = decode_lengths(src, n, &lengths, &nl);
    if (r.status != PQD_OK) { free(lengths); r.n_values = r.n_bytes = 0; return r; }
    const uint8_t* rest = src + r.consumed;
    uint64_t rest_n = n - r.consumed;
    r.n_values = r.n_bytes = 0;
    if (nl + 1 > offsets_cap) { free(lengths); return fail(r, PQD_ERR_OFFSETS_TOO_SMALL, (int64_t)(nl + 1)); }
    uint32_t last = 0;
    for (uint64_t k = 0; k < nl; k++) {
        if (lengths[k] < 0) { int64_t bad = lengths[k]; free(lengths); return fail(r, PQD_ERR_NEGATIVE_LENGTH, bad); }
        offsets[k] = last;
        last += (uint32_t)lengths[k];
    }
    offsets[nl] = last;
    free(lengths);
    r.n_values = nl;
    if ((uint64_t)last > rest_n) return fail(r, PQD_ERR_LENGTH_OUT_OF_BOUNDS, (int64_t)last);
    if ((uint64_t)last > cap) return fail(r, PQD_ERR_DST_TOO_SMALL, (int64_t)last);
    memmove(dst, rest, last); /* CanDecodeInPlace: dst may alias src */
    r.n_bytes = last;
    r.consumed += last;
    return r;
}

/* encoding/delta/length_byte_array.go:65-70, delta.go:28-33 (wrapping int32 sum). */
static int64_t sum_lengths(const int32_t* v, uint64_t n) {
    uint32_t s = 0;
    for (uint64_t k = 0; k < n; k++) s += (uint32_t)v[k];
    return (int64_t)(int32_t)s;
}

int64_t pqo_dlba_estimate(const uint8_t* src, uint64_t n) {
    int32_t* lengths;
    uint64_t nl;
    (void)decode_lengths(src, n, &lengths, &nl);
    int64_t s = sum_lengths(lengths, nl);
    free(lengths);
    return s;
}

/* encoding/delta/byte_array.go:176-183. */
int64_t pqo_dba_estimate(const uint8_t* src, uint64_t n) {
    int32_t* v;
    uint64_t nv;
    pqo_result r = decode_lengths(src, n, &v, &nv);
    int64_t s = sum_lengths(v, nv);
    free(v);
    (void)decode_lengths(src + r.consumed, n - r.consumed, &v, &nv);
    s += sum_lengths(v, nv);
    free(v);
    return s;
}

/* encoding/delta/byte_array.go:124-174 + byte_array_purego.go:5-63. */
static pqo_result dba_decode(uint8_t* dst, uint64_t cap, uint32_t* offsets, uint64_t offsets_cap, int with_offsets,
                             const uint8_t* src, uint64_t n) {
    int32_t *prefix = NULL, *suffix = NULL;
    uint64_t np = 0, ns = 0;
    pqo_result r = decode_lengths(src, n, &prefix, &np);
    if (r.status != PQD_OK) { free(prefix); r.n_values = r.n_bytes = 0; return r; }
    uint64_t used = r.consumed;
    r = decode_lengths(src + used, n - used, &suffix, &ns);
    if (r.status != PQD_OK) { free(prefix); free(suffix); r.n_values = r.n_bytes = 0; r.consumed += used; return r; }
    used += r.consumed;
    r = ok_result();
    r.consumed = used;
    if (np != ns) {
        free(prefix); free(suffix);
        return fail(r, PQD_ERR_PREFIX_SUFFIX_MISMATCH, (int64_t)((np << 32) | (ns & 0xffffffffu)));
    }
    const uint8_t* data = src + used;
    uint64_t data_n = n - used;
    uint64_t out = 0, last_off = 0, last_len = 0, nvals = 0;
    int32_t status = PQD_OK;
    int64_t arg = 0;
    for (uint64_t k = 0; k < ns; k++) {
        int64_t sl = suffix[k], pl = prefix[k];
        if (sl < 0) { status = PQD_ERR_NEGATIVE_LENGTH; arg = sl; break; }
        if ((uint64_t)sl > data_n) { status = PQD_ERR_LENGTH_OUT_OF_BOUNDS; arg = sl; break; }
        if (pl < 0) { status = PQD_ERR_NEGATIVE_PREFIX; arg = pl; break; }
        if ((uint64_t)pl > last_len) { status = PQD_ERR_PREFIX_OUT_OF_BOUNDS; arg = pl; break; }
        if (with_offsets) {
            if (nvals + 2 > offsets_cap) { status = PQD_ERR_OFFSETS_TOO_SMALL; arg = (int64_t)(ns + 1); break; }
            offsets[nvals] = (uint32_t)out;
        }
        if (out + (uint64_t)pl + (uint64_t)sl > cap) {
            /* report the full size so one retry suffices */
            uint64_t need = 0;
            for (uint64_t q = 0; q < ns; q++) need += (uint64_t)(prefix[q] > 0 ? prefix[q] : 0) + (uint64_t)(suffix[q] > 0 ? suffix[q] : 0);
            status = PQD_ERR_DST_TOO_SMALL; arg = (int64_t)need; break;
        }
        uint64_t j = out;
        memmove(dst + out, dst + last_off, (uint64_t)pl);
        out += (uint64_t)pl;
        memcpy(dst + out, data, (uint64_t)sl);
        out += (uint64_t)sl;
        data += sl;
        data_n -= (uint64_t)sl;
        last_off = j;
        last_len = out - j;
        nvals++;
    }
    free(prefix);
    free(suffix);
    r.n_values = nvals;
    r.n_bytes = out;
    if (status != PQD_OK) return fail(r, status, arg);
    if (with_offsets) {
        if (nvals + 1 > offsets_cap) return fail(r, PQD_ERR_OFFSETS_TOO_SMALL, (int64_t)(nvals + 1));
        offsets[nvals] = (uint32_t)out;
    }
    r.consumed = (uint64_t)(data - src);
    return r;
}

pqo_result pqo_dba_decode(uint8_t* dst, uint64_t cap, uint32_t* offsets, uint64_t offsets_cap,
                          const uint8_t* src, uint64_t n) {
    return dba_decode(dst, cap, offsets, offsets_cap, 1, src, n);
}

/* encoding/delta/byte_array.go:148-174. */
pqo_result pqo_dba_decode_flba(uint8_t* dst, uint64_t cap, const uint8_t* src, uint64_t n, int64_t size) {
    if (size < 0 || size > MAX_FLBA_SIZE) return fail(ok_result(), PQD_ERR_INVALID_ARGUMENT, size);
    return dba_decode(dst, cap, NULL, 0, 0, src, n);
}

/* ===================================================================== */
/* BYTE_STREAM_SPLIT, PLAIN                                               */
/* ===================================================================== */

/* encoding/bytestreamsplit/bytestreamsplit.go:35-105,
 * bytestreamsplit_purego.go:47-83, bytestreamsplit_fixedlen.go:13-21. */
pqo_result pqo_bss_decode(uint8_t* dst, uint64_t cap, const uint8_t* src, uint64_t n, int64_t size, int is_flba) {
    pqo_result r = ok_result();
    if (is_flba && (size <= 0 || size > MAX_FLBA_SIZE)) return fail(r, PQD_ERR_INVALID_ARGUMENT, size);
    if (n % (uint64_t)size) return fail(r, PQD_ERR_INVALID_ARGUMENT, (int64_t)n);
    if (n > cap) return fail(r, PQD_ERR_DST_TOO_SMALL, (int64_t)n);
    uint64_t count = n / (uint64_t)size;
    for (uint64_t s = 0; s < (uint64_t)size; s++)
        for (uint64_t k = 0; k < count; k++) dst[k * (uint64_t)size + s] = src[s * count + k];
    r.n_values = count;
    r.n_bytes = n;
    return r;
}

/* encoding/plain/plain.go:65-74,96-104, plain_le.go:26-52. width 1 = BOOLEAN
 * (plain copy), 4/8 = INT32/FLOAT/INT64/DOUBLE, 12 = INT96. */
pqo_result pqo_plain_decode_fixed(uint8_t* dst, uint64_t cap, const uint8_t* src, uint64_t n, int64_t width, int is_flba) {
    pqo_result r = ok_result();
    if (is_flba && (width <= 0 || width > MAX_FLBA_SIZE)) return fail(r, PQD_ERR_INVALID_ARGUMENT, width);
    if (n % (uint64_t)width) return fail(r, PQD_ERR_INVALID_ARGUMENT, (int64_t)n);
    if (n > cap) return fail(r, PQD_ERR_DST_TOO_SMALL, (int64_t)n);
    memmove(dst, src, n); /* CanDecodeInPlace */
    r.n_values = n / (uint64_t)width;
    r.n_bytes = n;
    return r;
}

/* encoding/plain/plain.go:76-94. */
pqo_result pqo_plain_decode_byte_array(uint8_t* dst, uint64_t cap, uint32_t* offsets, uint64_t offsets_cap,
                                       const uint8_t* src, uint64_t n) {
    pqo_result r = ok_result();
    uint64_t out = 0, nv = 0;
    for (uint64_t i = 0; i < n;) {
        r.n_values = nv;
        r.n_bytes = out;
        if (n - i < 4) return fail(r, PQD_ERR_UNEXPECTED_EOF, PQD_EOF_PLAIN_BYTE_ARRAY);
        uint64_t len = (uint64_t)src[i] | ((uint64_t)src[i + 1] << 8) | ((uint64_t)src[i + 2] << 16) | ((uint64_t)src[i + 3] << 24);
        if (len > n - 4) return fail(r, PQD_ERR_UNEXPECTED_EOF, PQD_EOF_PLAIN_BYTE_ARRAY);
        i += 4;
        if (len > n - i) return fail(r, PQD_ERR_UNEXPECTED_EOF, PQD_EOF_PLAIN_BYTE_ARRAY);
        if (nv + 2 > offsets_cap) return fail(r, PQD_ERR_OFFSETS_TOO_SMALL, (int64_t)(nv + 2));
        if (out + len > cap) return fail(r, PQD_ERR_DST_TOO_SMALL, (int64_t)(out + len));
        offsets[nv++] = (uint32_t)out;
        memmove(dst + out, src + i, len); /* in place: out <= i always */
        out += len;
        i += len;
    }
    if (nv + 1 > offsets_cap) return fail(r, PQD_ERR_OFFSETS_TOO_SMALL, (int64_t)(nv + 1));
    offsets[nv] = (uint32_t)out;
    r.n_values = nv;
    r.n_bytes = out;
    r.consumed = n;
    return r;
}

/* ===================================================================== */
/* Dictionary surface                                                     */
/* ===================================================================== */

/* dictionary_purego.go:12-38,54-66 / dictionary_amd64.s:578-711:
 * rows[i] = dict[indexes[i]] into a strided destination. */
pqo_result pqo_dict_lookup(uint8_t* rows, uint64_t row_stride, const uint8_t* dict, uint64_t dict_len,
                           uint32_t width, const int32_t* indexes, uint64_t n) {
    pqo_result r = ok_result();
    for (uint64_t k = 0; k < n; k++) {
        uint32_t j = (uint32_t)indexes[k];
        if ((uint64_t)j >= dict_len) { r.n_values = k; return fail(r, PQD_ERR_INDEX_OUT_OF_BOUNDS, (int64_t)k); }
        memcpy(rows + k * row_stride, dict + (uint64_t)j * width, width);
    }
    r.n_values = n;
    r.n_bytes = n * width;
    return r;
}

/* dictionary.go:136-162 (newIndexedPage): the decoded index buffer is cut to
 * num_values, or zero-extended when the stream held fewer. */
void pqo_indexed_page_fixup(int32_t* values, uint64_t decoded, uint64_t num_values) {
    for (uint64_t k = decoded; k < num_values; k++) values[k] = 0;
}

/* ===================================================================== */
/* Encoders -- fixture generation only                                    */
/* ===================================================================== */

#define ENC_FAIL UINT64_MAX

static int all_equal_u8(const uint8_t* p, uint64_t n, uint8_t v) {
    for (uint64_t i = 0; i < n; i++) if (p[i] != v) return 0;
    return 1;
}

/* encoding/rle/rle.go:141-201 (encodeBytes). */
uint64_t pqo_rle_encode_levels(uint8_t* dst, uint64_t cap, const uint8_t* src, uint64_t n, uint32_t bw) {
    if (bw > 8) return ENC_FAIL;
    if (cap < n * 2 + 32) return ENC_FAIL;
    uint64_t o = 0;
    if (bw == 0) {
        if (!all_equal_u8(src, n, 0)) return ENC_FAIL;
        return put_uvarint(dst, n << 1);
    }
    uint64_t nw = n / 8;
    const uint64_t mask = (1ull << bw) - 1;
    for (uint64_t i = 0; i < nw;) {
        uint64_t j = i;
        uint8_t pat = src[8 * i];
        while (j < nw && all_equal_u8(src + 8 * j, 8, pat)) j++;
        if (i < j) {
            o += put_uvarint(dst + o, (8 * (j - i)) << 1);
            dst[o++] = pat;
        } else {
            j++;
            /* extend while group j is not a broadcast of the first byte of group j-1 */
            while (j < nw && !all_equal_u8(src + 8 * j, 8, src[8 * (j - 1)])) j++;
            o += put_uvarint(dst + o, ((j - i) << 1) | 1);
            for (uint64_t g = i; g < j; g++) {
                uint64_t word = 0;
                for (unsigned v = 0; v < 8; v++) word |= ((uint64_t)src[8 * g + v] & mask) << (v * bw);
                for (unsigned b = 0; b < bw; b++) dst[o + b] = (uint8_t)(word >> (8 * b));
                o += bw;
            }
        }
        i = j;
    }
    for (uint64_t i = nw * 8; i < n;) {
        uint64_t j = i + 1;
        while (j < n && src[i] == src[j]) j++;
        o += put_uvarint(dst + o, (j - i) << 1);
        dst[o++] = src[i];
        i = j;
    }
    return o;
}

static int group_is_broadcast(const int32_t* g, int32_t v) {
    for (int k = 0; k < 8; k++) if (g[k] != v) return 0;
    return 1;
}

/* encoding/rle/rle.go:203-249 (encodeInt32), :478-492. */
uint64_t pqo_rle_encode_int32(uint8_t* dst, uint64_t cap, const int32_t* src, uint64_t n, uint32_t bw) {
    if (bw > 32) return ENC_FAIL;
    if (cap < n * 5 + 64) return ENC_FAIL;
    uint64_t o = 0;
    if (bw == 0) {
        for (uint64_t i = 0; i < n; i++) if (src[i] != 0) return ENC_FAIL;
        return put_uvarint(dst, n << 1);
    }
    const uint64_t vb = (bw + 7) / 8;
    uint64_t nw = n / 8;
    for (uint64_t i = 0; i < nw;) {
        uint64_t j = i;
        int32_t pat = src[8 * i];
        while (j < nw && group_is_broadcast(src + 8 * j, pat)) j++;
        if (i < j) {
            o += put_uvarint(dst + o, (8 * (j - i)) << 1);
            for (uint64_t b = 0; b < vb; b++) dst[o + b] = (uint8_t)((uint32_t)pat >> (8 * b));
            o += vb;
        } else {
            j++;
            /* encodeInt32IndexEqual8Contiguous: first group whose 8 values are all equal */
            while (j < nw && !group_is_broadcast(src + 8 * j, src[8 * j])) j++;
            o += put_uvarint(dst + o, ((j - i) << 1) | 1);
            uint64_t nv = 8 * (j - i);
            uint64_t nbytes = (nv * bw + 7) / 8;
            memset(dst + o, 0, nbytes);
            uint64_t fmask = bw == 32 ? 0xffffffffull : ((1ull << bw) - 1);
            for (uint64_t v = 0; v < nv; v++) pack_field(dst + o, v, bw, (uint64_t)(uint32_t)src[8 * i + v] & fmask);
            o += nbytes;
        }
        i = j;
    }
    for (uint64_t i = nw * 8; i < n;) {
        uint64_t j = i + 1;
        while (j < n && src[i] == src[j]) j++;
        o += put_uvarint(dst + o, (j - i) << 1);
        for (uint64_t b = 0; b < vb; b++) dst[o + b] = (uint8_t)((uint32_t)src[i] >> (8 * b));
        o += vb;
        i = j;
    }
    return o;
}

/* encoding/rle/rle.go:48-56,97-139 (EncodeBoolean, encodeBits). */
uint64_t pqo_rle_encode_boolean(uint8_t* dst, uint64_t cap, const uint8_t* src, uint64_t n) {
    if (cap < n * 2 + 32) return ENC_FAIL;
    uint64_t o = 4;
    if (n == 0 || all_equal_u8(src, n, 0) || all_equal_u8(src, n, 0xFF)) {
        o += put_uvarint(dst + o, (8 * n) << 1);
        if (n > 0) dst[o++] = src[0];
    } else {
        for (uint64_t i = 0; i < n;) {
            uint64_t j = i + 1;
            if (src[i] == 0 || src[i] == 0xFF) {
                while (j < n && src[i] == src[j]) j++;
                if (j - i > 1) {
                    o += put_uvarint(dst + o, (8 * (j - i)) << 1);
                    dst[o++] = src[i];
                    i = j;
                    continue;
                }
            }
            while (j < n && (src[j - 1] != src[j] || (src[j] != 0 && src[j] == 0xFF))) j++;
            if ((j - i) > 1 && j < n) j--;
            o += put_uvarint(dst + o, ((j - i) << 1) | 1);
            memcpy(dst + o, src + i, j - i);
            o += j - i;
            i = j;
        }
    }
    uint32_t len = (uint32_t)(o - 4);
    dst[0] = (uint8_t)len; dst[1] = (uint8_t)(len >> 8); dst[2] = (uint8_t)(len >> 16); dst[3] = (uint8_t)(len >> 24);
    return o;
}

static uint32_t bits_len64(uint64_t v) {
    uint32_t n = 0;
    while (v) { n++; v >>= 1; }
    return n;
}

/* encoding/rle/dictionary.go:23-28,52-59. */
uint64_t pqo_rle_dict_encode_int32(uint8_t* dst, uint64_t cap, const int32_t* src, uint64_t n) {
    uint32_t bw = 0;
    for (uint64_t i = 0; i < n; i++) {
        uint32_t l = bits_len64((uint32_t)src[i]);
        if (l > bw) bw = l;
    }
    if (cap < 1) return ENC_FAIL;
    dst[0] = (uint8_t)bw;
    uint64_t k = pqo_rle_encode_int32(dst + 1, cap - 1, src, n, bw);
    return k == ENC_FAIL ? ENC_FAIL : k + 1;
}

/* encoding/bitpacked/bitpacked.go:38-73 (encodeLevels), MSB-first. */
uint64_t pqo_bitpacked_encode_levels(uint8_t* dst, uint64_t cap, const uint8_t* src, uint64_t n, uint32_t bw) {
    if (bw == 0 || n == 0) {
        if (cap < 1) return ENC_FAIL;
        dst[0] = 0;
        return 1;
    }
    uint64_t nb = (bw * n + 7) / 8;
    if (cap < nb + 1) return ENC_FAIL;
    memset(dst, 0, nb + 1);
    uint64_t bit_offset = 0;
    for (uint64_t k = 0; k < n; k++) {
        uint64_t i = bit_offset / 8, j = bit_offset % 8, available = 8 - j;
        uint8_t value = src[k];
        if (bw <= available) {
            dst[i] |= shl8(value, available - bw);
        } else {
            dst[i] |= shr8(value, bw - available);
            dst[i + 1] |= shl8(value, 8 - (bw - available));
        }
        bit_offset += bw;
    }
    return nb;
}

/* encoding/delta/binary_packed.go:77-181 generalised to any (block_size,
 * num_miniblocks) the decoder accepts; the reference writer uses 128/4. */
static uint64_t delta_encode(uint8_t* dst, uint64_t cap, const void* srcv, uint64_t n, int is64,
                             uint32_t block_size, uint32_t num_mb) {
    const int64_t* s64 = (const int64_t*)srcv;
    const int32_t* s32 = (const int32_t*)srcv;
    const uint32_t elem_bits = is64 ? 64 : 32;
    const uint32_t vpm = block_size / num_mb;
    uint64_t worst = 64 + (n / block_size + 2) * (10ull + num_mb + (uint64_t)block_size * elem_bits / 8);
    if (cap < worst) return ENC_FAIL;
    uint64_t o = 0;
    int64_t first = n > 0 ? (is64 ? s64[0] : (int64_t)s32[0]) : 0;
    o += put_uvarint(dst + o, block_size);
    o += put_uvarint(dst + o, num_mb);
    o += put_uvarint(dst + o, n);
    o += put_varint(dst + o, first);
    if (n < 2) return o;
    uint64_t* block = (uint64_t*)malloc(sizeof(uint64_t) * block_size);
    uint64_t last = (uint64_t)first;
    for (uint64_t i = 1; i < n; i += block_size) {
        uint64_t blen = n - i < block_size ? n - i : block_size;
        /* the reference copies, deltas and mins over the whole zero-padded
         * block, then clears the padding (binary_packed.go:139-146) */
        for (uint64_t k = 0; k < block_size; k++) {
            uint64_t v = 0;
            if (k < blen) v = is64 ? (uint64_t)s64[i + k] : (uint64_t)(int64_t)s32[i + k];
            uint64_t d = v - last;
            if (!is64) d = (uint64_t)(int64_t)(int32_t)(uint32_t)d;
            block[k] = d;
            last = v;
        }
        int64_t min_delta = (int64_t)block[0];
        for (uint64_t k = 1; k < block_size; k++) if ((int64_t)block[k] < min_delta) min_delta = (int64_t)block[k];
        for (uint64_t k = 0; k < block_size; k++) {
            uint64_t d = block[k] - (uint64_t)min_delta;
            if (!is64) d = (uint32_t)d;
            block[k] = k < blen ? d : 0;
        }
        /* lastValue after a padded block is the padding's 0 minus chain in
         * the reference only for the final block, where it is unused */
        last = is64 ? (uint64_t)s64[i + blen - 1] : (uint64_t)(int64_t)s32[i + blen - 1];
        o += put_varint(dst + o, min_delta);
        uint8_t* widths = dst + o;
        o += num_mb;
        for (uint32_t m = 0; m < num_mb; m++) {
            uint32_t bw = 0;
            for (uint32_t k = 0; k < vpm; k++) {
                uint32_t l = bits_len64(block[m * vpm + k]);
                if (l > bw) bw = l;
            }
            widths[m] = (uint8_t)bw;
        }
        for (uint32_t m = 0; m < num_mb; m++) {
            uint32_t bw = widths[m];
            if (bw == 0) continue;
            uint64_t nbytes = (uint64_t)vpm * bw / 8;
            memset(dst + o, 0, nbytes);
            for (uint32_t k = 0; k < vpm; k++) pack_field(dst + o, k, bw, block[m * vpm + k]);
            o += nbytes;
        }
    }
    free(block);
    return o;
}

uint64_t pqo_delta_encode_int32(uint8_t* dst, uint64_t cap, const int32_t* src, uint64_t n, uint32_t block_size, uint32_t num_mb) {
    return delta_encode(dst, cap, src, n, 0, block_size, num_mb);
}

uint64_t pqo_delta_encode_int64(uint8_t* dst, uint64_t cap, const int64_t* src, uint64_t n, uint32_t block_size, uint32_t num_mb) {
    return delta_encode(dst, cap, src, n, 1, block_size, num_mb);
}

/* encoding/delta/length_byte_array.go:20-35. */
uint64_t pqo_dlba_encode(uint8_t* dst, uint64_t cap, const uint8_t* src, const uint32_t* offsets, uint64_t n_offsets) {
    if (n_offsets == 0) return 0;
    uint64_t nv = n_offsets - 1;
    int32_t* lengths = (int32_t*)malloc(4 * (nv + 1));
    for (uint64_t k = 0; k < nv; k++) lengths[k] = (int32_t)(offsets[k + 1] - offsets[k]);
    uint64_t o = delta_encode(dst, cap, lengths, nv, 0, 128, 4);
    free(lengths);
    if (o == ENC_FAIL) return ENC_FAIL;
    uint64_t total = offsets[nv];
    if (o + total > cap) return ENC_FAIL;
    memcpy(dst + o, src, total);
    return o + total;
}

/* encoding/delta/byte_array.go:26-74. */
uint64_t pqo_dba_encode(uint8_t* dst, uint64_t cap, const uint8_t* src, const uint32_t* offsets, uint64_t n_offsets) {
    uint64_t nv = n_offsets > 0 ? n_offsets - 1 : 0;
    int32_t* prefix = (int32_t*)malloc(4 * (nv + 1));
    int32_t* suffix = (int32_t*)malloc(4 * (nv + 1));
    uint64_t total = 0;
    const uint8_t* last = NULL;
    uint64_t last_n = 0;
    for (uint64_t k = 0; k < nv; k++) {
        const uint8_t* v = src + offsets[k];
        uint64_t len = offsets[k + 1] - offsets[k];
        uint64_t p = 0;
        while (p < last_n && p < len && last[p] == v[p]) p++;
        prefix[k] = (int32_t)p;
        suffix[k] = (int32_t)(len - p);
        total += len - p;
        last = v;
        last_n = len;
    }
    uint64_t o = delta_encode(dst, cap, prefix, nv, 0, 128, 4);
    if (o != ENC_FAIL) {
        uint64_t o2 = delta_encode(dst + o, cap - o, suffix, nv, 0, 128, 4);
        o = o2 == ENC_FAIL ? ENC_FAIL : o + o2;
    }
    if (o != ENC_FAIL && o + total <= cap) {
        for (uint64_t k = 0; k < nv; k++) {
            memcpy(dst + o, src + offsets[k] + prefix[k], (uint64_t)suffix[k]);
            o += (uint64_t)suffix[k];
        }
    } else {
        o = ENC_FAIL;
    }
    free(prefix);
    free(suffix);
    return o;
}

/* encoding/bytestreamsplit/bytestreamsplit_purego.go:7-45, bytestreamsplit_fixedlen.go:3-11. */
uint64_t pqo_bss_encode(uint8_t* dst, const uint8_t* src, uint64_t n, uint64_t size) {
    uint64_t count = n / size;
    for (uint64_t s = 0; s < size; s++)
        for (uint64_t k = 0; k < count; k++) dst[s * count + k] = src[k * size + s];
    return n;
}

/* encoding/plain/plain.go:43-56,182-188. */
uint64_t pqo_plain_encode_byte_array(uint8_t* dst, uint64_t cap, const uint8_t* src, const uint32_t* offsets, uint64_t n_offsets) {
    uint64_t o = 0;
    for (uint64_t k = 0; k + 1 < n_offsets; k++) {
        uint32_t len = offsets[k + 1] - offsets[k];
        if (o + 4 + len > cap) return ENC_FAIL;
        dst[o] = (uint8_t)len; dst[o + 1] = (uint8_t)(len >> 8); dst[o + 2] = (uint8_t)(len >> 16); dst[o + 3] = (uint8_t)(len >> 24);
        memcpy(dst + o + 4, src + offsets[k], len);
        o += 4 + len;
    }
    return o;
}

/* A level stream the way parquet-mr style writers lay it out (the shape of
 * testdata/alltypes_tiny_pages.parquet): runs of 8 or more equal values become
 * RLE runs of arbitrary length, everything else goes into bit-packed groups
 * of 8, the last group zero padded.  Not a restatement of reference code:
 * the reference writer never produces this shape, its decoder must read it. */
uint64_t pqo_rle_encode_levels_mr(uint8_t* dst, uint64_t cap, const uint8_t* src, uint64_t n, uint32_t bw) {
    if (bw == 0 || bw > 8 || cap < n * 2 + 64) return ENC_FAIL;
    uint64_t o = 0, i = 0;
    const uint64_t mask = (1ull << bw) - 1;
    while (i < n) {
        uint64_t run = 1;
        while (i + run < n && src[i + run] == src[i]) run++;
        if (run >= 8) {
            o += put_uvarint(dst + o, run << 1);
            dst[o++] = src[i];
            i += run;
            continue;
        }
        /* gather literal groups until a long run starts on a group boundary */
        uint64_t start = i, groups = 0;
        while (i < n && groups < 63) {
            uint64_t r2 = 1;
            while (i + r2 < n && src[i + r2] == src[i]) r2++;
            if (r2 >= 8) break;
            i += 8;
            groups++;
        }
        if (i > n) i = n;
        o += put_uvarint(dst + o, (groups << 1) | 1);
        for (uint64_t g = 0; g < groups; g++) {
            uint64_t word = 0;
            for (unsigned v = 0; v < 8; v++) {
                uint64_t idx = start + g * 8 + v;
                uint64_t val = idx < n ? src[idx] : 0;
                word |= (val & mask) << (v * bw);
            }
            for (unsigned b = 0; b < bw; b++) dst[o + b] = (uint8_t)(word >> (8 * b));
            o += bw;
        }
    }
    return o;
}
