// chunk.hip -- host side: from the bytes of a column chunk to the work items of a batch.
//
//   pqd_chunk_scan_pages   the header loop of FilePages.ReadPage (file.go:1193-1259): every
//                          format.PageHeader of the chunk (format/parquet.go:752-821,910-953),
//                          read with the thrift compact protocol
//                          (encoding/thrift/compact.go:132-170,196-245) and checked as
//                          validatePageHeader does (file.go:1414-1445);
//   pqd_chunk_plan_pages   how Column.decodeDataPageV1 / decodeDataPageV2 / decodeDictionary
//                          (column.go:733-843,963-993) cut an uncompressed page into
//                          repetition levels, definition levels and values, and which encoding
//                          object decodes each part (column.go:845-857,913-929,976-979,
//                          encoding.go:124-134).
//
// No GPU work here; the items go to pqd_batch_create.
#include <string.h>

#include "pqd_internal.h"
#include "thrift_compact.h"

namespace {

using pqd::Compact;

// crc32.ChecksumIEEE (polynomial 0xEDB88320, reflected), eight bytes per step.
struct Crc32 {
    uint32_t t[8][256];
    Crc32() {
        for (uint32_t i = 0; i < 256; i++) {
            uint32_t c = i;
            for (int k = 0; k < 8; k++) c = (c >> 1) ^ (0xEDB88320u & (0u - (c & 1)));
            t[0][i] = c;
        }
        for (uint32_t i = 0; i < 256; i++)
            for (int k = 1; k < 8; k++) t[k][i] = (t[k - 1][i] >> 8) ^ t[0][t[k - 1][i] & 0xff];
    }
    uint32_t sum(const uint8_t* p, uint64_t n) const {
        uint32_t c = 0xffffffffu;
        while (n >= 8) {
            uint32_t lo, hi;
            memcpy(&lo, p, 4);
            memcpy(&hi, p + 4, 4);
            lo ^= c;
            c = t[7][lo & 0xff] ^ t[6][(lo >> 8) & 0xff] ^ t[5][(lo >> 16) & 0xff] ^ t[4][lo >> 24] ^
                t[3][hi & 0xff] ^ t[2][(hi >> 8) & 0xff] ^ t[1][(hi >> 16) & 0xff] ^ t[0][hi >> 24];
            p += 8;
            n -= 8;
        }
        while (n--) c = (c >> 8) ^ t[0][(c ^ *p++) & 0xff];
        return ~c;
    }
};

const Crc32& crc32_table() {
    static const Crc32 table;
    return table;
}

int32_t bits_len8(int32_t v) { // bits.Len8
    int32_t n = 0;
    for (uint32_t x = (uint32_t)v & 0xff; x; x >>= 1) n++;
    return n;
}

} // namespace

extern "C" {

int pqd_chunk_scan_pages(const uint8_t* chunk, uint64_t chunk_len, int64_t num_values, pqd_page_header* out, uint64_t cap,
                         uint64_t* n_pages, int64_t* status_arg) {
    if ((chunk == nullptr && chunk_len > 0) || n_pages == nullptr || (out == nullptr && cap > 0)) return PQD_ERR_INVALID_ARGUMENT;
    *n_pages = 0;
    if (status_arg) *status_arg = 0;
    uint64_t pos = 0, count = 0;
    int64_t seen = 0;
    // a chunk ends with its bytes or, when the caller knows it, with its last value
    while (pos < chunk_len && (num_values < 0 || seen < num_values)) {
        Compact r{chunk + pos, chunk + chunk_len};
        pqd_page_header h;
        memset(&h, 0, sizeof h);
        h.num_nulls = h.num_rows = -1;
        h.is_compressed = 1;
        h.header_offset = pos;
        bool have_type = false, have_part = false;
        int id, type, last = 0;
        while (r.field(&id, &type, &last)) {
            if (id == 1) { h.type = r.i32(type); have_type = true; }
            else if (id == 2) h.uncompressed_page_size = r.i32(type);
            else if (id == 3) h.compressed_page_size = r.i32(type);
            else if (id == 4) { h.crc = r.i32(type); h.has_crc = 1; }
            else if ((id == 5 || id == 7 || id == 8) && type == 12) {
                have_part = true;
                int fid, ft, fl = 0;
                while (r.field(&fid, &ft, &fl)) {
                    if (id == 5) { // DataPageHeader
                        if (fid == 1) h.num_values = r.i32(ft);
                        else if (fid == 2) h.encoding = r.i32(ft);
                        else if (fid == 3) h.definition_level_encoding = r.i32(ft);
                        else if (fid == 4) h.repetition_level_encoding = r.i32(ft);
                        else r.skip(ft, false);
                    } else if (id == 7) { // DictionaryPageHeader
                        if (fid == 1) h.num_values = r.i32(ft);
                        else if (fid == 2) h.encoding = r.i32(ft);
                        else if (fid == 3 && (ft == 1 || ft == 2)) h.is_sorted = ft == 1;
                        else r.skip(ft, false);
                    } else { // DataPageHeaderV2
                        if (fid == 1) h.num_values = r.i32(ft);
                        else if (fid == 2) h.num_nulls = r.i32(ft);
                        else if (fid == 3) h.num_rows = r.i32(ft);
                        else if (fid == 4) h.encoding = r.i32(ft);
                        else if (fid == 5) h.definition_levels_byte_length = r.i32(ft);
                        else if (fid == 6) h.repetition_levels_byte_length = r.i32(ft);
                        else if (fid == 7 && (ft == 1 || ft == 2)) h.is_compressed = ft == 1;
                        else r.skip(ft, false);
                        h.definition_level_encoding = h.repetition_level_encoding = PQD_ENC_RLE;
                    }
                }
            } else r.skip(type, false);
        }
        if (!r.ok || !have_type) {
            if (status_arg) *status_arg = (int64_t)pos;
            *n_pages = count;
            return PQD_ERR_UNEXPECTED_EOF;
        }
        (void)have_part;
        // validatePageHeader (file.go:1414-1445)
        const bool negative = h.compressed_page_size < 0 || h.uncompressed_page_size < 0 || h.num_values < 0 ||
                              (h.type == PQD_PAGE_DATA_V2 && (h.num_nulls < 0 || h.num_rows < 0 || h.repetition_levels_byte_length < 0 ||
                                                              h.definition_levels_byte_length < 0));
        h.data_offset = (uint64_t)(r.p - chunk);
        if (negative || (uint64_t)h.compressed_page_size > chunk_len - h.data_offset) {
            if (status_arg) *status_arg = (int64_t)pos;
            *n_pages = count;
            return negative ? PQD_ERR_INVALID_ARGUMENT : PQD_ERR_UNEXPECTED_EOF;
        }
        // FilePages.readPage (file.go:1455-1475): a checksum other than zero is compared with
        // the IEEE CRC-32 of the page bytes as stored (compressed or not)
        if (h.has_crc && h.crc != 0 &&
            crc32_table().sum(chunk + h.data_offset, (uint64_t)h.compressed_page_size) != (uint32_t)h.crc) {
            if (status_arg) *status_arg = (int64_t)pos;
            *n_pages = count;
            return PQD_ERR_CHECKSUM;
        }
        if (count < cap) out[count] = h;
        count++;
        if (h.type == PQD_PAGE_DATA || h.type == PQD_PAGE_DATA_V2) seen += h.num_values;
        pos = h.data_offset + (uint64_t)h.compressed_page_size;
    }
    *n_pages = count;
    return count > cap ? PQD_ERR_DST_TOO_SMALL : PQD_OK;
}

int pqd_chunk_plan_pages(const pqd_page_header* headers, uint64_t n_headers, const uint8_t* chunk, uint64_t chunk_len,
                         const void* chunk_dev, const pqd_column_info* column, pqd_page* items, pqd_page_part* parts, uint64_t cap,
                         uint64_t* n_items, int64_t* status_arg) {
    if (headers == nullptr || column == nullptr || n_items == nullptr || (chunk == nullptr && chunk_len > 0) ||
        ((items == nullptr || parts == nullptr) && cap > 0))
        return PQD_ERR_INVALID_ARGUMENT;
    *n_items = 0;
    if (status_arg) *status_arg = 0;
    uint64_t count = 0;
    const uint8_t* dev = static_cast<const uint8_t*>(chunk_dev);
    auto emit = [&](uint64_t page, int32_t role, uint64_t off, uint64_t len, int32_t encoding, int32_t kind, int32_t bit_width,
                    int64_t num_values) {
        if (count < cap) {
            pqd_page it;
            memset(&it, 0, sizeof it);
            it.src = dev + off;
            it.src_len = len;
            it.num_values = num_values;
            it.encoding = encoding;
            it.kind = kind;
            it.bit_width = bit_width;
            items[count] = it;
            parts[count].page = page;
            parts[count].role = role;
            parts[count].reserved = 0;
            parts[count].src_offset = off;
        }
        count++;
    };
    auto fail = [&](uint64_t page, int status) {
        if (status_arg) *status_arg = (int64_t)page;
        *n_items = count;
        return status;
    };
    for (uint64_t k = 0; k < n_headers; k++) {
        const pqd_page_header& h = headers[k];
        const uint64_t size = (uint64_t)h.compressed_page_size;
        if (h.data_offset > chunk_len || size > chunk_len - h.data_offset) return fail(k, PQD_ERR_UNEXPECTED_EOF);
        uint64_t at = h.data_offset, rest = size;
        if (h.type == PQD_PAGE_DICTIONARY) {
            // column.go:976-979: a PLAIN_DICTIONARY dictionary page is PLAIN
            const int32_t enc = h.encoding == PQD_ENC_PLAIN_DICTIONARY ? PQD_ENC_PLAIN : h.encoding;
            emit(k, PQD_PART_DICTIONARY, at, rest, enc, column->kind, column->type_length, h.num_values);
            continue;
        }
        if (h.type != PQD_PAGE_DATA && h.type != PQD_PAGE_DATA_V2) continue; // index pages hold no values (file.go:1253-1254)
        int64_t n_present = h.num_values;
        if (h.type == PQD_PAGE_DATA) {
            // column.go:733-778, 913-924: each level section is a little-endian uint32 length and that many bytes
            for (int pass = 0; pass < 2; pass++) {
                const int32_t max_level = pass == 0 ? column->max_repetition_level : column->max_definition_level;
                if (max_level <= 0) continue;
                if (rest < 4) return fail(k, PQD_ERR_UNEXPECTED_EOF);
                const uint64_t len = (uint64_t)chunk[at] | ((uint64_t)chunk[at + 1] << 8) | ((uint64_t)chunk[at + 2] << 16) |
                                     ((uint64_t)chunk[at + 3] << 24);
                if (len > rest - 4) return fail(k, PQD_ERR_UNEXPECTED_EOF);
                const int32_t enc = pass == 0 ? h.repetition_level_encoding : h.definition_level_encoding;
                emit(k, pass == 0 ? PQD_PART_REPETITION_LEVELS : PQD_PART_DEFINITION_LEVELS, at + 4, len, enc, PQD_KIND_LEVELS,
                     bits_len8(max_level), h.num_values);
                at += 4 + len;
                rest -= 4 + len;
            }
            if (column->max_definition_level > 0) n_present = -1; // known once the definition levels are counted (column.go:772)
        } else {
            // column.go:780-843: repetition levels, then definition levels, lengths in the header, always RLE
            const uint64_t rl = (uint64_t)h.repetition_levels_byte_length, dl = (uint64_t)h.definition_levels_byte_length;
            if (rl > rest || dl > rest - rl) return fail(k, PQD_ERR_UNEXPECTED_EOF);
            if (column->max_repetition_level > 0)
                emit(k, PQD_PART_REPETITION_LEVELS, at, rl, PQD_ENC_RLE, PQD_KIND_LEVELS, bits_len8(column->max_repetition_level), h.num_values);
            if (column->max_definition_level > 0)
                emit(k, PQD_PART_DEFINITION_LEVELS, at + rl, dl, PQD_ENC_RLE, PQD_KIND_LEVELS, bits_len8(column->max_definition_level),
                     h.num_values);
            at += rl + dl;
            rest -= rl + dl;
            n_present = (int64_t)h.num_values - (int64_t)h.num_nulls; // column.go:841
            if (n_present < 0) return fail(k, PQD_ERR_INVALID_ARGUMENT);
        }
        // column.go:850-857: PLAIN_DICTIONARY data pages are RLE_DICTIONARY
        const int32_t enc = h.encoding == PQD_ENC_PLAIN_DICTIONARY ? PQD_ENC_RLE_DICTIONARY : h.encoding;
        emit(k, PQD_PART_VALUES, at, rest, enc, column->kind, column->type_length, n_present);
    }
    *n_items = count;
    return count > cap ? PQD_ERR_DST_TOO_SMALL : PQD_OK;
}

} // extern "C"
