// file.hip -- host side: from the last bytes of a parquet file to the byte ranges of its
// column chunks.
//
//   pqd_file_footer_range   the tail of OpenFile (file.go:108-131): magic, footer length;
//   pqd_file_parse_footer   format.FileMetaData (format/parquet.go:1386-1436) with its
//                           SchemaElement (:527-572), RowGroup (:1116-1144), ColumnChunk
//                           (:1084-1114) and ColumnMetaData (:987-1050) members, read with the
//                           thrift compact protocol; the schema walked depth first into leaf
//                           columns as columnLoader.open does (column.go:369-521), with the
//                           levels of Column.setLevels (column.go:234-280) and the chunk range
//                           FileColumnChunk reads its pages from (file.go:1119-1128).
//
// No GPU work here; the ranges go to pqd_chunk_scan_pages / pqd_chunk_plan_pages.
#include <string.h>

#include <string>
#include <vector>

#include "pqd_internal.h"
#include "thrift_compact.h"

namespace {

using pqd::Compact;

struct SchemaElement {
    bool has_type = false, has_children = false, has_repetition = false;
    int32_t type = 0, type_length = 0, repetition = 0, num_children = 0, converted_type = -1;
    const uint8_t* name = nullptr;
    uint64_t name_len = 0;
};

void read_schema_element(Compact& r, SchemaElement* e) {
    int id, type, last = 0;
    while (r.field(&id, &type, &last)) {
        switch (id) {
        case 1: e->type = r.i32(type); e->has_type = true; break;
        case 2: e->type_length = r.i32(type); break;
        case 3: e->repetition = r.i32(type); e->has_repetition = true; break;
        case 4: r.binary(type, &e->name, &e->name_len); break;
        case 5: e->num_children = r.i32(type); e->has_children = true; break;
        case 6: e->converted_type = r.i32(type); break;
        default: r.skip(type, false);
        }
    }
}

void read_column_metadata(Compact& r, pqd_file_chunk* c) {
    int id, type, last = 0;
    while (r.field(&id, &type, &last)) {
        switch (id) {
        case 1: c->kind = r.i32(type) + 1; break; // PQD_KIND_* counts from 1
        case 2: {
            int et;
            const uint64_t n = r.list(type, &et);
            for (uint64_t i = 0; i < n && r.ok; i++) {
                const int32_t e = r.i32(et);
                if (e >= 0 && e < 32) c->encodings |= 1u << e;
            }
            break;
        }
        case 4: c->codec = r.i32(type); break;
        case 5: c->num_values = r.i64(type); break;
        case 6: c->total_uncompressed_size = r.i64(type); break;
        case 7: c->total_compressed_size = r.i64(type); break;
        case 9: c->data_page_offset = r.i64(type); break;
        case 11: c->dictionary_page_offset = r.i64(type); break;
        default: r.skip(type, false);
        }
    }
    c->has_metadata = 1;
}

void read_column_chunk(Compact& r, pqd_file_chunk* c) {
    memset(c, 0, sizeof *c);
    int id, type, last = 0;
    while (r.field(&id, &type, &last)) {
        if (id == 3 && type == 12) read_column_metadata(r, c);
        else if (id == 9 && type == 8) { // encrypted_column_metadata: the metadata is not readable here
            c->is_encrypted = 1;
            r.skip(type, false);
        } else r.skip(type, false);
    }
    // FileColumnChunk.Pages (file.go:1119-1128): pages start at the dictionary page when there is one
    c->offset = c->dictionary_page_offset != 0 ? c->dictionary_page_offset : c->data_page_offset;
}

struct RowGroup {
    pqd_file_row_group info;
    std::vector<pqd_file_chunk> chunks;
};

void read_row_group(Compact& r, RowGroup* g) {
    memset(&g->info, 0, sizeof g->info);
    int id, type, last = 0;
    while (r.field(&id, &type, &last)) {
        switch (id) {
        case 1: {
            int et;
            const uint64_t n = r.list(type, &et);
            if (n > 0 && et != 12) { r.ok = false; break; }
            g->chunks.reserve(n < 4096 ? n : 4096); // counts are untrusted: no allocation sized by them
            for (uint64_t i = 0; i < n && r.ok; i++) {
                g->chunks.emplace_back();
                read_column_chunk(r, &g->chunks.back());
            }
            break;
        }
        case 2: g->info.total_byte_size = r.i64(type); break;
        case 3: g->info.num_rows = r.i64(type); break;
        case 5: g->info.file_offset = r.i64(type); break;
        case 6: g->info.total_compressed_size = r.i64(type); break;
        default: r.skip(type, false);
        }
    }
}

// columnLoader.open + Column.setLevels in one walk: schema element `*next` and its subtree.
struct Walk {
    const std::vector<SchemaElement>& schema;
    std::vector<pqd_file_column> columns;
    std::string names;
    uint64_t next = 0;
    int64_t children_left; // columnLoader.children: one slot per element but the root
    int status = PQD_OK;
    int64_t where = 0;

    explicit Walk(const std::vector<SchemaElement>& s) : schema(s), children_left((int64_t)s.size() - 1) {}

    bool fail(int64_t at) {
        status = PQD_ERR_INVALID_ARGUMENT;
        where = at;
        return false;
    }

    bool open(int depth, int repetition, int definition, const std::string& parent) {
        if (next >= schema.size()) return fail((int64_t)next);
        const uint64_t me = next++;
        const SchemaElement& e = schema[me];
        // limits.go:9-21
        if (depth > 255 || repetition > 255 || definition > 255 || columns.size() > 65534) return fail((int64_t)me);
        if (depth > 0 && e.has_repetition) {
            if (e.repetition == 1) definition++;
            else if (e.repetition == 2) { repetition++; definition++; }
        }
        std::string path = parent;
        if (depth > 0) { // Column.Path leaves the root's name out (column.go:70)
            if (!path.empty()) path += '\0'; // a separator no field name holds: names may contain '.' (Column.Path is a []string, column.go:378-383)
            path.append((const char*)e.name, e.name_len);
        }
        if (e.has_type) {
            pqd_file_column c;
            memset(&c, 0, sizeof c);
            c.kind = e.type + 1; // PQD_KIND_* = format.Type + 1
            c.type_length = e.type_length;
            c.repetition = e.has_repetition ? e.repetition : 0;
            c.converted_type = e.converted_type;
            c.max_definition_level = (uint8_t)definition;
            c.max_repetition_level = (uint8_t)repetition;
            c.depth = (uint8_t)depth;
            c.schema_index = (uint32_t)me;
            c.name_offset = (uint32_t)names.size();
            c.name_len = (uint32_t)path.size();
            names += path;
            columns.push_back(c);
            return true;
        }
        const int64_t n = e.has_children ? e.num_children : 0;
        if (n > children_left) return fail((int64_t)me);
        if (n > 0) children_left -= n;
        for (int64_t i = 0; i < n; i++) {
            if (next >= schema.size()) return fail((int64_t)me);
            if (!open(depth + 1, repetition, definition, path)) return false;
        }
        return true;
    }
};

} // namespace

extern "C" {

int pqd_file_footer_range(const uint8_t* tail, uint64_t tail_len, uint64_t file_size, uint64_t* footer_offset,
                          uint64_t* footer_len) {
    if (tail == nullptr || footer_offset == nullptr || footer_len == nullptr || tail_len > file_size) return PQD_ERR_INVALID_ARGUMENT;
    *footer_offset = *footer_len = 0;
    if (tail_len < 8) return PQD_ERR_UNEXPECTED_EOF;
    const uint8_t* b = tail + tail_len - 8;
    if (memcmp(b + 4, "PARE", 4) == 0) return PQD_ERR_NOT_SUPPORTED; // encrypted footer
    if (memcmp(b + 4, "PAR1", 4) != 0) return PQD_ERR_INVALID_ARGUMENT;
    uint32_t size;
    memcpy(&size, b, 4);
    if ((uint64_t)size + 8 > file_size) return PQD_ERR_UNEXPECTED_EOF;
    *footer_len = size;
    *footer_offset = file_size - 8 - size;
    return PQD_OK;
}

static int parse_footer(const uint8_t* footer, uint64_t footer_len, pqd_file_info* info, pqd_file_column* columns,
                        uint64_t column_cap, pqd_file_row_group* row_groups, uint64_t row_group_cap, pqd_file_chunk* chunks,
                        uint64_t chunk_cap, char* names, uint64_t names_cap, int64_t* status_arg) {
    if ((footer == nullptr && footer_len > 0) || info == nullptr || (columns == nullptr && column_cap > 0) ||
        (row_groups == nullptr && row_group_cap > 0) || (chunks == nullptr && chunk_cap > 0) || (names == nullptr && names_cap > 0))
        return PQD_ERR_INVALID_ARGUMENT;
    memset(info, 0, sizeof *info);
    if (status_arg) *status_arg = 0;

    Compact r{footer, footer + footer_len};
    std::vector<SchemaElement> schema;
    std::vector<RowGroup> groups;
    int id, type, last = 0;
    while (r.field(&id, &type, &last)) {
        switch (id) {
        case 1: info->version = r.i32(type); break;
        case 2: {
            int et;
            const uint64_t n = r.list(type, &et);
            if (n > 0 && et != 12) { r.ok = false; break; }
            schema.reserve(n < 4096 ? n : 4096);
            for (uint64_t i = 0; i < n && r.ok; i++) {
                schema.emplace_back();
                read_schema_element(r, &schema.back());
            }
            break;
        }
        case 3: info->num_rows = r.i64(type); break;
        case 4: {
            int et;
            const uint64_t n = r.list(type, &et);
            if (n > 0 && et != 12) { r.ok = false; break; }
            groups.reserve(n < 4096 ? n : 4096);
            for (uint64_t i = 0; i < n && r.ok; i++) {
                groups.emplace_back();
                read_row_group(r, &groups.back());
            }
            break;
        }
        case 8: info->is_encrypted = 1; r.skip(type, false); break; // encryption_algorithm: plaintext footer of an encrypted file
        default: r.skip(type, false);
        }
    }
    if (!r.ok) {
        if (status_arg) *status_arg = (int64_t)(r.p - footer);
        return PQD_ERR_UNEXPECTED_EOF;
    }
    info->footer_bytes_read = (uint64_t)(r.p - footer);

    Walk walk(schema);
    if (!schema.empty() && !walk.open(0, 0, 0, std::string())) {
        if (status_arg) *status_arg = walk.where;
        return walk.status;
    }
    // openColumns (column.go:223-228): every row group has exactly one chunk per leaf
    for (uint64_t g = 0; g < groups.size(); g++) {
        if (groups[g].chunks.size() != walk.columns.size()) {
            if (status_arg) *status_arg = (int64_t)g;
            return PQD_ERR_INVALID_ARGUMENT;
        }
    }

    info->n_schema = schema.size();
    info->n_columns = walk.columns.size();
    info->n_row_groups = groups.size();
    info->n_chunks = groups.size() * walk.columns.size();
    info->names_bytes = walk.names.size();
    if (info->n_columns > column_cap || info->n_row_groups > row_group_cap || info->n_chunks > chunk_cap || info->names_bytes > names_cap)
        return PQD_ERR_DST_TOO_SMALL;
    if (!walk.columns.empty()) memcpy(columns, walk.columns.data(), walk.columns.size() * sizeof(pqd_file_column));
    if (!walk.names.empty()) memcpy(names, walk.names.data(), walk.names.size());
    for (uint64_t g = 0; g < groups.size(); g++) {
        row_groups[g] = groups[g].info;
        for (uint64_t c = 0; c < walk.columns.size(); c++) chunks[g * walk.columns.size() + c] = groups[g].chunks[c];
    }
    return PQD_OK;
}

// No C++ exception may cross the C ABI (into cgo or ctypes it ends the process): a footer that makes
// the containers run out of memory comes back as a status.
int pqd_file_parse_footer(const uint8_t* footer, uint64_t footer_len, pqd_file_info* info, pqd_file_column* columns,
                          uint64_t column_cap, pqd_file_row_group* row_groups, uint64_t row_group_cap, pqd_file_chunk* chunks,
                          uint64_t chunk_cap, char* names, uint64_t names_cap, int64_t* status_arg) {
    try {
        return parse_footer(footer, footer_len, info, columns, column_cap, row_groups, row_group_cap, chunks, chunk_cap, names,
                            names_cap, status_arg);
    } catch (...) {
        if (status_arg) *status_arg = 0;
        return PQD_ERR_DEVICE;
    }
}

} // extern "C"
