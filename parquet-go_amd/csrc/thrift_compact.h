// thrift_compact.h -- reader for the thrift compact protocol (encoding/thrift/compact.go:132-170,
// 196-245), enough of it for the structures of format/parquet.go: field headers, varints,
// binaries, list headers, and skipping whatever a caller does not look at.  Host code.
#pragma once
#include <stdint.h>

namespace pqd {

// Thrift compact protocol reader over a byte range; `ok` drops to false at the first
// read past the end or malformed element and stays there.
struct Compact {
    const uint8_t* p;
    const uint8_t* end;
    bool ok = true;
    int depth = 0;
    static constexpr int MAX_DEPTH = 32; // nested structs, lists, sets and maps

    uint8_t byte() {
        if (p >= end) { ok = false; return 0; }
        return *p++;
    }
    uint64_t uvarint() {
        uint64_t x = 0;
        for (int s = 0; s < 70; s += 7) {
            const uint8_t c = byte();
            if (!ok) return 0;
            x |= (uint64_t)(c & 0x7f) << s;
            if (c < 0x80) return x;
        }
        ok = false;
        return 0;
    }
    int64_t zigzag() {
        const uint64_t u = uvarint();
        return (int64_t)(u >> 1) ^ -(int64_t)(u & 1);
    }
    void advance(uint64_t n) {
        if ((uint64_t)(end - p) < n) { ok = false; p = end; return; }
        p += n;
    }
    // compact.go type codes: 1 TRUE, 2 FALSE, 3 I8, 4 I16, 5 I32, 6 I64, 7 DOUBLE, 8 BINARY,
    // 9 LIST, 10 SET, 11 MAP, 12 STRUCT
    void skip(int t, bool in_collection) {
        if (!ok) return;
        switch (t) {
        case 1: case 2: if (in_collection) advance(1); break; // booleans of a collection take a byte
        case 3: advance(1); break;
        case 4: case 5: case 6: (void)uvarint(); break;
        case 7: advance(8); break;
        case 8: advance(uvarint()); break;
        // collections nest like structs do: the same depth limit, or a run of list headers
        // (0x19 0x19 ...) would recurse once per byte
        case 9: case 10: {
            if (++depth > MAX_DEPTH) { ok = false; return; }
            const uint8_t h = byte();
            uint64_t n = h >> 4;
            if (n == 15) n = uvarint();
            for (uint64_t i = 0; i < n && ok; i++) skip(h & 0xf, true);
            depth--;
            break;
        }
        case 11: {
            if (++depth > MAX_DEPTH) { ok = false; return; }
            const uint64_t n = uvarint();
            if (n > 0) {
                const uint8_t kv = byte();
                for (uint64_t i = 0; i < n && ok; i++) { skip(kv >> 4, true); skip(kv & 0xf, true); }
            }
            depth--;
            break;
        }
        case 12: skip_struct(); break;
        default: ok = false;
        }
    }
    // One field header: false at STOP.  Booleans carry their value in the type.
    bool field(int* id, int* type, int* last) {
        const uint8_t h = byte();
        if (!ok || h == 0) return false;
        *type = h & 0xf;
        const int delta = h >> 4;
        *id = delta ? *last + delta : (int)zigzag();
        *last = *id;
        return ok;
    }
    void skip_struct() {
        if (++depth > MAX_DEPTH) { ok = false; return; }
        int id, type, last = 0;
        while (field(&id, &type, &last)) skip(type, false);
        depth--;
    }
    int64_t i64(int type) {
        if (type != 4 && type != 5 && type != 6 && type != 3) { ok = false; return 0; }
        if (type == 3) return (int8_t)byte();
        return zigzag();
    }
    // A binary or string field: its bytes stay where they are.
    void binary(int type, const uint8_t** s, uint64_t* n) {
        *s = p;
        *n = 0;
        if (type != 8) { ok = false; return; }
        const uint64_t len = uvarint();
        if (!ok) return;
        *s = p;
        advance(len);
        if (ok) *n = len;
    }
    // The header of a list or set field: element type and count.  Every element takes at
    // least a byte, which bounds the count by the bytes left.
    uint64_t list(int type, int* elem_type) {
        *elem_type = 0;
        if (type != 9 && type != 10) { ok = false; return 0; }
        const uint8_t h = byte();
        uint64_t n = h >> 4;
        if (n == 15) n = uvarint();
        *elem_type = h & 0xf;
        if (!ok || n > (uint64_t)(end - p)) { ok = false; return 0; }
        return n;
    }
    int32_t i32(int type) {
        if (type != 4 && type != 5 && type != 6 && type != 3) { ok = false; return 0; }
        if (type == 3) return (int8_t)byte();
        return (int32_t)zigzag();
    }
};

} // namespace pqd
