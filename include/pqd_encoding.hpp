// pqd_encoding.hpp -- C++ host-side mirror of parquet-go's codec objects over the C ABI
// of pqd.h.  Header only.
//
// The reference is compiled Go; a Go toolchain is not part of this build, so the host side
// above the C ABI is given in C++ with the reference's names and argument meaning:
// `encoding.Encoding` (encoding/encoding.go:20-72) becomes pqd::Encoding, whose Decode
// methods take (dst, src), write from index 0 of dst, grow it when it is too small (Go's
// "may grow dst", encoding/rle/rle.go:510-522) and report errors as pqd::Error carrying the
// classes Go callers test with errors.Is (encoding/notsupported.go:11-39).  A codec answers
// NotSupported for the types its Go counterpart leaves to encoding.NotSupported
// (encoding/notsupported.go:116-205).
#pragma once

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "pqd.h"

namespace pqd {

enum class ErrorClass { NotSupported, InvalidArgument, UnexpectedEOF, IndexOutOfBounds, Other };

class Error : public std::runtime_error {
public:
    Error(const std::string& encoding, int status, int64_t arg)
        : std::runtime_error(encoding + ": " + describe(status, arg)), status(status), arg(arg), cls(classify(status)) {}
    int status;
    int64_t arg;
    ErrorClass cls;

    static ErrorClass classify(int st) {
        switch (st) {
            case PQD_ERR_NOT_SUPPORTED: return ErrorClass::NotSupported;          // encoding.ErrNotSupported
            case PQD_ERR_INVALID_ARGUMENT:
            case PQD_ERR_INVALID_BIT_WIDTH: return ErrorClass::InvalidArgument;   // encoding.ErrInvalidArgument
            case PQD_ERR_UNEXPECTED_EOF:
            case PQD_ERR_MISSING_VALUES: return ErrorClass::UnexpectedEOF;        // io.ErrUnexpectedEOF
            case PQD_ERR_INDEX_OUT_OF_BOUNDS: return ErrorClass::IndexOutOfBounds; // errors.go:75-83 (a panic in Go)
            default: return ErrorClass::Other;
        }
    }
    static std::string describe(int st, int64_t arg) {
        if (st == PQD_ERR_NOT_SUPPORTED) return "encoding not supported";
        if (st == PQD_ERR_MISSING_VALUES) return std::to_string(arg) + " missing values: unexpected EOF";
        return std::string(pqd_status_name(st)) + " (" + std::to_string(arg) + ")";
    }
};

// One process-wide context per device, created on first use.
inline pqd_ctx* default_context(int device = 0) {
    static pqd_ctx* ctx[64] = {nullptr};
    if (device < 0 || device >= 64) throw Error("pqd", PQD_ERR_DEVICE, device);
    if (ctx[device] == nullptr) {
        int rc = pqd_ctx_create(device, &ctx[device]);
        if (rc != PQD_OK) throw Error("pqd", rc, device);
    }
    return ctx[device];
}

struct ByteArrays {               // encoding.Values of kind BYTE_ARRAY (encoding/values.go:236-276)
    std::vector<uint8_t> data;    // the values back to back
    std::vector<uint32_t> offsets; // n + 1 entries
};

class Encoding {
public:
    Encoding(const char* name, int code, int bit_width = 0) : BitWidth(bit_width), name_(name), code_(code) {}
    virtual ~Encoding() = default;
    int BitWidth; // rle.Encoding.BitWidth / bitpacked.Encoding.BitWidth

    std::string String() const { return name_; }
    int EncodingCode() const { return code_; }   // Encoding() format.Encoding in Go
    bool CanDecodeInPlace() const { return pqd_can_decode_in_place(code_) != 0; }

    // The decode methods of encoding.Encoding (encoding/encoding.go:48-56).  The library
    // answers PQD_ERR_NOT_SUPPORTED for (encoding, type) pairs the reference leaves to
    // encoding.NotSupported, so one implementation serves every codec.
    void DecodeLevels(std::vector<uint8_t>& dst, const uint8_t* src, size_t n) const {
        fixed(dst, [&](uint8_t* d, uint64_t cap, uint64_t* ob, int64_t* a) { return pqd_decode_levels(ctx(), code_, BitWidth, src, n, d, cap, ob, a); });
    }
    void DecodeBoolean(std::vector<uint8_t>& dst, const uint8_t* src, size_t n) const {
        fixed(dst, [&](uint8_t* d, uint64_t cap, uint64_t* ob, int64_t* a) { return pqd_decode_boolean(ctx(), code_, src, n, d, cap, ob, a); });
    }
    void DecodeInt32(std::vector<int32_t>& dst, const uint8_t* src, size_t n) const {
        fixed(dst, [&](int32_t* d, uint64_t cap, uint64_t* ob, int64_t* a) { return pqd_decode_int32(ctx(), code_, BitWidth, src, n, d, cap, ob, a); });
    }
    void DecodeInt64(std::vector<int64_t>& dst, const uint8_t* src, size_t n) const {
        fixed(dst, [&](int64_t* d, uint64_t cap, uint64_t* ob, int64_t* a) { return pqd_decode_int64(ctx(), code_, src, n, d, cap, ob, a); });
    }
    void DecodeInt96(std::vector<uint8_t>& dst, const uint8_t* src, size_t n) const {
        fixed(dst, [&](uint8_t* d, uint64_t cap, uint64_t* ob, int64_t* a) { return pqd_decode_int96(ctx(), code_, src, n, d, cap, ob, a); });
    }
    void DecodeFloat(std::vector<float>& dst, const uint8_t* src, size_t n) const {
        fixed(dst, [&](float* d, uint64_t cap, uint64_t* ob, int64_t* a) { return pqd_decode_float(ctx(), code_, src, n, d, cap, ob, a); });
    }
    void DecodeDouble(std::vector<double>& dst, const uint8_t* src, size_t n) const {
        fixed(dst, [&](double* d, uint64_t cap, uint64_t* ob, int64_t* a) { return pqd_decode_double(ctx(), code_, src, n, d, cap, ob, a); });
    }
    void DecodeFixedLenByteArray(std::vector<uint8_t>& dst, const uint8_t* src, size_t n, int size) const {
        fixed(dst, [&](uint8_t* d, uint64_t cap, uint64_t* ob, int64_t* a) { return pqd_decode_fixed_len_byte_array(ctx(), code_, size, src, n, d, cap, ob, a); });
    }
    void DecodeByteArray(ByteArrays& dst, const uint8_t* src, size_t n) const {
        if (dst.data.size() < 64) dst.data.resize(n * 4 + 64);
        if (dst.offsets.size() < 64) dst.offsets.resize(n + 64);
        uint64_t ob = 0, oo = 0;
        int64_t arg = 0;
        int rc = PQD_OK;
        for (int attempt = 0; attempt < 4; attempt++) {
            rc = pqd_decode_byte_array(ctx(), code_, src, n, dst.data.data(), dst.data.size(), dst.offsets.data(), dst.offsets.size(), &ob, &oo, &arg);
            if (rc == PQD_ERR_DST_TOO_SMALL && (uint64_t)arg > dst.data.size()) { dst.data.resize((size_t)arg); continue; }
            if (rc == PQD_ERR_OFFSETS_TOO_SMALL && (uint64_t)arg > dst.offsets.size()) { dst.offsets.resize((size_t)arg); continue; }
            break;
        }
        dst.data.resize((size_t)ob);   // what was decoded before a failure is still handed back
        dst.offsets.resize((size_t)oo);
        if (rc != PQD_OK) throw Error(name_, rc, arg);
    }
    int64_t EstimateDecodeByteArraySize(const uint8_t* src, size_t n) const {
        return pqd_estimate_decode_byte_array_size(ctx(), code_, src, n);
    }

private:
    static pqd_ctx* ctx() { return default_context(0); }

    template <typename T, typename F>
    void fixed(std::vector<T>& dst, F call) const {
        if (dst.size() < 16) dst.resize(16);
        uint64_t ob = 0;
        int64_t arg = 0;
        int rc = PQD_OK;
        for (int attempt = 0; attempt < 3; attempt++) {
            rc = call(dst.data(), (uint64_t)dst.size() * sizeof(T), &ob, &arg);
            if (rc == PQD_ERR_DST_TOO_SMALL && (uint64_t)arg > dst.size() * sizeof(T)) {
                dst.resize(((size_t)arg + sizeof(T) - 1) / sizeof(T));
                continue;
            }
            break;
        }
        dst.resize((size_t)(ob / sizeof(T)));
        if (rc != PQD_OK) throw Error(name_, rc, arg);
    }

    const char* name_;
    int code_;
};

// The codecs of the reference's `encodings` table (encoding.go:50-60).
inline Encoding Plain() { return Encoding("PLAIN", PQD_ENC_PLAIN); }
inline Encoding PlainDictionary() { return Encoding("PLAIN_DICTIONARY", PQD_ENC_PLAIN_DICTIONARY); }
inline Encoding RLE(int bit_width = 0) { return Encoding("RLE", PQD_ENC_RLE, bit_width); }
inline Encoding RLEDictionary() { return Encoding("RLE_DICTIONARY", PQD_ENC_RLE_DICTIONARY); }
inline Encoding BitPacked(int bit_width) { return Encoding("BIT_PACKED", PQD_ENC_BIT_PACKED, bit_width); }
inline Encoding DeltaBinaryPacked() { return Encoding("DELTA_BINARY_PACKED", PQD_ENC_DELTA_BINARY_PACKED); }
inline Encoding DeltaLengthByteArray() { return Encoding("DELTA_LENGTH_BYTE_ARRAY", PQD_ENC_DELTA_LENGTH_BYTE_ARRAY); }
inline Encoding DeltaByteArray() { return Encoding("DELTA_BYTE_ARRAY", PQD_ENC_DELTA_BYTE_ARRAY); }
inline Encoding ByteStreamSplit() { return Encoding("BYTE_STREAM_SPLIT", PQD_ENC_BYTE_STREAM_SPLIT); }

// Dictionary.Lookup for fixed-width dictionaries (dictionary.go:64-69).
template <typename T>
class Dictionary {
public:
    explicit Dictionary(const std::vector<T>& values) : n_(values.size()) {
        int rc = pqd_dict_create(default_context(0), values.data(), values.size(), sizeof(T), &d_);
        if (rc != PQD_OK) throw Error("DICTIONARY", rc, 0);
    }
    ~Dictionary() { pqd_dict_destroy(d_); }
    Dictionary(const Dictionary&) = delete;
    Dictionary& operator=(const Dictionary&) = delete;
    size_t Len() const { return n_; }
    void Lookup(const std::vector<int32_t>& indexes, std::vector<T>& rows) const {
        rows.resize(indexes.size());
        int64_t arg = 0;
        int rc = pqd_dict_lookup(d_, indexes.data(), indexes.size(), rows.data(), sizeof(T), &arg);
        if (rc != PQD_OK) throw Error("DICTIONARY", rc, arg);
    }

private:
    pqd_dict* d_ = nullptr;
    size_t n_;
};

// The page headers of a column chunk held in memory: the header loop of FilePages.ReadPage
// (file.go:1193-1259).  `num_values` as in ColumnMetaData, or -1 to walk to the end of the bytes.
inline std::vector<pqd_page_header> ScanPages(const uint8_t* chunk, size_t n, int64_t num_values = -1) {
    std::vector<pqd_page_header> out(16);
    for (;;) {
        uint64_t found = 0;
        int64_t arg = 0;
        int rc = pqd_chunk_scan_pages(chunk, n, num_values, out.data(), out.size(), &found, &arg);
        if (rc == PQD_ERR_DST_TOO_SMALL) { out.resize(found); continue; }
        if (rc != PQD_OK) throw Error("PAGE_HEADER", rc, arg);
        out.resize(found);
        return out;
    }
}

// The parts of the pages of an uncompressed chunk, as Column.decodeDataPageV1 / V2 /
// decodeDictionary cut them (column.go:733-843, 963-993): items[i] is ready for
// pqd_batch_create once the caller has set dst (and offsets, dict); parts[i] says which page
// and which part it is.  `chunk_dev` is where the chunk lies in device memory.
struct PagePlan {
    std::vector<pqd_page> items;
    std::vector<pqd_page_part> parts;
};
inline PagePlan PlanPages(const std::vector<pqd_page_header>& headers, const uint8_t* chunk, size_t n, const void* chunk_dev,
                          const pqd_column_info& column) {
    PagePlan plan;
    plan.items.resize(3 * headers.size() + 1);
    plan.parts.resize(plan.items.size());
    uint64_t found = 0;
    int64_t arg = 0;
    int rc = pqd_chunk_plan_pages(headers.data(), headers.size(), chunk, n, chunk_dev, &column, plan.items.data(), plan.parts.data(),
                                  plan.items.size(), &found, &arg);
    if (rc != PQD_OK) throw Error("PAGE", rc, arg);
    plan.items.resize(found);
    plan.parts.resize(found);
    return plan;
}

} // namespace pqd
