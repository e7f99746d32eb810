// Exercises include/pqd_encoding.hpp (the C++ mirror of parquet-go's codec objects) against
// the CPU checker under oracle/, the way encoding/encoding_test.go round-trips the Go codecs.
// Test infrastructure: links libpqoracle.so for encoders and expected values.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

#include "../../include/pqd_encoding.hpp"
extern "C" {
#include "../../oracle/pq_oracle.h"
}

#define CHECK(c) do { if (!(c)) { std::printf("FAILED %s:%d: %s\n", __FILE__, __LINE__, #c); return 1; } } while (0)

static uint64_t rng_state = 0x9E3779B97F4A7C15ull;
static uint64_t rnd() { rng_state ^= rng_state << 13; rng_state ^= rng_state >> 7; rng_state ^= rng_state << 17; return rng_state; }

int main() {
    // DELTA_BINARY_PACKED int64: round trip through a dst that has to grow
    std::vector<int64_t> v(100000);
    int64_t t = 1600000000000000000ll;
    for (auto& x : v) { t += 1000000 + (int64_t)(rnd() % 4096); x = t; }
    std::vector<uint8_t> enc(v.size() * 10 + 64);
    enc.resize(pqo_delta_encode_int64(enc.data(), enc.size(), v.data(), v.size(), 128, 4));
    std::vector<int64_t> got;
    pqd::Encoding dbp = pqd::DeltaBinaryPacked();
    CHECK(dbp.String() == "DELTA_BINARY_PACKED" && dbp.EncodingCode() == 5 && !dbp.CanDecodeInPlace());
    dbp.DecodeInt64(got, enc.data(), enc.size());
    CHECK(got == v);

    // truncated stream: io.ErrUnexpectedEOF class, the partially filled dst is still returned
    try {
        dbp.DecodeInt64(got, enc.data(), enc.size() / 2);
        CHECK(!"no error on a truncated stream");
    } catch (const pqd::Error& e) {
        std::vector<int64_t> want(v.size());
        pqo_result r = pqo_delta_decode_int64(want.data(), want.size() * 8, enc.data(), enc.size() / 2);
        CHECK(e.cls == pqd::ErrorClass::UnexpectedEOF && e.status == r.status && e.arg == r.arg);
        CHECK(got.size() == v.size() && got == want);
        CHECK(std::string(e.what()).find("DELTA_BINARY_PACKED: ") == 0);
    }

    // RLE levels
    std::vector<uint8_t> lv(70000);
    for (size_t i = 0; i < lv.size(); i++) lv[i] = (i / 37) % 3 == 0 ? (uint8_t)(rnd() % 8) : 5;
    std::vector<uint8_t> lenc(lv.size() * 2 + 64);
    lenc.resize(pqo_rle_encode_levels(lenc.data(), lenc.size(), lv.data(), lv.size(), 3));
    std::vector<uint8_t> lgot;
    pqd::RLE(3).DecodeLevels(lgot, lenc.data(), lenc.size());
    CHECK(lgot == lv);

    // a type the codec does not implement: encoding.ErrNotSupported
    try {
        pqd::Plain().DecodeLevels(lgot, lenc.data(), lenc.size());
        CHECK(!"PLAIN decoded levels");
    } catch (const pqd::Error& e) { CHECK(e.cls == pqd::ErrorClass::NotSupported); }
    try {
        std::vector<int32_t> d;
        pqd::Plain().DecodeInt32(d, (const uint8_t*)"12345", 5);
        CHECK(!"PLAIN decoded 5 bytes of int32");
    } catch (const pqd::Error& e) { CHECK(e.cls == pqd::ErrorClass::InvalidArgument); }

    // DELTA_BYTE_ARRAY strings, estimate never undershoots (encoding_test.go:486-502)
    std::vector<uint8_t> data;
    std::vector<uint32_t> offs{0};
    for (int i = 0; i < 3000; i++) {
        std::string s = "https://www.example.com/items/" + std::to_string(100000 + i * 7);
        data.insert(data.end(), s.begin(), s.end());
        offs.push_back((uint32_t)data.size());
    }
    std::vector<uint8_t> benc(data.size() * 2 + 4096);
    benc.resize(pqo_dba_encode(benc.data(), benc.size(), data.data(), offs.data(), offs.size()));
    pqd::ByteArrays ba;
    pqd::Encoding dba = pqd::DeltaByteArray();
    dba.DecodeByteArray(ba, benc.data(), benc.size());
    CHECK(ba.data == data && ba.offsets == offs);
    CHECK(dba.EstimateDecodeByteArraySize(benc.data(), benc.size()) == pqo_dba_estimate(benc.data(), benc.size()));
    CHECK(dba.EstimateDecodeByteArraySize(benc.data(), benc.size()) >= (int64_t)data.size());

    // Dictionary.Lookup, and the index that makes the reference panic
    std::vector<int64_t> dict(5000);
    for (auto& x : dict) x = (int64_t)rnd();
    std::vector<int32_t> idx(20000);
    for (auto& x : idx) x = (int32_t)(rnd() % dict.size());
    pqd::Dictionary<int64_t> D(dict);
    std::vector<int64_t> rows;
    D.Lookup(idx, rows);
    for (size_t i = 0; i < idx.size(); i++) CHECK(rows[i] == dict[idx[i]]);
    idx[17] = 5000;
    try {
        D.Lookup(idx, rows);
        CHECK(!"out-of-range index accepted");
    } catch (const pqd::Error& e) { CHECK(e.cls == pqd::ErrorClass::IndexOutOfBounds && e.arg == 17); }

    // a column chunk put together by hand: a dictionary page and a v1 data page with definition
    // levels; headers in the thrift compact protocol (0x15 = next field, i32; 0x2c / 0x4c = a struct
    // two / four fields on; 0x00 = stop)
    {
        auto zz = [](std::vector<uint8_t>& o, int32_t v) {
            uint32_t u = ((uint32_t)v << 1) ^ (uint32_t)(v >> 31);
            while (u >= 0x80) { o.push_back((uint8_t)(u | 0x80)); u >>= 7; }
            o.push_back((uint8_t)u);
        };
        std::vector<int64_t> dvals{11, 22, 33, 44};
        std::vector<uint8_t> dict_body((const uint8_t*)dvals.data(), (const uint8_t*)dvals.data() + 32);
        std::vector<uint8_t> def{1, 1, 0, 1, 0, 1, 1, 1, 1, 0};                      // 7 of 10 slots hold a value
        std::vector<uint8_t> def_enc(4096);
        def_enc.resize(pqo_rle_encode_levels(def_enc.data(), def_enc.size(), def.data(), def.size(), 1));
        std::vector<int32_t> idx{3, 0, 1, 1, 2, 3, 0};
        std::vector<uint8_t> idx_enc(4096);
        idx_enc.resize(pqo_rle_dict_encode_int32(idx_enc.data(), idx_enc.size(), idx.data(), idx.size()));
        std::vector<uint8_t> body;
        uint32_t dl = (uint32_t)def_enc.size();
        body.insert(body.end(), (uint8_t*)&dl, (uint8_t*)&dl + 4);
        body.insert(body.end(), def_enc.begin(), def_enc.end());
        const size_t values_at = body.size();
        body.insert(body.end(), idx_enc.begin(), idx_enc.end());
        std::vector<uint8_t> chunk;
        // PageHeader{type = DICTIONARY_PAGE, sizes, dictionary_page_header{num_values 4, PLAIN_DICTIONARY}}
        chunk.push_back(0x15); zz(chunk, 2); chunk.push_back(0x15); zz(chunk, 32); chunk.push_back(0x15); zz(chunk, 32);
        chunk.push_back(0x4c); chunk.push_back(0x15); zz(chunk, 4); chunk.push_back(0x15); zz(chunk, 2); chunk.push_back(0x00);
        chunk.push_back(0x00);
        const size_t dict_at = chunk.size();
        chunk.insert(chunk.end(), dict_body.begin(), dict_body.end());
        // PageHeader{type = DATA_PAGE, sizes, data_page_header{num_values 10, PLAIN_DICTIONARY, RLE, RLE}}
        const size_t second = chunk.size();
        chunk.push_back(0x15); zz(chunk, 0); chunk.push_back(0x15); zz(chunk, (int32_t)body.size()); chunk.push_back(0x15); zz(chunk, (int32_t)body.size());
        chunk.push_back(0x2c); chunk.push_back(0x15); zz(chunk, 10); chunk.push_back(0x15); zz(chunk, 2); chunk.push_back(0x15); zz(chunk, 3);
        chunk.push_back(0x15); zz(chunk, 3); chunk.push_back(0x00);
        chunk.push_back(0x00);
        const size_t data_at = chunk.size();
        chunk.insert(chunk.end(), body.begin(), body.end());

        std::vector<pqd_page_header> hs = pqd::ScanPages(chunk.data(), chunk.size(), 10);
        CHECK(hs.size() == 2);
        CHECK(hs[0].type == PQD_PAGE_DICTIONARY && hs[0].num_values == 4 && hs[0].data_offset == dict_at && hs[0].compressed_page_size == 32);
        CHECK(hs[1].type == PQD_PAGE_DATA && hs[1].num_values == 10 && hs[1].header_offset == second && hs[1].data_offset == data_at);
        CHECK(hs[1].encoding == PQD_ENC_PLAIN_DICTIONARY && hs[1].definition_level_encoding == PQD_ENC_RLE);
        pqd_column_info col = {PQD_KIND_INT64, 0, 1, 0};
        pqd::PagePlan plan = pqd::PlanPages(hs, chunk.data(), chunk.size(), chunk.data(), col);
        CHECK(plan.items.size() == 3);
        CHECK(plan.parts[0].role == PQD_PART_DICTIONARY && plan.items[0].encoding == PQD_ENC_PLAIN && plan.items[0].src_len == 32);
        CHECK(plan.parts[1].role == PQD_PART_DEFINITION_LEVELS && plan.items[1].kind == PQD_KIND_LEVELS && plan.items[1].bit_width == 1);
        CHECK(plan.parts[1].src_offset == data_at + 4 && plan.items[1].src_len == def_enc.size() && plan.items[1].num_values == 10);
        CHECK(plan.parts[2].role == PQD_PART_VALUES && plan.items[2].encoding == PQD_ENC_RLE_DICTIONARY && plan.items[2].num_values == -1);
        CHECK(plan.parts[2].src_offset == data_at + values_at && plan.items[2].src_len == idx_enc.size());
        // the parts decode to what went in
        std::vector<uint8_t> dgot;
        pqd::RLE(1).DecodeLevels(dgot, chunk.data() + plan.parts[1].src_offset, plan.items[1].src_len);
        dgot.resize(10);
        CHECK(dgot == def);
        std::vector<int32_t> igot;
        pqd::RLEDictionary().DecodeInt32(igot, chunk.data() + plan.parts[2].src_offset, plan.items[2].src_len);
        igot.resize(7);
        CHECK(igot == idx);
        // a chunk cut inside the second header
        try {
            pqd::ScanPages(chunk.data(), second + 2, -1);
            CHECK(!"a truncated page header was accepted");
        } catch (const pqd::Error& e) { CHECK(e.cls == pqd::ErrorClass::UnexpectedEOF && e.arg == (int64_t)second); }
    }

    std::printf("pqd_encoding.hpp ok\n");
    return 0;
}
