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

    std::printf("pqd_encoding.hpp ok\n");
    return 0;
}
