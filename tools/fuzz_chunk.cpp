// Host-side fuzz of the page-header reader: column chunks from files, damaged at random, through
// pqd_chunk_scan_pages and pqd_chunk_plan_pages.  Meant to be built with the address and
// undefined-behaviour sanitizers (see tools/fuzz_chunk.sh); needs no GPU.
//   fuzz_chunk <rounds> <seed> <chunk file>...
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "../include/pqd.h"

static uint64_t state;
static uint64_t next_u64() { // splitmix64
    uint64_t z = (state += 0x9e3779b97f4a7c15ull);
    z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
    z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
    return z ^ (z >> 31);
}

int main(int argc, char** argv) {
    if (argc < 4) return 2;
    const long rounds = atol(argv[1]);
    state = strtoull(argv[2], nullptr, 10);
    unsigned long long calls = 0, ok = 0, planned = 0;
    for (int f = 3; f < argc; f++) {
        FILE* fp = fopen(argv[f], "rb");
        if (!fp) continue;
        std::vector<uint8_t> whole;
        uint8_t buf[65536];
        size_t n;
        while ((n = fread(buf, 1, sizeof buf, fp)) > 0) whole.insert(whole.end(), buf, buf + n);
        fclose(fp);
        for (long r = 0; r < rounds; r++) {
            // an exact-size copy on the heap, so that a read past the end is seen
            std::vector<uint8_t> c(whole);
            const uint64_t how = r == 0 ? 0 : next_u64() % 5;
            if (how == 1 && !c.empty()) c.resize(next_u64() % c.size());
            if (how == 2 && !c.empty()) c[next_u64() % (c.size() < 64 ? c.size() : 64)] = (uint8_t)next_u64();
            if (how == 3 && !c.empty())
                for (int k = 0; k < 4; k++) c[next_u64() % c.size()] ^= (uint8_t)(1u << (next_u64() % 8));
            if (how == 4)
                for (int k = 0; k < 24; k++) c.push_back((uint8_t)next_u64());
            uint8_t* exact = (uint8_t*)malloc(c.size() ? c.size() : 1);
            memcpy(exact, c.data(), c.size());
            std::vector<pqd_page_header> h(64);
            uint64_t n_pages = 0;
            int64_t where = 0;
            int rc = pqd_chunk_scan_pages(exact, c.size(), -1, h.data(), h.size(), &n_pages, &where);
            if (rc == PQD_ERR_DST_TOO_SMALL && n_pages < (1u << 20)) {
                h.resize(n_pages);
                rc = pqd_chunk_scan_pages(exact, c.size(), -1, h.data(), h.size(), &n_pages, &where);
            }
            calls++;
            if (rc == PQD_OK) {
                ok++;
                for (int kind = PQD_KIND_BOOLEAN; kind <= PQD_KIND_FIXED_LEN_BYTE_ARRAY; kind++) {
                    pqd_column_info col;
                    memset(&col, 0, sizeof col);
                    col.kind = kind;
                    col.type_length = kind == PQD_KIND_FIXED_LEN_BYTE_ARRAY ? 12 : 0;
                    col.max_definition_level = (int32_t)(next_u64() % 4);
                    col.max_repetition_level = (int32_t)(next_u64() % 3);
                    std::vector<pqd_page> items(3 * n_pages + 1);
                    std::vector<pqd_page_part> parts(3 * n_pages + 1);
                    uint64_t n_items = 0;
                    const int rc2 = pqd_chunk_plan_pages(h.data(), n_pages, exact, c.size(), (const uint8_t*)0x100000, &col, items.data(),
                                                         parts.data(), items.size(), &n_items, &where);
                    if (rc2 == PQD_OK) {
                        planned++;
                        // every planned part lies inside the chunk
                        for (uint64_t i = 0; i < n_items; i++) {
                            const uint64_t off = (uint64_t)((const uint8_t*)items[i].src - (const uint8_t*)0x100000);
                            if (off > c.size() || items[i].src_len > c.size() - off) {
                                printf("part %llu of %s round %ld lies outside the chunk\n", (unsigned long long)i, argv[f], r);
                                return 1;
                            }
                        }
                    }
                }
            }
            free(exact);
        }
    }
    printf("fuzz_chunk: %llu scans, %llu whole, %llu plans accepted, no fault\n", calls, ok, planned);
    return 0;
}
