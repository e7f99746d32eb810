// encode.hip -- page encoders for gfx950, byte for byte what the reference's writers produce:
//
//   DELTA_BINARY_PACKED int32 / int64   delta.BinaryPackedEncoding.EncodeInt32 / EncodeInt64
//                                       (encoding/delta/binary_packed.go:26-34, 77-181, 183-277):
//                                       blocks of 128 values in 4 miniblocks of 32
//   RLE_DICTIONARY index pages          rle.DictionaryEncoding.EncodeInt32
//                                       (encoding/rle/dictionary.go:23-28, 52-59) over encodeInt32
//                                       (encoding/rle/rle.go:203-249)
//   BYTE_STREAM_SPLIT                   bytestreamsplit.Encoding.Encode{Int32,Int64,Float,Double,
//                                       FixedLenByteArray} (encoding/bytestreamsplit/bytestreamsplit.go:19-33,
//                                       bytestreamsplit_purego.go:7-45, bytestreamsplit_amd64.s:56-208)
//
// and the page statistics of dictionary-encoded columns:
//
//   dictionaryBounds*                   min and max of dict[indexes[i]] (dictionary_purego.go:68-170,
//                                       dictionary_amd64.s:8-575)
//
// The variable-length encoders run in one pass over the values.  A page is cut into tiles; a
// workgroup encodes its tile into LDS, which tells it the tile's size; the place of the tile in
// the page comes from the sizes of the tiles before it, which the workgroups of a page publish
// in a per-tile state word and look back over (aggregate first, inclusive prefix once known);
// then the staged bytes are copied to the page.  The workgroups of a page have consecutive
// block indexes and a workgroup only ever waits for lower ones.
#include <algorithm>
#include <cstring>
#include <vector>

#include "pqd_internal.h"

namespace pqd {
namespace {

constexpr int ET = 256;              // threads per workgroup
constexpr uint32_t DE_BLOCK = 128;   // values per block (binary_packed.go:18-22)
constexpr uint32_t DE_MINI = 32;     // values per miniblock
constexpr uint32_t DE_TILE_BLOCKS = 16; // blocks per workgroup

// One page to encode, as the kernels see it.
struct EncPage {
    const uint8_t* src;
    uint64_t n_values;
    uint8_t* dst;
    uint64_t dst_cap;
    int32_t encoding;
    int32_t kind;
    int32_t width;     // bytes per value
    int32_t family;
    uint64_t tile_base; // first tile of the page in its family's grid
    uint64_t n_tiles;
    // results
    int32_t status;
    int32_t bw;        // RLE_DICTIONARY: OR of all indexes while the page is scanned, then unused
    int64_t arg;
    uint64_t out_bytes;
    // RLE_DICTIONARY scratch
    uint32_t* first_start; // per tile: first group of the tile that starts a run, or ~0
    uint32_t* next_start;  // per tile: first run start after the tile
    uint32_t* or_word;     // OR of all indexes of the page
};

enum EncFamily : int32_t { EF_NONE = 0, EF_DELTA32 = 1, EF_DELTA64 = 2, EF_RLE_DICT = 3, EF_BSS = 4, EF_N = 5 };

constexpr uint64_t ST_AGGREGATE = 1ull << 62, ST_INCLUSIVE = 2ull << 62, ST_FLAGS = 3ull << 62;

__device__ inline void lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

__device__ inline uint32_t bits_len64(uint64_t v) { return v ? 64u - (uint32_t)__builtin_clzll(v) : 0u; }

__device__ inline uint32_t uvarint_len(uint64_t v) { return v ? (bits_len64(v) + 6) / 7 : 1; }

__device__ inline uint64_t zigzag(int64_t v) { return ((uint64_t)v << 1) ^ (uint64_t)(v >> 63); }

// ORs `nbits` (<= 64) bits of v into the staging area at bit position `at`: the area is an array
// of little-endian dwords, cleared beforehand, so fields may be written in any order by any lane.
__device__ inline void stage_or(uint32_t* stage, uint64_t at, uint64_t v, uint32_t nbits) {
    if (nbits == 0) return;
    const uint32_t w = (uint32_t)(at >> 5), s = (uint32_t)(at & 31);
    const unsigned __int128 x = (unsigned __int128)v << s;
    const uint32_t x0 = (uint32_t)x, x1 = (uint32_t)(x >> 32), x2 = (uint32_t)(x >> 64);
    if (x0) atomicOr(&stage[w], x0);
    if (x1) atomicOr(&stage[w + 1], x1);
    if (x2) atomicOr(&stage[w + 2], x2);
}

// Writes a uvarint at byte `at` of the staging area; returns its length.
__device__ inline uint32_t stage_uvarint(uint32_t* stage, uint64_t at, uint64_t v) {
    uint32_t n = 0;
    while (v >= 0x80) {
        stage_or(stage, 8 * (at + n), (v & 0x7f) | 0x80, 8);
        v >>= 7;
        n++;
    }
    stage_or(stage, 8 * (at + n), v, 8);
    return n + 1;
}

// Copies `n` staged bytes to dst + base (any alignment) with dword stores where they fit.
__device__ inline void stage_flush(gptr<uint8_t> dst, uint64_t base, const uint32_t* stage, uint32_t n) {
    gptr<uint8_t> out = dst + base;
    const uint32_t head = std::min<uint32_t>(n, (uint32_t)((4 - ((uintptr_t)out & 3)) & 3));
    const uint8_t* bytes = reinterpret_cast<const uint8_t*>(stage);
    if (threadIdx.x < head) out[threadIdx.x] = bytes[threadIdx.x];
    const uint32_t words = (n - head) / 4;
    const uint32_t sh = 8 * (head & 3);
    const uint32_t w0 = head / 4; // head < 4: 0
    for (uint32_t i = threadIdx.x; i < words; i += ET) {
        const uint32_t lo = stage[w0 + i], hi = stage[w0 + i + 1];
        const uint32_t v = sh ? (lo >> sh) | (hi << (32 - sh)) : lo;
        *(gptr<uint32_t>)(out + head + 4 * i) = v;
    }
    const uint32_t tail0 = head + 4 * words;
    if (threadIdx.x < n - tail0) out[tail0 + threadIdx.x] = bytes[tail0 + threadIdx.x];
}

// The prefix of a tile: publishes the tile's own size, then adds up the tiles of the page before
// it.  Called by the first wavefront of the workgroup; returns the bytes in front of the tile.
// `tile` is the tile's number within its page, `state` the page's state words.
__device__ inline uint64_t tile_prefix(unsigned long long* state, uint64_t tile, uint64_t own, uint32_t lane) {
    if (tile == 0) {
        if (lane == 0) __hip_atomic_store(&state[0], ST_INCLUSIVE | own, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        return 0;
    }
    if (lane == 0) __hip_atomic_store(&state[tile], ST_AGGREGATE | own, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    uint64_t before = 0;
    uint64_t hi = tile; // tiles [.., hi) are still to be added
    while (hi > 0) {
        const uint32_t look = (uint32_t)std::min<uint64_t>(64, hi);
        unsigned long long s = ST_INCLUSIVE; // lanes with no tile to look at do not hold the others up
        if (lane < look) {
            do {
                s = __hip_atomic_load(&state[hi - 1 - lane], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if ((s & ST_FLAGS) == 0) __builtin_amdgcn_s_sleep(2);
            } while ((s & ST_FLAGS) == 0);
        }
        // the nearest tile that already knows its inclusive prefix ends the walk
        const uint64_t incl = __ballot(lane < look && (s & ST_FLAGS) == ST_INCLUSIVE);
        const uint32_t stop = incl ? (uint32_t)__builtin_ctzll(incl) : look - 1;
        uint64_t part = (lane <= stop && lane < look) ? (s & ~ST_FLAGS) : 0;
        for (int o = 32; o > 0; o >>= 1) part += (uint64_t)__shfl_xor((unsigned long long)part, o);
        before += part;
        if (incl) break;
        hi -= look;
    }
    if (lane == 0) __hip_atomic_store(&state[tile], ST_INCLUSIVE | (before + own), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return before;
}

// Finds the page a tile belongs to: prefix[p] <= tile < prefix[p + 1].
__device__ inline uint32_t page_of_tile(const uint64_t* __restrict__ prefix, uint32_t n_pages, uint64_t tile) {
    uint32_t lo = 0, hi = n_pages;
    while (hi - lo > 1) {
        const uint32_t mid = (lo + hi) / 2;
        if (prefix[mid] <= tile) lo = mid; else hi = mid;
    }
    return lo;
}

// The page (its place in the family's list) and the tile of it that this workgroup takes; false when there is none.
// Two grids: (tiles of all pages) with `prefix`, where the page is searched for -- a dozen dependent scalar loads
// and some four hundred scalar instructions per wavefront, half of what a tile of 16 KB costs otherwise --, and
// (most tiles a page has, pages) without, for batches whose pages are about equally long.  Either way the
// workgroups of a page are numbered in the order of its tiles, which is what the look-back needs.
__device__ inline bool page_and_tile(const EncPage* __restrict__ pages, const uint32_t* __restrict__ items,
                                     const uint64_t* __restrict__ prefix, uint32_t n_pages, uint32_t& p, uint64_t& tile) {
    if (prefix != nullptr) {
        p = page_of_tile(prefix, n_pages, blockIdx.x);
        tile = blockIdx.x - prefix[p];
        return true;
    }
    p = blockIdx.y;
    tile = blockIdx.x;
    return tile < pages[items[p]].n_tiles;
}

// ---------------------------------------------------------- DELTA_BINARY_PACKED --

template <typename T> struct DeltaOf;
template <> struct DeltaOf<int32_t> {
    // differences wrap in 32 bits (binary_packed.go:192-204: blockDeltaInt32 on int32)
    __device__ static int64_t diff(int32_t a, int32_t b) { return (int64_t)(int32_t)((uint32_t)a - (uint32_t)b); }
    __device__ static uint64_t rebase(int64_t d, int64_t m) { return (uint32_t)((uint32_t)d - (uint32_t)m); }
};
template <> struct DeltaOf<int64_t> {
    __device__ static int64_t diff(int64_t a, int64_t b) { return (int64_t)((uint64_t)a - (uint64_t)b); }
    __device__ static uint64_t rebase(int64_t d, int64_t m) { return (uint64_t)d - (uint64_t)m; }
};

struct DeltaShared {
    uint32_t size[DE_TILE_BLOCKS];   // encoded bytes of each block of the tile
    uint32_t at[DE_TILE_BLOCKS];     // where each block starts in the staging area
    uint32_t total;                  // bytes of the tile
    uint64_t before;                 // bytes of the page in front of this tile
    // worst case per block: 10 (min delta) + 4 (widths) + 128 * 8; header: 2 + 1 + 10 + 10; two spare dwords
    uint32_t stage[(DE_TILE_BLOCKS * (14 + DE_BLOCK * 8) + 24) / 4 + 4];
};

// Eight neighbouring values of a block from src + k0, or zeros where the page has ended.
template <typename T>
__device__ inline void load_eight(gptr<const T> src, uint64_t k0, uint64_t n, T (&x)[8]) {
    if (k0 + 8 <= n) {
        if (sizeof(T) == 8) {
#pragma unroll
            for (uint32_t i = 0; i < 4; i++) { // a Go []int64 is 8-byte aligned
                const u64x2_a8 q = *(gptr<const u64x2_a8>)(src + k0 + 2 * i);
                x[2 * i] = (T)q.x; x[2 * i + 1] = (T)q.y;
            }
        } else {
#pragma unroll
            for (uint32_t i = 0; i < 2; i++) {
                const u32x4_a4 q = *(gptr<const u32x4_a4>)(src + k0 + 4 * i);
                x[4 * i] = (T)q.x; x[4 * i + 1] = (T)q.y; x[4 * i + 2] = (T)q.z; x[4 * i + 3] = (T)q.w;
            }
        }
    } else {
#pragma unroll
        for (uint32_t i = 0; i < 8; i++) x[i] = k0 + i < n ? src[k0 + i] : (T)0;
    }
}

// stage_or for a field of at most 32 bits: two dwords at most.
__device__ inline void stage_or32(uint32_t* stage, uint32_t at, uint32_t v) {
    const uint32_t w = at >> 5, s = at & 31;
    const uint64_t x = (uint64_t)v << s;
    const uint32_t x0 = (uint32_t)x, x1 = (uint32_t)(x >> 32);
    atomicOr(&stage[w], x0);     // (no test for zero: a branch costs more than the OR of nothing)
    atomicOr(&stage[w + 1], x1);
}

// One workgroup encodes up to 16 blocks (binary_packed.go:105-176): every wavefront takes four
// of them at once, sixteen lanes to a block, a lane eight neighbouring values -- a quarter of a
// miniblock, whose packed form is a whole number of bytes.  (With two values to a lane and the
// blocks one after the other the reductions over the lanes were most of the kernel, which spent
// a hundred vector instructions per value and was bound by issuing them.)
template <typename T>
__global__ __launch_bounds__(ET) void delta_encode_kernel(EncPage* __restrict__ pages, const uint32_t* __restrict__ items,
                                                          const uint64_t* __restrict__ prefix, uint32_t n_pages,
                                                          unsigned long long* __restrict__ state) {
    __shared__ DeltaShared sh;
    uint32_t p;
    uint64_t tile;
    if (!page_and_tile(pages, items, prefix, n_pages, p, tile)) return;
    EncPage& pg = pages[items[p]];
    gptr<const T> src = (gptr<const T>)pg.src;
    const uint64_t n = pg.n_values;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint64_t n_blocks = n < 2 ? 0 : (n - 1 + DE_BLOCK - 1) / DE_BLOCK;
    const uint64_t block0 = tile * DE_TILE_BLOCKS;
    const uint32_t blocks = (uint32_t)std::min<uint64_t>(DE_TILE_BLOCKS, n_blocks - std::min(n_blocks, block0));

    constexpr uint32_t PER_WAVE = DE_TILE_BLOCKS / (ET / 64);
    static_assert(PER_WAVE * 16 == 64 && DE_BLOCK == 16 * 8 && DE_MINI == 4 * 8, "sixteen lanes to a block, four to a miniblock");
    const uint32_t l = lane & 15;                  // lane within its block
    const uint32_t b = wave * PER_WAVE + (lane >> 4); // the lane's block of the tile
    const bool live = b < blocks;
    uint64_t r[8] = {0, 0, 0, 0, 0, 0, 0, 0};      // the lane's differences, less the block's least
    int64_t min_delta = 0;
    uint32_t widths = 0;                           // the four miniblock widths of the block, a byte each
    {
        const uint64_t k0 = 1 + (block0 + b) * DE_BLOCK + 8 * l; // index of the lane's first value
        // values of the page from k0 on, as far as it matters here: value i of the lane exists when i < left, and
        // its difference counts when i <= left (the first zero of the padding is a difference too)
        const uint32_t left = !live || k0 > n ? 0u : (uint32_t)std::min<uint64_t>(n - k0, 16);
        const bool counted = live && k0 <= n;
        T x[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        T prev = 0;
        if (live && n > 0) {
            prev = src[std::min(k0 - 1, n - 1)];
            load_eight<T>(src, k0, n, x);
        }
        // the block is padded with zeros before the differences and their minimum are taken (binary_packed.go:139-146)
        int64_t d[8];
#pragma unroll
        for (uint32_t i = 0; i < 8; i++) d[i] = counted && i <= left ? DeltaOf<T>::diff(x[i], i == 0 ? prev : x[i - 1]) : 0;
        int64_t m = d[0];
#pragma unroll
        for (uint32_t i = 1; i < 8; i++) m = std::min(m, d[i]);
        for (int o = 8; o > 0; o >>= 1) m = std::min(m, (int64_t)__shfl_xor((long long)m, o)); // the sixteen lanes of the block
        uint64_t any = 0;
#pragma unroll
        for (uint32_t i = 0; i < 8; i++) {
            r[i] = i < left ? DeltaOf<T>::rebase(d[i], m) : 0;
            any |= r[i];
        }
        for (int o = 2; o > 0; o >>= 1) any |= __shfl_xor((unsigned long long)any, o);         // the four lanes of the miniblock
        const uint32_t bw = bits_len64(any);
#pragma unroll
        for (uint32_t mb = 0; mb < 4; mb++) widths |= (uint32_t)__shfl((int)bw, (int)((lane & 48) + 4 * mb)) << (8 * mb);
        min_delta = m;
        if (live && l == 0)
            sh.size[b] = uvarint_len(zigzag(m)) + 4 + 4 * ((widths & 0xff) + ((widths >> 8) & 0xff) + ((widths >> 16) & 0xff) + (widths >> 24));
    }
    const T first = n > 0 ? src[0] : (T)0;
    const uint32_t head = tile == 0 ? 2 + 1 + uvarint_len(n) + uvarint_len(zigzag((int64_t)first)) : 0;
    lds_sync();
    // where every block starts in the staging area: a scan over sixteen lanes
    if (wave == 0) {
        const uint32_t mine = lane < blocks ? sh.size[lane] : 0;
        const uint32_t incl = wave_inclusive_scan<uint32_t>(mine);
        if (lane < DE_TILE_BLOCKS) sh.at[lane] = head + incl - mine;
        if (lane == 63) sh.total = head + incl;
    }
    lds_sync();
    const uint32_t total = sh.total;
    // fields are ORed into the staging area: what will be used of it starts out as zeros
    for (uint32_t i = threadIdx.x; i < (total + 3) / 4 + 2; i += ET) sh.stage[i] = 0;
    lds_sync();
    const uint32_t at = sh.at[b]; // where the lane's block starts in the staging area
    if (wave == 0) {
        const uint64_t before = tile_prefix(state + pg.tile_base, tile, total, lane);
        if (lane == 0) sh.before = before;
    }
    if (tile == 0 && threadIdx.x == 0) { // binary_packed.go:93-103
        uint32_t o = 0;
        o += stage_uvarint(sh.stage, o, DE_BLOCK);
        o += stage_uvarint(sh.stage, o, DE_BLOCK / DE_MINI);
        o += stage_uvarint(sh.stage, o, n);
        o += stage_uvarint(sh.stage, o, zigzag((int64_t)first));
    }
    if (live) {
        const uint32_t w = widths;
        uint32_t o = at;
        const uint32_t vl = uvarint_len(zigzag(min_delta));
        if (l == 0) {
            stage_uvarint(sh.stage, o, zigzag(min_delta));
            stage_or(sh.stage, 8ull * (o + vl), w, 32);
        }
        o += vl + 4;
        const uint32_t mb = l >> 2;
        const uint32_t bw = (w >> (8 * mb)) & 0xff;
        for (uint32_t q = 0; q < mb; q++) o += 4 * ((w >> (8 * q)) & 0xff);
        const uint32_t bit = 8 * o + 8 * (l & 3) * bw; // eight values of bw bits: bw bytes per lane
        if (bw <= 32) {
#pragma unroll
            for (uint32_t i = 0; i < 8; i++) stage_or32(sh.stage, bit + i * bw, (uint32_t)r[i]);
        } else {
#pragma unroll
            for (uint32_t i = 0; i < 8; i++) stage_or(sh.stage, bit + i * bw, r[i], bw);
        }
    }
    lds_sync();
    const uint64_t before = sh.before;
    if (before + total <= pg.dst_cap) stage_flush(as_global(pg.dst), before, sh.stage, total);
    if (tile + 1 == pg.n_tiles && threadIdx.x == 0) {
        const uint64_t bytes = before + total;
        pg.out_bytes = bytes <= pg.dst_cap ? bytes : 0;
        pg.status = bytes <= pg.dst_cap ? PQD_OK : PQD_ERR_DST_TOO_SMALL;
        pg.arg = bytes <= pg.dst_cap ? 0 : (int64_t)bytes;
    }
}

// ------------------------------------------------------------ BYTE_STREAM_SPLIT --

// dst[s * count + k] = src[k * W + s] (bytestreamsplit_purego.go:7-45).  A thread takes four
// values and writes one dword to each of the W streams.
template <int W>
__device__ inline void bss_encode_four(gptr<const uint8_t> src, gptr<uint8_t> dst, uint64_t count, uint64_t k) {
    uint32_t in[W];
#pragma unroll
    for (int d = 0; d < W; d++) in[d] = *(gptr<const uint32_t>)(src + k * W + 4 * d);
#pragma unroll
    for (int s = 0; s < W; s++) {
        uint32_t w = 0;
#pragma unroll
        for (int e = 0; e < 4; e++) {
            const int b = e * W + s; // byte of the 4 * W block
            w |= ((in[b / 4] >> (8 * (b % 4))) & 0xffu) << (8 * e);
        }
        gptr<uint8_t> o = dst + (uint64_t)s * count + k;
        if (((uintptr_t)o & 3) == 0) *(gptr<uint32_t>)o = w;
        else { o[0] = (uint8_t)w; o[1] = (uint8_t)(w >> 8); o[2] = (uint8_t)(w >> 16); o[3] = (uint8_t)(w >> 24); }
    }
}

__global__ __launch_bounds__(ET) void bss_encode_kernel(EncPage* __restrict__ pages, const uint32_t* __restrict__ items) {
    EncPage& pg = pages[items[blockIdx.x]];
    const uint64_t width = (uint64_t)pg.width;
    const uint64_t count = pg.n_values, bytes = count * width;
    const bool lead = blockIdx.y == 0 && threadIdx.x == 0;
    if (bytes > pg.dst_cap) {
        if (lead) { pg.status = PQD_ERR_DST_TOO_SMALL; pg.arg = (int64_t)bytes; pg.out_bytes = 0; }
        return;
    }
    gptr<const uint8_t> src = as_global(pg.src);
    gptr<uint8_t> dst = as_global(pg.dst);
    const uint64_t tile = (uint64_t)ET * 4;
    const bool aligned = ((uintptr_t)pg.src & 3) == 0;
    for (uint64_t t = blockIdx.y; t * tile < count; t += gridDim.y) {
        const uint64_t k = t * tile + threadIdx.x * 4;
        if (k >= count) continue;
        if (k + 4 <= count && aligned && width == 4) bss_encode_four<4>(src, dst, count, k);
        else if (k + 4 <= count && aligned && width == 8) bss_encode_four<8>(src, dst, count, k);
        else {
            for (uint64_t e = k; e < k + 4 && e < count; e++)
                for (uint64_t s = 0; s < width; s++) dst[s * count + e] = src[e * width + s];
        }
    }
    if (lead) { pg.status = PQD_OK; pg.arg = 0; pg.out_bytes = bytes; }
}


// ------------------------------------------------------ RLE_DICTIONARY indexes --

constexpr uint32_t RE_GROUPS = ET;            // groups of eight values per workgroup, one per thread
constexpr uint32_t RE_NONE = 0xffffffffu;

// Eight indexes of one group, and what encodeInt32 asks of a group (rle.go:203-249): whether
// its values are all the same, and whether it begins a run.  A run of repeated values goes on
// over whole groups of that one value; a bit-packed run goes on until a group of one value.
struct Group {
    uint32_t v[8];
    bool same;
};

__device__ inline Group load_group(gptr<const uint32_t> src, uint64_t g) {
    Group r;
    const u32x4_a4 a = *(gptr<const u32x4_a4>)(src + 8 * g), b = *(gptr<const u32x4_a4>)(src + 8 * g + 4); // a Go []int32 is 4-byte aligned
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
    r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
    r.same = a.y == a.x && a.z == a.x && a.w == a.x && b.x == a.x && b.y == a.x && b.z == a.x && b.w == a.x;
    return r;
}

__device__ inline bool starts_run(const Group& cur, bool have_prev, const Group& prev) {
    if (!have_prev) return true;
    if (cur.same != prev.same) return true;
    return cur.same && cur.v[0] != prev.v[0];
}

struct RleEncShared {
    unsigned long long starts[ET / 64]; // per wavefront: lanes whose group begins a run
    uint32_t wave_total[ET / 64];
    uint32_t first;
    uint64_t before;
    // worst case per group: a 10-byte run header and 32 bytes of packed values; the width byte, the all-zero
    // page's single header and seven one-value runs at the end; spare dwords for the flush
    uint32_t stage[(RE_GROUPS * 42 + 11 + 7 * 14) / 4 + 4];
};

// Pass one: the OR of all indexes of a page (its bit width) and, per tile, the first group that begins a run.
__global__ __launch_bounds__(ET) void rle_encode_scan_kernel(EncPage* __restrict__ pages, const uint32_t* __restrict__ items,
                                                             const uint64_t* __restrict__ prefix, uint32_t n_pages) {
    __shared__ uint32_t sh_first, sh_or;
    uint32_t p;
    uint64_t tile;
    if (!page_and_tile(pages, items, prefix, n_pages, p, tile)) return;
    EncPage& pg = pages[items[p]];
    gptr<const uint32_t> src = (gptr<const uint32_t>)pg.src;
    const uint64_t n = pg.n_values, groups = n / 8;
    const uint64_t g = tile * RE_GROUPS + threadIdx.x;
    if (threadIdx.x == 0) { sh_first = RE_NONE; sh_or = 0; }
    lds_sync();
    uint32_t any = 0;
    if (g < groups) {
        const Group cur = load_group(src, g);
        Group prev = cur;
        if (g > 0) prev = load_group(src, g - 1);
        if (starts_run(cur, g > 0, prev)) atomicMin(&sh_first, (uint32_t)g);
#pragma unroll
        for (int e = 0; e < 8; e++) any |= cur.v[e];
    }
    if (tile + 1 == pg.n_tiles && threadIdx.x < (n & 7)) any |= src[groups * 8 + threadIdx.x];
    for (int o = 32; o > 0; o >>= 1) any |= __shfl_xor(any, o);
    if ((threadIdx.x & 63) == 0 && any) atomicOr(&sh_or, any);
    lds_sync();
    if (threadIdx.x == 0) {
        pg.first_start[tile] = sh_first;
        if (sh_or) atomicOr(pg.or_word, sh_or);
    }
}

// Pass two, one wavefront per page: for every tile, the first run start in a later tile (or the number of groups).
__global__ __launch_bounds__(64) void rle_encode_next_kernel(EncPage* __restrict__ pages, const uint32_t* __restrict__ items) {
    EncPage& pg = pages[items[blockIdx.x]];
    const uint32_t lane = threadIdx.x;
    const uint64_t groups = pg.n_values / 8;
    uint32_t carry = (uint32_t)groups; // first start after the tiles looked at so far
    for (uint64_t hi = pg.n_tiles; hi > 0; hi -= std::min<uint64_t>(64, hi)) {
        const uint32_t look = (uint32_t)std::min<uint64_t>(64, hi);
        // lane l looks at tile hi - 1 - l: later tiles sit in lower lanes
        const uint64_t t = hi - 1 - lane;
        uint32_t mine = lane < look ? pg.first_start[t] : RE_NONE;
        // exclusive running minimum over the lower lanes, carried in from the tiles behind
        uint32_t incl = mine;
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t up = __shfl_up(incl, o);
            if (lane >= (uint32_t)o) incl = std::min(incl, up);
        }
        uint32_t excl = __shfl_up(incl, 1);
        if (lane == 0) excl = RE_NONE;
        if (lane < look) pg.next_start[t] = std::min(excl, carry);
        carry = std::min(carry, lane_u32(incl, 63));
    }
}

// Pass three: the page bytes.  A thread encodes its group into the staging area: the header of the run that begins
// there, the repeated value of a run of repeats, the eight packed fields of a group of a bit-packed run.
__global__ __launch_bounds__(ET) void rle_encode_write_kernel(EncPage* __restrict__ pages, const uint32_t* __restrict__ items,
                                                              const uint64_t* __restrict__ prefix, uint32_t n_pages,
                                                              unsigned long long* __restrict__ state) {
    __shared__ RleEncShared sh;
    uint32_t p;
    uint64_t tile;
    if (!page_and_tile(pages, items, prefix, n_pages, p, tile)) return;
    EncPage& pg = pages[items[p]];
    gptr<const uint32_t> src = (gptr<const uint32_t>)pg.src;
    const uint64_t n = pg.n_values, groups = n / 8;
    const uint64_t g = tile * RE_GROUPS + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t bw = bits_len64(*pg.or_word);  // rle/dictionary.go:52-59: the widest index
    const uint32_t vb = (bw + 7) / 8;
    const bool last_tile = tile + 1 == pg.n_tiles;

    Group cur;
    bool have = g < groups && bw > 0, start = false;
    if (have) {
        cur = load_group(src, g);
        Group prev = cur;
        if (g > 0) prev = load_group(src, g - 1);
        start = starts_run(cur, g > 0, prev);
    }
    const unsigned long long mask = __ballot(start);
    if (lane == 0) sh.starts[wave] = mask;
    for (uint32_t i = threadIdx.x; i < sizeof(sh.stage) / 4; i += ET) sh.stage[i] = 0;
    lds_sync();
    // where the run that begins here ends: the next start in the tile, or the first one behind it
    uint64_t run_groups = 0;
    if (start) {
        uint64_t next = pg.next_start[tile];
        unsigned long long m = lane == 63 ? 0ull : mask & (~0ull << (lane + 1));
        for (uint32_t w = wave; ; w++) {
            if (m) { next = tile * RE_GROUPS + 64ull * w + (uint32_t)__builtin_ctzll(m); break; }
            if (w + 1 >= ET / 64) break;
            m = sh.starts[w + 1];
        }
        run_groups = next - g;
    }
    // bytes this group adds to the page
    uint32_t hdr = 0, mine = 0;
    if (have) {
        if (start) hdr = uvarint_len(cur.same ? (8 * run_groups) << 1 : (run_groups << 1) | 1);
        mine = hdr + (cur.same ? (start ? vb : 0u) : bw);
    }
    // the values behind the last whole group are runs of repeats of their own (rle.go:236-246); the
    // last tile's first thread encodes them once the groups are placed
    uint32_t incl = wave_inclusive_scan(mine);
    if (lane == 63) sh.wave_total[wave] = incl;
    lds_sync();
    uint32_t off = incl - mine, groups_total = 0;
#pragma unroll
    for (uint32_t w = 0; w < ET / 64; w++) {
        if (w < wave) off += sh.wave_total[w];
        groups_total += sh.wave_total[w];
    }
    // the head of the page: the width byte; a page of zeros is one header and no value (rle.go:204-207)
    uint32_t head = 0;
    if (tile == 0) head = 1 + (bw == 0 ? uvarint_len(n << 1) : 0);
    uint32_t tail_bytes = 0;
    if (last_tile && bw > 0) {
        for (uint64_t i = groups * 8; i < n;) {
            uint64_t j = i + 1;
            while (j < n && src[j] == src[i]) j++;
            tail_bytes += uvarint_len((j - i) << 1) + vb;
            i = j;
        }
    }
    const uint32_t total = head + groups_total + tail_bytes;
    if (wave == 0) {
        const uint64_t before = tile_prefix(state + pg.tile_base, tile, total, lane);
        if (lane == 0) sh.before = before;
    }
    if (tile == 0 && threadIdx.x == 0) {
        stage_or(sh.stage, 0, bw, 8);
        if (bw == 0) stage_uvarint(sh.stage, 1, n << 1);
    }
    if (have) {
        uint32_t o = head + off;
        if (start) o += stage_uvarint(sh.stage, o, cur.same ? (8 * run_groups) << 1 : (run_groups << 1) | 1);
        if (cur.same) {
            if (start) stage_or(sh.stage, 8ull * o, cur.v[0], 8 * vb);
        } else {
#pragma unroll
            for (uint32_t e = 0; e < 8; e++) stage_or(sh.stage, 8ull * o + (uint64_t)e * bw, cur.v[e], bw);
        }
    }
    if (last_tile && bw > 0 && threadIdx.x == 0) {
        uint32_t o = head + groups_total;
        for (uint64_t i = groups * 8; i < n;) {
            uint64_t j = i + 1;
            while (j < n && src[j] == src[i]) j++;
            o += stage_uvarint(sh.stage, o, (j - i) << 1);
            stage_or(sh.stage, 8ull * o, src[i], 8 * vb);
            o += vb;
            i = j;
        }
    }
    lds_sync();
    const uint64_t before = sh.before;
    if (before + total <= pg.dst_cap) stage_flush(as_global(pg.dst), before, sh.stage, total);
    if (last_tile && threadIdx.x == 0) {
        const uint64_t bytes = before + total;
        pg.out_bytes = bytes <= pg.dst_cap ? bytes : 0;
        pg.status = bytes <= pg.dst_cap ? PQD_OK : PQD_ERR_DST_TOO_SMALL;
        pg.arg = bytes <= pg.dst_cap ? 0 : (int64_t)bytes;
    }
}

// ------------------------------------------------------------- dictionaryBounds --

// The smallest and the largest of dict[indexes[i]] with the comparisons of the reference's loop
// (dictionary_purego.go:68-170): `value < min` and `value > max` against the first value, so of
// values that compare equal (+0 and -0) the earliest stays, and a NaN is never taken unless it
// is the very first value, in which case nothing ever replaces it.
template <typename T>
struct Bound {
    T lo, hi;
    uint64_t lo_at, hi_at; // positions of the values; ~0: none yet
};

template <typename T>
__device__ inline void bound_take(Bound<T>& b, T v, uint64_t at) {
    if (v != v) return; // NaN
    if (b.lo_at == ~0ull || v < b.lo || (v == b.lo && at < b.lo_at)) { b.lo = v; b.lo_at = at; }
    if (b.hi_at == ~0ull || v > b.hi || (v == b.hi && at < b.hi_at)) { b.hi = v; b.hi_at = at; }
}

template <typename T>
__device__ inline void bound_merge(Bound<T>& b, const Bound<T>& o) {
    if (o.lo_at != ~0ull && (b.lo_at == ~0ull || o.lo < b.lo || (o.lo == b.lo && o.lo_at < b.lo_at))) { b.lo = o.lo; b.lo_at = o.lo_at; }
    if (o.hi_at != ~0ull && (b.hi_at == ~0ull || o.hi > b.hi || (o.hi == b.hi && o.hi_at < b.hi_at))) { b.hi = o.hi; b.hi_at = o.hi_at; }
}

template <typename T>
__device__ inline Bound<T> bound_reduce_workgroup(Bound<T> b, Bound<T>* scratch) {
    scratch[threadIdx.x] = b;
    __syncthreads();
    for (uint32_t s = ET / 2; s > 0; s >>= 1) {
        if (threadIdx.x < s) {
            Bound<T> mine = scratch[threadIdx.x];
            bound_merge(mine, scratch[threadIdx.x + s]);
            scratch[threadIdx.x] = mine;
        }
        __syncthreads();
    }
    return scratch[0];
}

template <typename T>
__global__ __launch_bounds__(ET) void dict_bounds_kernel(const T* __restrict__ dict, uint64_t dict_len,
                                                         const int32_t* __restrict__ indexes, uint64_t n, Bound<T>* __restrict__ partial,
                                                         unsigned long long* __restrict__ status) {
    __shared__ Bound<T> scratch[ET];
    Bound<T> b;
    b.lo = b.hi = T(); b.lo_at = b.hi_at = ~0ull;
    for (uint64_t i = (uint64_t)blockIdx.x * ET + threadIdx.x; i < n; i += (uint64_t)gridDim.x * ET) {
        const uint32_t j = (uint32_t)indexes[i];
        if ((uint64_t)j >= dict_len) { atomicMin(&status[1], (unsigned long long)i); continue; }
        bound_take(b, dict[j], i);
    }
    b = bound_reduce_workgroup(b, scratch);
    if (threadIdx.x == 0) partial[blockIdx.x] = b;
}

template <typename T>
__global__ __launch_bounds__(ET) void dict_bounds_finish_kernel(const T* __restrict__ dict, uint64_t dict_len,
                                                                const int32_t* __restrict__ indexes, uint64_t n,
                                                                const Bound<T>* __restrict__ partial, uint32_t n_partial, T* __restrict__ out,
                                                                unsigned long long* __restrict__ status) {
    __shared__ Bound<T> scratch[ET];
    Bound<T> b;
    b.lo = b.hi = T(); b.lo_at = b.hi_at = ~0ull;
    for (uint32_t i = threadIdx.x; i < n_partial; i += ET) bound_merge(b, partial[i]);
    b = bound_reduce_workgroup(b, scratch);
    if (threadIdx.x != 0) return;
    if (status[1] != ~0ull) { // errno indexOutOfBounds (dictionary_amd64.s:213-215)
        status[0] = PQD_ERR_INDEX_OUT_OF_BOUNDS;
        out[0] = out[1] = T();
        return;
    }
    status[0] = PQD_OK;
    status[1] = 0;
    T lo = T(), hi = T();
    if (n > 0) {
        const T first = dict[(uint32_t)indexes[0]];
        if (first != first) lo = hi = first; // the comparisons never replace a NaN
        else { lo = b.lo; hi = b.hi; }
    }
    out[0] = lo;
    out[1] = hi;
}

__global__ void dict_bounds_init_kernel(unsigned long long* status) {
    status[0] = 0;
    status[1] = ~0ull;
}

} // namespace

// The scratch an encode call needs, and where its parts lie.
struct EncodePlan {
    std::vector<EncPage> pages;
    std::vector<uint32_t> fam_items[EF_N];
    std::vector<uint64_t> fam_prefix[EF_N]; // per family: first tile of each of its pages, and the total
    uint64_t bss_max_values = 0;
    size_t o_pages = 0, o_items[EF_N] = {}, o_prefix[EF_N] = {}, o_state = 0, state_bytes = 0, upload_end = 0, total = 0;
    size_t o_or = 0, o_first = 0, o_next = 0; // RLE_DICTIONARY: a word per page (inside the cleared block), two per tile
};

static int32_t value_width(int32_t kind, int32_t bit_width) {
    switch (kind) {
        case PQD_KIND_INT32: case PQD_KIND_FLOAT: return 4;
        case PQD_KIND_INT64: case PQD_KIND_DOUBLE: return 8;
        case PQD_KIND_INT96: return 12;
        case PQD_KIND_FIXED_LEN_BYTE_ARRAY: return bit_width;
        default: return 0;
    }
}

static int32_t encode_family(int32_t encoding, int32_t kind, int32_t width) {
    switch (encoding) {
        case PQD_ENC_DELTA_BINARY_PACKED:
            return kind == PQD_KIND_INT32 ? EF_DELTA32 : kind == PQD_KIND_INT64 ? EF_DELTA64 : EF_NONE;
        case PQD_ENC_RLE_DICTIONARY:
            return kind == PQD_KIND_INT32 ? EF_RLE_DICT : EF_NONE;
        case PQD_ENC_BYTE_STREAM_SPLIT:
            if (kind == PQD_KIND_INT32 || kind == PQD_KIND_INT64 || kind == PQD_KIND_FLOAT || kind == PQD_KIND_DOUBLE) return EF_BSS;
            return kind == PQD_KIND_FIXED_LEN_BYTE_ARRAY && width > 0 && width <= 32767 ? EF_BSS : EF_NONE;
        default:
            return EF_NONE;
    }
}

static size_t carve(size_t& off, size_t bytes) {
    const size_t at = off;
    off = (off + bytes + 255) & ~(size_t)255;
    return at;
}

static void plan_encode(const pqd_encode_page* in, uint64_t n, EncodePlan& plan, bool with_pointers) {
    plan.pages.resize(n);
    for (int f = 0; f < EF_N; f++) plan.fam_prefix[f].assign(1, 0);
    for (uint64_t i = 0; i < n; i++) {
        EncPage& e = plan.pages[i];
        std::memset(&e, 0, sizeof e);
        e.src = static_cast<const uint8_t*>(in[i].src);
        e.n_values = in[i].n_values;
        e.dst = static_cast<uint8_t*>(in[i].dst);
        e.dst_cap = in[i].dst_cap;
        e.encoding = in[i].encoding;
        e.kind = in[i].kind;
        e.width = value_width(in[i].kind, in[i].bit_width);
        e.family = encode_family(e.encoding, e.kind, e.width);
        if (with_pointers && ((e.src == nullptr && e.n_values > 0) || (e.dst == nullptr && e.dst_cap > 0))) {
            e.family = EF_NONE;
            e.status = PQD_ERR_INVALID_ARGUMENT;
        } else if (e.family == EF_NONE) {
            e.status = PQD_ERR_NOT_SUPPORTED; // encoding.NotSupported answers these (encoding/notsupported.go:116-160)
        } else if (with_pointers && e.family != EF_BSS && ((uintptr_t)e.src & (uintptr_t)(e.width - 1)) != 0) {
            e.family = EF_NONE;
            e.status = PQD_ERR_INVALID_ARGUMENT; // typed values are aligned to their width, as a Go slice is
            e.arg = (int64_t)((uintptr_t)e.src & (uintptr_t)(e.width - 1));
        } else if (e.family != EF_BSS && e.n_values > 0x7fffffffull) {
            e.family = EF_NONE;
            e.status = PQD_ERR_INVALID_ARGUMENT; // the decoder refuses more values than MaxInt32 (binary_packed.go:447-449)
            e.arg = (int64_t)e.n_values;
        }
        if (e.family == EF_NONE) continue;
        if (e.family == EF_DELTA32 || e.family == EF_DELTA64) {
            const uint64_t blocks = e.n_values < 2 ? 0 : (e.n_values - 1 + DE_BLOCK - 1) / DE_BLOCK;
            e.n_tiles = std::max<uint64_t>(1, (blocks + DE_TILE_BLOCKS - 1) / DE_TILE_BLOCKS);
        } else if (e.family == EF_RLE_DICT) {
            e.n_tiles = std::max<uint64_t>(1, (e.n_values / 8 + RE_GROUPS - 1) / RE_GROUPS);
        } else if (e.family == EF_BSS) {
            plan.bss_max_values = std::max(plan.bss_max_values, e.n_values);
        }
        plan.fam_items[e.family].push_back((uint32_t)i);
        plan.fam_prefix[e.family].push_back(plan.fam_prefix[e.family].back() + e.n_tiles);
    }
    // the state words of all tiled families lie in one block, cleared before every launch
    uint64_t tiles = 0;
    for (int f : {EF_DELTA32, EF_DELTA64, EF_RLE_DICT}) {
        for (size_t k = 0; k < plan.fam_items[f].size(); k++) {
            plan.pages[plan.fam_items[f][k]].tile_base = tiles + plan.fam_prefix[f][k];
        }
        tiles += plan.fam_prefix[f].back();
    }
    size_t off = 0;
    plan.o_pages = carve(off, n * sizeof(EncPage));
    for (int f = 0; f < EF_N; f++) {
        plan.o_items[f] = carve(off, plan.fam_items[f].size() * sizeof(uint32_t));
        plan.o_prefix[f] = carve(off, plan.fam_prefix[f].size() * sizeof(uint64_t));
    }
    plan.upload_end = off;
    const size_t n_rle = plan.fam_items[EF_RLE_DICT].size(), rle_tiles = plan.fam_prefix[EF_RLE_DICT].back();
    plan.state_bytes = tiles * sizeof(unsigned long long) + n_rle * sizeof(uint32_t);
    plan.o_state = carve(off, plan.state_bytes);
    plan.o_or = plan.o_state + tiles * sizeof(unsigned long long);
    plan.o_first = carve(off, rle_tiles * sizeof(uint32_t));
    plan.o_next = carve(off, rle_tiles * sizeof(uint32_t));
    plan.total = off;
}

// Device addresses of the per-page scratch, once the workspace is known.
static void place_scratch(EncodePlan& plan, uint8_t* ws) {
    for (size_t k = 0; k < plan.fam_items[EF_RLE_DICT].size(); k++) {
        EncPage& e = plan.pages[plan.fam_items[EF_RLE_DICT][k]];
        e.or_word = reinterpret_cast<uint32_t*>(ws + plan.o_or) + k;
        e.first_start = reinterpret_cast<uint32_t*>(ws + plan.o_first) + plan.fam_prefix[EF_RLE_DICT][k];
        e.next_start = reinterpret_cast<uint32_t*>(ws + plan.o_next) + plan.fam_prefix[EF_RLE_DICT][k];
    }
}

namespace {

template <typename T>
int run_bounds(const void* dict, uint64_t dict_len, const int32_t* indexes, uint64_t n, void* out, void* ws, unsigned long long* status,
               hipStream_t s) {
    Bound<T>* partial = static_cast<Bound<T>*>(ws);
    const uint32_t grid = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(1, (n + ET * 8 - 1) / (ET * 8)), PQD_BOUNDS_WORKSPACE_BYTES / sizeof(Bound<T>));
    hipLaunchKernelGGL(dict_bounds_init_kernel, dim3(1), dim3(1), 0, s, status);
    hipLaunchKernelGGL(dict_bounds_kernel<T>, dim3(grid), dim3(ET), 0, s, static_cast<const T*>(dict), dict_len, indexes, n, partial, status);
    hipLaunchKernelGGL(dict_bounds_finish_kernel<T>, dim3(1), dim3(ET), 0, s, static_cast<const T*>(dict), dict_len, indexes, n, partial, grid,
                       static_cast<T*>(out), status);
    return hipGetLastError() == hipSuccess ? PQD_OK : PQD_ERR_DEVICE;
}

} // namespace

} // namespace pqd

using namespace pqd;

extern "C" {

uint64_t pqd_encode_bound(int encoding, int kind, int bit_width, uint64_t n_values) {
    const int32_t width = value_width(kind, bit_width);
    switch (encoding) {
        case PQD_ENC_DELTA_BINARY_PACKED: {
            // header, then per block of 128 values a min delta, four widths and at most 128 full-width values
            const uint64_t blocks = n_values < 2 ? 0 : (n_values - 1 + DE_BLOCK - 1) / DE_BLOCK;
            return 24 + blocks * (14 + (uint64_t)DE_BLOCK * (kind == PQD_KIND_INT64 ? 8 : 4));
        }
        case PQD_ENC_RLE_DICTIONARY:
            // the width byte; per group of eight at most a 5-byte run header and 32 bytes of packed values;
            // up to seven one-value runs at the end
            return kind == PQD_KIND_INT32 ? 16 + (n_values / 8) * 37 + 7 * 6 : 0;
        case PQD_ENC_BYTE_STREAM_SPLIT:
            return n_values * (uint64_t)(width > 0 ? width : 0);
        default:
            return 0;
    }
}

uint64_t pqd_encode_workspace_size(const pqd_encode_page* pages, uint64_t n_pages) {
    if (pages == nullptr && n_pages > 0) return 0;
    EncodePlan plan;
    plan_encode(pages, n_pages, plan, false);
    return plan.total + 256;
}

int pqd_encode_batch(pqd_ctx* ctx, pqd_encode_page* pages, uint64_t n_pages, void* workspace_dev, uint64_t workspace_bytes,
                     void* hip_stream) {
    if (ctx == nullptr || (pages == nullptr && n_pages > 0)) return PQD_ERR_INVALID_ARGUMENT;
    DeviceGuard guard(ctx->device);
    if (!guard.ok) return PQD_ERR_DEVICE;
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    EncodePlan plan;
    plan_encode(pages, n_pages, plan, true);
    uint8_t* ws = reinterpret_cast<uint8_t*>(((uintptr_t)workspace_dev + 255) & ~(uintptr_t)255);
    if (workspace_dev == nullptr || (uint64_t)(ws - static_cast<uint8_t*>(workspace_dev)) + plan.total > workspace_bytes)
        return PQD_ERR_DST_TOO_SMALL;
    place_scratch(plan, ws);
    std::vector<uint8_t> image(plan.upload_end, 0);
    std::memcpy(image.data() + plan.o_pages, plan.pages.data(), n_pages * sizeof(EncPage));
    for (int f = 0; f < EF_N; f++) {
        if (!plan.fam_items[f].empty())
            std::memcpy(image.data() + plan.o_items[f], plan.fam_items[f].data(), plan.fam_items[f].size() * sizeof(uint32_t));
        std::memcpy(image.data() + plan.o_prefix[f], plan.fam_prefix[f].data(), plan.fam_prefix[f].size() * sizeof(uint64_t));
    }
    PQD_HIP(hipMemcpyAsync(ws, image.data(), plan.upload_end, hipMemcpyHostToDevice, s));
    if (plan.state_bytes) PQD_HIP(hipMemsetAsync(ws + plan.o_state, 0, plan.state_bytes, s));
    EncPage* d_pages = reinterpret_cast<EncPage*>(ws + plan.o_pages);
    unsigned long long* d_state = reinterpret_cast<unsigned long long*>(ws + plan.o_state);
    auto items_of = [&](int f) { return reinterpret_cast<const uint32_t*>(ws + plan.o_items[f]); };
    auto prefix_of = [&](int f) { return reinterpret_cast<const uint64_t*>(ws + plan.o_prefix[f]); };
    // the grid of a family's tile kernels, and the prefix they search (none for the pages-by-tiles grid: see page_and_tile)
    auto tile_grid = [&](int f, dim3& grid, const uint64_t*& prefix) {
        const std::vector<uint64_t>& pre = plan.fam_prefix[f];
        const uint64_t tiles = pre.back(), np = plan.fam_items[f].size();
        uint64_t most = 0;
        for (size_t k = 0; k + 1 < pre.size(); k++) most = std::max(most, pre[k + 1] - pre[k]);
        if (np <= 65535 && most <= 0x7fffffffull && most * np <= tiles + tiles / 2) { grid = dim3((uint32_t)most, (uint32_t)np); prefix = nullptr; }
        else { grid = dim3((uint32_t)tiles); prefix = prefix_of(f); }
    };
    for (int f : {EF_DELTA32, EF_DELTA64}) {
        const uint64_t tiles = plan.fam_prefix[f].back();
        if (tiles == 0) continue;
        if (tiles > 0x7fffffffull) return PQD_ERR_INVALID_ARGUMENT;
        const uint32_t np = (uint32_t)plan.fam_items[f].size();
        dim3 grid;
        const uint64_t* prefix;
        tile_grid(f, grid, prefix);
        if (f == EF_DELTA32)
            hipLaunchKernelGGL(delta_encode_kernel<int32_t>, grid, dim3(ET), 0, s, d_pages, items_of(f), prefix, np, d_state);
        else
            hipLaunchKernelGGL(delta_encode_kernel<int64_t>, grid, dim3(ET), 0, s, d_pages, items_of(f), prefix, np, d_state);
    }
    if (const uint64_t tiles = plan.fam_prefix[EF_RLE_DICT].back()) {
        if (tiles > 0x7fffffffull) return PQD_ERR_INVALID_ARGUMENT;
        const int f = EF_RLE_DICT;
        const uint32_t np = (uint32_t)plan.fam_items[f].size();
        dim3 grid;
        const uint64_t* prefix;
        tile_grid(f, grid, prefix);
        hipLaunchKernelGGL(rle_encode_scan_kernel, grid, dim3(ET), 0, s, d_pages, items_of(f), prefix, np);
        hipLaunchKernelGGL(rle_encode_next_kernel, dim3(np), dim3(64), 0, s, d_pages, items_of(f));
        hipLaunchKernelGGL(rle_encode_write_kernel, grid, dim3(ET), 0, s, d_pages, items_of(f), prefix, np, d_state);
    }
    if (!plan.fam_items[EF_BSS].empty()) {
        const uint64_t per = (uint64_t)ET * 4;
        const uint32_t gy = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(1, (plan.bss_max_values + per - 1) / per), 1024);
        hipLaunchKernelGGL(bss_encode_kernel, dim3((uint32_t)plan.fam_items[EF_BSS].size(), gy), dim3(ET), 0, s, d_pages, items_of(EF_BSS));
    }
    if (hipGetLastError() != hipSuccess) return PQD_ERR_DEVICE;
    // results
    std::vector<EncPage> out(n_pages);
    if (n_pages) PQD_HIP(hipMemcpyAsync(out.data(), d_pages, n_pages * sizeof(EncPage), hipMemcpyDeviceToHost, s));
    PQD_HIP(hipStreamSynchronize(s));
    for (uint64_t i = 0; i < n_pages; i++) {
        pages[i].status = out[i].status;
        pages[i].status_arg = out[i].arg;
        pages[i].out_bytes = out[i].status == PQD_OK ? out[i].out_bytes : 0;
    }
    return PQD_OK;
}

int pqd_dict_bounds_device(pqd_ctx* ctx, int32_t kind, int32_t is_unsigned, const void* dict_dev, uint64_t dict_len,
                           const int32_t* indexes_dev, uint64_t n, void* bounds_dev, void* workspace_dev, uint64_t workspace_bytes,
                           int64_t* status_dev, void* hip_stream) {
    if (ctx == nullptr || bounds_dev == nullptr || status_dev == nullptr || workspace_dev == nullptr ||
        workspace_bytes < PQD_BOUNDS_WORKSPACE_BYTES || ((uintptr_t)workspace_dev & 15) != 0 || (indexes_dev == nullptr && n > 0))
        return PQD_ERR_INVALID_ARGUMENT;
    DeviceGuard guard(ctx->device);
    if (!guard.ok) return PQD_ERR_DEVICE;
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    unsigned long long* st = reinterpret_cast<unsigned long long*>(status_dev);
    switch (kind) {
        case PQD_KIND_INT32:
            return is_unsigned ? run_bounds<uint32_t>(dict_dev, dict_len, indexes_dev, n, bounds_dev, workspace_dev, st, s)
                               : run_bounds<int32_t>(dict_dev, dict_len, indexes_dev, n, bounds_dev, workspace_dev, st, s);
        case PQD_KIND_INT64:
            return is_unsigned ? run_bounds<uint64_t>(dict_dev, dict_len, indexes_dev, n, bounds_dev, workspace_dev, st, s)
                               : run_bounds<int64_t>(dict_dev, dict_len, indexes_dev, n, bounds_dev, workspace_dev, st, s);
        case PQD_KIND_FLOAT:
            return run_bounds<float>(dict_dev, dict_len, indexes_dev, n, bounds_dev, workspace_dev, st, s);
        case PQD_KIND_DOUBLE:
            return run_bounds<double>(dict_dev, dict_len, indexes_dev, n, bounds_dev, workspace_dev, st, s);
        default:
            return PQD_ERR_NOT_SUPPORTED;
    }
}

} // extern "C"
