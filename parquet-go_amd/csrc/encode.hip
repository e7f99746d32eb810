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
typedef uint64_t u64_a1e __attribute__((aligned(1)));
typedef uint32_t u32_a1e __attribute__((aligned(1)));
typedef uint32_t u32x4_a1e __attribute__((ext_vector_type(4), aligned(1)));
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
    // RLE levels scratch (first_start / next_start / or_word as above)
    uint32_t* first_start_b; // per tile: first run start if the tile is entered inside a bit-packed run (first_start: inside repeats)
    uint8_t* tile_fn;        // per tile: what the tile does to the run state (LV_B, LV_R, LV_I), then the state it is entered in
    uint32_t fixed_bw;       // RLE with the width given (levels, RLE int32): the width; RLE_DICTIONARY: ~0
    // byte arrays
    const uint32_t* offsets; // n_values + 1 offsets into src; null: FIXED_LEN values of `width` bytes (or no values at all)
    uint64_t src_len;        // bytes at src
    int32_t* len_a;          // scratch, n_values: DELTA_LENGTH lengths / DELTA_BYTE_ARRAY prefix lengths
    int32_t* len_b;          // scratch, n_values: DELTA_BYTE_ARRAY suffix lengths, then where each suffix starts in the output
    uint32_t chain_a, chain_b; // the internal DELTA int32 items that encode len_a / len_b (positions in the page array)
};

enum EncFamily : int32_t { EF_NONE = 0, EF_DELTA32 = 1, EF_DELTA64 = 2, EF_RLE_DICT = 3, EF_BSS = 4, EF_DELTA32_B = 5, EF_LEVELS = 6, EF_BOOL = 7,
                           EF_PLAIN_BA = 8, EF_DLBA = 9, EF_DBA = 10, EF_N = 11 };

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
    const bool dict_form = pg.fixed_bw == 0xffffffffu;
    const uint32_t bw = dict_form ? bits_len64(*pg.or_word) : pg.fixed_bw;  // rle/dictionary.go:52-59: the widest index
    const uint32_t vb = (bw + 7) / 8;
    const bool last_tile = tile + 1 == pg.n_tiles;
    if (!dict_form && bw == 0 && *pg.or_word != 0) { // rle.go:207-212: width 0 takes nothing but zeros
        if (last_tile && threadIdx.x == 0) { pg.status = PQD_ERR_INVALID_BIT_WIDTH; pg.arg = 0; pg.out_bytes = 0; }
        return;
    }

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
    const uint32_t wb = dict_form ? 1u : 0u; // the dictionary form starts with its width byte
    if (tile == 0) head = wb + (bw == 0 ? uvarint_len(n << 1) : 0);
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
        if (dict_form) stage_or(sh.stage, 0, bw, 8);
        if (bw == 0) stage_uvarint(sh.stage, wb, n << 1);
    }
    if (have) {
        uint32_t o = head + off;
        if (start) o += stage_uvarint(sh.stage, o, cur.same ? (8 * run_groups) << 1 : (run_groups << 1) | 1);
        // (with the width given, a value may be wider: packed fields keep its low bits, a repeated value its low bytes)
        const uint32_t fmask = bw >= 32 ? 0xffffffffu : (1u << bw) - 1, vmask = vb >= 4 ? 0xffffffffu : (1u << (8 * vb)) - 1;
        if (cur.same) {
            if (start) stage_or(sh.stage, 8ull * o, cur.v[0] & vmask, 8 * vb);
        } else {
#pragma unroll
            for (uint32_t e = 0; e < 8; e++) stage_or(sh.stage, 8ull * o + (uint64_t)e * bw, cur.v[e] & fmask, bw);
        }
    }
    if (last_tile && bw > 0 && threadIdx.x == 0) {
        uint32_t o = head + groups_total;
        for (uint64_t i = groups * 8; i < n;) {
            uint64_t j = i + 1;
            while (j < n && src[j] == src[i]) j++;
            o += stage_uvarint(sh.stage, o, (j - i) << 1);
            stage_or(sh.stage, 8ull * o, src[i] & (vb >= 4 ? 0xffffffffu : (1u << (8 * vb)) - 1), 8 * vb);
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

// ------------------------------------------------------------------- RLE levels --

// encodeBytes (rle.go:141-201) takes the levels eight at a time, as 64-bit words.  A word all of
// whose bytes are the same begins (or continues) a run of repeats; but a bit-packed run only ends
// at a word that equals the broadcast of the LAST BYTE-0 of the word before it (rle.go:176-178), so
// a uniform word inside a bit-packed run is swallowed unless the word before it starts with the
// same byte.  Where runs begin therefore depends on whether the word before lies in a run of repeats
// (R) or in a bit-packed run (B) -- a two-state machine whose per-word transition is one of three
// functions: "R whatever came before" (the word equals the broadcast of its predecessor's first
// byte), "B whatever came before" (it is not uniform), or "unchanged" (uniform, but another value:
// after repeats it begins new repeats, inside a bit-packed run it is just more packed data).  The
// state in front of a word is the value of the nearest word before it that is not "unchanged":
// a ballot and a count of leading zeros inside a wavefront, a word per tile across tiles.
enum : uint32_t { LV_B = 0, LV_R = 1, LV_I = 2 };
constexpr uint32_t LV_WORDS = ET; // words (of eight levels) per workgroup, one per thread

__device__ inline uint64_t lv_broadcast(uint64_t w) { return (w & 0xff) * 0x0101010101010101ull; }

__device__ inline uint32_t lv_function(uint64_t w, uint64_t prev, bool have_prev) {
    if (have_prev && w == lv_broadcast(prev)) return LV_R;
    if (w != lv_broadcast(w)) return LV_B;
    return have_prev ? LV_I : LV_R;
}

// The state in front of this lane's word: the nearest lower lane whose function is not LV_I decides, else `entered`.
__device__ inline uint32_t lv_state_before(uint32_t fn, uint32_t lane, uint32_t entered) {
    const unsigned long long set = __ballot(fn != LV_I), reps = __ballot(fn == LV_R);
    const unsigned long long below = set & ((1ull << lane) - 1);
    if (!below) return entered;
    return (uint32_t)((reps >> (63 - __builtin_clzll(below))) & 1);
}
// What the 64 words of the wavefront do to the state together.
__device__ inline uint32_t lv_wave_function(uint32_t fn) {
    const unsigned long long set = __ballot(fn != LV_I), reps = __ballot(fn == LV_R);
    if (!set) return LV_I;
    return (uint32_t)((reps >> (63 - __builtin_clzll(set))) & 1);
}

__device__ inline uint64_t lv_load(gptr<const uint8_t> src, uint64_t w) { return *(gptr<const u64_a1e>)(src + 8 * w); }

// Does a run begin at this word, entered in `before`?  And is it a run of repeats?
__device__ inline bool lv_starts(uint32_t fn, uint32_t before, bool first_word) { return first_word || ((before == LV_R) != (fn == LV_R)); }

struct LvShared {
    uint32_t wave_fn[ET / 64];
    unsigned long long starts[ET / 64];
    uint32_t wave_total[ET / 64];
    uint32_t first_r, first_b, any;
    uint64_t before;
    // per word at most a 5-byte run header and 8 packed bytes; a page of zeros' single header; seven one-value runs at the end
    uint32_t stage[(LV_WORDS * 13 + 10 + 7 * 6) / 4 + 4];
};

// The state in front of every word of the tile, given the state the tile is entered in.
__device__ inline uint32_t lv_tile_state_before(LvShared& sh, uint32_t fn, uint32_t entered) {
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t wf = lv_wave_function(fn);
    if (lane == 0) sh.wave_fn[wave] = wf;
    lds_sync();
    uint32_t in = entered;
    for (uint32_t w = 0; w < wave; w++) in = sh.wave_fn[w] == LV_I ? in : sh.wave_fn[w];
    return lv_state_before(fn, lane, in);
}

// Pass one: per tile what it does to the state, the first run start for either state it may be entered in, and the OR of the levels.
__global__ __launch_bounds__(ET) void levels_encode_scan_kernel(EncPage* __restrict__ pages, const uint32_t* __restrict__ items,
                                                                const uint64_t* __restrict__ prefix, uint32_t n_pages) {
    __shared__ LvShared sh;
    uint32_t p;
    uint64_t tile;
    if (!page_and_tile(pages, items, prefix, n_pages, p, tile)) return;
    EncPage& pg = pages[items[p]];
    gptr<const uint8_t> src = as_global(pg.src);
    const uint64_t n = pg.n_values, words = n / 8;
    const uint64_t g = tile * LV_WORDS + threadIdx.x;
    if (threadIdx.x == 0) { sh.first_r = RE_NONE; sh.first_b = RE_NONE; sh.any = 0; }
    uint32_t fn = LV_I;
    uint64_t w = 0;
    if (g < words) {
        w = lv_load(src, g);
        fn = lv_function(w, g > 0 ? lv_load(src, g - 1) : 0, g > 0);
    }
    uint32_t any = (uint32_t)(w | (w >> 32));
    if (tile + 1 == pg.n_tiles && threadIdx.x < (n & 7)) any |= src[words * 8 + threadIdx.x];
    for (int o = 32; o > 0; o >>= 1) any |= __shfl_xor(any, o);
    // entered in R and entered in B (the first word of the page begins a run either way)
    const uint32_t before_r = lv_tile_state_before(sh, fn, LV_R);
    lds_sync();
    const uint32_t before_b = lv_tile_state_before(sh, fn, LV_B);
    if (g < words) {
        if (lv_starts(fn, before_r, g == 0)) atomicMin(&sh.first_r, (uint32_t)g);
        if (lv_starts(fn, before_b, g == 0)) atomicMin(&sh.first_b, (uint32_t)g);
    }
    if ((threadIdx.x & 63) == 0 && any) atomicOr(&sh.any, any);
    lds_sync();
    if (threadIdx.x == 0) {
        uint32_t tf = LV_I;
        for (uint32_t k = 0; k < ET / 64; k++) tf = sh.wave_fn[k] == LV_I ? tf : sh.wave_fn[k];
        pg.tile_fn[tile] = (uint8_t)tf;
        pg.first_start[tile] = sh.first_r;
        pg.first_start_b[tile] = sh.first_b;
        if (sh.any) atomicOr(pg.or_word, sh.any);
    }
}

// Pass two, one wavefront per page: forwards, the state every tile is entered in (which picks its first run
// start); backwards, the first run start behind every tile.
__global__ __launch_bounds__(64) void levels_encode_next_kernel(EncPage* __restrict__ pages, const uint32_t* __restrict__ items) {
    EncPage& pg = pages[items[blockIdx.x]];
    const uint32_t lane = threadIdx.x;
    const uint64_t words = pg.n_values / 8;
    uint32_t entered = LV_B; // (the first word of a page begins a run whatever this says)
    for (uint64_t lo = 0; lo < pg.n_tiles; lo += 64) {
        const uint64_t t = lo + lane;
        const uint32_t fn = t < pg.n_tiles ? pg.tile_fn[t] : LV_I;
        const uint32_t before = lv_state_before(fn, lane, entered);
        const uint32_t all = lv_wave_function(fn);
        if (t < pg.n_tiles) {
            pg.tile_fn[t] = (uint8_t)before;
            if (before == LV_B) pg.first_start[t] = pg.first_start_b[t];
        }
        entered = all == LV_I ? entered : all;
    }
    __threadfence(); // other lanes read below what these lanes wrote
    uint32_t carry = (uint32_t)words;
    for (uint64_t hi = pg.n_tiles; hi > 0; hi -= std::min<uint64_t>(64, hi)) {
        const uint32_t look = (uint32_t)std::min<uint64_t>(64, hi);
        const uint64_t t = hi - 1 - lane;
        uint32_t mine = lane < look ? pg.first_start[t] : RE_NONE;
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

// Pass three: the page bytes (appendRunLengthBytes / appendBitPackedBytes, rle.go:470-484, 532-548).
__global__ __launch_bounds__(ET) void levels_encode_write_kernel(EncPage* __restrict__ pages, const uint32_t* __restrict__ items,
                                                                 const uint64_t* __restrict__ prefix, uint32_t n_pages,
                                                                 unsigned long long* __restrict__ state) {
    __shared__ LvShared sh;
    uint32_t p;
    uint64_t tile;
    if (!page_and_tile(pages, items, prefix, n_pages, p, tile)) return;
    EncPage& pg = pages[items[p]];
    gptr<const uint8_t> src = as_global(pg.src);
    const uint64_t n = pg.n_values, words = n / 8;
    const uint64_t g = tile * LV_WORDS + threadIdx.x;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t bw = pg.fixed_bw;
    const bool last_tile = tile + 1 == pg.n_tiles;
    if (bw > 8 || (bw == 0 && *pg.or_word != 0)) { // rle.go:142-149 (every tile of the page leaves here: nobody waits for another)
        if (last_tile && threadIdx.x == 0) { pg.status = PQD_ERR_INVALID_BIT_WIDTH; pg.arg = bw; pg.out_bytes = 0; }
        return;
    }
    const bool have = g < words && bw > 0;
    uint32_t fn = LV_I;
    uint64_t w = 0;
    if (have) {
        w = lv_load(src, g);
        fn = lv_function(w, g > 0 ? lv_load(src, g - 1) : 0, g > 0);
    }
    const uint32_t before = lv_tile_state_before(sh, fn, pg.tile_fn[tile]);
    const bool start = have && lv_starts(fn, before, g == 0);
    const bool repeats = fn != LV_B; // the kind of the run that begins here, if one does
    const bool in_repeats = start ? repeats : before == LV_R; // the kind of the run this word lies in
    const unsigned long long mask = __ballot(start);
    if (lane == 0) sh.starts[wave] = mask;
    for (uint32_t i = threadIdx.x; i < sizeof(sh.stage) / 4; i += ET) sh.stage[i] = 0;
    lds_sync();
    uint64_t run_words = 0;
    if (start) {
        uint64_t next = pg.next_start[tile];
        unsigned long long m = lane == 63 ? 0ull : mask & (~0ull << (lane + 1));
        for (uint32_t k = wave; ; k++) {
            if (m) { next = tile * LV_WORDS + 64ull * k + (uint32_t)__builtin_ctzll(m); break; }
            if (k + 1 >= ET / 64) break;
            m = sh.starts[k + 1];
        }
        run_words = next - g;
    }
    uint32_t hdr = 0, mine = 0;
    if (have) {
        if (start) hdr = uvarint_len(repeats ? (8 * run_words) << 1 : (run_words << 1) | 1);
        mine = hdr + (in_repeats ? (start ? 1u : 0u) : bw);
    }
    const uint32_t incl = wave_inclusive_scan(mine);
    if (lane == 63) sh.wave_total[wave] = incl;
    lds_sync();
    uint32_t off = incl - mine, words_total = 0;
#pragma unroll
    for (uint32_t k = 0; k < ET / 64; k++) {
        if (k < wave) off += sh.wave_total[k];
        words_total += sh.wave_total[k];
    }
    const uint32_t head = tile == 0 && bw == 0 ? uvarint_len(n << 1) : 0; // a page of zeros at width 0: one header, no value
    uint32_t tail_bytes = 0;
    if (last_tile && bw > 0) {
        for (uint64_t i = words * 8; i < n;) { // the levels behind the last whole word: runs of repeats (rle.go:188-198)
            uint64_t j = i + 1;
            while (j < n && src[j] == src[i]) j++;
            tail_bytes += uvarint_len((j - i) << 1) + 1;
            i = j;
        }
    }
    const uint32_t total = head + words_total + tail_bytes;
    if (wave == 0) {
        const uint64_t bytes_before = tile_prefix(state + pg.tile_base, tile, total, lane);
        if (lane == 0) sh.before = bytes_before;
    }
    if (tile == 0 && bw == 0 && threadIdx.x == 0) stage_uvarint(sh.stage, 0, n << 1);
    if (have) {
        uint32_t o = head + off;
        if (start) o += stage_uvarint(sh.stage, o, repeats ? (8 * run_words) << 1 : (run_words << 1) | 1);
        if (in_repeats) {
            if (start) stage_or(sh.stage, 8ull * o, w & 0xff, 8);
        } else { // encodeBytesBitpackDefault (rle.go:532-548): the low bw bits of every byte
            const uint64_t fm = (1ull << bw) - 1;
            uint64_t packed = 0;
#pragma unroll
            for (uint32_t e = 0; e < 8; e++) packed |= ((w >> (8 * e)) & fm) << (e * bw);
            stage_or(sh.stage, 8ull * o, packed, 8 * bw);
        }
    }
    if (last_tile && bw > 0 && threadIdx.x == 0) {
        uint32_t o = head + words_total;
        for (uint64_t i = words * 8; i < n;) {
            uint64_t j = i + 1;
            while (j < n && src[j] == src[i]) j++;
            o += stage_uvarint(sh.stage, o, (j - i) << 1);
            stage_or(sh.stage, 8ull * o, src[i], 8);
            o += 1;
            i = j;
        }
    }
    lds_sync();
    const uint64_t bytes_before = sh.before;
    if (bytes_before + total <= pg.dst_cap) stage_flush(as_global(pg.dst), bytes_before, sh.stage, total);
    if (last_tile && threadIdx.x == 0) {
        const uint64_t bytes = bytes_before + total;
        pg.out_bytes = bytes <= pg.dst_cap ? bytes : 0;
        pg.status = bytes <= pg.dst_cap ? PQD_OK : PQD_ERR_DST_TOO_SMALL;
        pg.arg = bytes <= pg.dst_cap ? 0 : (int64_t)bytes;
    }
}

// ----------------------------------------------------------------- RLE booleans --

// encodeBits (rle.go:97-139) over the packed bits, behind the 4-byte length EncodeBoolean puts in front
// (rle.go:48-56).  The reference walks the bytes: at a byte of zeros or ones that repeats it writes a run of
// repeats; anywhere else a bit-packed run that goes on until two equal bytes follow one another (0xff never
// stops it, rle.go:125) and then gives the first of the two back.  Where its runs start is nevertheless a local
// matter.  In terms of the stretches of equal bytes: a stretch of two or more zero bytes is always a run of
// repeats (whatever run comes before it stops at its first byte); a stretch of two or more 0xff bytes is one
// only if a run starts at its first byte, which is the case at the start of the page and right behind a run of
// repeated zeros, and nowhere else (a bit-packed run swallows it); every byte of a stretch of two or more other
// bytes starts a run (bit-packed runs of one byte, the last one going on); and a run starts behind every run of
// repeats.  So: flags per byte from its neighbours, "this stretch is a run of repeats" handed along each stretch,
// the starts counted and listed, the runs sized and placed by a prefix sum, the packed bytes copied.  A
// workgroup takes a page, 4 KB at a time, three times over (list, place, copy): boolean pages are small, and a
// page a workgroup is enough to keep many of them busy.
constexpr uint32_t BE_T = 256, BE_PER = 16, BE_CHUNK = BE_T * BE_PER;

struct BeShared {
    uint32_t wave_starts[BE_T / 64];
    uint32_t wave_has[BE_T / 64], wave_g[BE_T / 64];
    unsigned long long wave_bytes[BE_T / 64];
};

// The starts among the sixteen bytes [i0, i0 + 16) of the page (bit j: byte i0 + j) and which of them start runs of
// repeats; g_carry: whether the stretch that holds the last byte before this chunk is a run of repeats (in and out).
__device__ inline void be_flags(BeShared& sh, gptr<const uint8_t> src, uint32_t n, uint32_t c0, uint32_t& starts, uint32_t& repeats,
                                uint32_t& g_carry) {
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t i0 = c0 + tid * BE_PER;
    uint32_t w[5] = {0, 0, 0, 0, 0}; // bytes i0 - 2 .. i0 + 17 from bit 16 of w[0] on
    uint8_t by[BE_PER + 3];
#pragma unroll
    for (uint32_t j = 0; j < BE_PER + 3; j++) {
        const int64_t at = (int64_t)i0 + j - 2;
        by[j] = at >= 0 && at < (int64_t)n ? src[at] : 0;
    }
    (void)w;
    uint32_t run_start = 0, g0 = 0, vrun = 0, valid = 0;
#pragma unroll
    for (uint32_t j = 0; j < BE_PER; j++) {
        const uint32_t i = i0 + j;
        if (i >= n) break;
        const uint8_t c = by[j + 2];
        const bool eqp = i > 0 && by[j + 1] == c, eqn = i + 1 < n && by[j + 3] == c;
        const bool zz_before = i >= 2 && by[j + 1] == 0 && by[j] == 0;
        const bool rs = !eqp;
        const bool rep = rs && eqn && (c == 0 || (c == 0xff && (i == 0 || zz_before)));
        valid |= 1u << j;
        if (rs) run_start |= 1u << j;
        if (rep || (n == 1 && (c == 0 || c == 0xff))) g0 |= 1u << j; // (a page of one byte of zeros or ones: rle.go:98-104)
        if (c != 0 && c != 0xff && (eqp || eqn)) vrun |= 1u << j;
    }
    // ---- is the stretch a byte lies in a run of repeats: its first byte says, the others hear it from the byte before ----
    const bool has = run_start != 0;
    const bool last_g = has ? ((g0 >> (31 - __builtin_clz(run_start))) & 1) != 0 : false; // as far as this thread knows
    const unsigned long long hs = __ballot(has), lg = __ballot(last_g);
    if (lane == 0) { sh.wave_has[wave] = hs != 0; sh.wave_g[wave] = hs ? (uint32_t)((lg >> (63 - __builtin_clzll(hs))) & 1) : 0u; }
    __syncthreads();
    uint32_t g_in = g_carry; // of the byte before this thread's first
    for (uint32_t wv = 0; wv < wave; wv++) if (sh.wave_has[wv]) g_in = sh.wave_g[wv];
    {
        const unsigned long long below = hs & ((1ull << lane) - 1);
        if (below) g_in = (uint32_t)((lg >> (63 - __builtin_clzll(below))) & 1);
    }
    uint32_t g_out = g_carry;
    for (uint32_t wv = 0; wv < BE_T / 64; wv++) if (sh.wave_has[wv]) g_out = sh.wave_g[wv];
    g_carry = g_out;
    uint32_t after = 0, g = g_in;
#pragma unroll
    for (uint32_t j = 0; j < BE_PER; j++) {
        if (!((valid >> j) & 1)) break;
        if ((run_start >> j) & 1) { if (g) after |= 1u << j; g = (g0 >> j) & 1; }
    }
    starts = (g0 | after | vrun | (i0 == 0 ? 1u : 0u)) & valid;
    repeats = g0 & valid;
    __syncthreads(); // (sh.wave_* are written again by the next call)
}

__device__ inline uint32_t be_uvarint_len(uint64_t v) { uint32_t k = 1; while (v >= 0x80) { v >>= 7; k++; } return k; }

__global__ __launch_bounds__(BE_T) void bits_encode_kernel(EncPage* __restrict__ pages, const uint32_t* __restrict__ items) {
    __shared__ BeShared sh;
    EncPage& pg = pages[items[blockIdx.x]];
    gptr<const uint8_t> src = as_global(pg.src);
    gptr<uint8_t> dst = as_global(pg.dst);
    gptr<uint32_t> pos = (gptr<uint32_t>)as_global(pg.len_a);   // where run k starts (bit 31: a run of repeats), n + 1 entries
    gptr<uint32_t> shift = (gptr<uint32_t>)as_global(pg.len_b); // bit-packed run k: its bytes go to body position i + shift[k]
    const uint32_t n = (uint32_t)pg.n_values, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint64_t cap = pg.dst_cap;
    if (n == 0) { // rle.go:98-104: the header of an empty run
        if (tid == 0) {
            const bool ok = cap >= 5;
            if (ok) { dst[0] = 1; dst[1] = 0; dst[2] = 0; dst[3] = 0; dst[4] = 0; }
            pg.status = ok ? PQD_OK : PQD_ERR_DST_TOO_SMALL; pg.arg = ok ? 0 : 5; pg.out_bytes = ok ? 5 : 0;
        }
        return;
    }
    // ---- the runs' starts, listed ----
    uint32_t m = 0, g_carry = 0;
    for (uint32_t c0 = 0; c0 < n; c0 += BE_CHUNK) {
        uint32_t starts, repeats;
        be_flags(sh, src, n, c0, starts, repeats, g_carry);
        const uint32_t mine = (uint32_t)__builtin_popcount(starts);
        const uint32_t incl = wave_inclusive_scan<uint32_t>(mine);
        if (lane == 63) sh.wave_starts[wave] = incl;
        __syncthreads();
        uint32_t k = m + incl - mine, all = 0;
        for (uint32_t wv = 0; wv < BE_T / 64; wv++) { if (wv < wave) k += sh.wave_starts[wv]; all += sh.wave_starts[wv]; }
        for (uint32_t b = starts; b; b &= b - 1) {
            const uint32_t j = (uint32_t)__builtin_ctz(b);
            pos[k++] = (c0 + tid * BE_PER + j) | (((repeats >> j) & 1) << 31);
        }
        m += all;
        __syncthreads();
    }
    __threadfence_block();
    __syncthreads();
    // ---- their sizes and places; the headers ----
    uint64_t body = 0;
    for (uint32_t k0 = 0; k0 < m; k0 += BE_T) {
        const uint32_t k = k0 + tid;
        uint32_t p = 0, len = 0, hdr = 0;
        bool rep = false;
        uint64_t head = 0;
        if (k < m) {
            const uint32_t e = pos[k];
            p = e & 0x7fffffffu; rep = (e >> 31) != 0;
            len = (k + 1 < m ? pos[k + 1] & 0x7fffffffu : n) - p;
            head = rep ? ((uint64_t)8 * len) << 1 : ((uint64_t)len << 1) | 1; // rle.go:416-427
            hdr = be_uvarint_len(head);
        }
        const unsigned long long size = k < m ? (rep ? hdr + 1ull : (unsigned long long)hdr + len) : 0ull;
        const unsigned long long incl = wave_inclusive_scan<uint64_t>(size);
        if (lane == 63) sh.wave_bytes[wave] = incl;
        __syncthreads();
        unsigned long long at = body + incl - size, all = 0;
        for (uint32_t wv = 0; wv < BE_T / 64; wv++) { if (wv < wave) at += sh.wave_bytes[wv]; all += sh.wave_bytes[wv]; }
        if (k < m) {
            if (4 + at + hdr + (rep ? 1 : 0) <= cap) {
                uint64_t v = head;
                uint32_t o = 0;
                while (v >= 0x80) { dst[4 + at + o++] = (uint8_t)(v | 0x80); v >>= 7; }
                dst[4 + at + o] = (uint8_t)v;
                if (rep) dst[4 + at + hdr] = src[p];
            }
            shift[k] = (uint32_t)(at + hdr) - p;
        }
        body += all;
        __syncthreads();
    }
    const uint64_t bytes = 4 + body;
    if (bytes > cap) {
        if (tid == 0) { pg.status = PQD_ERR_DST_TOO_SMALL; pg.arg = (int64_t)bytes; pg.out_bytes = 0; }
        return;
    }
    if (tid < 4) dst[tid] = (uint8_t)((uint32_t)body >> (8 * tid));
    __threadfence_block();
    __syncthreads();
    // ---- the packed bytes of the bit-packed runs ----
    uint32_t rank = 0;
    g_carry = 0;
    for (uint32_t c0 = 0; c0 < n; c0 += BE_CHUNK) {
        uint32_t starts, repeats;
        be_flags(sh, src, n, c0, starts, repeats, g_carry);
        const uint32_t mine = (uint32_t)__builtin_popcount(starts);
        const uint32_t incl = wave_inclusive_scan<uint32_t>(mine);
        if (lane == 63) sh.wave_starts[wave] = incl;
        __syncthreads();
        uint32_t k = rank + incl - mine, all = 0; // runs that start before this thread's first byte
        for (uint32_t wv = 0; wv < BE_T / 64; wv++) { if (wv < wave) k += sh.wave_starts[wv]; all += sh.wave_starts[wv]; }
        const uint32_t i0 = c0 + tid * BE_PER;
        if (i0 < n) {
            const uint32_t have = n - i0 < BE_PER ? n - i0 : BE_PER;
            if (starts == 0 && have == BE_PER) {
                // sixteen bytes of one run (k >= 1: byte 0 starts a run)
                if (!(pos[k - 1] >> 31)) *(gptr<u32x4_a1e>)(dst + 4 + (uint32_t)(i0 + shift[k - 1])) = *(gptr<const u32x4_a1e>)(src + i0);
            } else {
                uint32_t sft = 0;
                bool packed = false;
                if (k > 0) { packed = !(pos[k - 1] >> 31); sft = shift[k - 1]; }
                for (uint32_t j = 0; j < have; j++) {
                    if ((starts >> j) & 1) { packed = !(pos[k] >> 31); sft = shift[k]; k++; }
                    if (packed) dst[4 + (uint32_t)(i0 + j + sft)] = src[i0 + j];
                }
            }
        }
        rank += all;
        __syncthreads();
    }
    if (tid == 0) { pg.status = PQD_OK; pg.arg = 0; pg.out_bytes = bytes; }
}

// ------------------------------------------------------------------ byte arrays --

// Where value k lies in src, and how long it is.
__device__ inline void ba_value(const EncPage& pg, uint64_t k, uint64_t& at, uint32_t& len) {
    if (pg.offsets != nullptr) {
        const u32x2_a4 e = *(gptr<const u32x2_a4>)(as_global(pg.offsets) + k);
        at = e.x; len = e.y - e.x;
    } else {
        at = k * (uint64_t)pg.width; len = (uint32_t)pg.width;
    }
}

// wordSearchPrefixLength (byte_array.go:206-228): the longest common prefix, eight bytes at a time.
__device__ inline uint32_t ba_common_prefix(gptr<const uint8_t> a, uint32_t na, gptr<const uint8_t> b, uint32_t nb) {
    const uint32_t n = na < nb ? na : nb;
    uint32_t i = 0;
    for (; i + 8 <= n; i += 8) {
        const uint64_t x = *(gptr<const u64_a1e>)(a + i) ^ *(gptr<const u64_a1e>)(b + i);
        if (x) return i + (uint32_t)(__builtin_ctzll(x) >> 3);
    }
    while (i < n && a[i] == b[i]) i++;
    return i;
}

// The length streams: DELTA_LENGTH_BYTE_ARRAY's lengths (encodeByteArrayLengths, length_byte_array_purego.go:5-9), or
// DELTA_BYTE_ARRAY's prefix and suffix lengths (byte_array.go:30-52, 97-105).  A thread per value.
__global__ __launch_bounds__(ET) void ba_lengths_kernel(EncPage* __restrict__ pages, const uint32_t* __restrict__ items) {
    const EncPage& pg = pages[items[blockIdx.x]];
    gptr<const uint8_t> src = as_global(pg.src);
    gptr<int32_t> la = as_global(pg.len_a), lb = as_global(pg.len_b);
    for (uint64_t k = (uint64_t)blockIdx.y * ET + threadIdx.x; k < pg.n_values; k += (uint64_t)gridDim.y * ET) {
        uint64_t at;
        uint32_t len;
        ba_value(pg, k, at, len);
        if (pg.family == EF_DLBA) { la[k] = (int32_t)len; continue; }
        uint32_t pre = 0;
        if (k > 0) {
            uint64_t pat;
            uint32_t plen;
            ba_value(pg, k - 1, pat, plen);
            pre = ba_common_prefix(src + pat, plen, src + at, len);
        }
        la[k] = (int32_t)pre;
        lb[k] = (int32_t)(len - pre);
    }
}

// What an internal length-stream item came to: its bytes, or -1 for an error that is not about room.
__device__ inline int64_t stream_size(const EncPage& s) {
    if (s.status == PQD_OK) return (int64_t)s.out_bytes;
    if (s.status == PQD_ERR_DST_TOO_SMALL) return s.arg;
    return -1;
}

// Between the two length streams of DELTA_BYTE_ARRAY pages: the second goes behind the first.
__global__ __launch_bounds__(64) void ba_chain_kernel(EncPage* __restrict__ pages, const uint32_t* __restrict__ items, uint32_t n_items) {
    const uint32_t k = blockIdx.x * 64 + threadIdx.x;
    if (k >= n_items) return;
    EncPage& pg = pages[items[k]];
    const EncPage& a = pages[pg.chain_a];
    EncPage& b = pages[pg.chain_b];
    const int64_t sa = stream_size(a);
    const bool fits = a.status == PQD_OK && (uint64_t)sa <= pg.dst_cap;
    b.dst = fits ? pg.dst + sa : pg.dst;
    b.dst_cap = fits ? pg.dst_cap - (uint64_t)sa : 0; // no room: the stream is only sized
}

constexpr uint32_t BA_IMG = 8192; // output bytes of 64 values that are put together in LDS

// 64 neighbouring values by one wavefront: lane l moves `len` bytes from src + from to dst + to, `head` says whether the
// value's length goes in front of it as four bytes (PLAIN).  The outputs of the 64 follow one another, so they are put
// together in an LDS image that agrees with the destination modulo 16 and leave in whole 16-byte stores.
__device__ inline void ba_copy64(gptr<const uint8_t> src, gptr<uint8_t> dst, uint64_t from, uint32_t len, uint64_t to, bool head, uint32_t cnt,
                                 uint8_t* img) {
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t hb = head ? 4u : 0u;
    const uint64_t to0 = lane_u64(to, 0);
    const uint64_t total = lane_u64(to, cnt - 1) + lane_u32(len, cnt - 1) + hb - to0;
    if (total > BA_IMG) { // long values: one after the other, a lane per byte
        for (uint32_t v = 0; v < cnt; v++) {
            const uint32_t n = lane_u32(len, v);
            gptr<const uint8_t> s = src + lane_u64(from, v);
            gptr<uint8_t> d = dst + lane_u64(to, v);
            if (head && lane < 4) d[lane] = (uint8_t)(n >> (8 * lane));
            for (uint32_t b = lane; b < n; b += 64) d[hb + b] = s[b];
        }
        return;
    }
    const uint32_t align = (uint32_t)((uintptr_t)(dst + to0) & 15);
    if (lane < cnt) {
        uint8_t* q = img + align + (uint32_t)(to - to0);
        // (LDS takes a dword at any address on this chip)
        if (head) { *reinterpret_cast<u32_a1e*>(q) = len; q += 4; }
        gptr<const uint8_t> s = src + from;
        uint32_t b = 0;
        for (; b + 4 <= len; b += 4) *reinterpret_cast<u32_a1e*>(q + b) = *(gptr<const u32_a1e>)(s + b);
        for (; b < len; b++) q[b] = s[b];
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // one wavefront: its LDS operations execute in order
    gptr<uint8_t> o = dst + to0;
    const uint32_t bytes = (uint32_t)total;
    const uint32_t lead = std::min(bytes, (16 - align) & 15);
    const uint32_t whole = (bytes - lead) / 16, tail0 = lead + whole * 16;
    if (lane < lead) o[lane] = img[align + lane];
    if (tail0 + lane < bytes) o[tail0 + lane] = img[align + tail0 + lane];
    for (uint32_t c = lane; c < whole; c += 64)
        *(gptr<u32x4>)(o + lead + c * 16) = *reinterpret_cast<const u32x4*>(img + align + lead + c * 16);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
}

// PLAIN byte arrays (plain.go:44-57, AppendByteArray): every value behind its length.  Value k goes to
// offsets[k] - offsets[0] + 4 k: no sum to wait for.
__global__ __launch_bounds__(ET) void plain_ba_encode_kernel(EncPage* __restrict__ pages, const uint32_t* __restrict__ items) {
    __shared__ __attribute__((aligned(16))) uint8_t img[ET / 64][BA_IMG + 32];
    EncPage& pg = pages[items[blockIdx.x]];
    const uint64_t n = pg.n_values;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    uint64_t first = 0, last = 0;
    uint32_t l0;
    if (n > 0) { ba_value(pg, 0, first, l0); ba_value(pg, n - 1, last, l0); last += l0; }
    const uint64_t bytes = (last - first) + 4 * n;
    if (bytes > pg.dst_cap) {
        if (blockIdx.y == 0 && threadIdx.x == 0) { pg.status = PQD_ERR_DST_TOO_SMALL; pg.arg = (int64_t)bytes; pg.out_bytes = 0; }
        return;
    }
    for (uint64_t k0 = ((uint64_t)blockIdx.y * (ET / 64) + wave) * 64; k0 < n; k0 += (uint64_t)gridDim.y * ET) {
        const uint64_t k = k0 + lane;
        uint64_t at = 0;
        uint32_t len = 0;
        if (k < n) ba_value(pg, k, at, len);
        ba_copy64(as_global(pg.src), as_global(pg.dst), at, len, (at - first) + 4 * k, true, (uint32_t)std::min<uint64_t>(64, n - k0), img[wave]);
    }
    if (blockIdx.y == 0 && threadIdx.x == 0) { pg.status = PQD_OK; pg.arg = 0; pg.out_bytes = bytes; }
}

// Behind the length streams of a DELTA_LENGTH_BYTE_ARRAY or DELTA_BYTE_ARRAY page, one workgroup per page: the size of
// the page, whether it fits, and for DELTA_BYTE_ARRAY where every suffix goes (an exclusive sum over len_b, in place).
__global__ __launch_bounds__(ET) void ba_place_kernel(EncPage* __restrict__ pages, const uint32_t* __restrict__ items) {
    __shared__ uint64_t wsum[ET / 64];
    EncPage& pg = pages[items[blockIdx.x]];
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const bool dba = pg.family == EF_DBA;
    const int64_t sa = stream_size(pages[pg.chain_a]), sb = dba ? stream_size(pages[pg.chain_b]) : 0;
    if (sa < 0 || sb < 0) { // an error of the length streams that is not about room (none is expected: lengths are non-negative int32)
        if (threadIdx.x == 0) { const EncPage& bad = sa < 0 ? pages[pg.chain_a] : pages[pg.chain_b]; pg.status = bad.status; pg.arg = bad.arg; pg.out_bytes = 0; }
        return;
    }
    uint64_t payload = pg.src_len; // DELTA_LENGTH_BYTE_ARRAY appends all of src (length_byte_array.go:33)
    if (dba) {
        gptr<int32_t> lb = as_global(pg.len_b);
        uint64_t carry = 0;
        for (uint64_t base = 0; base < pg.n_values; base += ET) {
            const uint64_t k = base + threadIdx.x;
            const uint64_t mine = k < pg.n_values ? (uint64_t)(uint32_t)lb[k] : 0;
            const uint64_t x = wave_inclusive_scan<uint64_t>(mine);
            if (lane == 63) wsum[wave] = x;
            __syncthreads();
            uint64_t before = carry, all = 0;
            for (uint32_t w = 0; w < ET / 64; w++) {
                if (w < wave) before += wsum[w];
                all += wsum[w];
            }
            if (k < pg.n_values) lb[k] = (int32_t)(uint32_t)(before + x - mine); // (a page has less than 4 GB of values: offsets are uint32)
            carry += all;
            __syncthreads();
        }
        payload = carry;
    }
    if (threadIdx.x == 0) {
        const uint64_t bytes = (uint64_t)sa + (uint64_t)sb + payload;
        const bool fits = bytes <= pg.dst_cap && pages[pg.chain_a].status == PQD_OK && (!dba || pages[pg.chain_b].status == PQD_OK);
        pg.status = fits ? PQD_OK : PQD_ERR_DST_TOO_SMALL;
        pg.arg = fits ? 0 : (int64_t)bytes;
        pg.out_bytes = fits ? bytes : 0;
    }
}

// The bytes behind the length streams.
__global__ __launch_bounds__(ET) void ba_payload_kernel(EncPage* __restrict__ pages, const uint32_t* __restrict__ items) {
    __shared__ __attribute__((aligned(16))) uint8_t img[ET / 64][BA_IMG + 32];
    const EncPage& pg = pages[items[blockIdx.x]];
    if (pg.status != PQD_OK) return;
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    gptr<const uint8_t> src = as_global(pg.src);
    if (pg.family == EF_DLBA) {
        gptr<uint8_t> out = as_global(pg.dst) + pages[pg.chain_a].out_bytes;
        const uint64_t n = pg.src_len;
        const uint64_t lead = std::min<uint64_t>(n, (16 - ((uintptr_t)out & 15)) & 15);
        const uint64_t whole = (n - lead) / 16, tail0 = lead + whole * 16;
        const uint64_t tid = (uint64_t)blockIdx.y * ET + threadIdx.x, step = (uint64_t)gridDim.y * ET;
        if (tid < lead) out[tid] = src[tid];
        if (tid < n - tail0) out[tail0 + tid] = src[tail0 + tid];
        for (uint64_t c = tid; c < whole; c += step) {
            const u32x4_a1e x = *(gptr<const u32x4_a1e>)(src + lead + c * 16);
            *(gptr<u32x4>)(out + lead + c * 16) = u32x4{x.x, x.y, x.z, x.w};
        }
        return;
    }
    gptr<uint8_t> out = as_global(pg.dst) + pages[pg.chain_a].out_bytes + pages[pg.chain_b].out_bytes;
    gptr<const int32_t> la = as_global((const int32_t*)pg.len_a), lb = as_global((const int32_t*)pg.len_b);
    const uint64_t n = pg.n_values;
    for (uint64_t k0 = ((uint64_t)blockIdx.y * (ET / 64) + wave) * 64; k0 < n; k0 += (uint64_t)gridDim.y * ET) {
        const uint64_t k = k0 + lane;
        uint64_t at = 0, to = 0;
        uint32_t len = 0, pre = 0;
        if (k < n) {
            ba_value(pg, k, at, len);
            pre = (uint32_t)la[k];
            to = (uint32_t)lb[k];
        }
        ba_copy64(src, out, at + pre, len - pre, to, false, (uint32_t)std::min<uint64_t>(64, n - k0), img[wave]);
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
        const uint64_t j = indexes != nullptr ? (uint64_t)(uint32_t)indexes[i] : i; // no indexes: the values of a page as they lie
        if (j >= dict_len) { atomicMin(&status[1], (unsigned long long)i); continue; }
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
        const T first = dict[indexes != nullptr ? (uint32_t)indexes[0] : 0u];
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
    std::vector<EncPage> pages;             // the caller's pages, then the internal length-stream items of byte-array pages
    std::vector<uint32_t> fam_items[EF_N];
    std::vector<uint64_t> fam_prefix[EF_N]; // per family: first tile of each of its pages, and the total
    uint64_t bss_max_values = 0, ba_max_values = 0, ba_max_bytes = 0;
    size_t o_pages = 0, o_items[EF_N] = {}, o_prefix[EF_N] = {}, o_state = 0, state_bytes = 0, upload_end = 0, total = 0;
    size_t o_or = 0, o_first = 0, o_next = 0; // RLE: a word per page (inside the cleared block), two per tile
    size_t o_first_b = 0, o_tile_fn = 0;      // RLE levels: another word and a byte per tile
    size_t o_len = 0;                         // byte arrays: the length streams' values
};

// Families whose pages are cut into tiles that look back over per-tile state words.
static const int TILED_FAMILIES[] = {EF_DELTA32, EF_DELTA64, EF_RLE_DICT, EF_DELTA32_B, EF_LEVELS};

static int32_t value_width(int32_t kind, int32_t bit_width) {
    switch (kind) {
        case PQD_KIND_LEVELS: case PQD_KIND_BOOLEAN: case PQD_KIND_BYTE_ARRAY: return 1;
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
        case PQD_ENC_RLE: // rle.Encoding: EncodeLevels, EncodeBoolean, EncodeInt32 (rle.go:43-61)
            return kind == PQD_KIND_LEVELS ? EF_LEVELS : kind == PQD_KIND_BOOLEAN ? EF_BOOL : kind == PQD_KIND_INT32 ? EF_RLE_DICT : EF_NONE;
        case PQD_ENC_BYTE_STREAM_SPLIT:
            if (kind == PQD_KIND_INT32 || kind == PQD_KIND_INT64 || kind == PQD_KIND_FLOAT || kind == PQD_KIND_DOUBLE) return EF_BSS;
            return kind == PQD_KIND_FIXED_LEN_BYTE_ARRAY && width > 0 && width <= 32767 ? EF_BSS : EF_NONE;
        case PQD_ENC_PLAIN:
            return kind == PQD_KIND_BYTE_ARRAY ? EF_PLAIN_BA : EF_NONE;
        case PQD_ENC_DELTA_LENGTH_BYTE_ARRAY:
            return kind == PQD_KIND_BYTE_ARRAY ? EF_DLBA : EF_NONE;
        case PQD_ENC_DELTA_BYTE_ARRAY: // EncodeByteArray and EncodeFixedLenByteArray (byte_array.go:26-122)
            return kind == PQD_KIND_BYTE_ARRAY || (kind == PQD_KIND_FIXED_LEN_BYTE_ARRAY && width > 0 && width <= 32767) ? EF_DBA : EF_NONE;
        default:
            return EF_NONE;
    }
}

static size_t carve(size_t& off, size_t bytes) {
    const size_t at = off;
    off = (off + bytes + 255) & ~(size_t)255;
    return at;
}

static uint64_t delta_tiles(uint64_t n_values) {
    const uint64_t blocks = n_values < 2 ? 0 : (n_values - 1 + DE_BLOCK - 1) / DE_BLOCK;
    return std::max<uint64_t>(1, (blocks + DE_TILE_BLOCKS - 1) / DE_TILE_BLOCKS);
}

static void plan_encode(const pqd_encode_page* in, uint64_t n, EncodePlan& plan, bool with_pointers) {
    plan.pages.resize(n);
    for (int f = 0; f < EF_N; f++) plan.fam_prefix[f].assign(1, 0);
    auto enlist = [&](uint32_t i) {
        const EncPage& e = plan.pages[i];
        plan.fam_items[e.family].push_back(i);
        plan.fam_prefix[e.family].push_back(plan.fam_prefix[e.family].back() + e.n_tiles);
    };
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
        e.offsets = in[i].offsets;
        e.src_len = in[i].src_bytes;
        e.chain_a = e.chain_b = 0xffffffffu;
        e.fixed_bw = e.encoding == PQD_ENC_RLE ? (uint32_t)in[i].bit_width : 0xffffffffu;
        const bool bytes_family = e.family == EF_PLAIN_BA || e.family == EF_DLBA || e.family == EF_DBA;
        const bool fixed_len = e.kind == PQD_KIND_FIXED_LEN_BYTE_ARRAY;
        auto refuse = [&](int32_t status, int64_t arg) { e.family = EF_NONE; e.status = status; e.arg = arg; };
        if (with_pointers && ((e.src == nullptr && e.n_values > 0 && !bytes_family) || (e.dst == nullptr && e.dst_cap > 0))) {
            refuse(PQD_ERR_INVALID_ARGUMENT, 0);
        } else if (e.family == EF_NONE) {
            refuse(PQD_ERR_NOT_SUPPORTED, 0); // encoding.NotSupported answers these (encoding/notsupported.go:116-160)
        } else if (e.encoding == PQD_ENC_RLE && e.kind != PQD_KIND_BOOLEAN && (in[i].bit_width < 0 || in[i].bit_width > (e.kind == PQD_KIND_LEVELS ? 8 : 32))) {
            refuse(PQD_ERR_INVALID_BIT_WIDTH, in[i].bit_width); // rle.go:142-144, 204-206
        } else if (with_pointers && e.family != EF_BSS && e.width > 1 && !bytes_family && ((uintptr_t)e.src & (uintptr_t)(e.width - 1)) != 0) {
            refuse(PQD_ERR_INVALID_ARGUMENT, (int64_t)((uintptr_t)e.src & (uintptr_t)(e.width - 1))); // typed values are aligned to their width, as a Go slice is
        } else if (e.family != EF_BSS && e.n_values > 0x7fffffffull) {
            refuse(PQD_ERR_INVALID_ARGUMENT, (int64_t)e.n_values); // the decoder refuses more values than MaxInt32 (binary_packed.go:447-449)
        } else if (bytes_family && !fixed_len && e.offsets == nullptr && e.n_values > 0) {
            refuse(PQD_ERR_INVALID_ARGUMENT, 0);
        } else if (bytes_family && with_pointers && e.src == nullptr && e.src_len > 0) {
            refuse(PQD_ERR_INVALID_ARGUMENT, 0);
        } else if (bytes_family && e.src_len > 0xffffffffull) {
            refuse(PQD_ERR_INVALID_ARGUMENT, (int64_t)e.src_len); // offsets are uint32
        } else if (e.family == EF_DBA && fixed_len && e.src_len != e.n_values * (uint64_t)e.width) {
            refuse(PQD_ERR_INVALID_ARGUMENT, (int64_t)e.src_len); // ErrEncodeInvalidInputSize (byte_array.go:85-87)
        } else if (e.family == EF_DLBA && e.offsets == nullptr) {
            refuse(PQD_OK, 0); // no offsets at all: an empty page (length_byte_array.go:21-23)
        }
        if (fixed_len) e.offsets = nullptr;
        if (e.family == EF_NONE) continue;
        if (e.family == EF_DELTA32 || e.family == EF_DELTA64) e.n_tiles = delta_tiles(e.n_values);
        else if (e.family == EF_RLE_DICT) e.n_tiles = std::max<uint64_t>(1, (e.n_values / 8 + RE_GROUPS - 1) / RE_GROUPS);
        else if (e.family == EF_LEVELS) e.n_tiles = std::max<uint64_t>(1, (e.n_values / 8 + LV_WORDS - 1) / LV_WORDS);
        else if (e.family == EF_BSS) plan.bss_max_values = std::max(plan.bss_max_values, e.n_values);
        if (bytes_family) {
            plan.ba_max_values = std::max(plan.ba_max_values, e.n_values);
            plan.ba_max_bytes = std::max(plan.ba_max_bytes, e.src_len);
        }
        enlist((uint32_t)i);
    }
    // the length streams of byte-array pages are DELTA int32 items of their own, behind the caller's pages: the first
    // (lengths, or prefix lengths) goes to the head of the page, the second (suffix lengths) behind it once its size is known
    size_t len_values = 0;
    for (int f : {EF_DLBA, EF_DBA}) {
        for (uint32_t i : plan.fam_items[f]) {
            const int streams = f == EF_DBA ? 2 : 1;
            for (int k = 0; k < streams; k++) {
                EncPage s;
                std::memset(&s, 0, sizeof s);
                s.n_values = plan.pages[i].n_values;
                s.dst = plan.pages[i].dst;
                s.dst_cap = plan.pages[i].dst_cap;
                s.encoding = PQD_ENC_DELTA_BINARY_PACKED;
                s.kind = PQD_KIND_INT32;
                s.width = 4;
                s.family = k == 0 ? EF_DELTA32 : EF_DELTA32_B;
                s.n_tiles = delta_tiles(s.n_values);
                s.chain_a = s.chain_b = 0xffffffffu;
                s.fixed_bw = 0xffffffffu;
                (k == 0 ? plan.pages[i].chain_a : plan.pages[i].chain_b) = (uint32_t)plan.pages.size();
                plan.pages.push_back(s);
                enlist((uint32_t)plan.pages.size() - 1);
                len_values += (s.n_values + 63) & ~(size_t)63;
            }
        }
    }
    // boolean pages: where each run starts, and where its bytes go (a run may be a single byte)
    for (uint32_t i : plan.fam_items[EF_BOOL]) len_values += 2 * ((plan.pages[i].n_values + 64) & ~(size_t)63);
    // the state words of all tiled families lie in one block, cleared before every launch
    uint64_t tiles = 0;
    for (int f : TILED_FAMILIES) {
        for (size_t k = 0; k < plan.fam_items[f].size(); k++) plan.pages[plan.fam_items[f][k]].tile_base = tiles + plan.fam_prefix[f][k];
        tiles += plan.fam_prefix[f].back();
    }
    size_t off = 0;
    plan.o_pages = carve(off, plan.pages.size() * sizeof(EncPage));
    for (int f = 0; f < EF_N; f++) {
        plan.o_items[f] = carve(off, plan.fam_items[f].size() * sizeof(uint32_t));
        plan.o_prefix[f] = carve(off, plan.fam_prefix[f].size() * sizeof(uint64_t));
    }
    plan.upload_end = off;
    const size_t n_rle = plan.fam_items[EF_RLE_DICT].size() + plan.fam_items[EF_LEVELS].size();
    const size_t rle_tiles = plan.fam_prefix[EF_RLE_DICT].back() + plan.fam_prefix[EF_LEVELS].back();
    plan.state_bytes = tiles * sizeof(unsigned long long) + n_rle * sizeof(uint32_t);
    plan.o_state = carve(off, plan.state_bytes);
    plan.o_or = plan.o_state + tiles * sizeof(unsigned long long);
    plan.o_first = carve(off, rle_tiles * sizeof(uint32_t));
    plan.o_next = carve(off, rle_tiles * sizeof(uint32_t));
    plan.o_first_b = carve(off, plan.fam_prefix[EF_LEVELS].back() * sizeof(uint32_t));
    plan.o_tile_fn = carve(off, plan.fam_prefix[EF_LEVELS].back());
    plan.o_len = carve(off, len_values * sizeof(int32_t));
    plan.total = off;
}

// Device addresses of the per-page scratch, once the workspace is known.
static void place_scratch(EncodePlan& plan, uint8_t* ws) {
    size_t page = 0, tile = 0;
    for (int f : {EF_RLE_DICT, EF_LEVELS}) {
        for (size_t k = 0; k < plan.fam_items[f].size(); k++) {
            EncPage& e = plan.pages[plan.fam_items[f][k]];
            e.or_word = reinterpret_cast<uint32_t*>(ws + plan.o_or) + page++;
            e.first_start = reinterpret_cast<uint32_t*>(ws + plan.o_first) + tile + plan.fam_prefix[f][k];
            e.next_start = reinterpret_cast<uint32_t*>(ws + plan.o_next) + tile + plan.fam_prefix[f][k];
            if (f == EF_LEVELS) {
                e.first_start_b = reinterpret_cast<uint32_t*>(ws + plan.o_first_b) + plan.fam_prefix[f][k];
                e.tile_fn = ws + plan.o_tile_fn + plan.fam_prefix[f][k];
            }
        }
        tile += plan.fam_prefix[f].back();
    }
    int32_t* len = reinterpret_cast<int32_t*>(ws + plan.o_len);
    for (uint32_t i : plan.fam_items[EF_BOOL]) {
        EncPage& e = plan.pages[i];
        const size_t room = (e.n_values + 64) & ~(size_t)63;
        e.len_a = len; len += room;
        e.len_b = len; len += room;
    }
    for (int f : {EF_DLBA, EF_DBA}) {
        for (uint32_t i : plan.fam_items[f]) {
            EncPage& e = plan.pages[i];
            const size_t room = (e.n_values + 63) & ~(size_t)63;
            e.len_a = len; len += room;
            plan.pages[e.chain_a].src = reinterpret_cast<const uint8_t*>(e.len_a);
            if (f == EF_DBA) {
                e.len_b = len; len += room;
                plan.pages[e.chain_b].src = reinterpret_cast<const uint8_t*>(e.len_b);
            }
        }
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
        case PQD_ENC_RLE:
            // levels: per word of eight at most a 5-byte run header and 8 packed bytes, then up to seven one-value runs;
            // booleans (n_values bytes of packed bits): the length, and per byte at most a one-byte header and the byte;
            // int32 with the width given: as RLE_DICTIONARY without the width byte
            if (kind == PQD_KIND_LEVELS) return 16 + (n_values / 8) * 13 + 7 * 6;
            if (kind == PQD_KIND_BOOLEAN) return 16 + 2 * n_values;
            return kind == PQD_KIND_INT32 ? 16 + (n_values / 8) * 37 + 7 * 6 : 0;
        case PQD_ENC_BYTE_STREAM_SPLIT:
            return n_values * (uint64_t)(width > 0 ? width : 0);
        default:
            return 0; // byte arrays: pqd_encode_bound_byte_array
    }
}

uint64_t pqd_encode_bound_byte_array(int encoding, uint64_t n_values, uint64_t src_bytes) {
    const uint64_t stream = 24 + (n_values < 2 ? 0 : (n_values - 1 + DE_BLOCK - 1) / DE_BLOCK) * (14 + (uint64_t)DE_BLOCK * 4);
    switch (encoding) {
        case PQD_ENC_PLAIN: return src_bytes + 4 * n_values;
        case PQD_ENC_DELTA_LENGTH_BYTE_ARRAY: return stream + src_bytes;
        case PQD_ENC_DELTA_BYTE_ARRAY: return 2 * stream + src_bytes;
        default: return 0;
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
    std::memcpy(image.data() + plan.o_pages, plan.pages.data(), plan.pages.size() * sizeof(EncPage));
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
    for (int f : TILED_FAMILIES)
        if (plan.fam_prefix[f].back() > 0x7fffffffull) return PQD_ERR_INVALID_ARGUMENT;
    // how many workgroups walk the values (or bytes) of one byte-array page
    auto ba_grid_y = [&](uint64_t units, uint64_t per) { return (uint32_t)std::min<uint64_t>(std::max<uint64_t>(1, (units + per - 1) / per), 2048); };
    // the values of the length streams first: DELTA int32 items encode them below
    for (int f : {EF_DLBA, EF_DBA})
        if (!plan.fam_items[f].empty())
            hipLaunchKernelGGL(ba_lengths_kernel, dim3((uint32_t)plan.fam_items[f].size(), ba_grid_y(plan.ba_max_values, ET)), dim3(ET), 0, s, d_pages, items_of(f));
    auto launch_delta = [&](int f) {
        const uint64_t tiles = plan.fam_prefix[f].back();
        if (tiles == 0) return;
        const uint32_t np = (uint32_t)plan.fam_items[f].size();
        dim3 grid;
        const uint64_t* prefix;
        tile_grid(f, grid, prefix);
        if (f == EF_DELTA64)
            hipLaunchKernelGGL(delta_encode_kernel<int64_t>, grid, dim3(ET), 0, s, d_pages, items_of(f), prefix, np, d_state);
        else
            hipLaunchKernelGGL(delta_encode_kernel<int32_t>, grid, dim3(ET), 0, s, d_pages, items_of(f), prefix, np, d_state);
    };
    launch_delta(EF_DELTA32);
    launch_delta(EF_DELTA64);
    if (const uint32_t nd = (uint32_t)plan.fam_items[EF_DBA].size()) {
        hipLaunchKernelGGL(ba_chain_kernel, dim3((nd + 63) / 64), dim3(64), 0, s, d_pages, items_of(EF_DBA), nd);
        launch_delta(EF_DELTA32_B);
    }
    for (int f : {EF_DLBA, EF_DBA}) {
        if (plan.fam_items[f].empty()) continue;
        const uint32_t np = (uint32_t)plan.fam_items[f].size();
        hipLaunchKernelGGL(ba_place_kernel, dim3(np), dim3(ET), 0, s, d_pages, items_of(f));
        const uint32_t gy = f == EF_DLBA ? ba_grid_y(plan.ba_max_bytes, (uint64_t)ET * 64) : ba_grid_y(plan.ba_max_values, ET);
        hipLaunchKernelGGL(ba_payload_kernel, dim3(np, gy), dim3(ET), 0, s, d_pages, items_of(f));
    }
    if (const uint32_t np = (uint32_t)plan.fam_items[EF_PLAIN_BA].size())
        hipLaunchKernelGGL(plain_ba_encode_kernel, dim3(np, ba_grid_y(plan.ba_max_values, ET)), dim3(ET), 0, s, d_pages, items_of(EF_PLAIN_BA));
    if (const uint32_t np = (uint32_t)plan.fam_items[EF_BOOL].size())
        hipLaunchKernelGGL(bits_encode_kernel, dim3(np), dim3(BE_T), 0, s, d_pages, items_of(EF_BOOL));
    if (plan.fam_prefix[EF_LEVELS].back()) {
        const int f = EF_LEVELS;
        const uint32_t np = (uint32_t)plan.fam_items[f].size();
        dim3 grid;
        const uint64_t* prefix;
        tile_grid(f, grid, prefix);
        hipLaunchKernelGGL(levels_encode_scan_kernel, grid, dim3(ET), 0, s, d_pages, items_of(f), prefix, np);
        hipLaunchKernelGGL(levels_encode_next_kernel, dim3(np), dim3(64), 0, s, d_pages, items_of(f));
        hipLaunchKernelGGL(levels_encode_write_kernel, grid, dim3(ET), 0, s, d_pages, items_of(f), prefix, np, d_state);
    }
    if (const uint64_t tiles = plan.fam_prefix[EF_RLE_DICT].back()) {
        (void)tiles;
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

static int bounds_device(pqd_ctx* ctx, int32_t kind, int32_t is_unsigned, const void* dict_dev, uint64_t dict_len,
                         const int32_t* indexes_dev, uint64_t n, void* bounds_dev, void* workspace_dev, uint64_t workspace_bytes,
                         int64_t* status_dev, void* hip_stream);

int pqd_dict_bounds_device(pqd_ctx* ctx, int32_t kind, int32_t is_unsigned, const void* dict_dev, uint64_t dict_len,
                           const int32_t* indexes_dev, uint64_t n, void* bounds_dev, void* workspace_dev, uint64_t workspace_bytes,
                           int64_t* status_dev, void* hip_stream) {
    if (indexes_dev == nullptr && n > 0) return PQD_ERR_INVALID_ARGUMENT;
    return bounds_device(ctx, kind, is_unsigned, dict_dev, dict_len, indexes_dev, n, bounds_dev, workspace_dev, workspace_bytes, status_dev, hip_stream);
}

int pqd_page_bounds_device(pqd_ctx* ctx, int32_t kind, int32_t is_unsigned, const void* values_dev, uint64_t n, void* bounds_dev,
                           void* workspace_dev, uint64_t workspace_bytes, int64_t* status_dev, void* hip_stream) {
    if (values_dev == nullptr && n > 0) return PQD_ERR_INVALID_ARGUMENT;
    return bounds_device(ctx, kind, is_unsigned, values_dev, n, nullptr, n, bounds_dev, workspace_dev, workspace_bytes, status_dev, hip_stream);
}

static int bounds_device(pqd_ctx* ctx, int32_t kind, int32_t is_unsigned, const void* dict_dev, uint64_t dict_len,
                         const int32_t* indexes_dev, uint64_t n, void* bounds_dev, void* workspace_dev, uint64_t workspace_bytes,
                         int64_t* status_dev, void* hip_stream) {
    if (ctx == nullptr || bounds_dev == nullptr || status_dev == nullptr || workspace_dev == nullptr ||
        workspace_bytes < PQD_BOUNDS_WORKSPACE_BYTES || ((uintptr_t)workspace_dev & 15) != 0)
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
