// delta.hip -- DELTA_BINARY_PACKED decode for gfx950.
//
// Replaces delta.BinaryPackedEncoding.DecodeInt32/DecodeInt64 and the length
// streams of the DELTA byte-array encodings (reference:
// encoding/delta/binary_packed.go:279-488, binary_packed_purego.go:51-67;
// on amd64 bitpack.Unpack + decodeBlockInt32/64, binary_packed_amd64.s:423-498,902-920).
//
// One 256-thread workgroup walks one page front to back.  Page bytes stream
// through a 16 KiB LDS ring in coalesced 16-byte loads; lane 0 walks the block
// headers out of LDS (the byte position of every header depends on all
// earlier ones); then all four waves unpack a tile of 1024 deltas, add each
// block's min delta, prefix-sum across the tile (thread -> wave -> workgroup)
// and store 32 contiguous bytes per thread.  The running value is carried in
// registers from tile to tile, so a page needs no cross-workgroup hand-off and
// is read exactly once.  Parallelism comes from the batch: one workgroup per
// page, eight workgroups per CU.
#include <algorithm>

#include "pqd_internal.h"

namespace pqd {
namespace {

constexpr int DT = 256;                       // threads per workgroup
constexpr uint32_t TILE = 1024;               // deltas per tile, 4 per thread
constexpr uint32_t RING_BYTES = 16384;
constexpr uint32_t RING_MASK = RING_BYTES - 1;
constexpr uint32_t RING_DW_MASK = RING_BYTES / 4 - 1;
constexpr uint32_t CHUNK = DT * 16;           // bytes one fill step brings in
constexpr uint32_t LOOK = 3 * CHUNK;          // bytes the prefetch brings the ring up to, ahead of the tile start
constexpr uint32_t LOOK_MIN = 2 * CHUNK;      // bytes ahead of the tile start that every tile finds resident
constexpr int MAX_SLOTS = 40;                 // miniblocks a tile can touch (1024/32 + partials)
constexpr int BW_CACHE = 64;                  // bit widths of one block kept in LDS

struct Slot {
    int64_t min_delta;
    uint32_t off;  // ring-space byte offset of the miniblock data, low 32 bits (only ring positions are derived from it)
    uint32_t bw;
};

struct Ctl {
    uint64_t lo;        // ring-space offset where the next tile's bytes start
    uint64_t total;     // values the header announces
    int64_t first;
    int64_t arg;
    uint64_t consumed;  // ring-space offset one past the last consumed byte
    uint32_t vpm;       // values per miniblock
    uint32_t count;     // deltas in the current tile
    uint32_t slot_base; // global miniblock index of slot 0
    int32_t status;
    int32_t stop;       // nonzero: no further tile
};

struct Shared {
    uint32_t ring[RING_BYTES / 4 + 4]; // + mirror of the first 16 bytes, so a field never wraps
    Slot slots[MAX_SLOTS];
    uint64_t wtot[DT / 64];
    Ctl ctl;
    uint8_t blk_bw[BW_CACHE];
};

// Go's binary.Uvarint over ring bytes [pos, end): bytes read, 0 when the input
// ends inside the varint, negative on overflow.
__device__ inline int ring_uvarint(const uint8_t* ring8, uint64_t pos, uint64_t end, uint64_t* out) {
    uint64_t x = 0;
    unsigned s = 0;
    for (int i = 0;; i++) {
        if (pos + i >= end) { *out = 0; return 0; }
        if (i == 10) return -(i + 1);
        uint32_t b = ring8[(pos + i) & RING_MASK];
        if (b < 0x80) {
            if (i == 9 && b > 1) return -(i + 1);
            *out = x | ((uint64_t)b << s);
            return i + 1;
        }
        x |= (uint64_t)(b & 0x7f) << s;
        s += 7;
    }
}

__device__ inline int64_t unzigzag(uint64_t u) {
    int64_t x = (int64_t)(u >> 1);
    return (u & 1) ? ~x : x;
}

// Writes 16 loaded bytes at ring-space offset v, zeroing what lies at or past
// `end` (a truncated miniblock is zero padded, binary_packed.go:380-389).  The
// first 16 bytes of the ring are mirrored behind its end, so a reader may
// fetch the two dwords after any ring position without wrapping.
__device__ inline void ring_put(uint32_t* ring, uint64_t v, uint64_t end, u32x4 w) {
    if (__builtin_expect(v + 16 > end, 0)) {
        const int64_t keep = (int64_t)end - (int64_t)v; // bytes of w that belong to the page (may be <= 0)
#pragma unroll
        for (int k = 0; k < 4; k++) {
            const int64_t kb = keep - 4 * k;
            const uint32_t m = kb >= 4 ? 0xffffffffu : (kb <= 0 ? 0u : ((1u << ((uint32_t)kb * 8)) - 1));
            w[k] &= m;
        }
    }
    const uint32_t at = (uint32_t)(v & RING_MASK) >> 2;
    *reinterpret_cast<u32x4*>(&ring[at]) = w;
    if (at == 0) *reinterpret_cast<u32x4*>(&ring[RING_BYTES / 4]) = w;
}

// Brings ring chunks in until [lo, lo + LOOK) is resident.  `base` is the
// 16-byte aligned address ring-space offset 0 maps to; bytes at or past
// `end` are stored as zero (a truncated miniblock is zero padded,
// binary_packed.go:380-389).  Uniform across the workgroup.
__device__ inline void ring_fill(uint32_t* ring, gptr<const uint8_t> base, uint64_t end, uint64_t lo, uint64_t& c_next) {
    uint64_t c_lo = lo / CHUNK;
    if (c_next < c_lo) c_next = c_lo;
    uint64_t c_stop = c_lo + LOOK / CHUNK; // inclusive last chunk needed
    while (c_next <= c_stop) {
        uint64_t v = c_next * CHUNK + threadIdx.x * 16u;
        u32x4 w = (u32x4)(0u);
        if (v < end) w = *(gptr<const u32x4>)(base + v);
        ring_put(ring, v, end, w);
        c_next++;
    }
}

template <typename U> struct Elem;
template <> struct Elem<uint32_t> { static constexpr uint32_t max_bw = 32; };
template <> struct Elem<uint64_t> { static constexpr uint32_t max_bw = 64; };

// Page header (binary_packed.go:409-452) read by one lane out of the ring; fills
// the control block, writes the first value and reports block geometry.
template <typename U>
__device__ inline void parse_header(const uint8_t* ring8, uint64_t mis, uint64_t end, uint64_t dst_cap, gptr<U> out, Ctl& c,
                                    int64_t& block_size, int64_t& num_mb) {
    c.status = PQD_OK; c.arg = 0; c.stop = 0; c.count = 0; c.total = 0; c.first = 0; c.vpm = 0; c.slot_base = 0;
    uint64_t p = mis, u[3] = {0, 0, 0};
    for (int f = 0; f < 3 && c.status == PQD_OK; f++) {
        int k = ring_uvarint(ring8, p, end, &u[f]);
        if (k == 0) { c.status = PQD_ERR_UNEXPECTED_EOF; c.arg = PQD_EOF_DELTA_HEADER; }
        else if (k < 0) { c.status = PQD_ERR_VARINT_OVERFLOW; c.arg = (int64_t)(p - mis); }
        else p += k;
    }
    if (c.status == PQD_OK) {
        uint64_t z;
        int k = ring_uvarint(ring8, p, end, &z);
        if (k == 0) { c.status = PQD_ERR_UNEXPECTED_EOF; c.arg = PQD_EOF_DELTA_HEADER; }
        else if (k < 0) { c.status = PQD_ERR_VARINT_OVERFLOW; c.arg = (int64_t)(p - mis); }
        else { p += k; c.first = unzigzag(z); }
    }
    block_size = (int64_t)u[0];
    num_mb = (int64_t)u[1];
    const int64_t total = (int64_t)u[2];
    if (c.status == PQD_OK) {
        // binary_packed.go:437-449, in the reference's order
        if (num_mb == 0) { c.status = PQD_ERR_BAD_DELTA_HEADER; c.arg = PQD_DELTA_BAD_NUM_MINIBLOCKS; }
        else if (block_size <= 0 || (block_size % 128) != 0) { c.status = PQD_ERR_BAD_DELTA_HEADER; c.arg = PQD_DELTA_BAD_BLOCK_SIZE; }
        else if (block_size > 65536) { c.status = PQD_ERR_BAD_DELTA_HEADER; c.arg = PQD_DELTA_BLOCK_TOO_LARGE; }
        else if (num_mb <= 0 || ((block_size / num_mb) % 32) != 0) { c.status = PQD_ERR_BAD_DELTA_HEADER; c.arg = PQD_DELTA_BAD_MINIBLOCK_SIZE; }
        else if (total < 0) { c.status = PQD_ERR_BAD_DELTA_HEADER; c.arg = PQD_DELTA_NEGATIVE_TOTAL; }
        else if (total > 2147483647LL) { c.status = PQD_ERR_BAD_DELTA_HEADER; c.arg = PQD_DELTA_TOO_MANY_VALUES; }
    }
    c.consumed = p;
    if (c.status == PQD_OK && total > 0) {
        if (sizeof(U) == 4 && (c.first < -2147483647LL - 1 || c.first > 2147483647LL)) {
            c.status = PQD_ERR_FIRST_VALUE_RANGE; c.arg = c.first;
        } else if ((uint64_t)total * sizeof(U) > dst_cap) {
            c.status = PQD_ERR_DST_TOO_SMALL; c.arg = (int64_t)((uint64_t)total * sizeof(U));
        }
    }
    if (c.status == PQD_OK) {
        c.total = (uint64_t)total;
        c.vpm = (uint32_t)(block_size / num_mb);
        if (total > 0) out[0] = (U)c.first;
    }
    if (c.status != PQD_OK || total <= 1) c.stop = 1;
    c.lo = p;
}

// State of the header walk, kept in lane 0's registers across tiles.
struct Walk {
    uint64_t pos;       // ring-space offset of the next unread byte
    uint64_t end;       // ring-space offset one past the page
    uint64_t bw_pos;    // ring-space offset of the current block's bit widths
    int64_t min_delta;
    uint64_t cur_off;   // data offset of the current miniblock
    uint32_t cur_bw;
    uint32_t cur_rem;   // values left in the current miniblock
    uint32_t cur_within;
    uint32_t mb_in_block, nbw_in_block;
    uint32_t mb_global; // miniblocks started so far
};

// Unpacks and prefix-sums one tile: thread t owns deltas 4t..4t+3 of the tile,
// which sit in the miniblock `s` describes, starting at field `within`; r[]
// receives its four decoded values.  Adds the tile's sum to `carry`.  Contains
// one workgroup barrier.
// Fields are cut out of the ring with 32-bit funnel shifts (v_alignbit_b32); a
// 64-bit value is assembled from two halves, never shifted as a whole.
// Barrier for hand-offs through LDS only.  __syncthreads() also waits for every
// outstanding global store of the wave (vmcnt(0)); the fast kernel never hands
// global data from wave to wave, so it must not pay for that.
__device__ inline void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

template <typename U, bool LDS_ONLY>
__device__ inline U tile_values(const uint32_t* ring, uint64_t* wtot, const Slot s, uint32_t count, uint32_t within, U& carry,
                                U (&r)[4], uint32_t tid = threadIdx.x) {
    const uint32_t lane = tid & 63, wave = tid >> 6;
    const uint32_t j = tid * 4;
    U v[4] = {0, 0, 0, 0};
    if (j < count) {
        const uint32_t bw = s.bw;
        const U md = (U)s.min_delta;
        const uint32_t bit0 = within * bw;               // within < 65536, bw <= 64
        const uint32_t byte0 = s.off + (bit0 >> 3);
        const uint32_t dw = (byte0 & RING_MASK) >> 2;
        const uint32_t sft = (byte0 & 3) * 8 + (bit0 & 7);
        const uint32_t mask_lo = bw >= 32 ? 0xffffffffu : ((1u << bw) - 1);
        const uint32_t mask_hi = bw <= 32 ? 0u : (bw >= 64 ? 0xffffffffu : ((1u << (bw - 32)) - 1));
#pragma unroll
        for (int i = 0; i < 4; i++) {
            const uint32_t b = sft + i * bw;
            const uint32_t* q = ring + ((dw + (b >> 5)) & RING_DW_MASK); // q[1], q[2] exist thanks to the mirror
            const uint32_t w0 = q[0], w1 = q[1];
            const uint32_t f_lo = __builtin_amdgcn_alignbit(w1, w0, b) & mask_lo; // the shift uses b mod 32
            U d;
            if (sizeof(U) == 8) {
                const uint32_t w2 = q[2];
                const uint32_t f_hi = __builtin_amdgcn_alignbit(w2, w1, b) & mask_hi;
                d = (U)(((uint64_t)f_hi << 32) | f_lo);
            } else {
                d = (U)f_lo;
            }
            d += md;
            v[i] = (j + i < count) ? d : (U)0;
        }
    }
    // thread-local inclusive scan, then across the wave
    v[1] += v[0]; v[2] += v[1]; v[3] += v[2];
    const U x = wave_inclusive_scan<U>(v[3]);
    if (lane == 63) wtot[wave] = (uint64_t)x;
    if (LDS_ONLY) lds_barrier(); else __syncthreads(); // B3: wave totals visible
    U wave_prefix = 0, all = 0;
#pragma unroll
    for (int w = 0; w < DT / 64; w++) {
        const U t = (U)wtot[w];
        if (w < (int)wave) wave_prefix += t;
        all += t;
    }
    const U excl = carry + wave_prefix + (x - v[3]);
    r[0] = excl + v[0]; r[1] = excl + v[1]; r[2] = excl + v[2]; r[3] = excl + v[3];
    carry += all;
    return excl; // the value just before r[0]
}

// Stores the four values of this thread: 32 contiguous bytes (16 for INT32).
template <typename U>
__device__ inline void tile_store(gptr<U> out, uint64_t done, uint32_t count, const U (&r)[4]) {
    const uint32_t j = threadIdx.x * 4;
    if (j < count) {
        // tile base kept scalar (one SGPR pair, advanced per tile) + a small per-thread
        // offset, instead of a 64-bit per-thread pointer living in VGPRs across the loop
        const uint64_t ot = (uint64_t)(uintptr_t)out + (1 + done) * sizeof(U);
        const uint32_t ot_lo = __builtin_amdgcn_readfirstlane((uint32_t)ot);
        const uint32_t ot_hi = __builtin_amdgcn_readfirstlane((uint32_t)(ot >> 32));
        gptr<U> o = (gptr<U>)(((uint64_t)ot_hi << 32) | ot_lo) + j;
        if (j + 4 <= count) {
            if (sizeof(U) == 8) {
                // two 16-byte stores; the array starts one element in, so 8-byte aligned only
                u64x2_a8 a, b2;
                a.x = (uint64_t)r[0]; a.y = (uint64_t)r[1];
                b2.x = (uint64_t)r[2]; b2.y = (uint64_t)r[3];
                *(gptr<u64x2_a8>)(o) = a;
                *(gptr<u64x2_a8>)(o + 2) = b2;
            } else {
                u32x4_a4 a;
                a.x = (uint32_t)r[0]; a.y = (uint32_t)r[1];
                a.z = (uint32_t)r[2]; a.w = (uint32_t)r[3];
                *(gptr<u32x4_a4>)(o) = a;
            }
        } else {
            for (uint32_t i = 0; i < 4 && j + i < count; i++) o[i] = r[i];
        }
    }
}

// Aligned stores for the fast kernel.  The value just before a thread's four
// is its exclusive prefix `e`, so thread t can store {e, r0, r1, r2} at element
// done + 4t of the output instead of {r0..r3} at element done + 4t + 1: the
// same bytes overall, but every 16-byte store is 16-byte aligned.  Element
// done + TILE (r3 of the last thread) is written by the next tile, or by the
// caller after the last tile.
template <typename U>
__device__ inline void tile_store_shifted_full(gptr<U> out, uint32_t done, U e, const U (&r)[4], uint32_t tid = threadIdx.x) {
    const uint64_t ot = (uint64_t)(uintptr_t)out + (uint64_t)done * sizeof(U);
    const uint32_t ot_lo = __builtin_amdgcn_readfirstlane((uint32_t)ot);
    const uint32_t ot_hi = __builtin_amdgcn_readfirstlane((uint32_t)(ot >> 32));
    gptr<U> o = (gptr<U>)(((uint64_t)ot_hi << 32) | ot_lo) + tid * 4;
    if (sizeof(U) == 8) {
        u64x2_a8 a, b2;
        a.x = (uint64_t)e; a.y = (uint64_t)r[0];
        b2.x = (uint64_t)r[1]; b2.y = (uint64_t)r[2];
        *(gptr<u64x2_a8>)(o) = a;
        *(gptr<u64x2_a8>)(o + 2) = b2;
    } else {
        u32x4_a4 a;
        a.x = (uint32_t)e; a.y = (uint32_t)r[0];
        a.z = (uint32_t)r[1]; a.w = (uint32_t)r[2];
        *(gptr<u32x4_a4>)(o) = a;
    }
}

// A tile of fewer than TILE deltas: elements done .. done + count.
template <typename U>
__device__ inline void tile_store_shifted_partial(gptr<U> out, uint32_t done, uint32_t count, U e, const U (&r)[4]) {
    const uint32_t j = threadIdx.x * 4;
    gptr<U> o = out + done + j;
    if (j <= count) o[0] = e;
    for (uint32_t k = 1; k < 4; k++)
        if (j + k <= count) o[k] = r[k - 1];
}

template <typename U>
__device__ inline void decode_tile(const uint32_t* ring, uint64_t* wtot, const Slot s, gptr<U> out, uint64_t done, uint32_t count,
                                   uint32_t within, U& carry) {
    U r[4];
    tile_values<U, false>(ring, wtot, s, count, within, carry, r);
    tile_store<U>(out, done, count, r);
}

// ---------------------------------------------------------------------------
// Fast path: block sizes 128/256/512/1024 with at most 32 miniblocks (every
// writer in practice; the reference writes 128/4).  A tile is then a whole
// number of blocks, and wave 0 finds block headers for up to four tiles at a
// time, in parallel: lane l guesses that block l starts l * S bytes after the
// first one, S being the size of the block before, parses the header it finds
// there, and the guess holds for every lane whose predecessors all measured S
// again.  With steady widths one round finds 32 blocks; when every block
// differs the walk degrades to one block per round, still all in LDS.
// Pages of any other geometry are left to the general kernel below.
// ---------------------------------------------------------------------------

constexpr uint64_t FAST_DONE = 0x46415354444f4e45ull; // marks a result the fast kernel produced
constexpr uint64_t FLAT_PENDING = 0x464c415450454e44ull; // marks a page the tile-parallel kernels (delta_flat_*) decode
constexpr uint32_t TBL_TILES = 4;    // tiles one walk can describe
constexpr uint32_t TBL_SLOTS = TBL_TILES * 32;  // miniblocks one walk can describe
constexpr uint32_t TBL_BLOCKS = TBL_TILES * 8; // blocks one walk can describe
constexpr uint32_t HDR_MAX = 10 + 32; // longest block header the fast path reads
constexpr uint64_t FAST_MAX_BYTES = 0xfff00000ull; // pages the fast path addresses with 32-bit offsets
constexpr uint32_t NO_POS = 0xffffffffu;
#ifndef PQD_DELTA_ROTATE
#define PQD_DELTA_ROTATE 4 // tiles between two changes of a workgroup's priority; 0: the priority is left alone
#endif

// Everything below is in ring space (byte offsets from the 16-byte aligned
// base of the page), 32 bits wide.
struct FastCtl {
    uint32_t tile_lo[TBL_TILES + 1]; // start of every tile's bytes; [tiles] = start of what follows
    uint32_t consumed;
    uint32_t tiles;      // tiles described by the table
    uint32_t got;        // deltas described by the table
    int32_t status;      // status to report once the table is used up
    int32_t stop;        // nonzero: nothing follows the table
    int64_t arg;
};

struct FastShared {
    uint32_t ring[RING_BYTES / 4 + 4]; // + mirror of the first 16 bytes, so a field never wraps
    Slot slots[TBL_SLOTS];
    uint32_t blk_end[TBL_BLOCKS];
    uint32_t blk_pos[TBL_BLOCKS]; // where the walk found each block; NO_POS once its slots are written
    uint64_t wtot[DT / 64];
    Ctl ctl;
    FastCtl fc;
};

// Go's binary.Uvarint over ring bytes [pos, end), 32-bit positions.
__device__ inline int ring_uvarint32(const uint8_t* ring8, uint32_t pos, uint32_t end, uint64_t* out) {
    uint64_t x = 0;
    unsigned s = 0;
    for (int i = 0;; i++) {
        if (pos + i >= end) { *out = 0; return 0; }
        if (i == 10) return -(i + 1);
        uint32_t b = ring8[(pos + i) & RING_MASK];
        if (b < 0x80) {
            if (i == 9 && b > 1) return -(i + 1);
            *out = x | ((uint64_t)b << s);
            return i + 1;
        }
        x |= (uint64_t)(b & 0x7f) << s;
        s += 7;
    }
}

// Synchronous top-up of the ring: chunks c_next .. lo/CHUNK + LOOK_MIN/CHUNK.
__device__ inline void ring_fill32(uint32_t* ring, gptr<const uint8_t> base, uint32_t end, uint32_t lo, uint32_t& c_next) {
    const uint32_t c_lo = lo / CHUNK;
    if (c_next < c_lo) c_next = c_lo;
    const uint32_t c_stop = c_lo + LOOK_MIN / CHUNK;
    while (c_next <= c_stop) {
        const uint32_t v = c_next * CHUNK + threadIdx.x * 16u;
        const uint32_t vs = v < end ? v : 0u; // always a readable address; ring_put discards what lies past the end
        const u32x4 w = *(gptr<const u32x4>)(base + vs);
        ring_put(ring, v, end, w);
        c_next++;
    }
}


// The sixteen bytes at ring position p, as two little-endian 64-bit halves.
__device__ inline void ring_16(const uint32_t* ring, uint32_t p, uint64_t& lo, uint64_t& hi) {
    const uint32_t d = (p & RING_MASK) >> 2, sh = (p & 3) * 8;
    const uint32_t w0 = ring[d], w1 = ring[d + 1], w2 = ring[d + 2], w3 = ring[d + 3], w4 = ring[d + 4]; // the mirror covers d + 4
    lo = ((uint64_t)__builtin_amdgcn_alignbit(w2, w1, sh) << 32) | __builtin_amdgcn_alignbit(w1, w0, sh);
    hi = ((uint64_t)__builtin_amdgcn_alignbit(w4, w3, sh) << 32) | __builtin_amdgcn_alignbit(w3, w2, sh);
}

// Header of a block with four miniblocks, out of sixteen bytes in registers: length of the
// min-delta varint (Go's binary.Uvarint rules: at most ten bytes, the tenth at most 1) and the
// four width bytes behind it.  False when the varint does not end or overflows.
__device__ inline bool quick_header(uint64_t lo, uint64_t hi, uint32_t& k, uint32_t& widths) {
    const uint64_t ends = ~lo & 0x8080808080808080ull; // bytes without a continuation bit
    if (ends) k = ((uint32_t)__builtin_ctzll(ends) >> 3) + 1;
    else if (!(hi & 0x80)) k = 9;
    else if (!(hi & 0x8000) && ((hi >> 8) & 0xff) <= 1) k = 10;
    else return false;
    widths = k < 8 ? (uint32_t)((lo >> (8 * k)) | (hi << (64 - 8 * k))) : (uint32_t)(hi >> (8 * k - 64));
    return true;
}

// The value of that varint (k bytes, k from quick_header).
__device__ inline uint64_t quick_uvarint(uint64_t lo, uint64_t hi, uint32_t k) {
    uint64_t x = lo & 0x7f7f7f7f7f7f7f7full;                                     // 8 x 7 bits in bytes
    x = (x & 0x007f007f007f007full) | ((x & 0x7f007f007f007f00ull) >> 1);        // 4 x 14 bits in halves
    x = (x & 0x00003fff00003fffull) | ((x & 0x3fff00003fff0000ull) >> 2);        // 2 x 28 bits in words
    x = (x & 0x000000000fffffffull) | ((x & 0x0fffffff00000000ull) >> 4);        // 56 bits
    if (k < 8) return x & ((1ull << (7 * k)) - 1);
    if (k == 8) return x;
    x |= (hi & 0x7f) << 56;
    if (k == 10) x |= ((hi >> 8) & 1) << 63;
    return x;
}

template <typename U>
__global__ __launch_bounds__(DT) __attribute__((amdgpu_waves_per_eu(8, 8), amdgpu_num_sgpr(80)))
void delta_fast_kernel(const DevPage* __restrict__ pages, DevResult* __restrict__ results,
                       const uint32_t* __restrict__ items) {
    __shared__ FastShared sh;
#ifdef PQD_DELTA_TIMING
    const uint64_t t_entry = __builtin_readcyclecounter();
#endif
    const uint32_t item = items[blockIdx.x];
    const DevPage& pg = pages[item];
    DevResult* res = &results[item];
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63;
    const uint8_t* ring8 = reinterpret_cast<const uint8_t*>(sh.ring);

    uint64_t src_skip = 0;
    if (pg.chain >= 0) src_skip = results[pg.chain].consumed;
    const uint8_t* src = pg.src + src_skip;
    const uint64_t src_len = pg.src_len > src_skip ? pg.src_len - src_skip : 0;
    const uint32_t mis = (uint32_t)(reinterpret_cast<uintptr_t>(src) & 15);
    gptr<const uint8_t> base = as_global(src - mis);
    gptr<U> out = (gptr<U>)(pg.dst);
    if (res->aux == FLAT_PENDING) return; // uniform: decoded tile by tile (delta_flat_*)
    if (src_len > FAST_MAX_BYTES) { // uniform: left to the general kernel, which addresses with 64 bits
        if (tid == 0) res->aux = 0;
        return;
    }
    const uint32_t end = mis + (uint32_t)src_len;

    uint32_t c_next = 0;
    ring_fill32(sh.ring, base, end, mis, c_next);
    __syncthreads();

    if (tid == 0) {
        Ctl c;
        int64_t block_size = 0, num_mb = 0;
        parse_header<U>(ring8, mis, end, pg.dst_cap, out, c, block_size, num_mb);
        const bool regular = c.status != PQD_OK || c.total <= 1 ||
                             ((block_size == 128 || block_size == 256 || block_size == 512 || block_size == 1024) && num_mb <= 32);
        // slot_base carries the verdict and the geometry to the other lanes
        c.slot_base = regular ? (uint32_t)num_mb : 0xffffffffu;
        c.count = (uint32_t)block_size;
        sh.ctl = c;
    }
    __syncthreads();
    if (sh.ctl.slot_base == 0xffffffffu) { // irregular geometry: the general kernel decodes this page
        if (tid == 0) res->aux = 0;
        return;
    }

    const uint32_t total = (uint32_t)sh.ctl.total;          // <= MaxInt32
    const uint32_t vpm = sh.ctl.vpm;
    const uint32_t vpm_shift = vpm ? 31 - __clz(vpm) : 0;
    const uint32_t slots_tile = vpm ? TILE >> vpm_shift : 1;           // miniblocks per tile
    int stop = sh.ctl.stop;
    int32_t status = sh.ctl.status;
    int64_t status_arg = sh.ctl.arg;
    uint32_t consumed = (uint32_t)sh.ctl.consumed;
    U carry = (U)sh.ctl.first;
    uint32_t done = 0;
    uint32_t lo = (uint32_t)sh.ctl.lo;   // start of the next tile's bytes
    uint32_t guess = 0, alt = 0;         // wave 0: the block size the walk expects; the last size that differed
    uint32_t tbl_tiles = 0, tbl_next = 0, tbl_got = 0;
    __syncthreads();

    // chunks loaded during the previous tile, still in registers: pf_n of them, the first at ring offset pf_v
    uint32_t pf_n = 0, pf_v = 0;
    u32x4 pf0 = (u32x4)(0u), pf1 = (u32x4)(0u);

    while (!stop) {
        // The chunks requested one tile ago go into the ring now: every wave is past the
        // ring reads of that tile (barrier inside tile_values), and the loads have had a
        // whole tile to land.  The tile's own stores were issued after them, so waiting
        // for these loads does not wait for those stores (memory ops retire in order).
        if (pf_n > 0) {
            ring_put(sh.ring, pf_v, end, pf0);
            if (pf_n > 1) ring_put(sh.ring, pf_v + CHUNK, end, pf1);
            c_next += pf_n;
            pf_n = 0;
        }
        // Top-up: normally a no-op, the prefetch keeps the ring ahead of the tile.
        ring_fill32(sh.ring, base, end, lo, c_next);
        lds_barrier(); // B1: ring bytes (prefetched or topped up) visible

        if (tbl_next == tbl_tiles) {
            if (tid < 64) {
                // ---- walk: block headers of up to TBL_TILES tiles, starting at lo ----
                // the page geometry is re-read from LDS here, so that it does not occupy registers across the tile loop
                const uint32_t w_vpm = sh.ctl.vpm, w_nmb = sh.ctl.slot_base, w_bs = sh.ctl.count;
                const uint32_t w_vpm_shift = w_vpm ? 31 - __clz(w_vpm) : 0, w_blk_shift = w_bs ? 31 - __clz(w_bs) : 0;
                const uint32_t w_nb_tile = w_bs ? TILE >> w_blk_shift : 1;
                const uint32_t w_total = (uint32_t)sh.ctl.total;
                const uint32_t w_n_deltas = w_total > 0 ? w_total - 1 : 0;
                const uint32_t win_hi = c_next * CHUNK; // resident bytes end here
                const uint32_t rem = w_n_deltas - done;
                const uint32_t blocks_left = (rem + w_bs - 1) >> w_blk_shift;
                const uint32_t target = blocks_left < TBL_TILES * w_nb_tile ? blocks_left : TBL_TILES * w_nb_tile;
                const uint32_t enough = target < w_nb_tile ? target : w_nb_tile; // one tile, or what the page still has
                uint32_t pos = lo;
                uint32_t found = 0, last_vals = 0;
                int32_t wstatus = PQD_OK;
                int64_t warg = 0;
                bool ended = false;
                // One tile's blocks are always inside the resident window; the walk goes on past them,
                // up to the whole table, for as long as the next header is.
                while (found < target) {
                    if (found >= enough && pos + HDR_MAX > win_hi && win_hi < end) break;
                    const uint32_t bidx = found + lane;
                    const uint32_t cand = pos + lane * guess;
                    const bool parsed = bidx < target && (lane == 0 || cand + HDR_MAX <= win_hi || win_hi >= end);
                    // values this block still has to deliver, and the miniblocks that hold them
                    const uint32_t before = bidx << w_blk_shift;
                    const uint32_t bvals = parsed ? (rem - before < w_bs ? rem - before : w_bs) : 0;
                    const uint32_t need_mb = (bvals + w_vpm - 1) >> w_vpm_shift;
                    {
                        // Quick round: only what the chain needs (the size of the block that would
                        // start at cand), out of sixteen bytes in registers.  Blocks with four
                        // miniblocks, all of them wanted, away from the end of the page; whatever
                        // else heads the chain is left to the full parse below.
                        bool quick = parsed && w_nmb == 4 && need_mb == 4 && cand + 16 <= end;
                        uint32_t qsize = 0;
                        if (quick) {
                            uint64_t qlo, qhi;
                            ring_16(sh.ring, cand, qlo, qhi);
                            uint32_t k, wd;
                            quick = quick_header(qlo, qhi, k, wd);
                            const uint32_t b0 = wd & 0xff, b1 = (wd >> 8) & 0xff, b2 = (wd >> 16) & 0xff, b3 = wd >> 24;
                            const uint32_t top = b0 > b1 ? (b0 > b2 ? (b0 > b3 ? b0 : b3) : (b2 > b3 ? b2 : b3))
                                                         : (b1 > b2 ? (b1 > b3 ? b1 : b3) : (b2 > b3 ? b2 : b3));
                            if (top > Elem<U>::max_bw) quick = false;
                            qsize = k + 4 + (w_vpm >> 3) * (b0 + b1 + b2 + b3);
                        }
                        const unsigned long long qmask = __ballot(quick);
                        if (qmask & 1) {
                            const unsigned long long okmask = __ballot(quick && qsize == guess);
                            const uint32_t lead = okmask == ~0ull ? 64u : (uint32_t)__builtin_ctzll(~okmask);
                            uint32_t valid = lead < 64 ? lead + (uint32_t)((qmask >> lead) & 1) : 64u;
                            if (valid > target - found) valid = target - found;
                            if (lane < valid) sh.blk_pos[bidx] = cand;
                            const uint32_t last = valid - 1;
                            const uint32_t l_size = lane_u32(qsize, last);
                            found += valid;
                            pos = lane_u32(cand, last) + l_size;
                            // the size most blocks have stays the guess: an odd one (a varint one byte
                            // longer, one narrower miniblock) is adopted only when it comes twice in a row
                            if (l_size != guess) {
                                if (guess == 0 || l_size == alt) guess = l_size;
                                alt = l_size;
                            }
                            last_vals = lane_u32(bvals, last);
                            continue;
                        }
                    }
                    if (parsed) sh.blk_pos[bidx] = NO_POS; // the full parse writes the slots itself
                    int32_t err = PQD_OK;
                    int64_t err_arg = 0;
                    bool at_end = false;
                    uint32_t size = 0, have_mb = 0;
                    if (parsed) {
                        if (cand >= end) {
                            at_end = true;
                        } else {
                            uint64_t z;
                            int k = ring_uvarint32(ring8, cand, end, &z);
                            if (k == 0) { err = PQD_ERR_UNEXPECTED_EOF; err_arg = PQD_EOF_DELTA_MIN_DELTA; }
                            else if (k < 0) { err = PQD_ERR_VARINT_OVERFLOW; err_arg = (int64_t)(cand - mis); }
                            else {
                                const int64_t md = unzigzag(z);
                                const uint32_t wpos = cand + k;
                                const uint32_t nbw = end - wpos < w_nmb ? end - wpos : w_nmb;
                                uint32_t off = wpos + nbw;
                                have_mb = nbw < need_mb ? nbw : need_mb;
                                for (uint32_t m = 0; m < have_mb; m++) {
                                    const uint32_t bw = ring8[(wpos + m) & RING_MASK];
                                    if (bw > Elem<U>::max_bw) { err = PQD_ERR_INVALID_BIT_WIDTH; err_arg = bw; have_mb = m; break; }
                                    Slot sl;
                                    sl.min_delta = md; sl.off = off; sl.bw = bw;
                                    sh.slots[bidx * w_nmb + m] = sl;
                                    off += (w_vpm * bw) >> 3;
                                }
                                size = off - cand;
                                sh.blk_end[bidx] = off;
                            }
                        }
                    }
                    const bool bad = parsed && (err != PQD_OK || at_end);
                    const bool good = parsed && !bad;
                    const unsigned long long okmask = __ballot(good && size == guess);
                    const unsigned long long goodmask = __ballot(good);
                    const uint32_t lead = okmask == ~0ull ? 64u : (uint32_t)__builtin_ctzll(~okmask);
                    uint32_t valid = lead < 64 ? lead + (uint32_t)((goodmask >> lead) & 1) : 64u;
                    if (valid > target - found) valid = target - found;
                    if (valid == 0) { // the block the stream really continues with is unreadable
                        const int32_t e0 = __shfl(err, 0);
                        const int64_t a0 = __shfl(err_arg, 0);
                        if (e0 != PQD_OK) { wstatus = e0; warg = a0; }
                        else { wstatus = PQD_ERR_MISSING_VALUES; }
                        ended = true;
                        break;
                    }
                    const uint32_t last = valid - 1;
                    const uint32_t l_have = lane_u32(have_mb, last);
                    const uint32_t l_need = lane_u32(need_mb, last);
                    const uint32_t l_vals = lane_u32(bvals, last);
                    const uint32_t l_size = lane_u32(size, last);
                    const uint32_t l_cand = lane_u32(cand, last);
                    found += valid;
                    pos = l_cand + l_size;
                    guess = l_size;
                    last_vals = l_have < l_need ? l_have << w_vpm_shift : l_vals;
                    if (l_have < l_need) { // the widths ran out: nothing follows this block
                        wstatus = PQD_ERR_MISSING_VALUES;
                        ended = true;
                        break;
                    }
                }
                // blocks found by quick rounds: every lane parses its block in full and writes its slots
                if (lane < found && sh.blk_pos[lane] != NO_POS) {
                    const uint32_t at = sh.blk_pos[lane];
                    uint64_t qlo, qhi;
                    ring_16(sh.ring, at, qlo, qhi);
                    uint32_t k, wd;
                    quick_header(qlo, qhi, k, wd);
                    // The four width bytes are fetched again from the ring (two dwords and a funnel shift) and they and
                    // the offsets go to LDS before the min delta is worked out: taken out of the sixteen bytes with
                    // 64-bit shifts that depend on k and kept across the varint, they did not fit the kernel's 64
                    // registers, the compiler parked them in scratch memory, and the reload waited for every global
                    // store this wavefront still had in flight, with the other three wavefronts at the barrier behind it.
                    {
                        const uint32_t wp = at + k, wdw = (wp & RING_MASK) >> 2;
                        wd = __builtin_amdgcn_alignbit(sh.ring[wdw + 1], sh.ring[wdw], (wp & 3) * 8); // the mirror covers wdw + 1
                    }
                    uint32_t off = at + k + 4;
#pragma unroll
                    for (uint32_t m = 0; m < 4; m++) {
                        const uint32_t bw = (wd >> (8 * m)) & 0xff;
                        sh.slots[lane * 4 + m].off = off;
                        sh.slots[lane * 4 + m].bw = bw;
                        off += (w_vpm >> 3) * bw;
                    }
                    sh.blk_end[lane] = off;
                    asm volatile("" ::: "memory");
                    const int64_t md = unzigzag(quick_uvarint(qlo, qhi, k));
#pragma unroll
                    for (uint32_t m = 0; m < 4; m++) sh.slots[lane * 4 + m].min_delta = md;
                }
                // keep whole tiles unless the stream or the page ends inside the table
                const bool page_end = found == blocks_left && !ended;
                if (!ended && !page_end) found = (found / w_nb_tile) * w_nb_tile;
                const uint32_t got = found ? ((found - 1) << w_blk_shift) + (found == blocks_left || ended ? last_vals : w_bs) : 0;
                const uint32_t tiles = (got + TILE - 1) / TILE;
                if (tid == 0) {
                    FastCtl& f = sh.fc;
                    f.tiles = tiles;
                    f.got = got;
                    f.tile_lo[0] = lo;
                    for (uint32_t t = 1; t <= tiles; t++) {
                        uint32_t lastblk = t * w_nb_tile < found ? t * w_nb_tile : found;
                        f.tile_lo[t] = sh.blk_end[lastblk - 1];
                    }
                    const uint32_t fin = found ? sh.blk_end[found - 1] : lo;
                    f.consumed = fin < end ? fin : end;
                    f.status = wstatus;
                    f.arg = wstatus == PQD_ERR_MISSING_VALUES ? (int64_t)(rem - got) : warg;
                    f.stop = (ended || page_end) ? 1 : 0;
                }
            }
            lds_barrier(); // B2: table visible
            tbl_tiles = sh.fc.tiles;
            tbl_got = sh.fc.got;
            tbl_next = 0;
            if (tbl_tiles == 0) { // nothing decodable: the pending status ends the page
                status = sh.fc.status; status_arg = sh.fc.arg; consumed = sh.fc.consumed;
                break;
            }
        }

        {
            // a tile that reaches past what is resident (wide miniblocks after narrow ones): bring the rest in now
            const uint32_t tile_end = sh.fc.tile_lo[tbl_next + 1];
            if (tile_end > c_next * CHUNK) {
                const uint32_t c_last = (tile_end - 1) / CHUNK;
                while (c_next <= c_last) {
                    const uint32_t v = c_next * CHUNK + threadIdx.x * 16u;
                    const uint32_t vs = v < end ? v : 0u;
                    const u32x4 w = *(gptr<const u32x4>)(base + vs);
                    ring_put(sh.ring, v, end, w);
                    c_next++;
                }
                lds_barrier();
            }
        }
        // The thread number, opaque to the optimiser from here on: whatever the tile derives from it is
        // computed again per tile (a few instructions) instead of being kept in registers across the walk.
        uint32_t tq = tid;
        asm volatile("" : "+v"(tq));
        const uint32_t left = tbl_got - tbl_next * TILE;
        const uint32_t count = left < TILE ? left : TILE;
        const uint32_t j = tq * 4;
        const uint32_t slot = tbl_next * slots_tile + (j >> vpm_shift);
        U r[4];
        if (count < TILE) {
            // The last tile of the page, kept off the loop's back edge: the pipelined path
            // below relies on every thread issuing the same memory operations per iteration.
            const U e = tile_values<U, true>(sh.ring, sh.wtot, j < count ? sh.slots[slot] : Slot{0, 0, 0}, count, j & (vpm - 1), carry, r);
            tile_store_shifted_partial<U>(out, done, count, e, r);
            done += count;
            status = sh.fc.status; status_arg = sh.fc.arg; consumed = sh.fc.consumed;
            break;
        }

#if PQD_DELTA_ROTATE > 0
        // The hardware lets the oldest wavefront of a SIMD issue first: of the workgroups that share a compute unit
        // the first to arrive then finishes first (in 0.72 M cycles of the last one's 1.28 M, eight to a unit) and
        // the last ones decode their tails with the unit half empty.  A priority that goes round the workgroups
        // keeps them together; those of one unit are 256 apart in a grid that fills the chip once, which is what
        // the term from blockIdx tells apart (any other grid gets some other spread over the four levels, as good).
        if ((done & (PQD_DELTA_ROTATE * TILE - 1)) == 0) {
            // (the turn a workgroup has, doubled, plus its place among the eight: between two workgroups of one level the
            // older one still goes first, so the later a workgroup arrived the more of its time it spends a level up)
            const uint32_t slot8 = (blockIdx.x >> 8) & 7, rot8 = ((done / (PQD_DELTA_ROTATE * TILE)) + slot8) & 7;
            const uint32_t level = ((2 * rot8 + slot8) * 4) / 22;
            if (level == 0) __builtin_amdgcn_s_setprio(0);
            else if (level == 1) __builtin_amdgcn_s_setprio(1);
            else if (level == 2) __builtin_amdgcn_s_setprio(2);
            else __builtin_amdgcn_s_setprio(3);
        }
#endif
        // ---- prefetch: chunks the NEXT tile wants resident, requested now and written to
        // the ring at the top of the next iteration.  When there is something to fetch both
        // loads are issued (the second of a dummy address if one chunk is enough) and every
        // thread then issues exactly two stores, so the wait for these loads at the top of
        // the next iteration is a fixed "all but the two youngest operations"; a tile that
        // fetches nothing issues no load, and the next one then waits for nothing: its
        // stores go out behind this tile's without anybody having waited for these.
        // Addresses are clamped, so a load is always legal; ring_put discards what lies
        // past the end. ----
        const uint32_t lo_next = sh.fc.tile_lo[tbl_next + 1];
        const uint32_t c_want = lo_next / CHUNK + LOOK / CHUNK; // last chunk the next tile wants
        pf_n = c_next > c_want ? 0u : (c_want + 1 - c_next > 2 ? 2u : c_want + 1 - c_next);
        // Fewer, larger reads: while tiles are small nothing is fetched as long as the ring is still
        // LOOK_MIN ahead of the next tile, and then two chunks at once.  A read inside a saturated
        // stream of stores is dear whatever its size, so 8 KB every few tiles beats 4 KB twice as often.
        if (c_next > lo_next / CHUNK + LOOK_MIN / CHUNK && lo_next - lo <= 3072) pf_n = 0;
        pf_v = c_next * CHUNK + tq * 16u;
        if (pf_n > 0) {
            const uint32_t pv1 = pf_v + CHUNK;
            pf0 = *(gptr<const u32x4>)(base + (pf_v < end ? pf_v : 0u));
            pf1 = *(gptr<const u32x4>)(base + (pf_n > 1 && pv1 < end ? pv1 : 0u));
        }

        const U e = tile_values<U, true>(sh.ring, sh.wtot, sh.slots[slot], TILE, j & (vpm - 1), carry, r, tq);
        tile_store_shifted_full<U>(out, done, e, r, tq);
        done += count;
        tbl_next++;
        lo = lo_next;
        if (tbl_next == tbl_tiles && sh.fc.stop) {
            status = sh.fc.status; status_arg = sh.fc.arg; consumed = sh.fc.consumed;
            stop = 1;
        }
    }

    __syncthreads();
    if (status == PQD_ERR_MISSING_VALUES || status == PQD_ERR_UNEXPECTED_EOF || status == PQD_ERR_VARINT_OVERFLOW ||
        status == PQD_ERR_INVALID_BIT_WIDTH) {
        for (uint32_t k = 1 + done + tid; k < total; k += DT) out[k] = 0;
    }
    if (tid == 0) {
        if (done > 0) out[done] = carry; // the element a full last tile leaves to its successor
        res->status = status;
        res->arg = status_arg;
        res->total = total;
        res->out_values = total;
        res->out_bytes = (uint64_t)total * sizeof(U);
        res->consumed = src_skip + (consumed - mis);
        res->n_runs = 0;
        res->aux = FAST_DONE;
#ifdef PQD_DELTA_TIMING
        // where and when this workgroup ran, in place of the fields a caller reads (tools/delta_timing.py)
        uint32_t xcc, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        res->arg = (int64_t)t_entry;
        res->consumed = __builtin_readcyclecounter();
        res->out_values = ((uint64_t)(xcc & 0xf) << 32) | hw;
#endif
    }
}

template <typename U>
__global__ __launch_bounds__(DT) void delta_decode_kernel(const DevPage* __restrict__ pages, DevResult* __restrict__ results,
                                                         const uint32_t* __restrict__ items) {
    __shared__ Shared sh;
    const uint32_t item = items[blockIdx.x];
    const DevPage& pg = pages[item];
    DevResult* res = &results[item];
    const uint32_t tid = threadIdx.x;
    const uint8_t* ring8 = reinterpret_cast<const uint8_t*>(sh.ring);

    // A chained item starts where the previous stream of the same page ended.
    uint64_t src_skip = 0;
    if (pg.chain >= 0) src_skip = results[pg.chain].consumed;
    const uint8_t* src = pg.src + src_skip;
    const uint64_t src_len = pg.src_len > src_skip ? pg.src_len - src_skip : 0;
    const uint64_t mis = reinterpret_cast<uintptr_t>(src) & 15;
    gptr<const uint8_t> base = as_global(src - mis);       // ring-space 0
    const uint64_t end = mis + src_len;    // ring-space end of page
    gptr<U> out = (gptr<U>)(pg.dst);
    if (res->aux == FAST_DONE || res->aux == FLAT_PENDING) return; // uniform: the fast or the tile-parallel kernels decoded this page

    uint64_t c_next = 0;
    ring_fill(sh.ring, base, end, mis, c_next);
    __syncthreads();

    Walk wk;
    uint64_t num_mb_l0 = 0; // lane 0 only: miniblocks per block
    if (tid == 0) {
        Ctl c;
        int64_t block_size = 0, num_mb = 0;
        parse_header<U>(ring8, mis, end, pg.dst_cap, out, c, block_size, num_mb);
        wk.pos = c.lo; wk.end = end; wk.bw_pos = 0; wk.min_delta = 0; wk.cur_off = c.lo; wk.cur_bw = 0; wk.cur_rem = 0;
        wk.cur_within = 0; wk.mb_in_block = 0; wk.nbw_in_block = 0; wk.mb_global = 0;
        num_mb_l0 = (uint64_t)num_mb;
        sh.ctl = c;
    }
    __syncthreads();

    const uint64_t total = sh.ctl.total;
    const uint32_t vpm = sh.ctl.vpm;
    int stop = sh.ctl.stop;
    U carry = (U)sh.ctl.first;
    uint64_t done = 0;                 // deltas decoded so far
    const uint64_t n_deltas = total > 0 ? total - 1 : 0;

    while (!stop) {
        ring_fill(sh.ring, base, end, sh.ctl.lo, c_next);
        __syncthreads(); // B1: ring bytes visible

        if (tid == 0) {
            // ---- header walk for the next tile (binary_packed.go:367-400) ----
            Ctl& c = sh.ctl;
            uint32_t want = (uint32_t)(n_deltas - done < TILE ? n_deltas - done : TILE);
            uint32_t got = 0, nslots = 0;
            c.slot_base = wk.cur_rem > 0 ? wk.mb_global - 1 : wk.mb_global;
            if (vpm == 0) {
                // Degenerate header (more miniblocks than values per block): every
                // miniblock holds no value and no byte; the reference reads block
                // headers until the input ends, then reports the missing values.
                while (c.status == PQD_OK && wk.pos < wk.end) {
                    uint64_t z;
                    int k = ring_uvarint(ring8, wk.pos, wk.end, &z);
                    if (k == 0) { c.status = PQD_ERR_UNEXPECTED_EOF; c.arg = PQD_EOF_DELTA_MIN_DELTA; break; }
                    if (k < 0) { c.status = PQD_ERR_VARINT_OVERFLOW; c.arg = (int64_t)(wk.pos - mis); break; }
                    wk.pos += k;
                    uint64_t nb = wk.end - wk.pos < num_mb_l0 ? wk.end - wk.pos : num_mb_l0;
                    wk.pos += nb;
                    if (wk.pos + 64 > c.lo + LOOK) break; // come back after a refill
                }
                c.lo = wk.pos;
                if (c.status == PQD_OK && wk.pos >= wk.end) { c.status = PQD_ERR_MISSING_VALUES; c.arg = (int64_t)(n_deltas - done); }
                if (c.status != PQD_OK) c.stop = 1;
                c.count = 0;
                c.consumed = wk.pos;
            } else {
                while (got < want && c.status == PQD_OK) {
                    if (wk.cur_rem == 0) {
                        if (wk.mb_in_block == wk.nbw_in_block) {
                            if (wk.pos >= wk.end) break; // input exhausted: values are missing
                            uint64_t z;
                            int k = ring_uvarint(ring8, wk.pos, wk.end, &z);
                            if (k == 0) { c.status = PQD_ERR_UNEXPECTED_EOF; c.arg = PQD_EOF_DELTA_MIN_DELTA; break; }
                            if (k < 0) { c.status = PQD_ERR_VARINT_OVERFLOW; c.arg = (int64_t)(wk.pos - mis); break; }
                            wk.pos += k;
                            wk.min_delta = unzigzag(z);
                            uint64_t nb = wk.end - wk.pos < num_mb_l0 ? wk.end - wk.pos : num_mb_l0;
                            wk.bw_pos = wk.pos;
                            wk.pos += nb;
                            wk.mb_in_block = 0;
                            wk.nbw_in_block = (uint32_t)nb;
                            if (nb <= BW_CACHE)
                                for (uint32_t m = 0; m < nb; m++) sh.blk_bw[m] = ring8[(wk.bw_pos + m) & RING_MASK];
                            if (nb == 0) continue; // nothing but a min delta was left
                        }
                        uint32_t bw = wk.nbw_in_block <= BW_CACHE ? sh.blk_bw[wk.mb_in_block]
                                                                  : (uint32_t)base[wk.bw_pos + wk.mb_in_block];
                        wk.mb_in_block++;
                        if (bw > Elem<U>::max_bw) { c.status = PQD_ERR_INVALID_BIT_WIDTH; c.arg = bw; break; }
                        uint64_t mbsize = (uint64_t)vpm * bw / 8;
                        uint64_t avail = wk.end - wk.pos < mbsize ? wk.end - wk.pos : mbsize;
                        wk.cur_off = wk.pos;
                        wk.pos += avail;
                        wk.cur_bw = bw;
                        wk.cur_rem = vpm;
                        wk.cur_within = 0;
                        wk.mb_global++;
                    }
                    uint32_t take = wk.cur_rem < want - got ? wk.cur_rem : want - got;
                    Slot s;
                    s.min_delta = wk.min_delta; s.off = (uint32_t)wk.cur_off; s.bw = wk.cur_bw;
                    sh.slots[nslots++] = s;
                    wk.cur_rem -= take;
                    wk.cur_within += take;
                    got += take;
                }
                c.count = got;
                c.consumed = wk.pos;
                if (c.status == PQD_OK && got < want) { c.status = PQD_ERR_MISSING_VALUES; c.arg = (int64_t)(n_deltas - done - got); }
                if (c.status != PQD_OK || done + got >= n_deltas) c.stop = 1;
                // where the next tile's bytes start
                if (wk.cur_rem > 0) c.lo = wk.cur_off + (((uint64_t)wk.cur_within * wk.cur_bw) >> 3);
                else c.lo = wk.pos;
                // a finished tile whose values are all decoded leaves the last
                // miniblock consumed in full (binary_packed.go:392-397)
            }
        }
        __syncthreads(); // B2: slots and ctl visible

        const uint32_t count = sh.ctl.count;
        stop = sh.ctl.stop;
        const uint32_t j = tid * 4;
        uint32_t slot = 0, within = 0;
        if (j < count) {
            uint64_t g = done + j;                       // global delta index
            uint32_t mb = (uint32_t)(g / vpm);
            within = (uint32_t)(g - (uint64_t)mb * vpm);
            slot = mb - sh.ctl.slot_base;
        }
        decode_tile<U>(sh.ring, sh.wtot, j < count ? sh.slots[slot] : Slot{0, 0, 0}, out, done, count, within, carry);
        done += count;
    }

    __syncthreads();
    // Values the stream did not provide stay zero (the reference zero-fills the
    // output before decoding, delta.go:61-72).
    const int32_t status = sh.ctl.status;
    if (status == PQD_ERR_MISSING_VALUES || status == PQD_ERR_UNEXPECTED_EOF || status == PQD_ERR_VARINT_OVERFLOW ||
        status == PQD_ERR_INVALID_BIT_WIDTH) {
        for (uint64_t k = 1 + done + tid; k < total; k += DT) out[k] = 0;
    }
    if (tid == 0) {
        res->status = status;
        res->arg = sh.ctl.arg;
        res->total = total;
        // the reference returns the full-length slice even when it fails part way
        res->out_values = total;
        res->out_bytes = total * sizeof(U);
        res->consumed = src_skip + (sh.ctl.consumed - mis);
        res->n_runs = 0;
        res->aux = 0;
    }
}

// ---------------------------------------------------------------------------
// Tile-parallel path for batches of few pages.  One workgroup per page leaves most of the
// chip idle when a batch holds a few hundred pages, and its tile loop is a chain of LDS round
// trips and barriers (about 1.3 us per 1024 values whatever the width).  Pages with the
// writers' usual geometry (blocks of 128 values, four miniblocks) are decoded here instead:
//   delta_flat_walk   per page, finds where every block starts out of a 16 KB window of the
//                     stream in LDS: lane l guesses that block l starts l * S bytes after the
//                     first (S: the size most blocks have had), reads the bytes around there,
//                     and the guess holds for the lanes whose predecessors all measured S;
//                     behind a block a few bytes longer or shorter (a min delta whose varint
//                     grew) the lanes look again, moved by the difference, out of the bytes
//                     they already hold.  64 blocks per round when the widths are steady.
//                     Anything unusual -- a
//                     header problem, a width over the value width, data cut short, sizes that
//                     keep changing -- leaves the page to the kernels above, untouched.
//   delta_flat_tile   a workgroup per tile of 1024 deltas (8 blocks), twice: first only the
//                     sum of the tile's deltas, then the values themselves, each tile starting
//                     from the first value plus the sums of the tiles before it.  A tile
//                     loads its bytes, unpacks, scans, stores and is done: the form in which
//                     reads do not queue behind the page's own stores.
// ---------------------------------------------------------------------------

constexpr uint32_t FLAT_STAGE = 8 * (10 + 4 + 4 * 256) + 64; // most bytes eight blocks take, and slack

__device__ inline uint32_t lane_u32_or_u64(uint32_t v) { return lane_u32(v, 63); }
__device__ inline uint64_t lane_u32_or_u64(uint64_t v) { return lane_u64(v, 63); }

constexpr uint32_t FLAT_STRIDE = 16384;            // the walk's windows start this far apart ...
constexpr uint32_t FLAT_WIN = FLAT_STRIDE + 64;    // ... and overlap by 64 bytes, so that a block header lies wholly in one of them
constexpr uint32_t FLAT_WIN_LOADS = (FLAT_WIN + DT * 16 - 1) / (DT * 16);

struct FlatWalkState {
    uint32_t pos, found, guess, alt, rounds, mark, again;
    int32_t ok, done;
};

template <typename U>
__global__ __launch_bounds__(DT) void delta_flat_walk(const DevPage* __restrict__ pages, DevResult* __restrict__ results,
                                                      const uint32_t* __restrict__ items) {
    __shared__ __attribute__((aligned(16))) uint32_t wins[2][FLAT_WIN_LOADS * DT * 4 + 8];
    __shared__ uint8_t head[64];
    __shared__ FlatWalkState st;
    const uint32_t item = items[blockIdx.x];
    const DevPage& pg = pages[item];
    DevResult* res = &results[item];
    const uint32_t tid = threadIdx.x, lane = tid & 63;
    uint64_t src_skip = 0;
    if (pg.chain >= 0) src_skip = results[pg.chain].consumed;
    gptr<const uint8_t> src = as_global(pg.src) + src_skip;
    const uint64_t src_len = pg.src_len > src_skip ? pg.src_len - src_skip : 0;
    gptr<uint32_t> pos_out = as_global(pg.flat_pos);
    bool ok = pg.flat_pos != nullptr && src_len >= 4 && src_len <= FAST_MAX_BYTES;
    // ---- page header (binary_packed.go:409-452): anything but the plain case is not for this path ----
    if (tid < 64) head[tid] = ok && tid < src_len ? src[tid] : (uint8_t)0;
    __syncthreads();
    uint64_t hv[4] = {0, 0, 0, 0};
    uint32_t p = 0;
    for (int f = 0; f < 4 && ok; f++) {
        uint64_t x = 0;
        uint32_t sft = 0, i = 0;
        for (;; i++) {
            if (p + i >= 40 || p + i >= src_len || i >= 9) { ok = false; break; }
            const uint32_t b = head[p + i];
            x |= (uint64_t)(b & 0x7f) << sft;
            sft += 7;
            if (b < 0x80) break;
        }
        hv[f] = x;
        p += i + 1;
    }
    const uint64_t total = hv[2];
    const int64_t first = unzigzag(hv[3]);
    ok = ok && hv[0] == 128 && hv[1] == 4 && total >= 2 && total <= 0x7fffffffull && total * sizeof(U) <= pg.dst_cap;
    if (sizeof(U) == 4) ok = ok && first >= -2147483647LL - 1 && first <= 2147483647LL;
    const uint32_t n_deltas = ok ? (uint32_t)total - 1 : 0;
    const uint32_t n_blocks = (n_deltas + 127) / 128;
    ok = ok && (uint64_t)n_blocks + 1 <= pg.flat_cap;
    if (!ok) { // uniform
        if (tid == 0) res->aux = 0;
        return;
    }
    // ---- block starts.  Positions below are in "window space": stream offset + mis, so that
    // window space 0 is the 16-byte boundary at or before the stream's first byte. ----
    const uint32_t mis = (uint32_t)((uintptr_t)(pg.src + src_skip) & 15);
    gptr<const uint8_t> base = src - mis;
    const uint32_t end = mis + (uint32_t)src_len;
    if (tid == 0) { st.pos = mis + p; st.found = 0; st.guess = 0; st.alt = 0; st.rounds = 0; st.mark = 0; st.again = 0; st.ok = 1; st.done = 0; }
    __syncthreads();
    // Window i holds window-space bytes [i * FLAT_STRIDE, i * FLAT_STRIDE + FLAT_WIN); the walk goes
    // through the windows in order, and window i + 1 is on its way from memory while window i is walked.
    auto load_window = [&](uint32_t wi, u32x4 (&regs)[FLAT_WIN_LOADS]) {
#pragma unroll
        for (uint32_t u = 0; u < FLAT_WIN_LOADS; u++) {
            const uint64_t at = (uint64_t)wi * FLAT_STRIDE + (uint64_t)(u * DT + tid) * 16;
            regs[u] = (u32x4)(0u);
            if (at < end) regs[u] = *(gptr<const u32x4>)(base + at);
        }
    };
    auto put_window = [&](uint32_t wi, const u32x4 (&regs)[FLAT_WIN_LOADS]) {
#pragma unroll
        for (uint32_t u = 0; u < FLAT_WIN_LOADS; u++) *reinterpret_cast<u32x4*>(&wins[wi & 1][(u * DT + tid) * 4]) = regs[u];
    };
    uint32_t wi = (mis + p) / FLAT_STRIDE;
    {
        u32x4 regs[FLAT_WIN_LOADS];
        load_window(wi, regs);
        put_window(wi, regs);
    }
    __syncthreads();
    for (;;) {
        const uint32_t w0 = wi * FLAT_STRIDE;
        const uint32_t* win = wins[wi & 1];
        u32x4 next[FLAT_WIN_LOADS];
        load_window(wi + 1, next);
        if (tid < 64) {
            uint32_t pos = st.pos, found = st.found, guess = st.guess, alt = st.alt, rounds = st.rounds, mark = st.mark, again = st.again;
            bool good_walk = true, done = false;
            const bool to_end = w0 + FLAT_WIN >= end; // the window reaches the end of the stream
            while (found < n_blocks) {
                const uint32_t bidx = found + lane;
                const uint64_t cand64 = (uint64_t)pos + (uint64_t)lane * guess;
                const bool resident = cand64 + 24 <= (uint64_t)w0 + FLAT_WIN || (to_end && cand64 + 5 <= end);
                if (!lane_u32((uint32_t)resident, 0)) break; // the next block lies past the window: move it
                uint32_t cand = (uint32_t)cand64;
                const bool live = bidx < n_blocks && resident;
                // 28 bytes from the dword boundary at or before cand - 4: room to look at the header a few
                // bytes before or after the place the regular spacing says
                const uint32_t back = live && cand >= w0 + 4 ? 4u : 0u;
                const uint32_t rs = live ? cand - back - w0 : 0u, d = rs >> 2;
                uint32_t x[7];
#pragma unroll
                for (int i = 0; i < 7; i++) x[i] = win[d + i];
                int32_t shift = 0;      // this lane's block starts `shift` bytes after its candidate
                uint32_t size = 0;
                bool good = false;
                uint32_t from = 0;      // lanes from here on are (re)parsed
                uint32_t valid = 0;
                for (int pass = 0; pass < 16; pass++) {
                    if (lane >= from) {
                        size = 0;
                        good = false;
                        const int32_t o = (int32_t)((rs & 3) + back) + shift;   // header's byte offset in x[]
                        if (live && o >= 0 && o <= 11 && (uint64_t)cand + shift + 5 <= end) {
                            const uint32_t q = (uint32_t)o >> 2, sh8 = ((uint32_t)o & 3) * 8;
                            const uint32_t a0 = q == 0 ? x[0] : (q == 1 ? x[1] : x[2]), a1 = q == 0 ? x[1] : (q == 1 ? x[2] : x[3]),
                                           a2 = q == 0 ? x[2] : (q == 1 ? x[3] : x[4]), a3 = q == 0 ? x[3] : (q == 1 ? x[4] : x[5]),
                                           a4 = q == 0 ? x[4] : (q == 1 ? x[5] : x[6]);
                            const uint64_t lo = ((uint64_t)__builtin_amdgcn_alignbit(a2, a1, sh8) << 32) | __builtin_amdgcn_alignbit(a1, a0, sh8);
                            const uint64_t hi = ((uint64_t)__builtin_amdgcn_alignbit(a4, a3, sh8) << 32) | __builtin_amdgcn_alignbit(a3, a2, sh8);
                            uint32_t k, wd;
                            if (quick_header(lo, hi, k, wd)) {
                                const uint32_t bvals = n_deltas - bidx * 128 < 128 ? n_deltas - bidx * 128 : 128;
                                const uint32_t need = (bvals + 31) / 32;
                                uint32_t bits = 0, top = 0;
                                for (uint32_t m = 0; m < need; m++) {
                                    const uint32_t bw = (wd >> (8 * m)) & 0xff;
                                    bits += bw;
                                    top = bw > top ? bw : top;
                                }
                                size = k + 4 + 4 * bits;
                                // a miniblock cut short is zero padded by the reference: not on this path
                                good = top <= Elem<U>::max_bw && (uint64_t)cand + shift + size <= end;
                            }
                        }
                    }
                    // lanes up to the first whose block is not of the usual size stand where the stream has them
                    const unsigned long long goodmask = __ballot(good);
                    const unsigned long long same = __ballot(good && size == guess) | ((1ull << from) - 1);
                    const uint32_t lead = same == ~0ull ? 64u : (uint32_t)__builtin_ctzll(~same);
                    if (lead >= 64) { valid = 64; break; }
                    if (!((goodmask >> lead) & 1)) { valid = lead; break; }
                    // that block is a true one of another size: the lanes after it look again, moved by the difference
                    const int32_t moved = lane_u32((uint32_t)shift, lead) + (int32_t)lane_u32(size, lead) - (int32_t)guess;
                    valid = lead + 1;
                    if (guess == 0 || moved < -4 || moved > 4 || lead == 63) break;
                    if (lane > lead) shift = moved;
                    from = lead + 1;
                    again++;
                }
                cand = (uint32_t)((int32_t)cand + shift);
                if (valid > n_blocks - found) valid = n_blocks - found;
                if (valid == 0) { good_walk = false; break; }
                if (lane < valid) pos_out[bidx] = cand - mis;
                const uint32_t last = valid - 1;
                const uint32_t l_size = lane_u32(size, last);
                pos = lane_u32(cand, last) + l_size;
                found += valid;
                // The size most blocks have stays the guess.  Another one is adopted when the last two
                // blocks of a round agree on it, or when a round of one block meets it twice in a row.
                if (l_size != guess) {
                    const uint32_t before = valid >= 2 ? lane_u32(size, valid - 2) : alt;
                    if (guess == 0 || before == l_size) guess = l_size;
                    alt = valid >= 2 ? 0u : l_size;
                } else {
                    alt = 0;
                }
                // Block sizes that keep changing make this walk slower than decoding the page the
                // usual way (a round is a dependent chain of LDS reads and cross-lane moves, half a
                // microsecond and more).  Rounds that end for another reason than the window's or the
                // page's end are counted, and every four of them must have brought 16 blocks.
                // every block of another size costs a pass of its own: with more than one in four the page is
                // decoded sooner the usual way
                // wide blocks: few fit a window, the walk crawls, and the page-per-workgroup kernel is at its best
                if (guess > 640) { good_walk = false; break; }
                const bool early = found <= 256; // the verdict on a page's regularity is reached at its start: giving up late wastes the walk so far
                if (early && again > 8 + found / 4) { good_walk = false; break; }
                const bool cut_short = valid < 64 && found < n_blocks && lane_u32((uint32_t)resident, valid) != 0;
                if (cut_short && (++rounds & 3) == 0) {
                    if (early && found < mark + 16) { good_walk = false; break; }
                    mark = found;
                }
            }
            if (found == n_blocks && good_walk) done = true;
            if (tid == 0) {
                st.pos = pos; st.found = found; st.guess = guess; st.alt = alt; st.rounds = rounds; st.mark = mark; st.again = again;
                st.ok = good_walk ? 1 : 0;
                st.done = done ? 1 : 0;
            }
        }
        put_window(wi + 1, next); // the other buffer: nobody reads it during this walk
        __syncthreads();
        if (!st.ok || st.done) break;
        wi++;
    }
    if (tid == 0) {
        if (st.ok && st.done) {
            pos_out[n_blocks] = st.pos - mis;
            as_global(pg.flat_sum)[(n_deltas + TILE - 1) / TILE] = (uint64_t)first; // behind the tile sums
            res->status = PQD_OK;
            res->arg = 0;
            res->out_values = total;
            res->out_bytes = total * sizeof(U);
            res->bw = 0;
            res->total = total;
            res->n_runs = n_blocks;
            res->consumed = src_skip + (st.pos - mis);
            res->aux = FLAT_PENDING;
        } else {
            res->aux = 0;
        }
    }
}

// grid.x = pages, grid.y = tile slots.  STORE false: flat_sum[tile] = sum of the tile's deltas;
// STORE true: the tile starts from the first value plus the sums of the tiles before it, and the
// values are written.
template <typename U, bool STORE>
__global__ __launch_bounds__(DT) void delta_flat_tile(const DevPage* __restrict__ pages, DevResult* __restrict__ results,
                                                      const uint32_t* __restrict__ items) {
    __shared__ __attribute__((aligned(16))) uint32_t stage[FLAT_STAGE / 4 + 8];
    __shared__ uint64_t wtot[DT / 64];
    __shared__ uint32_t bp[9];
    const uint32_t item = items[blockIdx.x];
    const DevResult& rr = results[item];
    if (rr.aux != FLAT_PENDING) return;
    const DevPage& pg = pages[item];
    uint64_t src_skip = 0;
    if (pg.chain >= 0) src_skip = results[pg.chain].consumed;
    gptr<const uint8_t> src = as_global(pg.src) + src_skip;
    gptr<const uint32_t> bpos = (gptr<const uint32_t>)as_global(pg.flat_pos);
    gptr<uint64_t> sums = as_global(pg.flat_sum);
    gptr<U> out = (gptr<U>)(pg.dst);
    const uint32_t n_deltas = (uint32_t)rr.total - 1, n_blocks = (uint32_t)rr.n_runs;
    const uint32_t n_tiles = (n_deltas + TILE - 1) / TILE;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (uint32_t tile = blockIdx.y; tile < n_tiles; tile += gridDim.y) {
        const uint32_t b0 = tile * 8;
        const uint32_t nb = n_blocks - b0 < 8 ? n_blocks - b0 : 8;
        // one trip to memory for everything that is known before the bytes: the block starts of the
        // tile and (second pass) the sums of the tiles before it
        if (tid <= 8) bp[tid] = bpos[b0 + (tid < nb ? tid : nb)];
        U part = 0;
        if (STORE) for (uint32_t t = tid; t < tile; t += DT) part += (U)sums[t];
        __syncthreads();
        const uint32_t p0 = bp[0], p1 = bp[nb];
        // the tile's bytes, from the 16-byte boundary at or before p0; what is read past p1 is never used
        const uint64_t a0 = ((uintptr_t)(pg.src + src_skip) + p0) & 15;
        gptr<const uint8_t> from = src + p0 - a0;
        const uint32_t bytes = (uint32_t)a0 + (p1 - p0);
        for (uint32_t c = tid * 16; c < bytes; c += DT * 16)
            *reinterpret_cast<u32x4*>(&stage[c / 4]) = *(gptr<const u32x4>)(from + c);
        __syncthreads();
        const uint32_t count = n_deltas - tile * TILE < TILE ? n_deltas - tile * TILE : TILE;
        const uint32_t j = tid * 4;
        U v[4] = {0, 0, 0, 0};
        if (j < count) {
            const uint32_t blk = tid >> 5, mb = (tid >> 3) & 3, within = (tid & 7) * 4;
            const uint32_t at = (uint32_t)a0 + (bp[blk] - p0);     // block header in the stage
            const uint32_t d = at >> 2, sh8 = (at & 3) * 8;
            const uint32_t w0 = stage[d], w1 = stage[d + 1], w2 = stage[d + 2], w3 = stage[d + 3], w4 = stage[d + 4];
            const uint64_t lo = ((uint64_t)__builtin_amdgcn_alignbit(w2, w1, sh8) << 32) | __builtin_amdgcn_alignbit(w1, w0, sh8);
            const uint64_t hi = ((uint64_t)__builtin_amdgcn_alignbit(w4, w3, sh8) << 32) | __builtin_amdgcn_alignbit(w3, w2, sh8);
            uint32_t k, wd;
            quick_header(lo, hi, k, wd);
            const U md = (U)unzigzag(quick_uvarint(lo, hi, k));
            uint32_t off = at + k + 4;
            for (uint32_t m = 0; m < mb; m++) off += 4 * ((wd >> (8 * m)) & 0xff);
            const uint32_t bw = (wd >> (8 * mb)) & 0xff;
            const uint32_t bit0 = within * bw;
            const uint32_t byte0 = off + (bit0 >> 3);
            const uint32_t dw = byte0 >> 2;
            const uint32_t sft = (byte0 & 3) * 8 + (bit0 & 7);
            const uint32_t mask_lo = bw >= 32 ? 0xffffffffu : ((1u << bw) - 1);
            const uint32_t mask_hi = bw <= 32 ? 0u : (bw >= 64 ? 0xffffffffu : ((1u << (bw - 32)) - 1));
#pragma unroll
            for (int i = 0; i < 4; i++) {
                const uint32_t b = sft + i * bw;
                const uint32_t* q = stage + dw + (b >> 5);
                const uint32_t x0 = q[0], x1 = q[1];
                const uint32_t f_lo = __builtin_amdgcn_alignbit(x1, x0, b) & mask_lo;
                U dlt;
                if (sizeof(U) == 8) {
                    const uint32_t x2 = q[2];
                    const uint32_t f_hi = __builtin_amdgcn_alignbit(x2, x1, b) & mask_hi;
                    dlt = (U)(((uint64_t)f_hi << 32) | f_lo);
                } else {
                    dlt = (U)f_lo;
                }
                dlt += md;
                v[i] = (j + i < count) ? dlt : (U)0;
            }
        }
        v[1] += v[0]; v[2] += v[1]; v[3] += v[2];
        const U x = wave_inclusive_scan<U>(v[3]);
        if (lane == 63) wtot[wave] = (uint64_t)x;
        __syncthreads();
        U wave_prefix = 0, all = 0;
#pragma unroll
        for (int w = 0; w < DT / 64; w++) {
            const U t = (U)wtot[w];
            if (w < (int)wave) wave_prefix += t;
            all += t;
        }
        if (!STORE) {
            if (tid == 0) sums[tile] = (uint64_t)all;
        } else {
            // the value the tile starts from: the first value plus the sums of the tiles before it
            part = lane_u32_or_u64(wave_inclusive_scan<U>(part));
            __syncthreads(); // everybody has read the wave totals of the tile's own scan
            if (lane == 63) wtot[wave] = (uint64_t)part;
            __syncthreads();
            U carry = (U)sums[n_tiles]; // the first value, left there by the walk
            for (int w = 0; w < DT / 64; w++) carry += (U)wtot[w];
            const U e = carry + wave_prefix + (x - v[3]);
            U r[4] = {(U)(e + v[0]), (U)(e + v[1]), (U)(e + v[2]), (U)(e + v[3])};
            const uint32_t done = tile * TILE;
            if (count == TILE) {
                tile_store_shifted_full<U>(out, done, e, r);
                if (tile + 1 == n_tiles && tid == DT - 1) out[done + TILE] = r[3];
            } else {
                tile_store_shifted_partial<U>(out, done, count, e, r);
            }
        }
        __syncthreads(); // the stage and the wave totals are free again
    }
}

} // namespace

void launch_delta(const DevPage* pages, DevResult* results, const uint32_t* items, uint32_t n_items, bool is64,
                  uint64_t flat_max_blocks, hipStream_t stream) {
    if (n_items == 0) return;
    if (flat_max_blocks > 0) {
        // few pages: the tile-parallel kernels take the pages they can, the kernels below the rest
        uint64_t tiles = (flat_max_blocks + 7) / 8;
        const uint32_t slots = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(tiles, 1), std::max<uint64_t>(1, 16384 / n_items));
        if (is64) {
            hipLaunchKernelGGL(delta_flat_walk<uint64_t>, dim3(n_items), dim3(DT), 0, stream, pages, results, items);
            hipLaunchKernelGGL((delta_flat_tile<uint64_t, false>), dim3(n_items, slots), dim3(DT), 0, stream, pages, results, items);
            hipLaunchKernelGGL((delta_flat_tile<uint64_t, true>), dim3(n_items, slots), dim3(DT), 0, stream, pages, results, items);
        } else {
            hipLaunchKernelGGL(delta_flat_walk<uint32_t>, dim3(n_items), dim3(DT), 0, stream, pages, results, items);
            hipLaunchKernelGGL((delta_flat_tile<uint32_t, false>), dim3(n_items, slots), dim3(DT), 0, stream, pages, results, items);
            hipLaunchKernelGGL((delta_flat_tile<uint32_t, true>), dim3(n_items, slots), dim3(DT), 0, stream, pages, results, items);
        }
    }
    if (is64) {
        hipLaunchKernelGGL(delta_fast_kernel<uint64_t>, dim3(n_items), dim3(DT), 0, stream, pages, results, items);
        hipLaunchKernelGGL(delta_decode_kernel<uint64_t>, dim3(n_items), dim3(DT), 0, stream, pages, results, items);
    } else {
        hipLaunchKernelGGL(delta_fast_kernel<uint32_t>, dim3(n_items), dim3(DT), 0, stream, pages, results, items);
        hipLaunchKernelGGL(delta_decode_kernel<uint32_t>, dim3(n_items), dim3(DT), 0, stream, pages, results, items);
    }
}

} // namespace pqd
