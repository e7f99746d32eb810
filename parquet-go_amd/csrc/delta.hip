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
#include "pqd_internal.h"

namespace pqd {
namespace {

constexpr int DT = 256;                       // threads per workgroup
constexpr uint32_t TILE = 1024;               // deltas per tile, 4 per thread
constexpr uint32_t RING_BYTES = 16384;
constexpr uint32_t RING_MASK = RING_BYTES - 1;
constexpr uint32_t RING_DW_MASK = RING_BYTES / 4 - 1;
constexpr uint32_t CHUNK = DT * 16;           // bytes one fill step brings in
constexpr uint32_t LOOK = 3 * CHUNK;          // bytes kept ahead of the tile start
constexpr int MAX_SLOTS = 40;                 // miniblocks a tile can touch (1024/32 + partials)
constexpr int BW_CACHE = 64;                  // bit widths of one block kept in LDS

struct Slot {
    int64_t min_delta;
    uint64_t off;  // ring-space byte offset of the miniblock data
    uint32_t bw;
    uint32_t pad;
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
    uint32_t ring[RING_BYTES / 4];
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

// Brings ring chunks in until [lo, lo + LOOK) is resident.  `base` is the
// 16-byte aligned address ring-space offset 0 maps to; bytes at or past
// `end` are stored as zero (a truncated miniblock is zero padded,
// binary_packed.go:380-389).  Uniform across the workgroup.
__device__ inline void ring_fill(uint32_t* ring, const uint8_t* base, uint64_t end, uint64_t lo, uint64_t& c_next) {
    uint64_t c_lo = lo / CHUNK;
    if (c_next < c_lo) c_next = c_lo;
    uint64_t c_stop = c_lo + LOOK / CHUNK; // inclusive last chunk needed
    while (c_next <= c_stop) {
        uint64_t v = c_next * CHUNK + threadIdx.x * 16u;
        uint4 w = make_uint4(0, 0, 0, 0);
        if (v < end) {
            w = *reinterpret_cast<const uint4*>(base + v);
            if (v + 16 > end) { // zero the bytes past the end of the page
                uint32_t keep = (uint32_t)(end - v);
                uint32_t* p = reinterpret_cast<uint32_t*>(&w);
                for (int k = 0; k < 4; k++) {
                    uint32_t lo_b = k * 4;
                    if (keep <= lo_b) p[k] = 0;
                    else if (keep < lo_b + 4) p[k] &= (1u << ((keep - lo_b) * 8)) - 1;
                }
            }
        }
        *reinterpret_cast<uint4*>(&ring[(v & RING_MASK) >> 2]) = w;
        c_next++;
    }
}

template <typename U> struct Elem;
template <> struct Elem<uint32_t> { static constexpr uint32_t max_bw = 32; };
template <> struct Elem<uint64_t> { static constexpr uint32_t max_bw = 64; };

// Page header (binary_packed.go:409-452) read by one lane out of the ring; fills
// the control block, writes the first value and reports block geometry.
template <typename U>
__device__ inline void parse_header(const uint8_t* ring8, uint64_t mis, uint64_t end, uint64_t dst_cap, U* out, Ctl& c,
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

// Unpacks, prefix-sums and stores one tile: thread t owns deltas 4t..4t+3 of the
// tile, which sit in miniblock `slot` starting at field `within`.  Adds the
// tile's sum to `carry`.  Contains one workgroup barrier.
template <typename U>
__device__ inline void decode_tile(Shared& sh, U* out, uint64_t done, uint32_t count, uint32_t slot, uint32_t within,
                                   U& carry) {
    const uint32_t tid = threadIdx.x;
    const uint32_t lane = tid & 63, wave = tid >> 6;
    // ---- unpack 4 consecutive deltas per thread ----
    const uint32_t j = tid * 4;
    U v[4] = {0, 0, 0, 0};
    if (j < count) {
            const Slot s = sh.slots[slot];
            const uint32_t bw = s.bw;
            const U md = (U)s.min_delta;
            uint64_t bit0 = (uint64_t)within * bw;
            uint64_t byte0 = s.off + (bit0 >> 3);
            uint32_t dw = (uint32_t)((byte0 & RING_MASK) >> 2);
            uint32_t sft = (uint32_t)((byte0 & 3) * 8 + (bit0 & 7));
            const U mask = bw >= Elem<U>::max_bw ? ~(U)0 : (((U)1 << bw) - 1);
#pragma unroll
            for (int i = 0; i < 4; i++) {
                uint32_t b = sft + i * bw;
                uint32_t k = (dw + (b >> 5)) & RING_DW_MASK;
                uint32_t sh_ = b & 31;
                uint64_t w01 = (uint64_t)sh.ring[k] | ((uint64_t)sh.ring[(k + 1) & RING_DW_MASK] << 32);
                uint64_t f = w01 >> sh_;
                if (sizeof(U) == 8) {
                    uint64_t w2 = sh.ring[(k + 2) & RING_DW_MASK];
                    if (sh_) f |= w2 << (64 - sh_);
                }
                U d = ((U)f & mask) + md;
                v[i] = (j + i < count) ? d : (U)0;
            }
        }
        // thread-local inclusive scan
        v[1] += v[0]; v[2] += v[1]; v[3] += v[2];
        // wave inclusive scan of the thread totals
        U x = v[3];
#pragma unroll
        for (int d = 1; d < 64; d <<= 1) {
            U y = __shfl_up(x, d);
            if ((int)lane >= d) x += y;
        }
        if (lane == 63) sh.wtot[wave] = (uint64_t)x;
        __syncthreads(); // B3: wave totals visible
        U wave_prefix = 0, all = 0;
#pragma unroll
        for (int w = 0; w < DT / 64; w++) {
            U t = (U)sh.wtot[w];
            if (w < (int)wave) wave_prefix += t;
            all += t;
        }
        U excl = carry + wave_prefix + (x - v[3]);
        if (j < count) {
            U* o = out + 1 + done + j;
            if (j + 4 <= count) {
                if (sizeof(U) == 8) {
                    // two 16-byte stores; the array starts one element in, so 8-byte aligned only
                    uint64_t* o64 = reinterpret_cast<uint64_t*>(o);
                    ulonglong2 a, b2;
                    a.x = (uint64_t)(excl + v[0]); a.y = (uint64_t)(excl + v[1]);
                    b2.x = (uint64_t)(excl + v[2]); b2.y = (uint64_t)(excl + v[3]);
                    __builtin_memcpy(o64, &a, 16);
                    __builtin_memcpy(o64 + 2, &b2, 16);
                } else {
                    uint4 a;
                    a.x = (uint32_t)(excl + v[0]); a.y = (uint32_t)(excl + v[1]);
                    a.z = (uint32_t)(excl + v[2]); a.w = (uint32_t)(excl + v[3]);
                    __builtin_memcpy(o, &a, 16);
                }
            } else {
                for (uint32_t i = 0; i < 4 && j + i < count; i++) o[i] = excl + v[i];
            }
        }
    carry += all;
}


// ---------------------------------------------------------------------------
// Fast path: block sizes 128/256/512/1024 with at most 32 miniblocks (every
// writer in practice; the reference writes 128/4).  A tile is then a whole
// number of blocks, and wave 0 finds the tile's block headers in parallel:
// lane l guesses that block l starts l * S bytes after the first one, S being
// the size of the block before, parses the header it finds there and the
// guess holds for every lane whose predecessors all measured S again.  With
// steady widths one round finds all blocks of a tile; when every block differs
// the walk degrades to one block per round, still all in LDS.
// Pages of any other geometry are left to the general kernel below.
// ---------------------------------------------------------------------------

constexpr uint64_t FAST_DONE = 0x46415354444f4e45ull; // marks a result the fast kernel produced

template <typename U>
__global__ __launch_bounds__(DT) void delta_fast_kernel(const DevPage* __restrict__ pages, DevResult* __restrict__ results,
                                                       const uint32_t* __restrict__ items) {
    __shared__ Shared sh;
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
    const uint64_t mis = reinterpret_cast<uintptr_t>(src) & 15;
    const uint8_t* base = src - mis;
    const uint64_t end = mis + src_len;
    U* out = reinterpret_cast<U*>(pg.dst);

    uint64_t c_next = 0;
    ring_fill(sh.ring, base, end, mis, c_next);
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

    const uint64_t total = sh.ctl.total;
    const uint32_t vpm = sh.ctl.vpm;
    const uint32_t nmb = sh.ctl.slot_base;
    const uint32_t block_size = sh.ctl.count;
    const uint32_t vpm_shift = vpm ? 31 - __clz(vpm) : 0;
    const uint32_t blk_shift = block_size ? 31 - __clz(block_size) : 0;
    int stop = sh.ctl.stop;
    U carry = (U)sh.ctl.first;
    uint64_t done = 0;
    const uint64_t n_deltas = total > 0 ? total - 1 : 0;
    // wave 0 only: start of the next block and the size guess
    uint64_t pos = sh.ctl.lo;
    uint32_t guess = 0;
    __syncthreads();

    while (!stop) {
        ring_fill(sh.ring, base, end, sh.ctl.lo, c_next);
        __syncthreads(); // B1

        const uint32_t want = (uint32_t)(n_deltas - done < TILE ? n_deltas - done : TILE);
        if (tid < 64) {
            const uint32_t nb_want = (want + block_size - 1) >> blk_shift;
            uint32_t found = 0, got = 0;
            int32_t status = PQD_OK;
            int64_t arg = 0;
            while (found < nb_want) {
                const bool active = lane < nb_want - found;
                const uint64_t cand = pos + (uint64_t)lane * guess;
                // values this block still has to deliver, and the miniblocks that hold them
                const uint32_t bidx = found + lane;
                const uint32_t bvals = active ? (want - (bidx << blk_shift) < block_size ? want - (bidx << blk_shift) : block_size) : 0;
                const uint32_t need_mb = (bvals + vpm - 1) >> vpm_shift;
                int32_t err = PQD_OK;
                int64_t err_arg = 0;
                bool at_end = false;
                uint32_t size = 0, have_mb = 0;
                if (active) {
                    if (cand >= end) {
                        at_end = true;
                    } else {
                        uint64_t z;
                        int k = ring_uvarint(ring8, cand, end, &z);
                        if (k == 0) { err = PQD_ERR_UNEXPECTED_EOF; err_arg = PQD_EOF_DELTA_MIN_DELTA; }
                        else if (k < 0) { err = PQD_ERR_VARINT_OVERFLOW; err_arg = (int64_t)(cand - mis); }
                        else {
                            const int64_t md = unzigzag(z);
                            const uint64_t wpos = cand + k;
                            const uint32_t nbw = end - wpos < nmb ? (uint32_t)(end - wpos) : nmb;
                            uint64_t off = wpos + nbw;
                            have_mb = nbw < need_mb ? nbw : need_mb;
                            for (uint32_t m = 0; m < have_mb; m++) {
                                const uint32_t bw = ring8[(wpos + m) & RING_MASK];
                                if (bw > Elem<U>::max_bw) { err = PQD_ERR_INVALID_BIT_WIDTH; err_arg = bw; have_mb = m; break; }
                                Slot sl;
                                sl.min_delta = md; sl.off = off; sl.bw = bw; sl.pad = 0;
                                sh.slots[(bidx << (blk_shift - vpm_shift)) + m] = sl;
                                off += ((uint64_t)vpm * bw) >> 3;
                            }
                            size = (uint32_t)(off - cand);
                        }
                    }
                }
                const bool bad = err != PQD_OK || at_end;
                const bool ok = active && !bad && size == guess;
                const unsigned long long okmask = __ballot(ok);
                const unsigned long long badmask = __ballot(active && bad);
                uint32_t lead = okmask == ~0ull ? 64u : (uint32_t)__builtin_ctzll(~okmask);
                uint32_t valid = lead + 1 < nb_want - found ? lead + 1 : nb_want - found;
                if ((badmask >> (valid - 1)) & 1) {
                    if (valid == 1) { // the block the stream really continues with is unreadable
                        const int32_t e0 = __shfl(err, 0);
                        const int64_t a0 = __shfl(err_arg, 0);
                        if (e0 != PQD_OK) { status = e0; arg = a0; }
                        else { status = PQD_ERR_MISSING_VALUES; arg = (int64_t)(n_deltas - done - got); }
                        break;
                    }
                    valid--; // report it when it is first in line
                }
                const uint32_t last = valid - 1;
                const uint32_t last_have = __shfl(have_mb, last);
                const uint32_t last_need = __shfl(need_mb, last);
                const uint32_t last_vals = __shfl(bvals, last);
                const uint32_t last_size = __shfl(size, last);
                const uint64_t last_cand = __shfl(cand, last);
                found += valid;
                got = ((found - 1) << blk_shift) + (last_have < last_need ? last_have << vpm_shift : last_vals);
                pos = last_cand + last_size;
                guess = last_size;
                if (last_have < last_need) { // the widths ran out: nothing follows this block
                    status = PQD_ERR_MISSING_VALUES;
                    arg = (int64_t)(n_deltas - done - got);
                    break;
                }
            }
            if (tid == 0) {
                Ctl& c = sh.ctl;
                c.count = got;
                c.lo = pos;
                c.consumed = pos < end ? pos : end;
                if (status != PQD_OK) { c.status = status; c.arg = arg; }
                if (status != PQD_OK || done + got >= n_deltas) c.stop = 1;
            }
        }
        __syncthreads(); // B2

        const uint32_t count = sh.ctl.count;
        stop = sh.ctl.stop;
        const uint32_t j = tid * 4;
        decode_tile<U>(sh, out, done, count, j >> vpm_shift, j & (vpm - 1), carry);
        done += count;
    }

    __syncthreads();
    const int32_t status = sh.ctl.status;
    if (status == PQD_ERR_MISSING_VALUES || status == PQD_ERR_UNEXPECTED_EOF || status == PQD_ERR_VARINT_OVERFLOW ||
        status == PQD_ERR_INVALID_BIT_WIDTH) {
        for (uint64_t k = 1 + done + tid; k < total; k += DT) out[k] = 0;
    }
    if (tid == 0) {
        res->status = status;
        res->arg = sh.ctl.arg;
        res->total = total;
        res->out_values = total;
        res->out_bytes = total * sizeof(U);
        res->consumed = src_skip + (sh.ctl.consumed - mis);
        res->n_runs = 0;
        res->aux = FAST_DONE;
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
    const uint32_t lane = tid & 63, wave = tid >> 6;
    const uint8_t* ring8 = reinterpret_cast<const uint8_t*>(sh.ring);

    // A chained item starts where the previous stream of the same page ended.
    uint64_t src_skip = 0;
    if (pg.chain >= 0) src_skip = results[pg.chain].consumed;
    const uint8_t* src = pg.src + src_skip;
    const uint64_t src_len = pg.src_len > src_skip ? pg.src_len - src_skip : 0;
    const uint64_t mis = reinterpret_cast<uintptr_t>(src) & 15;
    const uint8_t* base = src - mis;       // ring-space 0
    const uint64_t end = mis + src_len;    // ring-space end of page
    U* out = reinterpret_cast<U*>(pg.dst);
    if (res->aux == FAST_DONE) return; // uniform: the fast kernel decoded this page

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
                                                                  : base[wk.bw_pos + wk.mb_in_block];
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
                    s.min_delta = wk.min_delta; s.off = wk.cur_off; s.bw = wk.cur_bw; s.pad = 0;
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
        decode_tile<U>(sh, out, done, count, slot, within, carry);
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

} // namespace

void launch_delta(const DevPage* pages, DevResult* results, const uint32_t* items, uint32_t n_items, bool is64,
                  hipStream_t stream) {
    if (n_items == 0) return;
    if (is64) {
        hipLaunchKernelGGL(delta_fast_kernel<uint64_t>, dim3(n_items), dim3(DT), 0, stream, pages, results, items);
        hipLaunchKernelGGL(delta_decode_kernel<uint64_t>, dim3(n_items), dim3(DT), 0, stream, pages, results, items);
    } else {
        hipLaunchKernelGGL(delta_fast_kernel<uint32_t>, dim3(n_items), dim3(DT), 0, stream, pages, results, items);
        hipLaunchKernelGGL(delta_decode_kernel<uint32_t>, dim3(n_items), dim3(DT), 0, stream, pages, results, items);
    }
}

} // namespace pqd
