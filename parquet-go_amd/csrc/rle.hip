// rle.hip -- RLE / bit-packed hybrid decode for gfx950, with the dictionary
// gather fused in.
//
// Replaces rle.Encoding.DecodeLevels / DecodeBoolean / DecodeInt32,
// rle.DictionaryEncoding.DecodeInt32 (reference: encoding/rle/rle.go:63-88,
// 251-419,550-576; encoding/rle/dictionary.go:30-37; on amd64
// decodeBytesBitpackBMI2, encoding/rle/rle_amd64.s:16-35, and bitpack.Unpack)
// and, for dictionary pages, newIndexedPage + Dictionary.Lookup
// (dictionary.go:136-162,230-243; dictionaryLookup32/64,
// dictionary_amd64.s:578-711).
//
// Two kernels per batch:
//   rle_index   one wavefront per page walks the run headers (each header's
//               position depends on the ones before it) out of an LDS ring and
//               writes a run table {out_start, in_off, value, count|kind}.
//   rle_expand  a grid over OUTPUT tiles of 2048 values, so work is balanced
//               by output size, not by run count.  A tile inside one bit-packed
//               run stages its packed bytes through LDS with aligned 16-byte
//               loads and every lane extracts its fields from two LDS dwords;
//               mixed tiles look their runs up in an LDS copy of the table.
//               With a dictionary the index goes straight to the gather and is
//               never written to memory.
#include <algorithm>

#include "pqd_internal.h"

namespace pqd {
namespace {

// A run table entry is moved as one 16-byte vector.
__device__ inline Run load_run(gptr<const Run> runs, uint64_t i) {
    const u32x4 w = *(gptr<const u32x4>)(runs + i);
    Run r;
    r.out_start = w.x; r.in_off = w.y; r.value = w.z; r.count_kind = w.w;
    return r;
}

// The first entry is also kept in the page's result, where the expand kernels find it without
// another dependent look-up.  The table holds `cap` entries: as many as the page could have runs, or
// as the values that will be written could lie in (the run of value p is at most the p-th); entries
// past that are counted but not kept, and nobody asks for them.
__device__ inline void store_run(gptr<Run> runs, uint64_t cap, uint64_t i, const Run& r, DevResult* res) {
    if (i >= cap) return;
    u32x4 w;
    w.x = r.out_start; w.y = r.in_off; w.z = r.value; w.w = r.count_kind;
    *(gptr<u32x4>)(runs + i) = w;
    if (i == 0) *(gptr<u32x4>)(as_global(res)->first) = w;
}

// ------------------------------------------------------------------ index --

constexpr int IT = 64;                       // one wavefront
constexpr uint32_t IRING_BYTES = 4096;
constexpr uint32_t IRING_MASK = IRING_BYTES - 1;
constexpr uint32_t ICHUNK = IT * 16;
constexpr uint32_t ILOOK = 3 * ICHUNK;
constexpr uint32_t MAX_TOTAL = 0x7fffffffu;  // values one page may decode to

enum Mode : int32_t { MODE_LEVELS = 0, MODE_INT32 = 1, MODE_BOOL = 2 };

struct ICtl {
    uint64_t pos;
    uint64_t n_runs;
    uint64_t total;
    int64_t arg;
    int32_t status;
    int32_t done;
};

template <uint32_t MASK = IRING_MASK>
__device__ inline int iring_uvarint(const uint8_t* ring8, uint64_t pos, uint64_t end, uint64_t* out) {
    uint64_t x = 0;
    unsigned s = 0;
    for (int i = 0;; i++) {
        if (pos + i >= end) { *out = 0; return 0; }
        if (i == 10) return -(i + 1);
        uint32_t b = ring8[(pos + i) & MASK];
        if (b < 0x80) {
            if (i == 9 && b > 1) return -(i + 1);
            *out = x | ((uint64_t)b << s);
            return i + 1;
        }
        x |= (uint64_t)(b & 0x7f) << s;
        s += 7;
    }
}

// What a run header at ring position p says (rle.go:295-419): the entry it adds to the run
// table (kind 0 repeated value, 1 bit-packed, 2 none: an empty run), where the next header
// starts, or the error the reference reports there.  With `registers_only` the header is
// decoded from 12 bytes held in registers and anything that needs more (a varint longer
// than four bytes, the last bytes of the page) is left for the caller: `slow` is set.
struct RunHeader {
    uint64_t next;    // position of the next header; where the serial decoder would stop on error
    uint64_t nvals;   // values (bytes for BOOLEAN) the run adds
    int64_t err_arg;
    int32_t err;
    uint32_t kind, in_off, value;
    bool slow;
};

template <uint32_t MASK = IRING_MASK>
__device__ inline RunHeader parse_run(const uint32_t* ring, uint64_t p, uint64_t end, uint64_t mis, uint64_t start, int32_t mode,
                                      uint32_t bw, bool registers_only) {
    const uint8_t* ring8 = reinterpret_cast<const uint8_t*>(ring);
    RunHeader h;
    h.next = p; h.nvals = 0; h.err_arg = 0; h.err = PQD_OK; h.kind = 2; h.in_off = 0; h.value = 0; h.slow = false;
    uint64_t q = p, u = 0;
    int k;
    uint64_t after = 0; // the bytes that follow the header, when it was read from registers
    bool have_after = false;
    if (end - p >= 12) {
        const uint32_t d = ((uint32_t)p & MASK) >> 2, sh = ((uint32_t)p & 3) * 8;
        const uint32_t w0 = ring[d], w1 = ring[(d + 1) & (MASK >> 2)], w2 = ring[(d + 2) & (MASK >> 2)];
        const uint32_t a0 = __builtin_amdgcn_alignbit(w1, w0, sh), a1 = __builtin_amdgcn_alignbit(w2, w1, sh);
        // Go's binary.Uvarint on up to four bytes
        if (!(a0 & 0x80)) { u = a0 & 0x7f; k = 1; }
        else if (!(a0 & 0x8000)) { u = (a0 & 0x7f) | ((a0 >> 1) & 0x3f80); k = 2; }
        else if (!(a0 & 0x800000)) { u = (a0 & 0x7f) | ((a0 >> 1) & 0x3f80) | ((a0 >> 2) & 0x1fc000); k = 3; }
        else if (!(a0 & 0x80000000u)) { u = (a0 & 0x7f) | ((a0 >> 1) & 0x3f80) | ((a0 >> 2) & 0x1fc000) | ((a0 >> 3) & 0xfe00000); k = 4; }
        else if (registers_only) { h.slow = true; return h; }
        else k = iring_uvarint<MASK>(ring8, q, end, &u);
        if (k >= 1 && k <= 4) { after = (((uint64_t)a1 << 32) | a0) >> (8 * k); have_after = true; }
    } else if (registers_only) {
        h.slow = true;
        return h;
    } else {
        k = iring_uvarint<MASK>(ring8, q, end, &u);
    }
    if (k == 0) { h.err = PQD_ERR_UNEXPECTED_EOF; h.err_arg = PQD_EOF_RUN_HEADER; return h; }
    if (k < 0) { h.err = PQD_ERR_VARINT_OVERFLOW; h.err_arg = (int64_t)(p - mis - start); return h; }
    q += k;
    const uint64_t count = u >> 1;
    const bool bitpacked = (u & 1) != 0;
    if (count == 0) { // empty runs carry no value byte (rle.go:311-313)
    } else if (count > 0x7fffffffull) { h.err = PQD_ERR_RUN_TOO_LARGE; h.err_arg = (int64_t)count; }
    else if (bitpacked) {
        uint64_t nbytes;
        if (mode == MODE_BOOL) { nbytes = count; h.nvals = count; }      // header counts bytes
        else { nbytes = count * bw; h.nvals = count * 8; }
        if (nbytes > end - q) { h.err = PQD_ERR_UNEXPECTED_EOF; h.err_arg = PQD_EOF_BITPACKED_RUN; h.nvals = 0; }
        else { h.kind = 1; h.in_off = (uint32_t)(q - mis); q += nbytes; }
    } else if (mode == MODE_BOOL) {
        if (q < end) { h.value = have_after ? (uint32_t)after & 0xff : ring8[q & MASK]; q++; }
        h.nvals = (count + 7) / 8; // bytes of output
        h.kind = 0;
    } else {
        const uint32_t vb = mode == MODE_LEVELS ? (bw != 0 ? 1u : 0u) : (bw + 7) / 8;
        if (vb > end - q) { h.err = PQD_ERR_UNEXPECTED_EOF; h.err_arg = PQD_EOF_RLE_VALUE; }
        else {
            if (have_after) h.value = vb == 0 ? 0u : (uint32_t)after & (vb >= 4 ? 0xffffffffu : ((1u << (8 * vb)) - 1));
            else for (uint32_t b = 0; b < vb; b++) h.value |= (uint32_t)ring8[(q + b) & MASK] << (8 * b);
            q += vb;
            h.nvals = count;
            h.kind = 0;
        }
    }
    h.next = q;
    return h;
}

// Records, for every step of RLE_TILE values whose first value lies in the run that starts at
// value `o` and holds `v` values, that run's table index.  Short spans are written by the lane
// that owns the run, long ones by the whole wave.
__device__ inline void publish_steps(gptr<uint32_t> tile_run, uint32_t n_steps, bool has, uint64_t o, uint64_t v, uint32_t idx,
                                     uint32_t lane) {
    uint64_t tf = 0, cnt = 0;
    if (has && v > 0) {
        tf = (o + RLE_TILE - 1) / RLE_TILE;
        const uint64_t tl = (o + v - 1) / RLE_TILE;
        cnt = tl >= tf ? tl - tf + 1 : 0;
        if (tf >= n_steps) cnt = 0;
        else if (tf + cnt > n_steps) cnt = n_steps - tf;
    }
    if (cnt > 0 && cnt <= 4)
        for (uint64_t t = 0; t < cnt; t++) tile_run[tf + t] = idx;
    unsigned long long big = __ballot(cnt > 4);
    while (big) {
        const uint32_t l = (uint32_t)__builtin_ctzll(big);
        big &= big - 1;
        const uint64_t f = lane_u64(tf, l), c = lane_u64(cnt, l);
        const uint32_t i = lane_u32(idx, l);
        for (uint64_t t = lane; t < c; t += 64) tile_run[f + t] = i;
    }
}

// What the stream of a page is, before any run header: where it starts, how long it is, the
// decoding mode and bit width, or the error its framing already holds.
struct PageSetup {
    uint64_t src_len;  // bytes of the page that belong to the stream (from offset 0: framing included)
    uint64_t start;    // offset of the first run header
    int64_t arg;
    int32_t mode;
    int32_t status;
    uint32_t bw;
};

__device__ inline PageSetup page_setup(const DevPage& pg) {
    gptr<const uint8_t> src = as_global(pg.src);
    uint64_t src_len = pg.src_len;
    int32_t mode = pg.kind == PQD_KIND_LEVELS ? MODE_LEVELS : (pg.kind == PQD_KIND_BOOLEAN ? MODE_BOOL : MODE_INT32);
    const bool dict_framed = pg.encoding == PQD_ENC_RLE_DICTIONARY || pg.encoding == PQD_ENC_PLAIN_DICTIONARY;
    uint32_t bw = (uint32_t)pg.bit_width;
    uint64_t start = 0;
    int32_t status0 = PQD_OK;
    int64_t arg0 = 0;
    if (dict_framed) {
        // rle/dictionary.go:30-37: first byte is the bit width; empty input is an empty page
        mode = MODE_INT32;
        if (src_len == 0) { bw = 0; } else { bw = src[0]; start = 1; }
    } else if (mode == MODE_BOOL) {
        // rle.go:68-82: 4-byte little-endian length prefix
        bw = 8;
        if (src_len == 4) { src_len = 0; }
        else if (src_len < 4) { status0 = PQD_ERR_UNEXPECTED_EOF; arg0 = PQD_EOF_LENGTH_PREFIX; src_len = 0; }
        else {
            uint64_t n = (uint64_t)src[0] | ((uint64_t)src[1] << 8) | ((uint64_t)src[2] << 16) | ((uint64_t)src[3] << 24);
            if (n > src_len - 4) { status0 = PQD_ERR_UNEXPECTED_EOF; arg0 = PQD_EOF_LENGTH_PREFIX; src_len = 0; }
            else { start = 4; src_len = 4 + n; }
        }
    }
    if (status0 == PQD_OK) {
        if (mode == MODE_LEVELS && bw > 8) { status0 = PQD_ERR_INVALID_BIT_WIDTH; arg0 = bw; }
        if (mode == MODE_INT32 && bw > 32) { status0 = PQD_ERR_INVALID_BIT_WIDTH; arg0 = bw; }
    }
    if (status0 != PQD_OK) src_len = 0;

    PageSetup ps;
    ps.src_len = src_len; ps.start = start; ps.arg = arg0; ps.mode = mode; ps.status = status0; ps.bw = bw;
    return ps;
}

// Hand-off through LDS inside ONE wavefront: its LDS operations execute in order, so it is enough
// that the compiler keeps them in order and the earlier ones have landed.
__device__ inline void isync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }

// The run table of one page, built by one wavefront (`tid` = its lane) with a ring and a control
// block of its own in LDS.  EXPAND_FIELDS: also reset the result fields the expand kernels fill
// in (not when the same launch's expand code may already be at work on the page).
template <bool EXPAND_FIELDS>
__device__ inline void rle_index_page(uint32_t* ring, ICtl& ctl, const DevPage& pg, DevResult* res, const uint32_t tid) {
    gptr<const uint8_t> src = as_global(pg.src);
    const PageSetup ps = page_setup(pg);
    const uint64_t src_len = ps.src_len, start = ps.start;
    const int32_t mode = ps.mode, status0 = ps.status;
    const uint32_t bw = ps.bw;
    const int64_t arg0 = ps.arg;
    const uint64_t mis = (uintptr_t)src & 15;
    gptr<const uint8_t> base = src - mis;
    const uint64_t end = mis + src_len;
    if (tid == 0) {
        ctl.pos = mis + start;
        ctl.n_runs = 0;
        ctl.total = 0;
        ctl.status = status0;
        ctl.arg = arg0;
        ctl.done = (mis + start >= end) ? 1 : 0;
    }
    isync();

    uint64_t c_next = 0;
    gptr<Run> runs = as_global(pg.runs);
    gptr<uint32_t> tile_run = as_global(pg.tile_run);
    const uint32_t n_steps = pg.n_steps;
    while (!ctl.done) {
        // fill the ring so that [pos, pos + ILOOK) is resident
        const uint64_t pos0 = ctl.pos;
        uint64_t c_lo = pos0 / ICHUNK;
        if (c_next < c_lo) c_next = c_lo;
        while (c_next <= c_lo + ILOOK / ICHUNK) {
            uint64_t v = c_next * ICHUNK + tid * 16u;
            u32x4 w = (u32x4)(0u);
            if (v < end) w = *(gptr<const u32x4>)(base + v); // bytes past `end` are never parsed
            *reinterpret_cast<u32x4*>(&ring[(v & IRING_MASK) >> 2]) = w;
            c_next++;
        }
        isync();
        {
            // Every lane parses the run header that WOULD start at its byte of a 64-byte
            // window; the headers that really follow one another from `pos` are then picked
            // out by a scalar walk over the lanes' "next position" (v_readlane hops), and
            // the lanes on that chain write their run-table entries together.
            uint64_t pos = pos0;
            const uint64_t covered = (c_lo + ILOOK / ICHUNK + 1) * ICHUNK;
            uint64_t n_runs = ctl.n_runs, total = ctl.total;
            int32_t status = PQD_OK;
            int64_t arg = 0;
            while (pos < end && status == PQD_OK) {
                if (pos + 64 + 16 > covered && covered < end) break; // a header might straddle the resident window
                const uint64_t p = pos + tid;
                const bool valid = p < end;
                RunHeader h = parse_run(ring, valid ? p : pos, end, mis, start, mode, bw, true);
                // ---- scalar walk: which lanes hold the headers the stream really has ----
                const uint32_t rel = (h.next - pos) >= 64 ? 64u : (uint32_t)(h.next - pos);
                const unsigned long long okm = __ballot(valid && !h.slow && h.err == PQD_OK);
                unsigned long long chain = 0;
                uint32_t cur = 0;
                while (cur < 64 && ((okm >> cur) & 1)) {
                    chain |= 1ull << cur;
                    cur = lane_u32(rel, cur);
                }
                // cur < 64 here: that lane is past the end of the page, failed to parse, or needs the slow reader
                const bool on = (chain >> tid) & 1;
                // running total along the chain; the first entry that would pass MAX_TOTAL fails (rle.go:373-375)
                const uint64_t x = wave_inclusive_scan<uint64_t>(on ? h.nvals : 0);
                const unsigned long long overm = __ballot(on && total + x > MAX_TOTAL);
                uint32_t stop_lane = 64; // lane whose header ends the decode with an error
                if (overm) { // cut the chain before the entry that overflows
                    stop_lane = (uint32_t)__builtin_ctzll(overm);
                    status = PQD_ERR_RUN_TOO_LARGE;
                    arg = (int64_t)(total + lane_u64(x, stop_lane));
                    chain &= (1ull << stop_lane) - 1;
                } else if (cur < 64 && lane_u32((uint32_t)(valid && !h.slow), cur)) { // the stream continues with a header that does not parse
                    stop_lane = cur;
                    status = (int32_t)lane_u32((uint32_t)h.err, cur);
                    arg = (int64_t)lane_u64((uint64_t)h.err_arg, cur);
                }
                const bool keep = ((chain >> tid) & 1) != 0;
                const unsigned long long recm = __ballot(keep && h.kind != 2);
                if (keep && h.kind != 2) {
                    Run r;
                    r.out_start = (uint32_t)(total + x - h.nvals);
                    r.in_off = h.in_off;
                    r.value = h.value;
                    r.count_kind = ((uint32_t)h.nvals << 1) | (h.kind == 1 ? 1u : 0u);
                    store_run(runs, pg.runs_cap, n_runs + __popcll(recm & ((1ull << tid) - 1)), r, res);
                }
                publish_steps(tile_run, n_steps, keep && h.kind != 2, total + x - h.nvals, h.nvals,
                              (uint32_t)(n_runs + __popcll(recm & ((1ull << tid) - 1))), tid);
                n_runs += __popcll(recm);
                if (chain) {
                    const uint32_t l = 63 - (uint32_t)__builtin_clzll(chain);
                    total += lane_u64(x, l);
                    pos = lane_u64(h.next, l);
                }
                // a failing header leaves the position where the serial decoder would stop reading
                if (stop_lane < 64) pos = lane_u64(h.next, stop_lane);
                if (status != PQD_OK) break;
                // The chain stopped at a header the register reader left alone: every lane reads
                // that one header the slow way (same result in all lanes), lane 0 records it.
                if (cur < 64 && pos < end && lane_u32((uint32_t)(valid && h.slow), cur)) {
                    const RunHeader g = parse_run(ring, pos, end, mis, start, mode, bw, false);
                    if (g.err != PQD_OK) { status = g.err; arg = g.err_arg; pos = g.next; break; }
                    if (total + g.nvals > MAX_TOTAL) { status = PQD_ERR_RUN_TOO_LARGE; arg = (int64_t)(total + g.nvals); pos = g.next; break; }
                    if (g.kind != 2) {
                        if (tid == 0) {
                            Run r;
                            r.out_start = (uint32_t)total;
                            r.in_off = g.in_off;
                            r.value = g.value;
                            r.count_kind = ((uint32_t)g.nvals << 1) | (g.kind == 1 ? 1u : 0u);
                            store_run(runs, pg.runs_cap, n_runs, r, res);
                        }
                        publish_steps(tile_run, n_steps, tid == 0, total, g.nvals, (uint32_t)n_runs, tid);
                        n_runs++;
                    }
                    total += g.nvals;
                    pos = g.next;
                }
            }
            if (tid == 0) {
                ctl.pos = pos;
                ctl.n_runs = n_runs;
                ctl.total = total;
                if (status != PQD_OK) { ctl.status = status; ctl.arg = arg; ctl.done = 1; }
                else if (pos >= end) ctl.done = 1;
            }
        }
        isync();
    }
    if (tid == 0) {
        res->status = ctl.status;
        res->bw = (int32_t)bw;
        res->arg = ctl.arg;
        res->n_runs = ctl.n_runs < pg.runs_cap ? ctl.n_runs : pg.runs_cap;
        res->total = ctl.total;
        res->consumed = ctl.pos >= mis ? ctl.pos - mis : 0;
        if (EXPAND_FIELDS) {
            res->out_values = 0;
            res->out_bytes = 0;
            res->aux = ~0ull; // position of the first out-of-range dictionary index
        }
    }
}

__global__ __launch_bounds__(IT) void rle_index_kernel(const DevPage* __restrict__ pages, DevResult* __restrict__ results,
                                                      const uint32_t* __restrict__ items, const uint32_t* pk_flag) {
    __shared__ uint32_t ring[IRING_BYTES / 4];
    __shared__ ICtl ctl;
    const uint32_t item = items[blockIdx.x];
    const DevPage& pg = pages[item];
    if (pg.ck_index >= 0 && pk_flag[pg.ck_index] == 0) return; // indexed by the chunked kernels
    rle_index_page<true>(ring, ctl, pg, &results[item], threadIdx.x);
}


// ---------------------------------------------------------- chunked index --
//
// A long page handled by many workgroups instead of one wavefront.  The page is cut into
// chunks of CK bytes.  Where the run headers lie inside a chunk depends on where the chain of
// headers enters it, which only the chunks before can tell; so each workgroup first parses a
// header at EVERY byte of its chunk (what the header would be if one started there) and
// follows those "next header" links to the end of the chunk by pointer jumping: a table from
// entry position to exit position.  The chunks of a page then resolve one after the other
// with a single table look-up each (resolve kernel); knowing its entry, a chunk marks the
// headers the stream really has (pointer jumping again), counts them, adds up what the
// chunks before it hold (every chunk publishes its totals as soon as it has them), and writes
// its part of the run table (emit kernel).  Anything out of the ordinary -- a header that
// does not parse, more values than a page may hold -- sends the whole page to the
// one-wavefront kernel, which reports errors exactly as the serial decoder does.

constexpr uint32_t CK = 8192;            // bytes of a chunk
constexpr uint32_t CKT = 256;            // threads of a chunk's workgroup
constexpr uint32_t CKP = CK / CKT;       // positions per thread
constexpr uint32_t CK_BUF = 2 * CK;      // LDS bytes holding the chunk and the header bytes that follow it
constexpr uint32_t CK_MASK = CK_BUF - 1;
constexpr uint32_t CK_LOOK = 16;         // bytes past the chunk a header that starts in it may need
constexpr uint32_t CK_ROUNDS = 13;       // 2^13 links: as many as a chunk can hold
constexpr uint32_t POS_STOP = 0xffffffffu; // exit position: the chain hit something the chunked path leaves alone
constexpr uint32_t NO_ENTRY = 0xffffffffu;
constexpr unsigned long long CK_READY = 1ull << 63; // a chunk's totals are there
constexpr uint32_t CK_SPINS = 1u << 22;  // polls before a waiting workgroup gives up (the page then goes the serial way)

// Workgroups take the work items in the order of their numbers.  A chunk only ever waits for chunks
// with lower numbers, and the hardware starts the workgroups of a grid in the order of their numbers
// (on every XCD: in the order of the numbers it is dealt), so the lowest unfinished one is always
// running; a wait that still does not end gives up after CK_SPINS polls and the page goes the serial way.

// The chunk's bytes [cs, cs + CK + CK_LOOK) of ring space into LDS (zero past the end of the stream).
__device__ inline void chunk_load(uint32_t* buf, gptr<const uint8_t> base, uint32_t cs, uint32_t end) {
    for (uint32_t c = threadIdx.x * 16; c < CK + CK_LOOK; c += CKT * 16) {
        const uint32_t v = cs + c;
        u32x4 w = (u32x4)(0u);
        if (v < end) w = *(gptr<const u32x4>)(base + v);
        *reinterpret_cast<u32x4*>(&buf[(v & CK_MASK) >> 2]) = w;
    }
}

// The header that would start at ring position p (p < end); false when it does not parse.
__device__ inline bool chunk_header(const uint32_t* buf, uint32_t p, uint32_t end, uint32_t mis, uint32_t start, int32_t mode,
                                    uint32_t bw, RunHeader& h) {
    // Headers are read out of registers; the byte-wise reader only serves the last bytes of the page.
    // A varint of more than four bytes elsewhere counts as "does not parse" (a run of 2^27 values or
    // more, or an overlong encoding: if the stream really has one, the page goes the serial way).
    h = parse_run<CK_MASK>(buf, p, end, mis, start, mode, bw, true);
    if (h.slow) {
        if (end - p >= 12) return false;
        h = parse_run<CK_MASK>(buf, p, end, mis, start, mode, bw, false);
    }
    return h.err == PQD_OK;
}

// Where the header after the one at ring position p starts, for a header with at least twelve
// bytes of page behind its first byte and a varint of at most four; 0 when the header does not
// parse (rle.go:295-419, the same rules as parse_run, cut down to what the chain needs).
__device__ inline uint32_t quick_next(const uint32_t* buf, uint32_t p, uint32_t end, int32_t mode, uint32_t bw) {
    const uint32_t d = (p & CK_MASK) >> 2, sh = (p & 3) * 8;
    const uint32_t w0 = buf[d], w1 = buf[(d + 1) & (CK_MASK >> 2)];
    const uint32_t a0 = __builtin_amdgcn_alignbit(w1, w0, sh);
    uint32_t u, k;
    if (!(a0 & 0x80)) { u = a0 & 0x7f; k = 1; }
    else if (!(a0 & 0x8000)) { u = (a0 & 0x7f) | ((a0 >> 1) & 0x3f80); k = 2; }
    else if (!(a0 & 0x800000)) { u = (a0 & 0x7f) | ((a0 >> 1) & 0x3f80) | ((a0 >> 2) & 0x1fc000); k = 3; }
    else if (!(a0 & 0x80000000u)) { u = (a0 & 0x7f) | ((a0 >> 1) & 0x3f80) | ((a0 >> 2) & 0x1fc000) | ((a0 >> 3) & 0xfe00000); k = 4; }
    else return 0;
    const uint32_t q = p + k, count = u >> 1;
    if (count == 0) return q;                                   // an empty run has no value byte
    if (u & 1) {                                                // bit-packed: groups of eight values (bytes for BOOLEAN)
        const uint64_t nbytes = mode == MODE_BOOL ? (uint64_t)count : (uint64_t)count * bw;
        return nbytes > end - q ? 0 : q + (uint32_t)nbytes;
    }
    if (mode == MODE_BOOL) return q + 1;                        // q < end here: twelve bytes follow p
    const uint32_t vb = mode == MODE_LEVELS ? (bw != 0 ? 1u : 0u) : (bw + 7) / 8;
    return q + vb;                                              // vb <= 4 fits as well
}

// Where the header after the one at ring position p (p < end) starts; 0 when that one does not parse.
__device__ inline uint32_t chunk_next(const uint32_t* buf, uint32_t p, uint32_t end, uint32_t mis, uint32_t start, int32_t mode,
                                      uint32_t bw) {
    if (end - p >= 12) return quick_next(buf, p, end, mode, bw);
    RunHeader h; // the last bytes of the page: the byte-wise reader
    return chunk_header(buf, p, end, mis, start, mode, bw, h) ? (uint32_t)h.next : 0u;
}

// First links of a chunk: nxt[q] = position (relative to the chunk) of the header after the one
// at q; q itself when that lies past the chunk or the header does not parse.
__device__ inline void chunk_links(const uint32_t* buf, uint16_t* nxt, uint32_t cs, uint32_t ce, uint32_t end, uint32_t mis,
                                   uint32_t start, int32_t mode, uint32_t bw) {
    for (uint32_t i = 0; i < CKP; i++) {
        const uint32_t q = threadIdx.x + i * CKT, p = cs + q;
        uint32_t n = q;
        if (p < ce) {
            const uint32_t next = chunk_next(buf, p, end, mis, start, mode, bw);
            if (next != 0 && next < ce) n = next - cs;
        }
        nxt[q] = (uint16_t)n;
    }
}

struct ChunkWhere {
    uint32_t k;      // page, as an index into the list of chunked pages
    uint32_t cs, ce; // ring-space bytes of the chunk
    uint32_t end, mis;
    bool first;
};

__device__ inline ChunkWhere chunk_where(const DevPage& pg, const PageSetup& ps, uint32_t chunk, uint32_t k) {
    ChunkWhere w;
    w.k = k;
    w.mis = (uint32_t)((uintptr_t)pg.src & 15);
    w.end = w.mis + (uint32_t)ps.src_len;
    w.cs = chunk * CK;
    w.ce = w.cs + CK < w.end ? w.cs + CK : w.end;
    w.first = chunk == 0;
    return w;
}

__global__ __launch_bounds__(CKT) void rle_chunk_resolve_kernel(const DevPage* __restrict__ pages, const uint32_t* __restrict__ ck_items,
                                                              const uint32_t* __restrict__ ck_prefix, const uint32_t* __restrict__ ck_page,
                                                              const uint32_t* __restrict__ ck_chunk, uint32_t* page_pos, uint32_t* ck_exit,
                                                              uint32_t* ck_entry, uint16_t* ck_links, uint32_t* pk_flag) {
    __shared__ uint32_t buf[CK_BUF / 4];
    __shared__ __attribute__((aligned(16))) uint16_t nxt[CK];
    const uint32_t id = blockIdx.x;
    const uint32_t k = ck_page[id], chunk = ck_chunk[id];
    const uint32_t at = ck_prefix[k] + chunk; // this chunk's place in the per-chunk arrays
    const DevPage& pg = pages[ck_items[k]];
    const PageSetup ps = page_setup(pg);
    const ChunkWhere w = chunk_where(pg, ps, chunk, k);
    const uint32_t tid = threadIdx.x;
    const bool usable = ps.status == PQD_OK && w.mis + ps.start < w.end;
    if (usable) {
        chunk_load(buf, as_global(pg.src) - w.mis, w.cs, w.end);
        __syncthreads();
        chunk_links(buf, nxt, w.cs, w.ce, w.end, w.mis, (uint32_t)ps.start, ps.mode, ps.bw);
        __syncthreads();
        // the first links are kept for the emit kernel
        for (uint32_t c = threadIdx.x * 8; c < CK; c += CKT * 8)
            *(gptr<u32x4>)(ck_links + (size_t)at * CK + c) = *reinterpret_cast<const u32x4*>(&nxt[c]);
        __syncthreads(); // ... before anyone replaces them
        // Pointer jumping: every position learns the last header of the chain that starts there.  Links
        // are replaced in place, two per dword; a link read while its owner replaces it is the old or the
        // new one, and either lies further along the same chain, so no round needs a second barrier.
        uint32_t* nxt32 = reinterpret_cast<uint32_t*>(nxt);
        for (uint32_t round = 0; round < CK_ROUNDS; round++) {
            int moved = 0;
#pragma unroll
            for (uint32_t i = 0; i < CKP / 2; i++) {
                const uint32_t d = tid + i * CKT;
                const uint32_t pair = nxt32[d];
                const uint32_t a = pair & 0xffff, b2 = pair >> 16;
                const uint32_t na = nxt[a], nb = nxt[b2];
                if (na != a || nb != b2) { nxt32[d] = na | (nb << 16); moved = 1; }
            }
            if (!__syncthreads_or(moved)) break;
        }
    }
    if (tid == 0) {
        // page_pos[k]: how far the chain of the page's run headers is known -- the position of the first
        // header no chunk has looked up yet.  A chunk waits until that position reaches it; chunks the
        // chain passes over (inside a long bit-packed run) see it go past and have nothing to do.
        uint32_t from = POS_STOP;
        if (!usable) {
            from = POS_STOP;
        } else if (w.first) {
            from = w.mis + (uint32_t)ps.start;
        } else {
            for (uint32_t spin = 0; spin < CK_SPINS; spin++) {
                const uint32_t v = __hip_atomic_load(&page_pos[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (v >= w.cs) { from = v; break; }
                __builtin_amdgcn_s_sleep(1);
            }
        }
        uint32_t entry = NO_ENTRY, to = from;
        if (from != POS_STOP && from < w.ce) {
            entry = from - w.cs;
            // the last header of the chain that enters here, and where that one points
            const uint32_t next = chunk_next(buf, w.cs + nxt[entry], w.end, w.mis, (uint32_t)ps.start, ps.mode, ps.bw);
            to = next != 0 ? next : POS_STOP;
        }
        const bool mine = entry != NO_ENTRY || from == POS_STOP; // the chain stands in this chunk, or nobody can move it any more
        if (mine) __hip_atomic_store(&page_pos[k], to, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (to == POS_STOP) pk_flag[k] = 1;
        ck_entry[at] = entry;
        ck_exit[at] = to;
    }
}

__global__ __launch_bounds__(CKT) void rle_chunk_emit_kernel(const DevPage* __restrict__ pages, DevResult* __restrict__ results,
                                                           const uint32_t* __restrict__ ck_items, const uint32_t* __restrict__ ck_prefix,
                                                           const uint32_t* __restrict__ ck_page, const uint32_t* __restrict__ ck_chunk,
                                                           const uint32_t* __restrict__ ck_exit, const uint32_t* __restrict__ ck_entry,
                                                           const uint16_t* __restrict__ ck_links, unsigned long long* ck_runs,
                                                           unsigned long long* ck_vals, uint32_t* pk_flag) {
    __shared__ uint32_t buf[CK_BUF / 4];
    __shared__ __attribute__((aligned(16))) uint16_t nxt[CK];
    __shared__ uint16_t list[CK];   // the headers of the stream in this chunk, in order
    __shared__ uint32_t n_list;
    __shared__ unsigned long long wave_runs[CKT / 64], wave_vals[CKT / 64], base_runs, base_vals;
    const uint32_t id = blockIdx.x;
    const uint32_t k = ck_page[id], chunk = ck_chunk[id];
    const uint32_t first_at = ck_prefix[k], at = first_at + chunk; // places in the per-chunk arrays
    const uint32_t item = ck_items[k];
    const DevPage& pg = pages[item];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint32_t entry = ck_entry[at];
    if (pk_flag[k] != 0) return;        // the page goes the serial way: nobody waits for this chunk
    if (entry == NO_ENTRY) {            // the chain passes over this chunk
        if (tid == 0) {
            __hip_atomic_store(&ck_vals[at], CK_READY, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&ck_runs[at], CK_READY, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    const PageSetup ps = page_setup(pg);
    const ChunkWhere w = chunk_where(pg, ps, chunk, k);
    const uint32_t start = (uint32_t)ps.start;
    chunk_load(buf, as_global(pg.src) - w.mis, w.cs, w.end);
    for (uint32_t c = tid * 8; c < CK; c += CKT * 8) // the first links, as the resolve kernel left them
        *reinterpret_cast<u32x4*>(&nxt[c]) = *(gptr<const u32x4>)(ck_links + (size_t)at * CK + c);
    __syncthreads();
    // The headers of the stream, found by wave 0: 64 links at a time in registers, the chain followed
    // through them by lane reads; the lanes it visits append their positions to the list.
    if (wave == 0) {
        uint32_t pos = entry, count = 0;
        for (uint32_t guard = 0; guard < CK; guard++) { // every pass moves on by at least one position
            const uint32_t q = pos + lane;
            const uint32_t n = q < CK ? (uint32_t)nxt[q] : q; // a position's own number: the chain ends there
            unsigned long long chain = 0;
            uint32_t cur = 0, last = 0;
            bool ended = false;
            while (cur < 64) {
                chain |= 1ull << cur;
                last = cur;
                const uint32_t to_ = lane_u32(n, cur);
                if (to_ == pos + cur) { ended = true; break; }
                cur = to_ - pos;
            }
            if ((chain >> lane) & 1) list[count + __popcll(chain & ((1ull << lane) - 1))] = (uint16_t)q;
            count += __popcll(chain);
            if (ended) break;
            pos = lane_u32(n, last);
        }
        if (lane == 0) n_list = count;
    }
    __syncthreads();
    const uint32_t n_hdr = n_list;
    // ---- what the chunk adds: a header per thread, 256 at a time ----
    unsigned long long all_runs = 0, all_vals = 0;
    for (uint32_t b0 = 0; b0 < n_hdr; b0 += CKT) {
        unsigned long long r = 0, v = 0;
        if (b0 + tid < n_hdr) {
            RunHeader h;
            chunk_header(buf, w.cs + list[b0 + tid], w.end, w.mis, start, ps.mode, ps.bw, h);
            r = h.kind != 2 ? 1 : 0;
            v = h.nvals;
        }
        all_runs += r; all_vals += v;
    }
    all_runs = wave_inclusive_scan<uint64_t>(all_runs);
    all_vals = wave_inclusive_scan<uint64_t>(all_vals);
    if (lane == 63) { wave_runs[wave] = all_runs; wave_vals[wave] = all_vals; }
    __syncthreads();
    all_runs = 0; all_vals = 0;
    for (uint32_t v = 0; v < CKT / 64; v++) { all_runs += wave_runs[v]; all_vals += wave_vals[v]; }
    // this chunk's totals for the chunks after it; the totals of the chunks before it
    if (tid == 0) {
        __hip_atomic_store(&ck_vals[at], CK_READY | all_vals, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(&ck_runs[at], CK_READY | all_runs, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (wave == 0) {
        bool gave_up = false;
        unsigned long long sum_runs = 0, sum_vals = 0;
        for (uint32_t c = first_at + lane; c < at; c += 64) {
            unsigned long long r = 0;
            uint32_t spin = 0;
            for (; spin < CK_SPINS; spin++) {
                r = __hip_atomic_load(&ck_runs[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                if (r & CK_READY) break;
                __builtin_amdgcn_s_sleep(1);
            }
            if (spin == CK_SPINS) { gave_up = true; break; }
            __atomic_thread_fence(__ATOMIC_ACQUIRE); // the values were stored before the runs
            sum_runs += r & ~CK_READY;
            sum_vals += __hip_atomic_load(&ck_vals[c], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & ~CK_READY;
        }
        sum_runs = wave_inclusive_scan<uint64_t>(sum_runs);
        sum_vals = wave_inclusive_scan<uint64_t>(sum_vals);
        gave_up = __ballot(gave_up) != 0;
        if (lane == 63) { base_runs = gave_up ? ~0ull : sum_runs; base_vals = sum_vals; }
    }
    __syncthreads();
    if (base_runs == ~0ull) { // a chunk before this one never reported: leave the page to the serial kernel
        if (tid == 0) pk_flag[k] = 1;
        return;
    }
    // ---- the run-table entries of this chunk, and the per-step hints ----
    gptr<Run> runs = as_global(pg.runs);
    gptr<uint32_t> tile_run = as_global(pg.tile_run);
    const uint32_t n_steps = pg.n_steps;
    unsigned long long r_base = base_runs, v_base = base_vals;
    for (uint32_t b0 = 0; b0 < n_hdr; b0 += CKT) {
        RunHeader h;
        h.kind = 2; h.nvals = 0; h.in_off = 0; h.value = 0;
        if (b0 + tid < n_hdr) chunk_header(buf, w.cs + list[b0 + tid], w.end, w.mis, start, ps.mode, ps.bw, h);
        const bool rec = h.kind != 2;
        const unsigned long long in_r = wave_inclusive_scan<uint64_t>(rec ? 1 : 0), in_v = wave_inclusive_scan<uint64_t>(h.nvals);
        __syncthreads(); // the totals of the batch before have been read
        if (lane == 63) { wave_runs[wave] = in_r; wave_vals[wave] = in_v; }
        __syncthreads();
        unsigned long long r_at = r_base + in_r - (rec ? 1 : 0), v_at = v_base + in_v - h.nvals;
        for (uint32_t v = 0; v < CKT / 64; v++) {
            if (v < wave) { r_at += wave_runs[v]; v_at += wave_vals[v]; }
            r_base += wave_runs[v]; v_base += wave_vals[v];
        }
        if (rec) {
            Run r;
            r.out_start = (uint32_t)v_at;
            r.in_off = h.in_off;
            r.value = h.value;
            r.count_kind = ((uint32_t)h.nvals << 1) | (h.kind == 1 ? 1u : 0u);
            store_run(runs, pg.runs_cap, r_at, r, &results[item]);
        }
        publish_steps(tile_run, n_steps, rec, v_at, h.nvals, (uint32_t)r_at, lane);
    }
    // the chunk in which the chain reaches the end of the stream closes the page
    const uint32_t to = ck_exit[at];
    if (to == w.end && tid == 0) {
        const unsigned long long n_runs = base_runs + all_runs, total = base_vals + all_vals;
        if (total > MAX_TOTAL) {
            pk_flag[k] = 1; // the serial kernel reports which run passes the limit (rle.go:373-375)
        } else {
            DevResult* res = &results[item];
            res->status = PQD_OK;
            res->bw = (int32_t)ps.bw;
            res->arg = 0;
            res->n_runs = n_runs < pg.runs_cap ? n_runs : pg.runs_cap;
            res->total = total;
            res->consumed = w.end - w.mis;
            res->out_values = 0;
            res->out_bytes = 0;
            res->aux = ~0ull;
        }
    }
}

// ----------------------------------------------------------------- expand --

constexpr int ET = 256;
constexpr uint32_t STAGE_BYTES = RLE_TILE * 4 + 64; // packed bytes of one tile at 32 bits per value, plus slack
constexpr uint32_t MAX_TILE_RUNS = RLE_TILE + 2;
constexpr uint32_t RT_CAP = MAX_TILE_RUNS * 4 / 16; // whole run-table entries the run_start buffer can hold instead

constexpr uint32_t MAX_GROUP = 8;             // steps one expand workgroup may take

struct EShared {
    uint32_t stage[STAGE_BYTES / 4];
    uint32_t run_start[MAX_TILE_RUNS];
    u32x4 first_run[MAX_GROUP];   // per step: the run that holds its first value ...
    uint32_t first_idx[MAX_GROUP]; // ... and that run's index
    uint32_t span_lo, span_hi;     // byte range of the packed data being staged
};

// Number of values the item writes, after the caller's length fix-up
// (dictionary.go:136-162; column.go:940-945).
__device__ inline uint64_t output_count(const DevPage& pg, uint64_t total, bool fused) {
    if (pg.num_values < 0) return total;
    uint64_t nv = (uint64_t)pg.num_values;
    // the indexes of a dictionary-encoded page are cut to num_values or zero-extended to it
    // (newIndexedPage), whether they go on through the dictionary or are returned as they are
    const bool indexes = pg.encoding == PQD_ENC_RLE_DICTIONARY || pg.encoding == PQD_ENC_PLAIN_DICTIONARY;
    if (fused || indexes) return nv;
    return total < nv ? total : nv;  // levels and plain RLE values are only ever cut (column.go:940-945)
}

template <int ELEM, bool FUSED>
__device__ inline void store_value(const DevPage& pg, DevResult* res, uint64_t pos, uint32_t v, uint32_t stride) {
    gptr<uint8_t> p = as_global(pg.dst) + pos * stride;
    if (!FUSED) {
        if (ELEM == 1) *p = (uint8_t)v;
        else *(gptr<uint32_t>)(p) = v;
        return;
    }
    // Dictionary.Lookup: unsigned bounds check, then gather (dictionary_amd64.s:578-711)
    const bool ok = (uint64_t)v < pg.dict_len;
    if (!ok) atomicMin(reinterpret_cast<unsigned long long*>(&res->aux), (unsigned long long)pos);
    gptr<const uint8_t> d = as_global(pg.dict) + (uint64_t)(ok ? v : 0) * ELEM;
    if (ELEM == 4) {
        uint32_t x = ok ? *(gptr<const uint32_t>)(d) : 0u;
        *(gptr<uint32_t>)(p) = x;
    } else if (ELEM == 8) {
        uint64_t x = ok ? *(gptr<const uint64_t>)(d) : 0ull;
        u32x2_a4 w;
        w.x = (uint32_t)x; w.y = (uint32_t)(x >> 32);
        *(gptr<u32x2_a4>)(p) = w; // rows of a strided destination are only 4-byte aligned
    } else {
        gptr<const uint32_t> d32 = (gptr<const uint32_t>)(d);
        gptr<uint32_t> p32 = (gptr<uint32_t>)(p);
        for (int k = 0; k < ELEM / 4; k++) p32[k] = ok ? d32[k] : 0u;
    }
}

// Writes `value` to the values [s0, s1) of a dense, non-dictionary destination: 16 bytes per lane
// and step once the address is aligned, single elements at the ragged ends.
template <int ELEM>
__device__ inline void fill_dense(const DevPage& pg, uint64_t s0, uint64_t s1, uint32_t value, uint32_t lid = threadIdx.x,
                                  uint32_t team = ET) {
    gptr<uint8_t> base = as_global(pg.dst);
    const uint64_t b0 = s0 * ELEM, b1 = s1 * ELEM;
    uint64_t a0 = (((uintptr_t)(pg.dst + b0) + 15) & ~(uintptr_t)15) - (uintptr_t)pg.dst; // first 16-byte boundary at or after b0
    if (a0 > b1) a0 = b1;
    const uint64_t a1 = a0 + ((b1 - a0) & ~(uint64_t)15);
    const uint32_t w = ELEM == 1 ? (value & 0xff) * 0x01010101u : value;
    u32x4 v = {w, w, w, w};
    for (uint64_t c = a0 + (uint64_t)lid * 16; c < a1; c += (uint64_t)team * 16) *(gptr<u32x4>)(base + c) = v;
    for (uint64_t c = b0 + (uint64_t)lid * ELEM; c < a0; c += (uint64_t)team * ELEM) {
        if (ELEM == 1) base[c] = (uint8_t)value; else *(gptr<uint32_t>)(base + c) = value;
    }
    for (uint64_t c = a1 + (uint64_t)lid * ELEM; c < b1; c += (uint64_t)team * ELEM) {
        if (ELEM == 1) base[c] = (uint8_t)value; else *(gptr<uint32_t>)(base + c) = value;
    }
}

// Decodes the tile [t0, t0 + n) of one item.
template <int ELEM, bool FUSED>
__device__ inline void expand_tile(EShared& sh, const DevPage& pg, DevResult* res, uint64_t t0, uint32_t n,
                                   uint64_t total, uint64_t n_runs, uint32_t bw, uint64_t r0, Run first) {
    const uint32_t tid = threadIdx.x;
    gptr<const Run> runs = as_global((const Run*)pg.runs);
    const uint32_t stride = pg.dst_stride > 0 ? (uint32_t)pg.dst_stride : (uint32_t)ELEM;
    const bool dense = stride == (uint32_t)ELEM;
    gptr<const uint8_t> src = as_global(pg.src);

    // values past the end of the stream: zero-extended indices (FUSED only)
    uint32_t n_stream = n;
    if (t0 >= total) n_stream = 0;
    else if (t0 + n > total) n_stream = (uint32_t)(total - t0);

    // r0 / first: the run holding t0, looked up by the caller (the index kernel left a hint per step)
    const uint64_t first_end = (uint64_t)first.out_start + (first.count_kind >> 1);
    const bool single = n_stream > 0 && first_end >= t0 + n_stream;

    // thread -> value mapping: G consecutive values per thread and repetition,
    // chosen so one store instruction of a wave covers contiguous memory
    constexpr int G = ELEM == 1 ? 8 : (ELEM == 8 ? 2 : 4);
    constexpr int R = 8 / G;
    const uint32_t mask = bw >= 32 ? 0xffffffffu : ((1u << bw) - 1);

    // ---- the G values v of repetition r, values [s0 + j, s0 + j + G) of the page with j + g < sn:
    // through the dictionary if there is one, then one vector store where the destination allows ----
    auto emit = [&](uint64_t s0, uint32_t sn, int r, const uint32_t (&v)[G]) {
        const uint32_t j = r * (ET * G) + tid * G;
        if (!FUSED && dense && j + G <= sn) {
            gptr<uint8_t> p = as_global(pg.dst) + (s0 + j) * ELEM;
            if (ELEM == 1) {
                u32x2_a4 w;
                w.x = v[0] | (v[1 % G] << 8) | (v[2 % G] << 16) | (v[3 % G] << 24);
                w.y = v[4 % G] | (v[5 % G] << 8) | (v[6 % G] << 16) | (v[7 % G] << 24);
                if (((uintptr_t)p & 3) == 0) *(gptr<u32x2_a4>)(p) = w;
                else for (int g = 0; g < 8; g++) p[g] = (uint8_t)v[g % G];
            } else {
                u32x4_a4 w;
                w.x = v[0]; w.y = v[1 % G]; w.z = v[2 % G]; w.w = v[3 % G];
                *(gptr<u32x4_a4>)(p) = w;
            }
        } else if (FUSED && dense && ELEM == 8 && j + G <= sn) {
            // gather two 8-byte values, store 16 contiguous bytes
            uint64_t x[2];
#pragma unroll
            for (int g = 0; g < 2; g++) {
                const uint32_t idx = v[g % G];
                const bool ok = (uint64_t)idx < pg.dict_len;
                if (!ok) atomicMin(reinterpret_cast<unsigned long long*>(&res->aux), (unsigned long long)(s0 + j + g));
                x[g] = ok ? ((gptr<const uint64_t>)(pg.dict))[idx] : 0ull;
            }
            u64x2_a8 w;
            w.x = x[0]; w.y = x[1];
            *(gptr<u64x2_a8>)(as_global(pg.dst) + (s0 + j) * 8) = w;
        } else if (FUSED && dense && ELEM == 4 && j + G <= sn) {
            uint32_t x[4];
#pragma unroll
            for (int g = 0; g < 4; g++) {
                const uint32_t idx = v[g % G];
                const bool ok = (uint64_t)idx < pg.dict_len;
                if (!ok) atomicMin(reinterpret_cast<unsigned long long*>(&res->aux), (unsigned long long)(s0 + j + g));
                x[g] = ok ? ((gptr<const uint32_t>)(pg.dict))[idx] : 0u;
            }
            u32x4_a4 w;
            w.x = x[0]; w.y = x[1]; w.z = x[2]; w.w = x[3];
            *(gptr<u32x4_a4>)(as_global(pg.dst) + (s0 + j) * 4) = w;
        } else {
#pragma unroll
            for (int g = 0; g < G; g++)
                if (j + g < sn) store_value<ELEM, FUSED>(pg, res, s0 + j + g, v[g], stride);
        }
    };

    if (single && (first.count_kind & 1)) {
        // ---- the whole tile lies inside one bit-packed run: packed bytes staged through LDS with
        // aligned 16-byte loads, every lane cuts its fields out of two LDS dwords ----
        const uint64_t k0 = t0 - first.out_start;            // first field of the tile within the run
        const uint64_t bit0 = k0 * bw;
        const uint64_t byte0 = (uint64_t)first.in_off + (bit0 >> 3);
        const uint64_t byte1 = (uint64_t)first.in_off + ((bit0 + (uint64_t)n_stream * bw + 7) >> 3);
        const uint64_t mis = (uintptr_t)(src + byte0) & 15;
        gptr<const uint8_t> abase = src + byte0 - mis;
        const uint64_t span = mis + (byte1 - byte0);        // bytes to stage, from the aligned base
        for (uint32_t c = tid * 16; c < span; c += ET * 16) {
            u32x4 w = *(gptr<const u32x4>)(abase + c);
            *reinterpret_cast<u32x4*>(&sh.stage[c >> 2]) = w;
        }
        __syncthreads();
        const uint32_t bitbase = (uint32_t)(mis * 8 + (bit0 & 7));
#pragma unroll
        for (int r = 0; r < R; r++) {
            const uint32_t j = r * (ET * G) + tid * G;
            uint32_t v[G];
#pragma unroll
            for (int g = 0; g < G; g++) {
                const uint32_t b = bitbase + (j + g) * bw;
                const uint32_t k = b >> 5, s = b & 31;
                const uint64_t w01 = (uint64_t)sh.stage[k] | ((uint64_t)sh.stage[k + 1] << 32);
                v[g] = (uint32_t)(w01 >> s) & mask;
            }
            emit(t0, n_stream, r, v);
        }
    } else if (single) {
        // ---- the whole tile lies inside one repeated run ----
        if (!FUSED && dense && (ELEM == 1 || ELEM == 4)) fill_dense<(ELEM == 1 ? 1 : 4)>(pg, t0, t0 + n_stream, first.value);
        else for (uint32_t j = tid; j < n_stream; j += ET) store_value<ELEM, FUSED>(pg, res, t0 + j, first.value, stride);
    } else if (n_stream > 0) {
        // ---- mixed tile: the run-table entries that meet it and the packed bytes of its bit-packed
        // runs are copied to LDS; every lane then finds the run of its values by bisection there ----
        const uint64_t avail = n_runs - r0;
        const uint32_t nr = avail < RT_CAP ? (uint32_t)avail : RT_CAP;
        u32x4* tab = reinterpret_cast<u32x4*>(sh.run_start);
        for (uint32_t k = tid; k < nr; k += ET) tab[k] = *(gptr<const u32x4>)(runs + r0 + k);
        if (tid == 0) { sh.span_lo = 0xffffffffu; sh.span_hi = 0; }
        __syncthreads();
        uint32_t in_tile; // staged runs that start before the end of the tile (the first one starts at or before t0)
        {
            uint32_t lo = 0, hi = nr;
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if ((uint64_t)tab[mid].x < t0 + n_stream) lo = mid; else hi = mid;
            }
            in_tile = lo + 1;
        }
        const u32x4 last = tab[in_tile - 1];
        bool in_lds = (uint64_t)last.x + (last.w >> 1) >= t0 + n_stream; // else the tile goes on past the staged entries
        {
            uint32_t my_lo = 0xffffffffu, my_hi = 0;
            for (uint32_t k = tid; k < in_tile; k += ET) {
                const u32x4 e = tab[k];
                if (e.w & 1) {
                    const uint64_t rb = e.x, re = rb + (e.w >> 1);
                    const uint64_t s0 = rb > t0 ? rb : t0, s1 = re < t0 + n_stream ? re : t0 + n_stream;
                    if (s1 > s0) {
                        my_lo = min(my_lo, e.y + (uint32_t)(((s0 - rb) * bw) >> 3));
                        my_hi = max(my_hi, e.y + (uint32_t)(((s1 - rb) * bw + 7) >> 3));
                    }
                }
            }
            if ((tid & ~63u) < in_tile) { // one atomic per wavefront that saw an entry
                const uint32_t w_lo = wave_least_u32(my_lo), w_hi = wave_greatest_u32(my_hi);
                if ((tid & 63) == 0 && w_lo != 0xffffffffu) { atomicMin(&sh.span_lo, w_lo); atomicMax(&sh.span_hi, w_hi); }
            }
        }
        __syncthreads();
        const uint32_t b_lo = sh.span_lo, b_hi = sh.span_hi;
        int64_t base = 0; // src offset of stage byte 0: up to 15 bytes before the first packed byte, so the loads are aligned
        if (b_hi > b_lo && b_lo != 0xffffffffu) {
            base = (int64_t)b_lo - (int64_t)((uintptr_t)(pg.src + b_lo) & 15);
            const uint32_t span = (uint32_t)((int64_t)b_hi - base);
            if (span > STAGE_BYTES - 64) in_lds = false;
            else if (in_lds)
                for (uint32_t c = tid * 16; c < span; c += ET * 16)
                    *reinterpret_cast<u32x4*>(&sh.stage[c >> 2]) = *(gptr<const u32x4>)(src + base + c);
        }
        __syncthreads();
        if (in_lds) {
#pragma unroll
            for (int r = 0; r < R; r++) {
                const uint32_t j = r * (ET * G) + tid * G;
                uint32_t v[G];
#pragma unroll
                for (int g = 0; g < G; g++) v[g] = 0;
                if (j < n_stream) {
                    const uint64_t pos = t0 + j;
                    uint32_t lo = 0, hi = in_tile; // last staged run with out_start <= pos
                    while (hi - lo > 1) {
                        const uint32_t mid = (lo + hi) >> 1;
                        if ((uint64_t)tab[mid].x <= pos) lo = mid; else hi = mid;
                    }
                    u32x4 e = tab[lo];
#pragma unroll
                    for (int g = 0; g < G; g++) {
                        const uint64_t q = pos + g;
                        if (j + g >= n_stream) break; // past the end of the stream: stays zero
                        if (q >= (uint64_t)e.x + (e.w >> 1) && lo + 1 < in_tile) e = tab[++lo]; // every run holds at least one value
                        if (e.w & 1) {
                            const uint64_t bits = (q - e.x) * bw;
                            const uint32_t b = (uint32_t)((int64_t)e.y + (int64_t)(bits >> 3) - base) * 8 + (uint32_t)(bits & 7);
                            const uint32_t k = b >> 5;
                            const uint64_t w01 = (uint64_t)sh.stage[k] | ((uint64_t)sh.stage[k + 1] << 32);
                            v[g] = (uint32_t)(w01 >> (b & 31)) & mask;
                        } else {
                            v[g] = e.z;
                        }
                    }
                }
                emit(t0, n_stream, r, v);
            }
        } else {
            // ---- more runs or packed bytes than the LDS buffers hold: the out_start of every run,
            // a bisection per value, entries and bytes read from memory ----
            const uint32_t nr2 = avail < MAX_TILE_RUNS ? (uint32_t)avail : MAX_TILE_RUNS;
            for (uint32_t k = tid; k < nr2; k += ET) sh.run_start[k] = runs[r0 + k].out_start;
            __syncthreads();
            for (uint32_t j = tid; j < n_stream; j += ET) {
                const uint64_t pos = t0 + j;
                uint32_t lo = 0, hi = nr2; // last staged run with out_start <= pos
                while (hi - lo > 1) {
                    uint32_t mid = (lo + hi) >> 1;
                    if ((uint64_t)sh.run_start[mid] <= pos) lo = mid; else hi = mid;
                }
                const Run rr = load_run(runs, r0 + lo);
                uint32_t v;
                if (rr.count_kind & 1) {
                    // bytes touched all belong to this run, which the index kernel bounds-checked
                    uint64_t bit = (pos - rr.out_start) * bw;
                    gptr<const uint8_t> q = src + rr.in_off + (bit >> 3);
                    uint32_t s = (uint32_t)(bit & 7);
                    uint32_t nb = (s + bw + 7) >> 3;
                    uint64_t acc = 0;
                    for (uint32_t b = 0; b < nb; b++) acc |= (uint64_t)q[b] << (8 * b);
                    v = (uint32_t)(acc >> s) & mask;
                } else {
                    v = rr.value;
                }
                store_value<ELEM, FUSED>(pg, res, pos, v, stride);
            }
        }
    }
    // zero-extended tail of a dictionary page (n > n_stream only there: see output_count)
    if (FUSED || ELEM == 4) {
        for (uint32_t j = n_stream + tid; j < n; j += ET) store_value<ELEM, FUSED>(pg, res, t0 + j, 0u, stride);
    }
}

constexpr uint32_t GROUP_SHORT_RUN = 1024;          // runs of at most this many values are expanded by one wavefront
constexpr uint32_t GROUP_STAGE_PAD = 16;           // bytes left free before the staged packed data

// Expands the staged runs tab[0 .. nr) over the values [G0, G0 + GS) of a dense one-byte destination.
// `stage` holds the packed bytes of the bit-packed runs: stage byte GROUP_STAGE_PAD is byte `base` of
// the page.  Long runs are shared by the whole workgroup, short ones are dealt out to the
// wavefronts: no run waits for another, they write disjoint parts of the destination.
__device__ __forceinline__ void expand_runs_bytes(const u32x4* tab, uint32_t nr, const uint32_t* stage, int64_t base, const DevPage& pg,
                                         uint64_t G0, uint32_t GS, uint32_t bw) {
    const uint32_t tid = threadIdx.x;
    gptr<uint8_t> dst = as_global(pg.dst);
    const uint32_t mask = (1u << bw) - 1;
    const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    for (uint32_t k = 0; k < nr; k++) {
        const u32x4 r = tab[k];
        const uint64_t rb = r.x, re = rb + (r.w >> 1);
        const uint64_t s0 = rb > G0 ? rb : G0, s1 = re < G0 + GS ? re : G0 + GS;
        if (s1 <= s0) continue;
        const uint32_t sn = (uint32_t)(s1 - s0);
        const bool shared = sn > GROUP_SHORT_RUN;
        if (!shared && (k & 3) != wave) continue;
        const uint32_t lid = shared ? tid : (tid & 63), team = shared ? ET : 64;
        if (!(r.w & 1)) { fill_dense<1>(pg, s0, s1, r.z, lid, team); continue; }
        const uint64_t skip = (s0 - rb) * bw; // bits of the run before this group
        const uint32_t bit_base = (uint32_t)((int64_t)r.y + (int64_t)(skip >> 3) - base + GROUP_STAGE_PAD) * 8 + (uint32_t)(skip & 7);
        // Every lane owns eight values that start on an 8-byte boundary of the destination: its
        // fields lie inside 96 bits from the dword the first one starts in.  At the ragged ends of
        // the span the same window is cut out (it may start before the first field: the staged
        // bytes sit GROUP_STAGE_PAD bytes into the buffer) and only the values inside are stored.
        const uint32_t a = (uint32_t)((uintptr_t)(pg.dst + s0) & 7);
        for (uint32_t c = lid * 8; c < sn + a; c += team * 8) {
            const int32_t j0 = (int32_t)c - (int32_t)a;
            const uint32_t b = bit_base + (uint32_t)(j0 * (int32_t)bw);
            const uint32_t d = b >> 5;
            const uint32_t w0 = stage[d], w1 = stage[d + 1], w2 = stage[d + 2];
            const uint64_t w01 = (uint64_t)w0 | ((uint64_t)w1 << 32);
            const uint64_t w12 = (uint64_t)w1 | ((uint64_t)w2 << 32);
            uint32_t v[8];
#pragma unroll
            for (uint32_t e = 0; e < 8; e++) {
                const uint32_t off = (b & 31) + e * bw;
                v[e] = (uint32_t)(off < 32 ? w01 >> off : w12 >> (off - 32)) & mask;
            }
            gptr<uint8_t> o = dst + (s0 + (uint64_t)(int64_t)j0);
            if (j0 >= 0 && (uint32_t)j0 + 8 <= sn) {
                u32x2 w;
                w.x = v[0] | (v[1] << 8) | (v[2] << 16) | (v[3] << 24);
                w.y = v[4] | (v[5] << 8) | (v[6] << 16) | (v[7] << 24);
                *(gptr<u32x2>)(o) = w;
            } else {
#pragma unroll
                for (uint32_t e = 0; e < 8; e++)
                    if ((uint32_t)(j0 + (int32_t)e) < sn) o[e] = (uint8_t)v[e]; // unsigned: values before the span fail too
            }
        }
    }
}

template <int NW = ET / 64>
__device__ __forceinline__ void page_expand(const u32x4* tab, uint32_t nr, const uint32_t* win, int64_t base, const DevPage& pg, uint32_t G0,
                                   uint32_t G1, uint32_t bw, uint32_t share = 0, uint32_t shares = 1);

// One-byte outputs (levels, booleans): a whole group of steps at once.  The runs that meet the
// group are copied to LDS in one go, the packed bytes of all its bit-packed runs in another,
// and the runs are then expanded one after the other out of LDS: three dependent round trips
// to memory per workgroup instead of three per run.  Returns false when the group does not
// fit these buffers; the caller then takes it step by step.
__device__ __forceinline__ bool expand_group_bytes(EShared& sh, const DevPage& pg, uint64_t G0, uint32_t GN, uint64_t total,
                                          uint64_t n_runs, uint32_t bw) {
    const uint32_t tid = threadIdx.x;
    gptr<const uint8_t> src = as_global(pg.src);
    gptr<const Run> runs = as_global((const Run*)pg.runs);
    if (G0 >= total) return true; // nothing of the stream reaches this group
    const uint32_t GS = total - G0 < GN ? (uint32_t)(total - G0) : GN;
    u32x4* tab = reinterpret_cast<u32x4*>(sh.run_start);
    const uint64_t r_first = n_runs > 1 ? as_global((const uint32_t*)pg.tile_run)[G0 / RLE_TILE] : 0;
    const uint32_t K = n_runs - r_first < RT_CAP ? (uint32_t)(n_runs - r_first) : RT_CAP;
    for (uint32_t k = tid; k < K; k += ET) tab[k] = *(gptr<const u32x4>)(runs + r_first + k);
    if (tid == 0) { sh.span_lo = 0xffffffffu; sh.span_hi = 0; }
    __syncthreads();
    // runs that start before the end of the group
    uint32_t nr;
    {
        uint32_t lo = 0, hi = K;
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if ((uint64_t)tab[mid].x < G0 + GS) lo = mid; else hi = mid;
        }
        nr = lo + 1;
    }
    {
        const u32x4 last = tab[nr - 1];
        if ((uint64_t)last.x + (last.w >> 1) < G0 + GS) return false; // the group runs past the staged entries
    }
    {
        uint32_t my_lo = 0xffffffffu, my_hi = 0;
        for (uint32_t k = tid; k < nr; k += ET) {
            const u32x4 r = tab[k];
            if (r.w & 1) {
                const uint64_t rb = r.x, re = rb + (r.w >> 1);
                const uint64_t s0 = rb > G0 ? rb : G0, s1 = re < G0 + GS ? re : G0 + GS;
                if (s1 > s0) {
                    my_lo = min(my_lo, r.y + (uint32_t)(((s0 - rb) * bw) >> 3));
                    my_hi = max(my_hi, r.y + (uint32_t)(((s1 - rb) * bw + 7) >> 3));
                }
            }
        }
        if ((tid & ~63u) < nr) { // one atomic per wavefront that saw an entry
            const uint32_t w_lo = wave_least_u32(my_lo), w_hi = wave_greatest_u32(my_hi);
            if ((tid & 63) == 0 && w_lo != 0xffffffffu) { atomicMin(&sh.span_lo, w_lo); atomicMax(&sh.span_hi, w_hi); }
        }
    }
    __syncthreads();
    const uint32_t b_lo = sh.span_lo, b_hi = sh.span_hi;
    int64_t base = 0; // src offset of stage byte 0: up to 15 bytes before the first packed byte, so the loads are aligned
    if (b_hi > b_lo && b_lo != 0xffffffffu) {
        base = (int64_t)b_lo - (int64_t)((uintptr_t)(pg.src + b_lo) & 15);
        const uint32_t span = (uint32_t)((int64_t)b_hi - base);
        if (span > STAGE_BYTES - 64 - GROUP_STAGE_PAD) { __syncthreads(); return false; }
        for (uint32_t c = tid * 16; c < span; c += ET * 16)
            *reinterpret_cast<u32x4*>(&sh.stage[(c + GROUP_STAGE_PAD) >> 2]) = *(gptr<const u32x4>)(src + base + c);
    }
    __syncthreads();
    // Levels take the piece-by-piece form (every lane looks up the run of its sixteen bytes): the run-by-run
    // form spends some thirty scalar instructions per run in every wavefront, and level runs are short.
    if (pg.kind == PQD_KIND_LEVELS && bw >= 1 && bw <= 8 && G0 + GS <= 0xffffffffull)
        page_expand(tab, nr, sh.stage, base, pg, (uint32_t)G0, (uint32_t)(G0 + GS), bw);
    else
        expand_runs_bytes(tab, nr, sh.stage, base, pg, G0, GS, bw);
    return true;
}

// One instantiation per RLE hybrid family, so that each carries only the code and the registers
// of its own element type: FAM_RLE_BYTES one byte per element (levels, booleans), FAM_RLE_DICT8 /
// FAM_RLE_DICT4 through a dictionary of 8- / 4-byte values, FAM_RLE the rest (4-byte values,
// 12-byte dictionary entries).
// Tile `tile` of an item's grid (pg.group steps of RLE_TILE values), by one workgroup, out of the run
// table the index left.  Every return is taken by the whole workgroup.
template <int FAMILY>
__device__ __forceinline__ void expand_item_tile(EShared& sh, const DevPage& pg, DevResult* res, uint64_t tile) {
    const uint64_t total = res->total;
    const uint64_t n_runs = res->n_runs;
    const uint32_t bw = (uint32_t)res->bw;
    const bool fused = pg.fused != 0;
    const uint32_t elem = (uint32_t)pg.elem_width;
    const uint32_t stride = pg.dst_stride > 0 ? (uint32_t)pg.dst_stride : elem;
    const uint32_t group = pg.group;
    uint64_t n_out = output_count(pg, total, fused);
    const uint64_t fit = pg.dst_cap / stride + ((pg.dst_cap % stride) >= elem ? 1 : 0);
    const uint64_t need_bytes = n_out == 0 ? 0 : (n_out - 1) * stride + elem;
    bool too_small = false;
    if (n_out > fit) { too_small = true; n_out = fit; }
    if (tile == 0 && threadIdx.x == 0) {
        res->out_values = n_out;
        res->out_bytes = n_out == 0 ? 0 : (n_out - 1) * (uint64_t)stride + elem;
        if (too_small && res->status == PQD_OK) { res->status = PQD_ERR_DST_TOO_SMALL; res->arg = (int64_t)need_bytes; }
    }
    if (FAMILY == FAM_RLE_BYTES && group > 1 && !fused && elem == 1 && stride == 1 && bw <= 8) {
        const uint64_t G0 = tile * group * (uint64_t)RLE_TILE;
        if (G0 >= n_out) return;
        const uint32_t GN = n_out - G0 < (uint64_t)group * RLE_TILE ? (uint32_t)(n_out - G0) : group * RLE_TILE;
        if (expand_group_bytes(sh, pg, G0, GN, total, n_runs, bw)) return;
        __syncthreads();
    }
    // the first run of every step, fetched together: one round trip instead of two per step
    if (threadIdx.x < group) {
        const uint64_t ts = (tile * group + threadIdx.x) * RLE_TILE;
        uint32_t idx = 0;
        u32x4 w = (u32x4)(0u);
        if (ts < n_out && ts < total) {
            if (n_runs > 1) idx = as_global((const uint32_t*)pg.tile_run)[ts / RLE_TILE]; // a single run needs no hint
            w = *(gptr<const u32x4>)(as_global((const Run*)pg.runs) + idx);
        }
        sh.first_idx[threadIdx.x] = idx;
        sh.first_run[threadIdx.x] = w;
    }
    __syncthreads();
    for (uint32_t step = 0; step < group; step++) {
        const uint64_t t0 = (tile * group + step) * RLE_TILE;
        if (t0 >= n_out) return;
        const uint64_t r0 = sh.first_idx[step];
        const u32x4 fw = sh.first_run[step];
        Run first;
        first.out_start = fw.x; first.in_off = fw.y; first.value = fw.z; first.count_kind = fw.w;
        const uint32_t n = n_out - t0 < RLE_TILE ? (uint32_t)(n_out - t0) : RLE_TILE;
        if (FAMILY == FAM_RLE_BYTES) expand_tile<1, false>(sh, pg, res, t0, n, total, n_runs, bw, r0, first);
        else if (FAMILY == FAM_RLE_DICT8) expand_tile<8, true>(sh, pg, res, t0, n, total, n_runs, bw, r0, first);
        else if (FAMILY == FAM_RLE_DICT4) expand_tile<4, true>(sh, pg, res, t0, n, total, n_runs, bw, r0, first);
        else if (fused) expand_tile<12, true>(sh, pg, res, t0, n, total, n_runs, bw, r0, first);
        else expand_tile<4, false>(sh, pg, res, t0, n, total, n_runs, bw, r0, first);
        if (step + 1 < group) __syncthreads(); // the next step restages the LDS buffers
    }
}

template <int FAMILY>
__global__ __launch_bounds__(ET) __attribute__((amdgpu_waves_per_eu(FAMILY == FAM_RLE_BYTES ? 8 : 4, 8))) void rle_expand_kernel(const DevPage* __restrict__ pages, DevResult* __restrict__ results,
                                                       const uint32_t* __restrict__ items, uint32_t n_items,
                                                       const uint64_t* __restrict__ tile_prefix,
                                                       const uint32_t* __restrict__ tile_item) {
    __shared__ EShared sh;
    const uint64_t b = blockIdx.x;
    uint32_t lo = 0;
    if (tile_item != nullptr) {
        lo = tile_item[b];
    } else { // which item does this tile belong to: last i with tile_prefix[i] <= blockIdx.x
        uint32_t hi = n_items;
        while (hi - lo > 1) {
            uint32_t mid = (lo + hi) >> 1;
            if (tile_prefix[mid] <= b) lo = mid; else hi = mid;
        }
    }
    const uint32_t item = items[lo];
    expand_item_tile<FAMILY>(sh, pages[item], &results[item], b - tile_prefix[lo]);
}


// ------------------------------------------------------ a page per workgroup --
//
// Level pages in one pass (rle.go:295-350 decodeBytes, for the widths levels have: 1 to 8 bits).
// The index + expand pair above goes to memory five times in a row for every tile (the run hint, the
// runs, the packed bytes, ...) and a lone wavefront follows run headers at about a microsecond
// each: it issues an instruction every five cycles or so, and a header takes two hundred.  A page
// of levels is a few kilobytes with a few dozen headers.  Here a workgroup takes a page, window
// by window (PW_WIN bytes in LDS):
//   A  every thread works out, for eight byte positions of the window, how far the next header
//      would be if a header started there (the varint, the run's payload) -- 0 for a position
//      that cannot start a run the round can use;
//   B  the first wavefront follows those distances from the known position: the distances of the
//      64 positions around it sit in one register, a hop inside them is a v_readlane and a
//      select that notes the header's position in the lane whose turn it is, a hop out of them
//      one LDS read more;
//   C  a thread per header found reads it in full, a scan over the workgroup gives every run its
//      first value, the runs go to a table in LDS;
//   D  all wavefronts expand the table: sixteen values per lane on 16-byte boundaries of the
//      destination; the runs are looked up in registers (a ballot over 64 table entries).
// A bit-packed run that reaches past the window is listed piecewise.  Anything the walk does not
// expect -- a varint over three bytes, a run that runs past the page, more than 2^31-1 values, a
// width of 0 or above 8 -- ends the attempt: the page is then indexed with rle_index_page and
// expanded tile by tile by rle_page_retry_kernel, exactly what the kernel pair does, so statuses
// and the bytes written are the same on every path.

constexpr uint32_t PW_WIN = 4096;                 // bytes of the page resident at a time
// (headers followed per round: one per thread of the workgroup)
constexpr uint32_t PW_SLACK = 32;                 // LDS bytes behind the window that reads may touch
#ifndef PQD_PAGE_HOPS_LOG2
#define PQD_PAGE_HOPS_LOG2 3
#endif
constexpr uint32_t PAGE_HOPS_LOG2 = PQD_PAGE_HOPS_LOG2;            // wide workgroups: the walk takes 2^3 headers per step (see page_chase_hops)
#ifndef PQD_PAGE_SHARE_WGS
#define PQD_PAGE_SHARE_WGS 256 // workgroups a batch's pages are shared out to at most
#endif
#ifndef PQD_PAGE_MAX_SHARES
#define PQD_PAGE_MAX_SHARES 8
#endif
constexpr uint32_t PAGE_MAX_SHARES = PQD_PAGE_MAX_SHARES; // workgroups that may share one page's expansion
constexpr uint64_t PAGE_RETRY = 0x5041474552455452ull; // DevResult::aux of a page rle_page_kernel left to the retry kernel
static_assert(PW_WIN % ET == 0, "a whole number of positions per thread");

template <int T>
struct PageShared {
    uint32_t win[(GROUP_STAGE_PAD + PW_WIN + PW_SLACK) / 4];
    uint16_t adv[PW_WIN];                         // per position of the window: bytes to the next header, 0: none
    uint16_t hdr[T];                              // positions (in the window) of the headers the chain holds
    u32x4 tab[T + 1];                             // the round's runs (Run layout; in_off counts from the page's first byte)
    uint64_t wsum[T / 64];
    // wide workgroups only: per position, where PAGE_HOPS headers on would be (low half: distance, saturated; high half:
    // headers passed, fewer where the chain stops first), and the positions the walk itself stood on
    uint32_t jc[2][T >= 512 ? PW_WIN : 1];     // (two copies: a round of doubling reads one and writes the other)
    uint16_t anchor[T >= 512 ? T / 8 + 2 : 1];
    // the walk, between rounds.  Positions count from the 16-byte boundary at or before the page's first byte.
    uint32_t pos;                                 // next header, or next group of the bit-packed run in progress
    uint32_t pending;                             // groups of eight values that run still holds
    uint32_t total;                               // values listed so far
    uint32_t state;                               // 0 going on, 1 page finished, 2 give up
    uint32_t n_runs;
    // a round
    uint32_t n_hdr, cur_end, reason, n_tab;
};
static_assert(sizeof(EShared::stage) >= IRING_BYTES, "the retry kernel's index ring fits the stage");

__device__ inline uint32_t pw_uniform(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ inline void pw_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); } // LDS only: stores stay in flight

// The varint at the head of a0 (up to three bytes: counts below 2^20).
__device__ inline bool pw_varint(uint32_t a0, uint32_t& k, uint32_t& u) {
    if (!(a0 & 0x80)) { u = a0 & 0x7f; k = 1; return true; }
    if (!(a0 & 0x8000)) { u = (a0 & 0x7f) | ((a0 >> 1) & 0x3f80); k = 2; return true; }
    if (!(a0 & 0x800000)) { u = (a0 & 0x7f) | ((a0 >> 1) & 0x3f80) | ((a0 >> 2) & 0x1fc000); k = 3; return true; }
    return false;
}

// Bytes from position p to the next header if a header starts at p (a0: the four bytes there); 0 when
// the round cannot use a header at p: behind the page's end, too close to the window's end to be
// read, or not a header the fast path takes (then the walk gives up when it gets there).
__device__ inline uint32_t pw_advance(uint32_t a0, uint32_t p, uint32_t w_hi, uint32_t end, uint32_t bw) {
    if (p >= end || (p + 4 > w_hi && w_hi < end)) return 0;
    uint32_t k, u;
    if (!pw_varint(a0, k, u) || p + k > end) return 0;
    const uint32_t count = u >> 1;
    if (count == 0) return k; // an empty run carries no value byte (rle.go:311-313)
    if (u & 1) {
        const uint32_t payload = count * bw; // < 2^23
        return payload > end - (p + k) ? 0 : k + payload;
    }
    return p + k >= end ? 0 : k + 1;
}

// Sixteen fields of BW bits out of the 128 bits x[0..3] (field i at bit i * BW), one byte each.
// Four fields at a time: the 4 BW bits are cut out, their upper half moves up to bit 16, and the upper
// field of either half up to its byte: two shift-and-merge steps for four fields, whatever the width.
template <uint32_t BW>
__device__ inline u32x4 page_fields16(const uint32_t (&x)[4]) {
    if (BW == 8) return u32x4{x[0], x[1], x[2], x[3]};
    constexpr uint32_t M1 = (1u << BW) - 1, M2 = (1u << (2 * BW)) - 1, M4 = BW == 8 ? 0xffffffffu : (1u << (4 * BW)) - 1;
    uint32_t o[4];
#pragma unroll
    for (uint32_t q = 0; q < 4; q++) {
        const uint32_t bit = q * 4 * BW, k = bit >> 5, off = bit & 31;
        uint32_t t;
        if (off + 4 * BW <= 32) t = x[k] >> off;
        else t = __builtin_amdgcn_alignbit(x[(k + 1) & 3], x[k], off);
        t &= M4;
        t = (t & M2) | ((t << (16 - 2 * BW)) & (M2 << 16));
        t = (t & (M1 * 0x00010001u)) | ((t << (8 - BW)) & (M1 * 0x01000100u));
        o[q] = t;
    }
    return u32x4{o[0], o[1], o[2], o[3]};
}

// Sixteen fields of a bit-packed run (packed bytes from page offset e_y on) from its field `first` on; `first`
// may lie up to 15 fields before the run and the sixteen may reach past its end: the buffer has room
// on both sides, what is read there is not used.
__device__ inline u32x4 page_unpack16(const uint32_t* win, uint32_t e_y, int64_t base, int32_t first, uint32_t bw) {
    const uint32_t b = (uint32_t)((int32_t)(((uint32_t)((int64_t)e_y - base) + GROUP_STAGE_PAD) * 8) + first * (int32_t)bw); // bit in the window buffer
    const uint32_t d = b >> 5, sft = b & 31;
    const uint32_t w0 = win[d], w1 = win[d + 1], w2 = win[d + 2], w3 = win[d + 3], w4 = win[d + 4];
    const uint32_t x[4] = {__builtin_amdgcn_alignbit(w1, w0, sft), __builtin_amdgcn_alignbit(w2, w1, sft),
                           __builtin_amdgcn_alignbit(w3, w2, sft), __builtin_amdgcn_alignbit(w4, w3, sft)};
    switch (bw) {
        case 1: return page_fields16<1>(x);
        case 2: return page_fields16<2>(x);
        case 3: return page_fields16<3>(x);
        case 4: return page_fields16<4>(x);
        case 5: return page_fields16<5>(x);
        case 6: return page_fields16<6>(x);
        case 7: return page_fields16<7>(x);
        default: return page_fields16<8>(x);
    }
}

// Expands the values [G0, G1) that the runs tab[0 .. nr) hold (tab[0] starts at G0, the runs follow
// one another and reach G1 or beyond) into a dense one-byte destination, sixteen values per lane
// on 16-byte boundaries of the destination.  64 runs at a time sit in registers, a lane each; a
// wavefront takes 64 pieces at a time and finds the run of the first by a ballot.  When all 64
// lie inside that run -- most values of a level stream belong to long runs -- every lane stores a
// splat or unpacks its sixteen fields without looking anything up; otherwise every lane finds
// the run of its piece in the table, and a piece that meets several runs, or the ragged ends of a
// stretch, goes value by value.
constexpr uint32_t PAGE_FEW_RUNS = 8; // up to so many runs behind the first that meet 64 pieces are counted out of registers

template <int NW>
__device__ __forceinline__ void page_expand(const u32x4* tab, uint32_t nr, const uint32_t* win, int64_t base, const DevPage& pg, uint32_t G0,
                                   uint32_t G1, uint32_t bw, uint32_t share, uint32_t shares) {
    // the NW wavefronts of `shares` workgroups share the pieces, 64 at a time in turn
    const uint32_t lane = threadIdx.x & 63, wave = (NW == 1 ? 0u : pw_uniform(threadIdx.x >> 6)) + share * NW;
    const uint32_t waves = NW * shares;
    gptr<uint8_t> dst = as_global(pg.dst);
    const uint32_t a = (uint32_t)((uintptr_t)pg.dst & 15); // piece c holds the values [16 c - a, 16 c - a + 16)
    for (uint32_t rb = 0; rb < nr; rb += 64) {
        u32x4 mine = u32x4{0xffffffffu, 0, 0, 0};
        if (rb + lane < nr) mine = tab[rb + lane];
        const uint32_t nb = nr - rb < 64 ? nr - rb : 64;
        const uint32_t B0 = rb == 0 ? G0 : pw_uniform(mine.x);
        const uint32_t last_end = (uint32_t)__builtin_amdgcn_readlane((int)(mine.x + (mine.w >> 1)), (int)(nb - 1));
        const uint32_t B1 = last_end < G1 ? last_end : G1; // this stretch of runs holds [B0, B1)
        if (B1 <= B0) break;
        const uint32_t c_lo = (B0 + a) >> 4, c_hi = (B1 + a + 15) >> 4;
        for (uint32_t c0 = c_lo + wave * 64; c0 < c_hi; c0 += waves * 64) {
            const uint32_t c1 = c0 + 64 < c_hi ? c0 + 64 : c_hi;
            const uint32_t q_lo = c0 * 16 > B0 + a ? c0 * 16 - a : B0; // first value of these pieces inside the stretch
            const unsigned long long le = __ballot(mine.x <= q_lo);
            const uint32_t r = (uint32_t)__popcll(le) - 1;              // lane that holds the run of q_lo
            const uint32_t e_x = (uint32_t)__builtin_amdgcn_readlane((int)mine.x, (int)r);
            const uint32_t e_w = (uint32_t)__builtin_amdgcn_readlane((int)mine.w, (int)r);
            const bool whole = c0 * 16 >= B0 + a && c1 == c0 + 64 && c1 * 16 - a <= B1; // 64 full pieces
            const uint32_t c = c0 + lane;
            if (whole && c1 * 16 - a <= e_x + (e_w >> 1)) {
                const uint32_t P = c * 16 - a;
                u32x4 v;
                if (e_w & 1) {
                    v = page_unpack16(win, (uint32_t)__builtin_amdgcn_readlane((int)mine.y, (int)r), base, (int32_t)(P - e_x), bw);
                } else {
                    const uint32_t w = ((uint32_t)__builtin_amdgcn_readlane((int)mine.z, (int)r) & 0xff) * 0x01010101u;
                    v = u32x4{w, w, w, w};
                }
                *(gptr<u32x4>)(dst + P) = v;
                continue;
            }
            // ---- the lane's own piece: [p0, p1) of it lie inside the stretch ----
            const bool live = c < c1;
            const int64_t Ps = (int64_t)(live ? c : c0) * 16 - a;
            const uint32_t p0 = Ps < (int64_t)B0 ? B0 : (uint32_t)Ps, p1 = Ps + 16 > (int64_t)B1 ? B1 : (uint32_t)(Ps + 16);
            // Which run holds the piece's first value: counted over the runs that meet these 64 pieces, uniformly, out of
            // the registers (a lane walking the table in LDS pays a dependent round trip per run it passes).
            const uint32_t q_hi = c1 * 16 - a < B1 ? c1 * 16 - a : B1;
            const uint32_t r_last = (uint32_t)__popcll(__ballot(mine.x < q_hi)) - 1; // (lanes without a run hold 0xffffffff)
            uint32_t rl = rb + r;
            if (r_last - r <= PAGE_FEW_RUNS) {
                for (uint32_t k = r + 1; k <= r_last; k++) rl += (uint32_t)__builtin_amdgcn_readlane((int)mine.x, (int)k) <= p0 ? 1u : 0u;
            } else {
                while (rl + 1 < rb + nb && tab[rl + 1].x <= p0) rl++;
            }
            if (!live || p1 <= p0) continue;
            // what each run that meets the piece puts into its sixteen bytes, merged under byte masks
            u32x4 acc = u32x4{0, 0, 0, 0};
            for (uint32_t q = p0;;) {
                const u32x4 e = tab[rl];
                const uint32_t re = e.x + (e.w >> 1), q1 = re < p1 ? re : p1; // this run holds [q, q1) of the piece
                u32x4 v;
                if (e.w & 1) v = page_unpack16(win, e.y, base, (int32_t)(Ps - (int64_t)e.x), bw);
                else { const uint32_t w = (e.z & 0xff) * 0x01010101u; v = u32x4{w, w, w, w}; }
                const uint32_t lo = (uint32_t)((int64_t)q - Ps), hi = (uint32_t)((int64_t)q1 - Ps); // bytes [lo, hi) of the piece
                uint32_t m[4];
#pragma unroll
                for (uint32_t j = 0; j < 4; j++) {
                    const uint32_t l4 = lo > 4 * j ? (lo - 4 * j < 4 ? lo - 4 * j : 4) : 0, h4 = hi > 4 * j ? (hi - 4 * j < 4 ? hi - 4 * j : 4) : 0;
                    const uint32_t below_h = h4 == 4 ? 0xffffffffu : (1u << (8 * h4)) - 1, below_l = l4 == 4 ? 0xffffffffu : (1u << (8 * l4)) - 1;
                    m[j] = below_h & ~below_l;
                }
                acc.x |= v.x & m[0]; acc.y |= v.y & m[1]; acc.z |= v.z & m[2]; acc.w |= v.w & m[3];
                q = q1;
                if (q >= p1) break;
                rl++; // every listed run holds a value, and the runs cover the stretch
            }
            if (p1 - p0 == 16) {
                *(gptr<u32x4>)(dst + p0) = acc;
            } else {
                const uint32_t words[4] = {acc.x, acc.y, acc.z, acc.w};
#pragma unroll
                for (uint32_t i = 0; i < 16; i++) {
                    const int64_t pi = Ps + i;
                    if (pi >= (int64_t)p0 && pi < (int64_t)p1) dst[pi] = (uint8_t)(words[i >> 2] >> (8 * (i & 3)));
                }
            }
        }
    }
}

// Phase B: follows the distances in sh.adv from window position `cur`; notes the headers it meets
// in sh.hdr.  One wavefront; everything but the registers of distances and positions is uniform.
// The distances of the 64 positions around the walk's position sit in one register (a lane
// each): hops inside that stretch are a v_readlane away, a hop out of it costs one LDS read.
template <int T>
__device__ inline void page_chase(PageShared<T>& sh, uint32_t cur, uint32_t w_lo, uint32_t w_hi, uint32_t end) {
    constexpr uint32_t PW_HDRS = T;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t size = w_hi - w_lo;
    uint32_t h = 0, n = 0;
    bool zero = false;
    cur = pw_uniform(cur);
    __builtin_amdgcn_s_setprio(3); // the workgroup's other wavefronts wait for this walk: it goes first on its SIMD
    // One loop, few branches: a taken branch costs this lone instruction stream more than the arithmetic of a hop.
    uint32_t seg = 0xffffffffu, av = 0;
    while (cur < size) {
        if ((cur >> 6) != seg) { // the hop left the 64 positions held in `av`: one LDS read
            seg = cur >> 6;
            av = sh.adv[seg * 64 + lane]; // (positions at or behind the window's last byte are never looked at)
        }
        const uint32_t d = (uint32_t)__builtin_amdgcn_readlane((int)av, (int)(cur & 63));
        if ((d == 0) | (n == PW_HDRS)) { zero = d == 0; break; }
        h = lane == (n & 63) ? cur : h;
        n++;
        cur += d;
        if ((n & 63) == 0) sh.hdr[n - 64 + lane] = (uint16_t)h;
    }
    __builtin_amdgcn_s_setprio(0);
    if (n & 63) sh.hdr[(n & ~63u) + lane] = (uint16_t)h;
    const uint32_t p = w_lo + cur;
    uint32_t reason = 0; // 0: the next round goes on from p, 1: p is the page's end, 2: p holds nothing the fast path takes
    if (p >= end) reason = 1;
    else if (zero && !(p + 4 > w_hi && w_hi < end)) reason = 2;
    if (lane == 0) { sh.n_hdr = n; sh.cur_end = p; sh.reason = reason; }
}

// Phase B for wide workgroups.  Phase A has also worked out, by three rounds of doubling over all positions at once, where
// the chain stands eight headers on from every position (sh.jc): the walk -- the one part of a page only one wavefront can
// do, a dependent LDS round trip per step -- takes eight headers per step and notes where it stood; the seven headers
// behind each of those places are then filled in by a lane per place.
template <int T>
__device__ inline void page_chase_hops(PageShared<T>& sh, uint32_t cur, uint32_t w_lo, uint32_t w_hi, uint32_t end) {
    constexpr uint32_t PW_HDRS = T, HOPS = 1u << PAGE_HOPS_LOG2;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t size = w_hi - w_lo;
    uint32_t n = 0, na = 0;
    bool zero = false;
    cur = pw_uniform(cur);
    __builtin_amdgcn_s_setprio(3);
    while (cur < size) {
        const uint32_t v = pw_uniform(sh.jc[PAGE_HOPS_LOG2 & 1][cur]);
        const uint32_t c = v >> 16;
        if (c == 0) { zero = true; break; }       // nothing the round can use starts here
        if (n + c > PW_HDRS) break;               // the table is full: the next round goes on from here
        if (lane == 0) sh.anchor[na] = (uint16_t)cur;
        na++;
        n += c;
        cur += v & 0xffffu; // (fewer than HOPS headers: the chain left the window, or stands on a position of distance 0)
    }
    __builtin_amdgcn_s_setprio(0);
    for (uint32_t a = lane; a < na; a += 64) {
        uint32_t p = sh.anchor[a];
        for (uint32_t i = 0; i < HOPS && p < size; i++) {
            const uint32_t d = sh.adv[p];
            if (d == 0) break;
            sh.hdr[a * HOPS + i] = (uint16_t)p;
            p += d;
        }
    }
    const uint32_t p = w_lo + cur;
    uint32_t reason = 0;
    if (p >= end) reason = 1;
    else if (zero && !(p + 4 > w_hi && w_hi < end)) reason = 2;
    if (lane == 0) { sh.n_hdr = n; sh.cur_end = p; sh.reason = reason; }
}

// T threads take a page: 256 (four wavefronts share the expansion: long pages), or 64 -- short pages are a chain of
// dependent steps (descriptor, window, distances, walk, table), and a wavefront that has a SIMD nearly to itself goes
// through them several times faster than four that share theirs with twenty-eight others and meet at barriers.
template <int T>
__global__ __launch_bounds__(T) __attribute__((amdgpu_waves_per_eu(T >= 256 ? 8 : 2, 8))) void rle_page_kernel(const DevPage* __restrict__ pages, DevResult* __restrict__ results,
                                                     const uint32_t* __restrict__ items, uint32_t* __restrict__ retry_word, uint32_t launch) {
    constexpr uint32_t ET = T; // (shadows the file's workgroup size inside this kernel)
    __shared__ PageShared<T> sh;
#ifdef PQD_PAGE_TIMING2
    const uint64_t t_entry = __builtin_readcyclecounter();
#endif
    const uint32_t item = items[blockIdx.x];
    const DevPage& pg = pages[item];
    DevResult* res = &results[item];
    const uint32_t tid = threadIdx.x, wave = pw_uniform(tid >> 6), lane = tid & 63;
    // gridDim.y workgroups to a page when the batch has few pages: each goes through the page's windows on its own (the
    // walk is the same work done several times over, on compute units that would stand idle) and expands every
    // gridDim.y-th stretch of the values -- a single compute unit writes a page of a million levels at 20 GB/s, whatever
    // else the chip does.  The first of them reports.
    const uint32_t share = blockIdx.y, shares = gridDim.y;
    // (everything the kernel wants of the descriptor is asked for at once, with no branch in between: a test per field
    // would be half a dozen trips to memory one after the other before the first byte of the page is on its way)
    const uint32_t bw = (uint32_t)pg.bit_width;
    const int32_t d_kind = pg.kind, d_enc = pg.encoding, d_stride = pg.dst_stride;
    const uint32_t d_elem = pg.elem_width, d_fused = pg.fused;
    const uint8_t* const d_src = pg.src;
    const uint64_t d_src_len = pg.src_len, d_cap = pg.dst_cap;
    const int64_t d_num = pg.num_values;
    const bool dense = (d_kind == PQD_KIND_LEVELS) & (d_enc == PQD_ENC_RLE) & (d_elem == 1) & (d_stride <= 1) & (d_fused == 0);
    if (dense & (bw >= 1) & (bw <= 8) & (d_src_len <= 0x7ff00000ull)) {
        const uint32_t mis = (uint32_t)((uintptr_t)d_src & 15);
        gptr<const uint8_t> base = as_global(d_src) - mis;
        const uint32_t end = mis + (uint32_t)d_src_len;
        const uint64_t limit64 = d_num >= 0 && (uint64_t)d_num < d_cap ? (uint64_t)d_num : d_cap;
        const uint32_t limit = limit64 > MAX_TOTAL ? MAX_TOTAL : (uint32_t)limit64; // values that may be written
        if (tid == 0) { sh.pos = mis; sh.pending = 0; sh.total = 0; sh.state = 0; sh.n_runs = 0; }
        pw_barrier();
        uint32_t state;
#ifdef PQD_PAGE_TIMING
        uint64_t t_load = 0, t_walk = 0, t_exp = 0, t_c = 0, t0 = __builtin_readcyclecounter(), t1;
        const uint64_t t_first = t0;
#define PQD_TICK(acc) t1 = __builtin_readcyclecounter(); acc += t1 - t0; t0 = t1;
#else
#define PQD_TICK(acc)
#endif
        for (;;) {
            const uint32_t pos = sh.pos, G0 = sh.total, pending = sh.pending;
            const uint32_t w_lo = pos & ~15u;
            const uint32_t w_hi = w_lo + PW_WIN < end ? w_lo + PW_WIN : end;
            // ---- the window: aligned chunks that hold a byte of the page ----
            for (uint32_t c = tid; c < (PW_WIN + PW_SLACK) / 16 && w_lo + c * 16 < end; c += ET)
                *reinterpret_cast<u32x4*>(&sh.win[(GROUP_STAGE_PAD + c * 16) >> 2]) = *(gptr<const u32x4>)(base + w_lo + c * 16);
            // the part of a bit-packed run in progress that the window holds
            const uint32_t room = (w_hi - pos) / bw;
            const uint32_t gp = pending < room ? pending : room, pend_left = pending - gp;
            const uint32_t cur0 = pos + gp * bw, pend_vals = gp * 8;
            pw_barrier();
            // ---- A: the distance to the next header, at every position the window holds ----
            if constexpr (T >= 512) {
                constexpr uint32_t PER = PW_WIN / T;
                const uint32_t* w = sh.win + GROUP_STAGE_PAD / 4;
                const uint32_t size = w_hi - w_lo;
                uint32_t jc[PER]; // distance (low half, saturated) and headers passed (high half), as in sh.jc
#pragma unroll
                for (uint32_t i = 0; i < PER; i++) {
                    const uint32_t rel = tid + i * ET;
                    jc[i] = 0;
                    if (rel < size) {
                        const uint32_t dw = rel >> 2;
                        const uint32_t a0 = __builtin_amdgcn_alignbit(w[dw + 1], w[dw], (rel & 3) * 8);
                        const uint32_t x = pw_advance(a0, w_lo + rel, w_hi, end, bw);
                        const uint32_t j = x < 0xffffu ? x : 0xffffu; // anything this long leaves the window
                        jc[i] = j | (x != 0 ? 0x10000u : 0u);
                        sh.adv[rel] = (uint16_t)j;
                        sh.jc[0][rel] = jc[i];
                    }
                }
                pw_barrier();
                // twice as many headers on per round: a chain that has stopped, or stands behind the window, stays
                for (uint32_t lvl = 0; lvl < PAGE_HOPS_LOG2; lvl++) {
#pragma unroll
                    for (uint32_t i = 0; i < PER; i++) {
                        const uint32_t rel = tid + i * ET, q = rel + (jc[i] & 0xffffu);
                        if ((jc[i] >> 16) == (1u << lvl) && q < size) {
                            const uint32_t v = sh.jc[lvl & 1][q];
                            const uint32_t j = (jc[i] & 0xffffu) + (v & 0xffffu);
                            jc[i] = (j < 0xffffu ? j : 0xffffu) | ((jc[i] & 0xffff0000u) + (v & 0xffff0000u));
                        }
                        if (rel < size) sh.jc[(lvl + 1) & 1][rel] = jc[i];
                    }
                    pw_barrier();
                }
            } else {
                const uint32_t* w = sh.win + GROUP_STAGE_PAD / 4;
                for (uint32_t rel = tid; rel < w_hi - w_lo; rel += ET) {
                    const uint32_t dw = rel >> 2;
                    const uint32_t a0 = __builtin_amdgcn_alignbit(w[dw + 1], w[dw], (rel & 3) * 8);
                    const uint32_t x = pw_advance(a0, w_lo + rel, w_hi, end, bw);
                    sh.adv[rel] = (uint16_t)(x < 0xffffu ? x : 0xffffu); // anything this long leaves the window
                }
                pw_barrier();
            }
            PQD_TICK(t_load)
            // ---- B: the chain of headers ----
            if (wave == 0) {
                if (pend_left > 0) { // the run in progress fills the window
                    if (lane == 0) { sh.n_hdr = 0; sh.cur_end = cur0; sh.reason = 0; }
                } else {
                    if constexpr (T >= 512) page_chase_hops(sh, cur0 - w_lo, w_lo, w_hi, end);
                    else page_chase(sh, cur0 - w_lo, w_lo, w_hi, end);
                }
            }
            pw_barrier();
            PQD_TICK(t_walk)
            // ---- C: the headers in full, their runs' first values, the table ----
            const uint32_t n_hdr = sh.n_hdr;
            uint32_t nvals = 0, listed = 0, in_off = 0, value = 0, kind = 0, more = 0, next_pos = 0;
            if (tid < n_hdr) {
                const uint32_t rel = sh.hdr[tid], dw = rel >> 2;
                const uint32_t* w = sh.win + GROUP_STAGE_PAD / 4;
                const uint32_t a0 = __builtin_amdgcn_alignbit(w[dw + 1], w[dw], (rel & 3) * 8);
                uint32_t k = 1, u = 0;
                pw_varint(a0, k, u); // it parsed in phase A
                const uint32_t count = u >> 1, in_abs = w_lo + rel + k;
                in_off = in_abs - mis;
                if (count == 0) {
                } else if (u & 1) {
                    const uint32_t fit = in_abs < w_hi ? (w_hi - in_abs) / bw : 0;
                    const uint32_t g = count < fit ? count : fit;
                    nvals = g * 8; listed = g > 0; kind = 1;
                    more = count - g; next_pos = in_abs + g * bw; // more > 0: the rest of the run is the next round's
                } else {
                    nvals = count; listed = 1; value = (a0 >> (8 * k)) & 0xff;
                }
            }
            const uint64_t packed = ((uint64_t)listed << 40) | nvals;
            const uint64_t incl = wave_inclusive_scan<uint64_t>(packed);
            if (lane == 63) sh.wsum[wave] = incl;
            pw_barrier();
            uint64_t before = 0, all = 0;
#pragma unroll
            for (uint32_t w = 0; w < ET / 64; w++) {
                const uint64_t t = sh.wsum[w];
                if (w < wave) before += t;
                all += t;
            }
            const uint64_t excl = before + incl - packed;
            const uint32_t has_pend = gp > 0 ? 1u : 0u;
            if (listed) {
                u32x4 e;
                e.x = G0 + pend_vals + (uint32_t)(excl & 0xffffffffffull);
                e.y = in_off; e.z = value; e.w = (nvals << 1) | kind;
                sh.tab[has_pend + (uint32_t)(excl >> 40)] = e;
            }
            if (tid == 0 && has_pend) {
                u32x4 e;
                e.x = G0; e.y = pos - mis; e.z = 0; e.w = (pend_vals << 1) | 1u;
                sh.tab[0] = e;
            }
            const uint64_t G1_64 = (uint64_t)G0 + pend_vals + (all & 0xffffffffffull);
            const uint32_t nr = has_pend + (uint32_t)(all >> 40);
            if (tid == (n_hdr > 0 ? n_hdr - 1 : 0)) { // one thread moves the walk on
                uint32_t st = sh.reason, np = sh.cur_end, npend = 0;
                if (n_hdr == 0 && pend_left > 0) { np = cur0; npend = pend_left; st = 0; }
                else if (n_hdr > 0 && more > 0) { np = next_pos; npend = more; st = 0; }
                if (G1_64 > MAX_TOTAL) st = 2;
                sh.pos = np; sh.pending = npend; sh.total = (uint32_t)G1_64; sh.state = st; sh.n_runs += n_hdr;
            }
            pw_barrier();
            state = sh.state;
            PQD_TICK(t_c)
            if (state == 2) break;
            // ---- D: the values ----
            const uint32_t G1 = (uint32_t)G1_64;
            if (G0 < limit && G1 > G0)
                page_expand<T / 64>(sh.tab, nr, sh.win, (int64_t)w_lo - (int64_t)mis, pg, G0, G1 < limit ? G1 : limit, bw, share, shares);
            if (state == 1) break;
            pw_barrier(); // the window and the table are rewritten
            PQD_TICK(t_exp)
        }
        PQD_TICK(t_exp)
        if (state == 1) {
            if (tid == 0 && share == 0) {
                const uint64_t total = sh.total;
                const uint64_t n_out = pg.num_values >= 0 && (uint64_t)pg.num_values < total ? (uint64_t)pg.num_values : total;
                const bool too_small = n_out > pg.dst_cap;
                const uint64_t w = too_small ? pg.dst_cap : n_out;
                res->status = too_small ? PQD_ERR_DST_TOO_SMALL : PQD_OK;
                res->arg = too_small ? (int64_t)n_out : 0;
                res->bw = (int32_t)bw;
                res->out_values = w;
                res->out_bytes = w;
                res->consumed = sh.pos - mis;
                res->n_runs = sh.n_runs < pg.runs_cap ? sh.n_runs : pg.runs_cap;
                res->total = total;
                res->aux = ~0ull;
#ifdef PQD_PAGE_TIMING
                res->arg = (int64_t)((t_c << 32) | (t_walk & 0xffffffffull));
                res->consumed = (t_load << 32) | (t_exp & 0xffffffffull);
                res->total = ((__builtin_readcyclecounter() - t_first) << 32) | (uint32_t)sh.n_runs;
#endif
#ifdef PQD_PAGE_TIMING2
                res->arg = (int64_t)t_entry;
                res->consumed = __builtin_readcyclecounter();
#endif
            }
            return;
        }
    }
    // ---- everything else is left to rle_page_retry_kernel: the page is marked, and so is the launch (launch numbers
    // only grow, so the word needs no clearing between launches: the retry kernel compares it with its own) ----
    if (tid == 0 && share == 0) {
        res->aux = PAGE_RETRY;
        atomicMax(retry_word, launch);
    }
}

// The pages rle_page_kernel gave up on: the page's run table by one wavefront, then tile by tile,
// as the index + expand kernels do it (a kernel of its own: the index needs twice the registers).
// Usually there are none, so the grid is small: workgroup b looks at the pages b, b + grid, ...,
// a thread per page, and takes the flagged ones in turn.
__global__ __launch_bounds__(ET) void rle_page_retry_kernel(const DevPage* __restrict__ pages, DevResult* __restrict__ results,
                                                            const uint32_t* __restrict__ items, uint32_t n_items,
                                                            const uint32_t* __restrict__ retry_word, uint32_t launch) {
    __shared__ EShared sh;
    __shared__ ICtl ctl;
    __shared__ unsigned long long flagged[ET / 64];
    if (*as_global(retry_word) != launch) return; // no page of this launch was left over: the usual case, one load and out
    for (uint32_t k0 = blockIdx.x; k0 < n_items; k0 += gridDim.x * ET) {
        const uint32_t k = k0 + threadIdx.x * gridDim.x;
        const bool retry = k < n_items && as_global((const DevResult*)&results[items[k]])->aux == PAGE_RETRY;
        const unsigned long long m = __ballot(retry);
        if ((threadIdx.x & 63) == 0) flagged[threadIdx.x >> 6] = m;
        __syncthreads();
        for (uint32_t w = 0; w < ET / 64; w++) {
            unsigned long long todo = flagged[w];
            while (todo) {
                const uint32_t t = w * 64 + (uint32_t)__builtin_ctzll(todo);
                todo &= todo - 1;
                const uint32_t item = items[k0 + t * gridDim.x];
                const DevPage& pg = pages[item];
                DevResult* res = &results[item];
                if (threadIdx.x < 64) rle_index_page<true>(sh.stage, ctl, pg, res, threadIdx.x);
                __syncthreads();
                const uint32_t n_tiles = (pg.n_steps + pg.group - 1) / pg.group;
                for (uint32_t tile = 0; tile < n_tiles; tile++) {
                    expand_item_tile<FAM_RLE_BYTES>(sh, pg, res, tile);
                    __syncthreads();
                }
            }
        }
        __syncthreads(); // the flags are rewritten
    }
}

// ------------------------------------------------- dictionary held in LDS --
//
// Dictionary pages through a dictionary that is kept in LDS (rle.DictionaryEncoding.DecodeInt32 +
// newIndexedPage + Dictionary.Lookup in one pass: encoding/rle/dictionary.go:30-37,
// dictionary.go:136-162, dictionaryLookup32/64 dictionary_amd64.s:578-711).  A gather from
// memory pulls a whole cache line over the CU's L1-L2 port for every index; a gather from LDS
// costs an LDS read.  So, for dense 4- and 8-byte values: a persistent workgroup of sixteen
// wavefronts per CU copies the dictionary (or as much of it as fits; indexes past that part
// still go to memory) into LDS once and then decodes a contiguous stretch of the family's
// tiles.  Inside the workgroup the wavefronts are independent: each takes whole tiles, reads
// the packed indexes of its values straight from memory (G consecutive values per lane, so a
// store instruction of the wavefront covers contiguous memory; the 8-byte loads of neighbouring
// lanes overlap inside a few cache lines and the now undisturbed L1 serves them), looks the
// values up in LDS and stores them.  There is no workgroup barrier except around a change of
// dictionary.  A tile that is not inside one run copies the run-table entries that meet it to
// a table of the wavefront's own in LDS and every lane bisects there.

constexpr int LDT = 1024;                 // threads of a persistent workgroup
constexpr uint32_t LD_WAVES = LDT / 64;
constexpr uint32_t LD_INDEX_BYTES = IRING_BYTES + 64; // LDS of a wavefront while it indexes a page: ring + control block
static_assert(sizeof(ICtl) <= 64, "the control block of the index fits its slot");
typedef uint64_t u64_a1 __attribute__((aligned(1)));
typedef uint32_t u32x4_a1 __attribute__((ext_vector_type(4), aligned(1)));

__device__ inline void ld_wave_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
__device__ inline uint32_t ld_uniform(uint32_t v) { return (uint32_t)__builtin_amdgcn_readfirstlane((int)v); }
__device__ inline uint64_t ld_uniform(uint64_t v) { return ((uint64_t)ld_uniform((uint32_t)(v >> 32)) << 32) | ld_uniform((uint32_t)v); }

// 64 bits of the stream starting at bit `bit` of src (little-endian).  `lim` is the end of the
// last aligned 16-byte chunk that holds a byte of the page: whole 8-byte loads stay below it,
// the last bytes are assembled one at a time.
__device__ inline uint64_t read_bits(gptr<const uint8_t> src, const uint8_t* lim, uint64_t bit) {
    gptr<const uint8_t> q = src + (bit >> 3);
    const uint32_t s = (uint32_t)bit & 7;
    uint64_t w = 0;
    if ((const uint8_t*)q + 8 <= lim) {
        w = *(gptr<const u64_a1>)q;
    } else {
        for (uint32_t b = 0; b < 8; b++)
            if ((const uint8_t*)q + b < lim) w |= (uint64_t)q[b] << (8 * b);
    }
    return w >> s;
}

// G consecutive fields of `bw` bits from bit `bit` on.
template <int G>
__device__ inline void read_fields(gptr<const uint8_t> src, const uint8_t* lim, uint64_t bit, uint32_t bw, uint32_t mask,
                                   uint32_t (&v)[G]) {
    if (G * bw + 7 <= 64) {
        const uint64_t w = read_bits(src, lim, bit);
#pragma unroll
        for (int g = 0; g < G; g++) v[g] = (uint32_t)(w >> (g * bw)) & mask;
    } else if (G > 2 && 2 * bw + 7 <= 64) {
#pragma unroll
        for (int g = 0; g < G; g += 2) {
            const uint64_t w = read_bits(src, lim, bit + (uint64_t)g * bw);
            v[g] = (uint32_t)w & mask;
            v[(g + 1) % G] = (uint32_t)(w >> bw) & mask;
        }
    } else {
#pragma unroll
        for (int g = 0; g < G; g++) v[g] = (uint32_t)read_bits(src, lim, bit + (uint64_t)g * bw) & mask;
    }
}

template <int ELEM> struct LdElem;
template <> struct LdElem<8> { typedef uint64_t T; };
template <> struct LdElem<4> { typedef uint32_t T; };

// What a wavefront knows about the item its current tile belongs to.
struct LdItem {
    const DevPage* pg;
    DevResult* res;
    uint64_t total, n_runs, n_out;
    uint64_t tile_lo, tile_hi;     // the item's tiles in the family's grid
    uint32_t bw;
    Run only;                      // the first run of the page (pages that are one run have no other)
};

template <int ELEM>
struct LdDict {
    const typename LdElem<ELEM>::T* lds;  // entries [lo, lo + lds_n)
    gptr<const typename LdElem<ELEM>::T> mem;
    uint32_t lds_n, lo;
    uint32_t launch;               // number of this launch of the batch (tags the out-of-range positions)
    uint64_t len;
};

// The first position of a page whose index lies outside the dictionary is kept as a maximum of
// (launch << 32 | ~position): a later launch of the batch outranks what an earlier one left, so
// nothing has to be cleared in between, and inside a launch the least position wins.
__device__ inline void ld_record_bad(DevResult* res, uint32_t launch, uint64_t pos) {
    atomicMax(reinterpret_cast<unsigned long long*>(&res->bad), ((unsigned long long)launch << 32) | (0xffffffffull - (pos & 0xffffffffull)));
}

// dict[idx] for G indexes at page positions pos.. (zero and a recorded position for an index outside the dictionary)
template <int ELEM, int G>
__device__ inline void ld_lookup(const LdDict<ELEM>& d, DevResult* res, uint64_t pos, const uint32_t (&v)[G],
                                 typename LdElem<ELEM>::T (&x)[G]) {
#pragma unroll
    for (int g = 0; g < G; g++) {
        const uint32_t idx = v[g], rel = idx - d.lo;
        if (rel < d.lds_n) x[g] = d.lds[rel];
        else if ((uint64_t)idx < d.len) x[g] = d.mem[idx];
        else {
            x[g] = 0;
            ld_record_bad(res, d.launch, pos + g);
        }
    }
}

// Stores the values of page positions [pos, pos + G) that lie below `end`.
template <int ELEM, int G>
__device__ inline void ld_store(const DevPage& pg, uint64_t pos, uint64_t end, const typename LdElem<ELEM>::T (&x)[G]) {
    gptr<uint8_t> p = as_global(pg.dst) + pos * ELEM;
    if (G * ELEM == 16 && pos + G <= end) { // a full step: one 16-byte store
        if (ELEM == 8) {
            u64x2_a8 w;
            w.x = x[0]; w.y = x[1 % G];
            *(gptr<u64x2_a8>)p = w;
        } else {
            u32x4_a4 w;
            w.x = (uint32_t)x[0]; w.y = (uint32_t)x[1 % G]; w.z = (uint32_t)x[2 % G]; w.w = (uint32_t)x[3 % G];
            *(gptr<u32x4_a4>)p = w;
        }
    } else {
#pragma unroll
        for (int g = 0; g < G; g++) {
            if (pos + g >= end) break;
            if (ELEM == 8) {
                u32x2_a4 w;
                w.x = (uint32_t)x[g]; w.y = (uint32_t)((uint64_t)x[g] >> 32);
                *(gptr<u32x2_a4>)(p + g * 8) = w;
            } else {
                *(gptr<uint32_t>)(p + g * 4) = (uint32_t)x[g];
            }
        }
    }
}

// One tile [t0, t0 + n) of an item, by one wavefront.
template <int ELEM>
__device__ inline void ld_wave_tile(const LdDict<ELEM>& dict, const LdItem& it, uint64_t t0, uint32_t n, u32x4* tab, uint32_t tab_cap) {
    typedef typename LdElem<ELEM>::T T;
    constexpr int G = 16 / ELEM;                 // values per lane and step: one 16-byte store
    constexpr uint32_t STEP = 64 * G;            // values per wavefront and step
    const DevPage& pg = *it.pg;
    DevResult* res = it.res;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t bw = it.bw;
    const uint32_t mask = bw >= 32 ? 0xffffffffu : ((1u << bw) - 1);
    gptr<const uint8_t> src = as_global(pg.src);
    const uint8_t* lim = (const uint8_t*)(((uintptr_t)(pg.src + pg.src_len) + 15) & ~(uintptr_t)15);
    gptr<const Run> runs = as_global((const Run*)pg.runs);
    const uint64_t end = t0 + n;

    uint32_t n_stream = n;                       // values of the tile the stream holds; the rest are zero indexes
    if (t0 >= it.total) n_stream = 0;
    else if (end > it.total) n_stream = (uint32_t)(it.total - t0);

    uint64_t r0 = 0;
    Run first = it.only;
    if (n_stream > 0 && it.n_runs > 1) {
        r0 = as_global((const uint32_t*)pg.tile_run)[t0 / RLE_TILE];
        first = load_run(runs, r0);
    }
    const uint64_t first_end = (uint64_t)first.out_start + (first.count_kind >> 1);
    const bool single = n_stream > 0 && first_end >= t0 + n_stream;

    if (single && (first.count_kind & 1)) {
        // ---- inside one bit-packed run ----
        const uint64_t k0 = t0 - first.out_start;
        gptr<const uint8_t> packed = src + first.in_off;
#pragma unroll 4
        for (uint32_t j0 = 0; j0 < RLE_TILE; j0 += STEP) {
            const uint32_t j = j0 + lane * G;
            if (j0 >= n) break;
            uint32_t v[G];
#pragma unroll
            for (int g = 0; g < G; g++) v[g] = 0;
            if (j + G <= n_stream) {
                read_fields<G>(packed, lim, (k0 + j) * bw, bw, mask, v);
            } else {
#pragma unroll
                for (int g = 0; g < G; g++)
                    if (j + g < n_stream) v[g] = (uint32_t)read_bits(packed, lim, (k0 + j + g) * bw) & mask;
            }
            if (j < n) {
                T x[G];
                ld_lookup<ELEM, G>(dict, res, t0 + j, v, x);
                ld_store<ELEM, G>(pg, t0 + j, end, x);
            }
        }
        return;
    }
    if (single) {
        // ---- inside one repeated run: one look-up, then a fill (the zero-extended rest of the tile after it) ----
        uint32_t v[1] = {first.value};
        T x1[1];
        ld_lookup<ELEM, 1>(dict, res, t0, v, x1);
        T x[G];
#pragma unroll
        for (int g = 0; g < G; g++) x[g] = x1[0];
        for (uint32_t j = lane * G; j < n_stream; j += STEP) ld_store<ELEM, G>(pg, t0 + j, t0 + n_stream, x);
    } else if (n_stream > 0) {
        // ---- several runs meet the tile: their entries go to the wavefront's table in LDS ----
        const uint64_t avail = it.n_runs - r0;
        const uint32_t nr = avail < tab_cap ? (uint32_t)avail : tab_cap;
        for (uint32_t k = lane; k < nr; k += 64) tab[k] = *(gptr<const u32x4>)(runs + r0 + k);
        ld_wave_sync();
        uint32_t in_tile; // staged runs that start before the end of the tile's stream part
        {
            uint32_t lo = 0, hi = nr;
            while (hi - lo > 1) {
                const uint32_t mid = (lo + hi) >> 1;
                if ((uint64_t)tab[mid].x < t0 + n_stream) lo = mid; else hi = mid;
            }
            in_tile = lo + 1;
        }
        const u32x4 last = tab[in_tile - 1];
        const bool in_lds = (uint64_t)last.x + (last.w >> 1) >= t0 + n_stream;
        for (uint32_t j0 = 0; j0 < n_stream; j0 += STEP) {
            const uint32_t j = j0 + lane * G;
            if (j >= n_stream) continue;
            const uint64_t pos = t0 + j;
            uint32_t v[G];
#pragma unroll
            for (int g = 0; g < G; g++) v[g] = 0;
            if (in_lds) {
                uint32_t lo = 0, hi = in_tile; // last staged run with out_start <= pos
                while (hi - lo > 1) {
                    const uint32_t mid = (lo + hi) >> 1;
                    if ((uint64_t)tab[mid].x <= pos) lo = mid; else hi = mid;
                }
                u32x4 e = tab[lo];
#pragma unroll
                for (int g = 0; g < G; g++) {
                    const uint64_t q = pos + g;
                    if (j + g >= n_stream) break;
                    if (q >= (uint64_t)e.x + (e.w >> 1) && lo + 1 < in_tile) e = tab[++lo]; // every run holds at least one value
                    v[g] = (e.w & 1) ? (uint32_t)read_bits(src + e.y, lim, (q - e.x) * bw) & mask : e.z;
                }
            } else {
                // more runs than the table holds: bisection over the run table in memory
#pragma unroll
                for (int g = 0; g < G; g++) {
                    const uint64_t q = pos + g;
                    if (j + g >= n_stream) break;
                    uint64_t lo = r0, hi = it.n_runs;
                    while (hi - lo > 1) {
                        const uint64_t mid = (lo + hi) >> 1;
                        if ((uint64_t)runs[mid].out_start <= q) lo = mid; else hi = mid;
                    }
                    const Run rr = load_run(runs, lo);
                    v[g] = (rr.count_kind & 1) ? (uint32_t)read_bits(src + rr.in_off, lim, (q - rr.out_start) * bw) & mask : rr.value;
                }
            }
            T x[G];
            // positions past the stream inside this step are written by the tail below
            uint32_t valid = n_stream - j < (uint32_t)G ? n_stream - j : (uint32_t)G;
            ld_lookup<ELEM, G>(dict, res, pos, v, x);
            ld_store<ELEM, G>(pg, pos, pos + valid, x);
        }
        ld_wave_sync(); // the table is rewritten for the next tile only after these reads
    }
    // ---- the zero-extended rest: index 0 ----
    if (n_stream < n) {
        uint32_t v[1] = {0};
        T x1[1];
        // an empty dictionary fails at the first such position
        ld_lookup<ELEM, 1>(dict, res, t0 + n_stream, v, x1);
        for (uint32_t j = n_stream + lane; j < n; j += 64) ld_store<ELEM, 1>(pg, t0 + j, end, x1);
    }
}

// A tile as a wavefront holds it between looking it up and decoding it (all wave-uniform, and
// small: two of these and the item of the tile looked up last live in scalar registers).
struct LdTile {
    uint64_t u;                 // the tile's number in the family's grid
    uint32_t wi;                // position of its item in the family list
    uint32_t n;                 // values to write; 0: nothing
    bool fast;                  // a whole tile inside one bit-packed run: its packed words are fetched a tile ahead
    // fast tiles only:
    gptr<const uint8_t> packed; // byte of the run's packed data that holds the tile's first field
    gptr<uint8_t> dst;          // where the tile's first value goes
    DevResult* res;
    uint64_t t0;                // first value of the tile in its page
    uint32_t s0, bw;            // bit of the first field inside that byte; field width
};

// Position inside the tile of the p-th pair of values of a lane: sixteen pairs per lane and tile
// (eight per half tile), laid out so that the values of one 16-byte store are consecutive.
template <int ELEM>
__device__ inline uint32_t ld_pair_pos(uint32_t p, uint32_t lane) {
    return ELEM == 8 ? p * 128 + lane * 2 : (p >> 1) * 256 + lane * 4 + (p & 1) * 2;
}
constexpr uint32_t LD_PAIRS = RLE_TILE / (64 * 2) / 2; // pairs per lane and HALF tile: what is fetched ahead at a time
constexpr uint32_t LD_FAST_MAX_BW = 28; // two fields and up to seven bits of shift fit one 8-byte load

// A page that is a width byte and ONE bit-packed run up to its last byte -- what indexes without
// repeats come as -- needs no run table: everything the decode wants to know stands in its
// first bytes.  Same rules as the index (rle.go:352-419, rle/dictionary.go:30-37), cut down to
// this shape; anything else answers false and goes through the index.
__device__ inline bool ld_simple_page(const DevPage& pg, uint32_t& bw, uint64_t& total, uint32_t& in_off) {
    if (pg.src_len < 3 || pg.src_len > 0xffffffffull) return false;
    const uint8_t* lim = (const uint8_t*)(((uintptr_t)(pg.src + pg.src_len) + 15) & ~(uintptr_t)15);
    const uint64_t w = ld_uniform(read_bits(as_global(pg.src), lim, 0));
    bw = (uint32_t)w & 0xff;
    if (bw > 32) return false;
    const uint32_t a0 = (uint32_t)(w >> 8);
    uint32_t u, k;
    if (!(a0 & 0x80)) { u = a0 & 0x7f; k = 1; }
    else if (!(a0 & 0x8000)) { u = (a0 & 0x7f) | ((a0 >> 1) & 0x3f80); k = 2; }
    else if (!(a0 & 0x800000)) { u = (a0 & 0x7f) | ((a0 >> 1) & 0x3f80) | ((a0 >> 2) & 0x1fc000); k = 3; }
    else if (!(a0 & 0x80000000u)) { u = (a0 & 0x7f) | ((a0 >> 1) & 0x3f80) | ((a0 >> 2) & 0x1fc000) | ((a0 >> 3) & 0xfe00000); k = 4; }
    else return false;
    const uint64_t count = u >> 1;
    if (!(u & 1) || count == 0 || count * 8 > MAX_TOTAL) return false;
    if (1 + k + count * bw != pg.src_len) return false;
    total = count * 8;
    in_off = 1 + k;
    return true;
}

// Is the page's run table the work of the chunked kernels (rle_chunk_resolve / rle_chunk_emit)?
__device__ inline bool ld_chunk_indexed(const DevPage& pg, const uint32_t* __restrict__ pk_flag) {
    return pg.ck_index >= 0 && pk_flag[pg.ck_index] == 0;
}

// Makes `it` the item that holds tile u (reloaded only when that is another one than before; `wi`
// follows the family list forwards) and returns how many values of the tile are to be written.
// The wavefront that takes an item's first tile also writes the item's result fields.
template <int ELEM>
__device__ inline uint32_t ld_item_of(const DevPage* __restrict__ pages, DevResult* __restrict__ results, const uint32_t* __restrict__ items,
                                      const uint64_t* __restrict__ tile_prefix, const uint32_t* __restrict__ pk_flag, uint64_t u,
                                      LdItem& it, uint32_t& wi) {
    const uint32_t lane = threadIdx.x & 63;
    bool simple = false;
    if (it.pg == nullptr || u >= it.tile_hi) {
        while (tile_prefix[wi + 1] <= u) wi++;
        const uint32_t item = items[wi];
        it.pg = &pages[item];
        it.res = &results[item];
        it.tile_lo = tile_prefix[wi];
        it.tile_hi = tile_prefix[wi + 1];
        uint32_t in_off = 0;
        simple = !ld_chunk_indexed(*it.pg, pk_flag) && ld_simple_page(*it.pg, it.bw, it.total, in_off);
        if (simple) {
            it.n_runs = 1;
            it.only.out_start = 0; it.only.in_off = in_off; it.only.value = 0; it.only.count_kind = ((uint32_t)it.total << 1) | 1u;
        } else { // what the index left (this workgroup's own wavefronts, or the chunked kernels before this launch)
            gptr<const DevResult> r = as_global((const DevResult*)it.res);
            it.total = ld_uniform(r->total);
            it.n_runs = ld_uniform(r->n_runs);
            it.bw = ld_uniform((uint32_t)r->bw);
            const u32x4 f = *(gptr<const u32x4>)(r->first);
            it.only.out_start = ld_uniform(f.x); it.only.in_off = ld_uniform(f.y); it.only.value = ld_uniform(f.z); it.only.count_kind = ld_uniform(f.w);
        }
        it.n_out = output_count(*it.pg, it.total, true);
    }
    const DevPage& pg = *it.pg;
    const uint64_t tile = u - it.tile_lo;
    uint64_t n_out = it.n_out;
    const uint64_t fit = pg.dst_cap / ELEM;
    if (tile == 0 && lane == 0) {
        // (tile 0 is the first tile of its item this wavefront sees, so `simple` was computed above)
        DevResult* res = it.res;
        if (simple) {
            res->status = PQD_OK;
            res->bw = (int32_t)it.bw;
            res->arg = 0;
            res->n_runs = 1;
            res->total = it.total;
            res->consumed = pg.src_len;
        }
        const bool too_small = n_out > fit;
        const uint64_t w = too_small ? fit : n_out;
        res->out_values = w;
        res->out_bytes = w * ELEM;
        res->need_bytes = too_small ? n_out * ELEM : 0; // the index may still be writing the status: kept apart from it
    }
    if (n_out > fit) n_out = fit;
    const uint64_t t0 = tile * RLE_TILE;
    return t0 >= n_out ? 0u : (n_out - t0 < RLE_TILE ? (uint32_t)(n_out - t0) : RLE_TILE);
}

// Looks tile u up and says whether the fast form applies.
template <int ELEM>
__device__ inline void ld_setup(const DevPage* __restrict__ pages, DevResult* __restrict__ results, const uint32_t* __restrict__ items,
                                const uint64_t* __restrict__ tile_prefix, const uint32_t* __restrict__ pk_flag, uint64_t u, LdItem& it,
                                uint32_t& wi, bool allow_fast, LdTile& out) {
    out.n = ld_item_of<ELEM>(pages, results, items, tile_prefix, pk_flag, u, it, wi);
    out.u = u;
    out.wi = wi;
    out.fast = false;
    const DevPage& pg = *it.pg;
    const uint64_t t0 = (u - it.tile_lo) * RLE_TILE;
    // the first run of the page covers the whole tile (the usual page has one run; tiles of later runs take the slow form)
    const Run& f = it.only;
    if (allow_fast && out.n == RLE_TILE && (f.count_kind & 1) && it.bw <= LD_FAST_MAX_BW &&
        (uint64_t)f.out_start + (f.count_kind >> 1) >= t0 + RLE_TILE && t0 + RLE_TILE <= it.total) {
        const uint64_t bit0 = (t0 - f.out_start) * it.bw;
        const uint64_t first_byte = (uint64_t)f.in_off + (bit0 >> 3);
        const uint64_t last_load_end = first_byte + ((7 + (uint64_t)RLE_TILE * it.bw) >> 3) + 8;
        const uint64_t lim = (((uintptr_t)(pg.src + pg.src_len) + 15) & ~(uintptr_t)15) - (uintptr_t)pg.src;
        if (last_load_end <= lim) {
            out.fast = true;
            out.packed = as_global(pg.src) + first_byte;
            out.dst = as_global(pg.dst) + t0 * ELEM;
            out.res = it.res;
            out.t0 = t0;
            out.s0 = (uint32_t)bit0 & 7;
            out.bw = it.bw;
        }
    }
}

template <int ELEM>
__device__ inline void ld_prefetch(const LdTile& tl, uint32_t half, uint64_t (&w)[LD_PAIRS]) {
    const uint32_t lane = threadIdx.x & 63;
#pragma unroll
    for (uint32_t p = 0; p < LD_PAIRS; p++) {
        const uint32_t bit = tl.s0 + ld_pair_pos<ELEM>(half * LD_PAIRS + p, lane) * tl.bw; // < 8 + 2048 * 28
        w[p] = *(gptr<const u64_a1>)(tl.packed + (bit >> 3));
    }
}

// Half of a fast tile out of its prefetched words: unpack, look up, store.  Straight-line code:
// every lane reads LDS (at entry 0 when its index lies elsewhere) and, when part of the
// dictionary stayed in memory (PART), memory too (again at entry 0 when the index lies
// elsewhere: those lanes share one cache line); an index outside the dictionary yields zero
// and the first such position of the lane is reported once, after the stores.
template <int ELEM, bool PART>
__device__ inline void ld_fast_half(const LdDict<ELEM>& dict, const LdTile& tl, uint32_t half, const uint64_t (&w)[LD_PAIRS]) {
    typedef typename LdElem<ELEM>::T T;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t bw = tl.bw;
    const uint32_t mask = (1u << bw) - 1; // bw <= 28 here
    const uint32_t len = dict.len > 0xffffffffull ? 0xffffffffu : (uint32_t)dict.len; // indexes have at most 28 bits
    uint32_t bad = 0xffffffffu; // first position of this lane, inside the tile, whose index is outside the dictionary
    auto look = [&](uint32_t idx, uint32_t j) -> T {
        const bool in_lds = idx < dict.lds_n;
        T x = dict.lds[in_lds ? idx : 0];
        if (PART) {
            const bool in_mem = !in_lds && idx < len;
            const T m = dict.mem[in_mem ? idx : 0];
            x = in_mem ? m : x;
        }
        const bool ok = idx < len;
        bad = ok || bad < j ? bad : j;
        return ok ? x : (T)0;
    };
    if (ELEM == 8) {
#pragma unroll
        for (uint32_t p = 0; p < LD_PAIRS; p++) {
            const uint32_t j = ld_pair_pos<8>(half * LD_PAIRS + p, lane);
            const uint32_t s = (tl.s0 + j * bw) & 7;
            u64x2_a8 o;
            o.x = look((uint32_t)(w[p] >> s) & mask, j);
            o.y = look((uint32_t)(w[p] >> (s + bw)) & mask, j + 1);
            *(gptr<u64x2_a8>)(tl.dst + (uint64_t)j * 8) = o;
        }
    } else {
#pragma unroll
        for (uint32_t p = 0; p < LD_PAIRS; p += 2) {
            const uint32_t j = ld_pair_pos<4>(half * LD_PAIRS + p, lane);
            const uint32_t s0 = (tl.s0 + j * bw) & 7, s1 = (tl.s0 + (j + 2) * bw) & 7;
            u32x4_a4 o;
            o.x = (uint32_t)look((uint32_t)(w[p] >> s0) & mask, j);
            o.y = (uint32_t)look((uint32_t)(w[p] >> (s0 + bw)) & mask, j + 1);
            o.z = (uint32_t)look((uint32_t)(w[p + 1] >> s1) & mask, j + 2);
            o.w = (uint32_t)look((uint32_t)(w[p + 1] >> (s1 + bw)) & mask, j + 3);
            *(gptr<u32x4_a4>)(tl.dst + (uint64_t)j * 4) = o;
        }
    }
    if (bad != 0xffffffffu) ld_record_bad(tl.res, dict.launch, tl.t0 + bad);
}

template <int ELEM, bool PART>
__global__ __launch_bounds__(LDT) void rle_dict_lds_kernel(const DevPage* __restrict__ pages, DevResult* __restrict__ results,
                                                           const uint32_t* __restrict__ items, uint32_t n_items,
                                                           const uint64_t* __restrict__ tile_prefix, uint64_t n_tiles,
                                                           const uint32_t* __restrict__ wg_item, const uint32_t* __restrict__ seg_last,
                                                           const uint32_t* __restrict__ pk_flag, uint32_t launch,
                                                           uint32_t dict_cap_bytes, uint32_t tab_cap) {
    typedef typename LdElem<ELEM>::T T;
    extern __shared__ __attribute__((aligned(16))) uint8_t ld_shared[];
    const uint32_t tid = threadIdx.x, wave = ld_uniform(tid >> 6);
    u32x4* tab = reinterpret_cast<u32x4*>(ld_shared + dict_cap_bytes) + (size_t)wave * tab_cap;
    // this workgroup's stretch of the family's tiles, and the items its first and last tile belong to
    const uint64_t T0 = n_tiles * blockIdx.x / gridDim.x, T1 = n_tiles * (blockIdx.x + 1) / gridDim.x;
    uint32_t li = wg_item[2 * blockIdx.x];
    // ---- the run tables of this workgroup's pages: a wavefront per page, in LDS that the dictionary
    // takes over afterwards.  Pages that are one bit-packed run need none, pages the chunked kernels
    // indexed have theirs.  A page that reaches into a neighbouring workgroup's stretch is indexed
    // there as well: both write the same table, and neither waits for the other. ----
    if (T0 < T1) {
        const uint32_t li_last = wg_item[2 * blockIdx.x + 1];
        uint32_t* ring = reinterpret_cast<uint32_t*>(ld_shared + (size_t)wave * LD_INDEX_BYTES);
        ICtl& ctl = *reinterpret_cast<ICtl*>(ld_shared + (size_t)wave * LD_INDEX_BYTES + IRING_BYTES);
        for (uint32_t k = li + wave; k <= li_last; k += LD_WAVES) {
            const uint32_t item = items[k];
            const DevPage& pg = pages[item];
            uint32_t bw0, off0;
            uint64_t total0;
            if (ld_chunk_indexed(pg, pk_flag) || ld_simple_page(pg, bw0, total0, off0)) continue;
            rle_index_page<false>(ring, ctl, pg, &results[item], tid & 63);
        }
    }
    __syncthreads(); // the tables are in place (and the rings free) before any wavefront looks a tile up
    uint64_t t = T0;
    while (t < T1) {
        // ---- a segment: the tiles of consecutive items that share one dictionary ----
        while (tile_prefix[li + 1] <= t) li++;
        const uint64_t seg_hi = tile_prefix[seg_last[li] + 1];
        const uint64_t seg_end = seg_hi < T1 ? seg_hi : T1;
        const DevPage& p0 = pages[items[li]];
        LdDict<ELEM> dict;
        dict.lds = reinterpret_cast<const T*>(ld_shared);
        dict.mem = (gptr<const T>)as_global(p0.dict);
        dict.len = p0.dict_len;
        dict.lds_n = dict.len < dict_cap_bytes / ELEM ? (uint32_t)dict.len : dict_cap_bytes / ELEM;
        dict.lo = 0;
        dict.launch = launch;
        // the fast form reads entry 0 for lanes whose index lies elsewhere: there has to be one, and all of
        // the dictionary has to be in LDS unless this is the kernel that also looks in memory
        const bool allow_fast = dict.len > 0 && (PART || dict.lds_n == dict.len);
        // ---- this wavefront's first tile: looked up, and its packed words asked for, while the dictionary comes in ----
        LdItem it;
        it.pg = nullptr; it.res = nullptr; it.total = it.n_runs = it.n_out = 0; it.tile_lo = it.tile_hi = 0; it.bw = 0;
        it.only.out_start = it.only.in_off = it.only.value = it.only.count_kind = 0;
        uint32_t wi = li;
        uint64_t u = t + wave;
        const bool have = u < seg_end;
        LdTile cur;
        uint64_t w_a[LD_PAIRS], w_b[LD_PAIRS]; // packed words of the first / second half of a fast tile
#pragma unroll
        for (uint32_t p = 0; p < LD_PAIRS; p++) w_a[p] = w_b[p] = 0;
        cur.fast = false; cur.n = 0; cur.u = 0; cur.wi = 0;
        if (have) {
            ld_setup<ELEM>(pages, results, items, tile_prefix, pk_flag, u, it, wi, allow_fast, cur);
            if (cur.fast) ld_prefetch<ELEM>(cur, 0, w_a);
        }
        {
            // The dictionary's bytes, 16 per thread and load, U loads in flight per thread.  Offsets past
            // the last whole 16 bytes are pulled back onto them (threads then copy the same bytes
            // twice): no branch around a load, so all of a pass's loads are in flight together.
            const uint32_t bytes = dict.lds_n * ELEM, body = bytes & ~15u;
            gptr<const uint8_t> g = as_global(p0.dict);
            constexpr uint32_t U = 4, PASS = U * LDT * 16;
            const uint32_t last = body >= 16 ? body - 16 : 0;
            u32x4_a1 wd[U];
            uint32_t at[U];
            if (body >= 16) {
#pragma unroll
                for (uint32_t k = 0; k < U; k++) {
                    const uint32_t c = tid * 16 + k * LDT * 16;
                    at[k] = c < last ? c : last;
                    wd[k] = *(gptr<const u32x4_a1>)(g + at[k]);
                }
            }
            __syncthreads(); // nobody still reads the dictionary of the segment before
            if (body >= 16) {
#pragma unroll
                for (uint32_t k = 0; k < U; k++) *reinterpret_cast<u32x4*>(ld_shared + at[k]) = wd[k];
                for (uint32_t base = PASS; base < body; base += PASS) {
#pragma unroll
                    for (uint32_t k = 0; k < U; k++) {
                        const uint32_t c = base + tid * 16 + k * LDT * 16;
                        at[k] = c < last ? c : last;
                        wd[k] = *(gptr<const u32x4_a1>)(g + at[k]);
                    }
#pragma unroll
                    for (uint32_t k = 0; k < U; k++) *reinterpret_cast<u32x4*>(ld_shared + at[k]) = wd[k];
                }
            }
            for (uint32_t b = body + tid; b < bytes; b += LDT) ld_shared[b] = g[b];
        }
        __syncthreads();
        // ---- the wavefronts take the segment's tiles in turn, each on its own.  First the fast ones: the
        // packed words of half a tile are asked for before the half tile before it is decoded, so that no
        // load waits behind stores ----
        bool any_slow = false;
        if (have) {
            for (;;) {
                const uint64_t un = u + LD_WAVES;
                const bool more = un < seg_end;
                LdTile nxt;
                nxt.fast = false; nxt.n = 0; nxt.u = 0; nxt.wi = 0;
                if (more) ld_setup<ELEM>(pages, results, items, tile_prefix, pk_flag, un, it, wi, allow_fast, nxt);
                if (cur.fast) {
                    ld_prefetch<ELEM>(cur, 1, w_b);
                    ld_fast_half<ELEM, PART>(dict, cur, 0, w_a);
                    if (nxt.fast) ld_prefetch<ELEM>(nxt, 0, w_a);
                    ld_fast_half<ELEM, PART>(dict, cur, 1, w_b);
                } else {
                    any_slow = any_slow || cur.n > 0;
                    if (nxt.fast) ld_prefetch<ELEM>(nxt, 0, w_a);
                }
                if (!more) break;
                cur = nxt;
                u = un;
            }
        }
        // ---- then whatever is left: tiles that several runs meet, that lie in a repeated run, partial
        // tiles, zero-extended ones (a pass of its own, so that the loop above carries none of this code's state) ----
        if (any_slow) {
            uint32_t wj = li;
            for (uint64_t v = t + wave; v < seg_end; v += LD_WAVES) {
                LdTile tl;
                LdItem it2;
                it2.pg = nullptr; it2.res = nullptr; it2.total = it2.n_runs = it2.n_out = 0; it2.tile_lo = it2.tile_hi = 0; it2.bw = 0;
                it2.only.out_start = it2.only.in_off = it2.only.value = it2.only.count_kind = 0;
                ld_setup<ELEM>(pages, results, items, tile_prefix, pk_flag, v, it2, wj, allow_fast, tl);
                if (!tl.fast && tl.n > 0) ld_wave_tile<ELEM>(dict, it2, (v - it2.tile_lo) * RLE_TILE, tl.n, tab, tab_cap);
            }
        }
        t = seg_end;
    }
}

// ---------------------------------------------------------------------------------------------
// A dictionary of a few LDS sizes, sliced over neighbouring CUs.  Random 8-byte gathers from a
// dictionary that LDS cannot hold each pull a whole cache line into the CU; the sliced form
// never gathers from memory: the persistent workgroups stand in groups of S, the s-th of a
// group keeps entries [s * E, (s + 1) * E) in LDS, ALL of a group walk the same tiles, and each
// writes just the values whose index falls into its slice (an index outside the dictionary is
// the first slice's).  The values of a tile therefore leave as S sets of masked stores from S
// CUs; the group's workgroups are b, b + 8, b + 16 ... (blocks eight apart have been observed to
// share an L2), so the partial lines meet there -- placement is a matter of speed only, every
// value is written exactly once wherever the workgroups run.  Lanes hold every 64th value of a
// tile, so one store instruction covers 64 neighbouring values.  The launch is only made for
// families all of whose dictionaries are larger than LDS and need at most S slices.  What it
// costs: every 64 bytes of output are written by S requests instead of one, and the write path
// of a CU takes a request per four or five cycles however few of its bytes are enabled: at two
// slices that is cheaper than the gathers it replaces, from four on it is not (measured).

typedef uint32_t u32_a1 __attribute__((aligned(1)));
constexpr uint32_t LD_SLICE_STEPS = RLE_TILE / 64 / 2; // values per lane and HALF tile
constexpr uint32_t LD_SLICE_MAX_BW = 25;               // a field and up to seven bits of shift fit one 4-byte load
constexpr uint32_t LD_MAX_SLICES = 8;

enum { LD_WHOLE = 0, LD_PART = 1, LD_SLICED = 2 };
template <int MODE> struct LdWords { uint64_t w[LD_PAIRS]; };
template <> struct LdWords<LD_SLICED> { uint32_t w[LD_SLICE_STEPS]; };

template <int ELEM, int MODE>
__device__ inline void ld_fetch(const LdTile& tl, uint32_t half, LdWords<MODE>& W) {
    if constexpr (MODE == LD_SLICED) {
        // a lane's fields lie 64 fields = 8 * bw bytes apart: one address per lane, a uniform step (all below 2^24)
        const uint32_t lane = threadIdx.x & 63;
        const uint32_t bit = tl.s0 + __umul24(half * LD_SLICE_STEPS * 64 + lane, tl.bw); // < 8 + 2048 * 25
        gptr<const uint8_t> p = tl.packed + (bit >> 3);
        const uint32_t step = 8 * tl.bw;
#pragma unroll
        for (uint32_t q = 0; q < LD_SLICE_STEPS; q++) W.w[q] = *(gptr<const u32_a1>)(p + q * step);
    } else {
        ld_prefetch<ELEM>(tl, half, W.w);
    }
}

// Half of a fast tile for the workgroup that holds dictionary entries [dict.lo, dict.lo + dict.lds_n):
// the values whose index lies there, and for the first slice (`owner`) zero for an index outside
// the dictionary, whose first position is reported as everywhere else.
template <int ELEM>
__device__ inline void ld_slice_half(const LdDict<ELEM>& dict, bool owner, const LdTile& tl, uint32_t half,
                                     const uint32_t (&w)[LD_SLICE_STEPS]) {
    typedef typename LdElem<ELEM>::T T;
    const uint32_t lane = threadIdx.x & 63;
    const uint32_t bw = tl.bw;
    const uint32_t mask = (1u << bw) - 1; // bw <= 25 here
    const uint32_t len = dict.len > 0xffffffffull ? 0xffffffffu : (uint32_t)dict.len;
    uint32_t bad = 0xffffffffu;
    const uint32_t j0 = half * LD_SLICE_STEPS * 64 + lane;
    const uint32_t s = (tl.s0 + __umul24(j0, bw)) & 7; // the same in every step: 64 fields are whole bytes
    // The tile's destination as a buffer: a lane that has nothing to write stores at an offset outside it, which
    // the hardware drops.  Every step is then one store instruction without a branch around it, so the waits
    // for the packed words fetched ahead count the memory operations in flight exactly and never wait for stores.
    const uint64_t dst_u = ld_uniform((uint64_t)(uintptr_t)(uint8_t*)tl.dst);
    const __amdgpu_buffer_rsrc_t rsrc = __builtin_amdgcn_make_buffer_rsrc((void*)(uintptr_t)dst_u, 0, (int)(RLE_TILE * ELEM), 0x00020000);
    constexpr uint32_t B = 8; // look-ups in flight together
#pragma unroll
    for (uint32_t q0 = 0; q0 < LD_SLICE_STEPS; q0 += B) {
        T x[B];
#pragma unroll
        for (uint32_t k = 0; k < B; k++) {
            const uint32_t rel = ((w[q0 + k] >> s) & mask) - dict.lo;
            x[k] = dict.lds[rel < dict.lds_n ? rel : 0];
        }
#pragma unroll
        for (uint32_t k = 0; k < B; k++) asm volatile("" : "+v"(x[k])); // all of them asked for before the first is stored
#pragma unroll
        for (uint32_t k = 0; k < B; k++) {
            const uint32_t q = q0 + k;
            const uint32_t idx = (w[q] >> s) & mask;
            const bool hit = idx - dict.lo < dict.lds_n;
            const bool out = owner && idx >= len;
            const uint32_t j = j0 + q * 64;
            bad = !out || bad < j ? bad : j;
            const uint32_t off = hit || out ? j * ELEM : 0xffffffffu;
            if (ELEM == 8) {
                u32x2 o;
                o.x = hit ? (uint32_t)x[k] : 0u; o.y = hit ? (uint32_t)((uint64_t)x[k] >> 32) : 0u;
                __builtin_amdgcn_raw_buffer_store_b64(o, rsrc, (int)off, 0, 0);
            } else {
                __builtin_amdgcn_raw_buffer_store_b32(hit ? (uint32_t)x[k] : 0u, rsrc, (int)off, 0, 0);
            }
        }
    }
    if (bad != 0xffffffffu) ld_record_bad(tl.res, dict.launch, tl.t0 + bad);
}

template <int ELEM, int MODE>
__device__ inline void ld_half(const LdDict<ELEM>& dict, bool owner, const LdTile& tl, uint32_t half, const LdWords<MODE>& W) {
    if constexpr (MODE == LD_SLICED) ld_slice_half<ELEM>(dict, owner, tl, half, W.w);
    else ld_fast_half<ELEM, MODE == LD_PART>(dict, tl, half, W.w);
}

// One segment (the tiles [.., seg_end) of consecutive items that share a dictionary) by one
// workgroup: entries [dict.lo, dict.lo + dict.lds_n) of the dictionary into LDS, then the tiles
// first, first + stride, ... by the wavefront.  In sliced mode those are every workgroup's of the
// group: `owner` (the first slice) writes the zeros of indexes outside the dictionary, and the
// tiles the fast form leaves are decoded whole by one workgroup each, the k-th such tile of the
// wavefront by workgroup k mod share_n (this one is share_k), which reads memory for the
// indexes outside its slice.
template <int ELEM, int MODE>
__device__ inline void ld_segment(uint8_t* ld_shared, const LdDict<ELEM>& dict, bool owner, uint32_t share_n, uint32_t share_k,
                                  const DevPage* __restrict__ pages,
                                  DevResult* __restrict__ results, const uint32_t* __restrict__ items,
                                  const uint64_t* __restrict__ tile_prefix, const uint32_t* __restrict__ pk_flag, uint32_t li,
                                  uint64_t first, uint64_t stride, uint64_t seg_end, u32x4* tab, uint32_t tab_cap) {
    const uint32_t tid = threadIdx.x;
    const bool allow_fast = dict.len > 0 && (MODE != LD_WHOLE || dict.lds_n == dict.len);
    const uint32_t max_bw = MODE == LD_SLICED ? LD_SLICE_MAX_BW : LD_FAST_MAX_BW;
    LdItem it;
    it.pg = nullptr; it.res = nullptr; it.total = it.n_runs = it.n_out = 0; it.tile_lo = it.tile_hi = 0; it.bw = 0;
    it.only.out_start = it.only.in_off = it.only.value = it.only.count_kind = 0;
    uint32_t wi = li;
    uint64_t u = first;
    const bool have = u < seg_end;
    LdTile cur;
    LdWords<MODE> Wa, Wb;
    for (auto& x : Wa.w) x = 0;
    for (auto& x : Wb.w) x = 0;
    cur.fast = false; cur.n = 0; cur.u = 0; cur.wi = 0;
    if (have) {
        ld_setup<ELEM>(pages, results, items, tile_prefix, pk_flag, u, it, wi, allow_fast, cur);
        if (cur.fast && cur.bw > max_bw) cur.fast = false;
        if (cur.fast) ld_fetch<ELEM, MODE>(cur, 0, Wa);
    }
    {
        // the slice's bytes, as rle_dict_lds_kernel copies a dictionary
        const uint32_t bytes = dict.lds_n * ELEM, body = bytes & ~15u;
        gptr<const uint8_t> g = (gptr<const uint8_t>)dict.mem + (uint64_t)dict.lo * ELEM;
        constexpr uint32_t U = 4, PASS = U * LDT * 16;
        const uint32_t last = body >= 16 ? body - 16 : 0;
        u32x4_a1 wd[U];
        uint32_t at[U];
        if (body >= 16) {
#pragma unroll
            for (uint32_t k = 0; k < U; k++) {
                const uint32_t c = tid * 16 + k * LDT * 16;
                at[k] = c < last ? c : last;
                wd[k] = *(gptr<const u32x4_a1>)(g + at[k]);
            }
        }
        __syncthreads(); // nobody still reads what LDS held for the segment before
        if (body >= 16) {
#pragma unroll
            for (uint32_t k = 0; k < U; k++) *reinterpret_cast<u32x4*>(ld_shared + at[k]) = wd[k];
            for (uint32_t base = PASS; base < body; base += PASS) {
#pragma unroll
                for (uint32_t k = 0; k < U; k++) {
                    const uint32_t c = base + tid * 16 + k * LDT * 16;
                    at[k] = c < last ? c : last;
                    wd[k] = *(gptr<const u32x4_a1>)(g + at[k]);
                }
#pragma unroll
                for (uint32_t k = 0; k < U; k++) *reinterpret_cast<u32x4*>(ld_shared + at[k]) = wd[k];
            }
        }
        for (uint32_t b = body + tid; b < bytes; b += LDT) ld_shared[b] = g[b];
    }
    __syncthreads();
    bool any_slow = false;
    if (have) {
        for (;;) {
            const uint64_t un = u + stride;
            const bool more = un < seg_end;
            LdTile nxt;
            nxt.fast = false; nxt.n = 0; nxt.u = 0; nxt.wi = 0;
            if (more) {
                ld_setup<ELEM>(pages, results, items, tile_prefix, pk_flag, un, it, wi, allow_fast, nxt);
                if (nxt.fast && nxt.bw > max_bw) nxt.fast = false;
            }
            if (cur.fast) {
                ld_fetch<ELEM, MODE>(cur, 1, Wb);
                ld_half<ELEM, MODE>(dict, owner, cur, 0, Wa);
                if (nxt.fast) ld_fetch<ELEM, MODE>(nxt, 0, Wa);
                ld_half<ELEM, MODE>(dict, owner, cur, 1, Wb);
            } else {
                any_slow = any_slow || cur.n > 0;
                if (nxt.fast) ld_fetch<ELEM, MODE>(nxt, 0, Wa);
            }
            if (!more) break;
            cur = nxt;
            u = un;
        }
    }
    if (any_slow) {
        uint32_t wj = li, k = 0;
        for (uint64_t v = first; v < seg_end; v += stride, k++) {
            if (MODE == LD_SLICED && k % share_n != share_k) continue;
            LdTile tl;
            LdItem it2;
            it2.pg = nullptr; it2.res = nullptr; it2.total = it2.n_runs = it2.n_out = 0; it2.tile_lo = it2.tile_hi = 0; it2.bw = 0;
            it2.only.out_start = it2.only.in_off = it2.only.value = it2.only.count_kind = 0;
            ld_setup<ELEM>(pages, results, items, tile_prefix, pk_flag, v, it2, wj, allow_fast, tl);
            if (tl.fast && tl.bw > max_bw) tl.fast = false;
            if (!tl.fast && tl.n > 0) ld_wave_tile<ELEM>(dict, it2, (v - it2.tile_lo) * RLE_TILE, tl.n, tab, tab_cap);
        }
    }
}

template <int ELEM>
__global__ __launch_bounds__(LDT) void rle_dict_slice_kernel(const DevPage* __restrict__ pages, DevResult* __restrict__ results,
                                                             const uint32_t* __restrict__ items, uint32_t n_items,
                                                             const uint64_t* __restrict__ tile_prefix, uint64_t n_tiles,
                                                             const uint32_t* __restrict__ wg_item, const uint32_t* __restrict__ seg_last,
                                                             const uint32_t* __restrict__ pk_flag, uint32_t launch,
                                                             uint32_t dict_cap_bytes, uint32_t tab_cap, uint32_t slice_bytes,
                                                             uint32_t n_slices, uint32_t groups_per_xcd) {
    typedef typename LdElem<ELEM>::T T;
    extern __shared__ __attribute__((aligned(16))) uint8_t ld_shared[];
    const uint32_t tid = threadIdx.x, wave = ld_uniform(tid >> 6);
    u32x4* tab = reinterpret_cast<u32x4*>(ld_shared + dict_cap_bytes) + (size_t)wave * tab_cap;
    // workgroups b, b + 8, ... , b + 8 (S - 1) form a group; the groups share out the family's tiles
    const uint32_t xcd = blockIdx.x & 7, in_xcd = blockIdx.x >> 3;
    const uint32_t slice = in_xcd % n_slices, group = xcd * groups_per_xcd + in_xcd / n_slices, n_groups = 8 * groups_per_xcd;
    const uint64_t T0 = n_tiles * group / n_groups, T1 = n_tiles * (group + 1) / n_groups;
    uint32_t li = wg_item[2 * group];
    // ---- run tables of the group's pages, by every workgroup of the group (same tables, same values; nobody waits) ----
    if (T0 < T1) {
        const uint32_t li_last = wg_item[2 * group + 1];
        uint32_t* ring = reinterpret_cast<uint32_t*>(ld_shared + (size_t)wave * LD_INDEX_BYTES);
        ICtl& ctl = *reinterpret_cast<ICtl*>(ld_shared + (size_t)wave * LD_INDEX_BYTES + IRING_BYTES);
        for (uint32_t k = li + wave; k <= li_last; k += LD_WAVES) {
            const uint32_t item = items[k];
            const DevPage& pg = pages[item];
            uint32_t bw0, off0;
            uint64_t total0;
            if (ld_chunk_indexed(pg, pk_flag) || ld_simple_page(pg, bw0, total0, off0)) continue;
            rle_index_page<false>(ring, ctl, pg, &results[item], tid & 63);
        }
    }
    __syncthreads();
    const uint32_t slice_entries = slice_bytes / ELEM;
    uint64_t t = T0;
    while (t < T1) {
        while (tile_prefix[li + 1] <= t) li++;
        const uint64_t seg_hi = tile_prefix[seg_last[li] + 1];
        const uint64_t seg_end = seg_hi < T1 ? seg_hi : T1;
        const DevPage& p0 = pages[items[li]];
        LdDict<ELEM> dict;
        dict.lds = reinterpret_cast<const T*>(ld_shared);
        dict.mem = (gptr<const T>)as_global(p0.dict);
        dict.len = p0.dict_len;
        dict.launch = launch;
        const uint64_t cap_entries = dict_cap_bytes / ELEM;
        if (dict.len > cap_entries && dict.len <= (uint64_t)slice_entries * n_slices) {
            const uint64_t lo = (uint64_t)slice * slice_entries;
            if (lo < dict.len) { // (a slice past the dictionary's end has nothing to write)
                dict.lo = (uint32_t)lo;
                dict.lds_n = dict.len - lo < slice_entries ? (uint32_t)(dict.len - lo) : slice_entries;
                const uint32_t n_active = (uint32_t)((dict.len + slice_entries - 1) / slice_entries);
                ld_segment<ELEM, LD_SLICED>(ld_shared, dict, slice == 0, n_active, slice, pages, results, items, tile_prefix, pk_flag, li,
                                            t + wave, LD_WAVES, seg_end, tab, tab_cap);
            }
        }
        t = seg_end;
    }
}

} // namespace

// Largest dynamic LDS allocation a workgroup of rle_dict_lds_kernel may ask for.
constexpr uint32_t LD_MAX_LDS = 160 * 1024 - 1024;
constexpr uint32_t LD_MIN_TAB = 32;    // run-table entries per wavefront, at least / at most
constexpr uint32_t LD_MAX_TAB = 512;
constexpr uint32_t LD_SLICE_BYTES = 128 * 1024; // of a dictionary per workgroup when it is sliced over a group of them

bool rle_dict_lds_init() {
    // dynamic LDS above the default limit has to be asked for once per kernel
    static const bool ok = [] {
        bool r = true;
        for (const void* fn : {reinterpret_cast<const void*>(&rle_dict_lds_kernel<8, false>), reinterpret_cast<const void*>(&rle_dict_lds_kernel<8, true>),
                               reinterpret_cast<const void*>(&rle_dict_lds_kernel<4, false>), reinterpret_cast<const void*>(&rle_dict_lds_kernel<4, true>),
                               reinterpret_cast<const void*>(&rle_dict_slice_kernel<8>), reinterpret_cast<const void*>(&rle_dict_slice_kernel<4>)})
            r = r && hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, (int)LD_MAX_LDS) == hipSuccess;
        (void)hipGetLastError();
        return r;
    }();
    return ok;
}

uint32_t rle_dict_lds_capacity() { return LD_MAX_LDS - LD_WAVES * LD_MIN_TAB * (uint32_t)sizeof(Run); }

// What LDS holds of a dictionary at most, and the size of a slice, under the experiment knobs.
static uint32_t ld_dict_cap(uint64_t cap_bytes) {
    uint32_t c = rle_dict_lds_capacity();
    if (cap_bytes > 0 && c > cap_bytes) c = (uint32_t)cap_bytes;
    c &= ~15u;
    return c < 16 ? 16 : c;
}
static uint32_t ld_slice_bytes(uint64_t cap_bytes, uint64_t slice_bytes_opt) {
    return (uint32_t)std::min<uint64_t>(slice_bytes_opt > 0 ? (slice_bytes_opt & ~15ull) : LD_SLICE_BYTES, ld_dict_cap(cap_bytes));
}

void rle_dict_slice_range(uint64_t cap_bytes, uint32_t n_cus, uint32_t max_slices, uint64_t slice_bytes_opt, uint64_t* above, uint64_t* upto) {
    const uint64_t sb = ld_slice_bytes(cap_bytes, slice_bytes_opt);
    const uint64_t most = std::min<uint64_t>(std::min<uint32_t>(max_slices, LD_MAX_SLICES), n_cus / 8);
    *above = ld_dict_cap(cap_bytes);
    *upto = most >= 2 && sb >= 16 ? most * sb : 0;
}

LdsPlan rle_dict_lds_plan(uint64_t n_tiles, uint64_t max_dict_bytes, uint64_t sliced_dict_bytes, bool all_sliced, uint64_t cap_bytes,
                          uint32_t n_cus, uint64_t slice_bytes_opt) {
    LdsPlan p;
    // LDS: the dictionary (or the part of it that fits), then a run table per wavefront from what is left
    p.dict_bytes = (uint32_t)std::min<uint64_t>((max_dict_bytes + 15) & ~15ull, ld_dict_cap(cap_bytes));
    if (p.dict_bytes < 16) p.dict_bytes = 16;
    p.part = max_dict_bytes > p.dict_bytes;
    p.tab_cap = std::min((LD_MAX_LDS - p.dict_bytes) / (LD_WAVES * (uint32_t)sizeof(Run)), LD_MAX_TAB);
    p.lds_bytes = std::max(p.dict_bytes + LD_WAVES * p.tab_cap * (uint32_t)sizeof(Run), LD_WAVES * LD_INDEX_BYTES);
    // workgroups: one per CU (sixteen wavefronts; their registers leave no room for a second), fewer when there
    // is little work: loading the dictionary has to pay
    uint64_t grid = n_cus;
    const uint64_t by_work = (n_tiles + 7) / 8;
    if (grid > by_work) grid = by_work;
    p.grid = (uint32_t)std::max<uint64_t>(grid, 1);
    p.groups = p.grid;
    p.slices = 1;
    // the largest dictionary of the family that is to be sliced says over how many workgroups
    if (all_sliced && sliced_dict_bytes > p.dict_bytes) {
        const uint32_t sb = ld_slice_bytes(cap_bytes, slice_bytes_opt);
        const uint64_t n_sl = (sliced_dict_bytes + sb - 1) / sb, per_xcd = n_cus / 8;
        if (n_sl >= 2 && n_sl <= LD_MAX_SLICES && n_sl <= per_xcd) {
            const uint64_t by_work_xcd = (by_work + 7) / 8;
            const uint32_t gpx = (uint32_t)std::max<uint64_t>(1, std::min<uint64_t>(per_xcd / n_sl, by_work_xcd));
            p.slices = (uint32_t)n_sl;
            p.slice_bytes = sb;
            p.groups = 8 * gpx;
            p.grid = p.groups * p.slices;
        }
    }
    return p;
}

void launch_rle_dict_lds(const DevPage* pages, DevResult* results, const uint32_t* items, uint32_t n_items,
                         const uint64_t* tile_prefix, uint64_t n_tiles, int32_t elem, const LdsPlan& plan, const uint32_t* wg_item,
                         const uint32_t* seg_last, const uint32_t* pk_flag, uint32_t launch, hipStream_t stream) {
    if (n_items == 0 || n_tiles == 0) return;
if (plan.slices > 1) {
#define PQD_SLICE_LAUNCH(E)                                                                                                     \
    hipLaunchKernelGGL((rle_dict_slice_kernel<E>), dim3(plan.grid), dim3(LDT), plan.lds_bytes, stream, pages, results, items,   \
                       n_items, tile_prefix, n_tiles, wg_item, seg_last, pk_flag, launch, plan.dict_bytes, plan.tab_cap,        \
                       plan.slice_bytes, plan.slices, plan.groups / 8)
        if (elem == 8) PQD_SLICE_LAUNCH(8); else PQD_SLICE_LAUNCH(4);
#undef PQD_SLICE_LAUNCH
        return;
    }
#define PQD_LDS_LAUNCH(E, P)                                                                                                   \
    hipLaunchKernelGGL((rle_dict_lds_kernel<E, P>), dim3(plan.grid), dim3(LDT), plan.lds_bytes, stream, pages, results, items, \
                       n_items, tile_prefix, n_tiles, wg_item, seg_last, pk_flag, launch, plan.dict_bytes, plan.tab_cap)
    if (elem == 8) { if (plan.part) PQD_LDS_LAUNCH(8, true); else PQD_LDS_LAUNCH(8, false); }
    else { if (plan.part) PQD_LDS_LAUNCH(4, true); else PQD_LDS_LAUNCH(4, false); }
#undef PQD_LDS_LAUNCH
}

void launch_rle_pages(const DevPage* pages, DevResult* results, const uint32_t* items, uint32_t n_items, uint32_t n_small, uint32_t threads,
                      uint32_t* retry_word, uint32_t launch, hipStream_t stream) {
    if (n_items == 0) return;
    if (n_small > n_items) n_small = n_items;
    if (n_small > 0) hipLaunchKernelGGL(rle_page_kernel<64>, dim3(n_small), dim3(64), 0, stream, pages, results, items, retry_word, launch);
    // The workgroup is sized to the batch: a page is a chain of dependent phases, and what the chip has left over when
    // there are few pages goes into wider phases (sixteen wavefronts share a window's distances and its expansion, and the
    // walk takes eight headers a step).  Measured on pages of 128 Ki levels: 1024 threads win up to two pages per compute
    // unit (512 pages: 53 -> 41 us), 256 threads beyond (640 pages: 58 against 60 us); 512 threads win nowhere.
    // (`threads`: 0, or the size a test or a measurement asks for)
    const uint32_t n_big = n_items - n_small;
    const uint32_t t = threads ? threads : n_big <= 512 ? 1024u : 256u;
    // ... and when even wide workgroups leave compute units idle (two of them fit one), several share a page's expansion
    const uint32_t shares = n_big > 0 ? std::min<uint32_t>(PAGE_MAX_SHARES, std::max<uint32_t>(1, PQD_PAGE_SHARE_WGS / n_big)) : 1;
    if (n_big > 0 && t == 1024)
        hipLaunchKernelGGL(rle_page_kernel<1024>, dim3(n_big, shares), dim3(1024), 0, stream, pages, results, items + n_small, retry_word, launch);
    else if (n_big > 0)
        hipLaunchKernelGGL(rle_page_kernel<256>, dim3(n_big), dim3(256), 0, stream, pages, results, items + n_small, retry_word, launch);
    hipLaunchKernelGGL(rle_page_retry_kernel, dim3(n_items < 256 ? n_items : 256), dim3(ET), 0, stream, pages, results, items, n_items, retry_word, launch);
}

void launch_rle_index(const DevPage* pages, DevResult* results, const uint32_t* items, uint32_t n_items, const uint32_t* pk_flag,
                      hipStream_t stream) {
    if (n_items == 0) return;
    hipLaunchKernelGGL(rle_index_kernel, dim3(n_items), dim3(IT), 0, stream, pages, results, items, pk_flag);
}

uint32_t rle_chunk_bytes() { return CK; }

void launch_rle_chunked(const DevPage* pages, DevResult* results, const RleChunks& c, hipStream_t stream) {
    if (c.n_chunks == 0) return;
    (void)hipMemsetAsync(c.state, 0, c.state_bytes, stream);
    hipLaunchKernelGGL(rle_chunk_resolve_kernel, dim3(c.n_chunks), dim3(CKT), 0, stream, pages, c.items, c.prefix, c.page, c.chunk,
                       c.page_pos, c.exit, c.entry, c.links, c.pk_flag);
    hipLaunchKernelGGL(rle_chunk_emit_kernel, dim3(c.n_chunks), dim3(CKT), 0, stream, pages, results, c.items, c.prefix, c.page,
                       c.chunk, c.exit, c.entry, c.links, c.runs, c.vals, c.pk_flag);
}

void launch_rle_expand(const DevPage* pages, DevResult* results, const uint32_t* items, uint32_t n_items,
                       const uint64_t* tile_prefix, const uint32_t* tile_item, uint64_t n_tiles, int32_t family,
                       hipStream_t stream) {
    if (n_items == 0 || n_tiles == 0) return;
    const dim3 grid((uint32_t)n_tiles), block(ET);
    switch (family) {
        case FAM_RLE_BYTES:
            hipLaunchKernelGGL(rle_expand_kernel<FAM_RLE_BYTES>, grid, block, 0, stream, pages, results, items, n_items, tile_prefix, tile_item);
            break;
        case FAM_RLE_DICT8:
            hipLaunchKernelGGL(rle_expand_kernel<FAM_RLE_DICT8>, grid, block, 0, stream, pages, results, items, n_items, tile_prefix, tile_item);
            break;
        case FAM_RLE_DICT4:
            hipLaunchKernelGGL(rle_expand_kernel<FAM_RLE_DICT4>, grid, block, 0, stream, pages, results, items, n_items, tile_prefix, tile_item);
            break;
        default:
            hipLaunchKernelGGL(rle_expand_kernel<FAM_RLE>, grid, block, 0, stream, pages, results, items, n_items, tile_prefix, tile_item);
            break;
    }
}

} // namespace pqd
