// plain_bytes.hip -- PLAIN byte arrays, many workgroups per page
// (plain.Encoding.DecodeByteArray, encoding/plain/plain.go:76-94).
//
// A page is a chain of `u32 length | bytes` records: where a record starts is only known once the
// record before it has been read.  The decoded values are the page with the length fields taken
// out, so byte q of record k goes to q - 4 (k + 1), and the offset of record k is its position
// minus 4 k: all it takes is the chain of record positions and their ranks.
//
// The chain is found speculatively.  A page is cut into tiles of 16 KB, a tile into 256 sub-chunks
// of 64 bytes, one per lane.  A lane guesses where the chain enters its sub-chunk -- the first
// position from which three records in a row have lengths that fit the page -- and walks the
// records that start in the sub-chunk.  A lane's walk ends at some position in a later
// sub-chunk; when that is the later lane's guess the two walks link up.  Pointer doubling over
// the 256 lanes then tells which lanes lie on the chain that starts at the tile's entry and how
// many records precede each; where the chain arrives at a position a lane had not guessed, that
// lane walks again from there and the doubling is redone.  Tiles are chained the same way, in the
// same pass: a workgroup resolves its tile from its own guess while the bytes sit in LDS, says in
// a state word of the tile what it found (entry assumed, exit, records), and looks back over the
// tiles before it for the nearest one that already knows where the chain really stands behind it;
// from there the tiles in between are added up as long as each one's guess is where the one
// before it ended (a tile that guessed wrong resolves again from its true entry and says so).
// Then the offsets are written and the bytes copied out of the same LDS tile: the page is read
// once.  Tiles are handed out by a counter per page, in order, so a tile only ever waits for
// tiles that are already running.  Wrong guesses cost time, never correctness: every position
// used in the end was reached from the start of the page.
#include <algorithm>

#include "pqd_internal.h"

namespace pqd {
namespace {

constexpr uint32_t PT = 256;            // threads per workgroup
constexpr uint32_t PSUB = 64;           // bytes of the tile a lane walks
constexpr uint32_t PTILE = PT * PSUB;   // bytes per tile
constexpr uint32_t P_NONE = 0xffffffffu;
constexpr uint32_t PLOOK = 256;         // bytes staged behind the tile: a length field may straddle its end, and the
                                        // last lanes want to see the record after their first
constexpr uint32_t PSTEPS = 6;          // records in a row that make a position a guess

// A page's state: PW_WORDS 32-bit words, then TS_WORDS 64-bit words per tile.  All of it is zero when a launch starts
// (cleared when the batch is made, and again by the last kernel of every launch).
enum : uint32_t { PW_STATUS = 0, PW_COUNT = 1, PW_END = 2, PW_DST_RANK = 3 /* record + 1; 0: none */, PW_DST_OUT = 4, PW_DST_LEN = 5,
                  PW_DST_POS = 6, PW_OFF_OUT = 7, PW_OFF_POS = 8, PW_TICKET = 9, PW_WORDS = 16 };
// A tile says what it found in two pairs of words, every word with TS_READY set once written (a pair counts when both are):
//   A0  what the tile found from its own guess: bit 62 a record of it does not fit the page, bits 32..45 its records, low half
//       the guess (page position; P_NONE: nothing in the tile looks like a record)
//   A1  bit 62 it has records, bits 32..45 the last one's tile position; low half where its chain ends
//   P0  the chain behind the tile, for good: bit 62 it has ended in a record that does not fit, bits 32..61 records so
//       far, low half where it stands (the record that failed, when one has)
//   P1  bit 62 there has been a record, low half the last one's page position
enum : uint32_t { TS_A0 = 0, TS_A1 = 1, TS_P0 = 2, TS_P1 = 3, TS_WORDS = 4 };
constexpr unsigned long long TS_READY = 1ull << 63, TS_BIT = 1ull << 62;

// how a tile's chain ends, and jump targets of the doubling beyond the lanes
constexpr uint32_t T_END = 256, T_ERR = 257, T_BROKEN = 258, T_NONE = 259;

typedef uint32_t u32x4_a1 __attribute__((ext_vector_type(4), aligned(1)));
typedef uint32_t u32x3_a1 __attribute__((ext_vector_type(3), aligned(1)));
typedef unsigned long long u64_a1 __attribute__((aligned(1)));
typedef uint16_t u16_a1 __attribute__((aligned(1)));
typedef uint32_t u32_a1 __attribute__((aligned(1)));

struct PbShared {
    alignas(16) uint32_t stage[(PTILE + PLOOK) / 4 + 16];
    unsigned long long starts[PT];  // per byte of a lane's sub-chunk: a record starts here
    uint32_t guess[PT];             // tile position where a lane starts, or P_NONE
    uint32_t exit_[PT];             // tile position where the lane's walk ended (the failing record's, when it failed)
    // Never alive together (a barrier lies between the last use of one and the first of the next): the marks
    // while the walks are checked against each other, the hops while the links are followed, the length-field
    // bytes while the tile is copied.  4 KB less, and a compute unit holds six workgroups instead of five.
    union {
        uint16_t hop[8][PT];        // lane reached after 1, 2, 4 ... links (or T_*)
        struct {
            uint32_t exits[PT * 2];         // per byte of the tile: some lane's walk ended here
            unsigned long long headers[PT]; // per byte of a lane's sub-chunk: the byte belongs to a length field
        };
    };
    uint16_t before[PT];            // records of the tile that start before the sub-chunk
    uint8_t err[PT];
    uint8_t reach[PT];
    uint32_t wave_sum[PT / 64];
    uint32_t part[4][PT / 64];      // per wavefront: least guess, last lane with one, records, last record
    uint32_t all_ok, first_guess, last_lane, term_lane, term_type, count, last_start;
    uint32_t ticket, chain[4];      // the tile taken; where the chain stands in front of it (Chain's fields)
};

__device__ inline void lds_sync() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Wavefront-wide least / greatest / sum, the same in every lane.  (An LDS atomic of all lanes on one address is
// turned by the compiler into a loop over the lanes, half a dozen scalar instructions each: four of them per tile
// and the compute unit's one scalar unit is what the kernel waits for.)
__device__ inline uint32_t wave_least(uint32_t v) { return wave_least_u32(v); }
__device__ inline uint32_t wave_greatest(uint32_t v) { return wave_greatest_u32(v); }
__device__ inline uint32_t wave_total(uint32_t v) { return lane_u32(wave_inclusive_scan<uint32_t>(v), 63); }

// sh.part[k] of the four wavefronts, combined; k as in the declaration.
__device__ inline uint32_t part_least(const PbShared& sh, uint32_t k) {
    uint32_t v = sh.part[k][0];
    for (uint32_t w = 1; w < PT / 64; w++) v = std::min(v, sh.part[k][w]);
    return v;
}
__device__ inline uint32_t part_greatest(const PbShared& sh, uint32_t k) {
    uint32_t v = sh.part[k][0];
    for (uint32_t w = 1; w < PT / 64; w++) v = std::max(v, sh.part[k][w]);
    return v;
}
__device__ inline uint32_t part_total(const PbShared& sh, uint32_t k) {
    uint32_t v = 0;
    for (uint32_t w = 0; w < PT / 64; w++) v += sh.part[k][w];
    return v;
}

// The page as a tile sees it; positions inside a tile are relative to its first byte.
struct Tile {
    gptr<const uint8_t> src;
    uint64_t base;     // page position of the tile's first byte
    uint32_t rem;      // bytes from there to the end of the page
    uint32_t bytes;    // bytes of the tile inside the page
    uint32_t staged;   // bytes in LDS (the tile and a little behind it)
    bool narrow;       // the page is short enough for the top byte of every length to be zero
};

__device__ inline Tile make_tile(const DevPage& pg, uint64_t tile) {
    Tile t;
    t.src = as_global(pg.src);
    t.base = tile * PTILE;
    t.rem = (uint32_t)(pg.src_len - t.base);
    t.bytes = std::min(PTILE, t.rem);
    t.staged = std::min(PTILE + PLOOK, t.rem);
    t.narrow = pg.src_len < (1u << 24);
    return t;
}

__device__ inline void stage_tile(PbShared& sh, const Tile& t) {
    // 16 bytes per lane and step, the steps' loads in flight together (a loop of load, wait, store keeps
    // 4 KB per workgroup under way and the memory idle); the page may start at any address
    constexpr uint32_t STEPS = (PTILE + PLOOK + 16 * PT - 1) / (16 * PT);
    const uint32_t body = t.staged / 16;
    u32x4_a1 w[STEPS];
#pragma unroll
    for (uint32_t u = 0; u < STEPS; u++) {
        const uint32_t c = threadIdx.x + u * PT;
        if (c < body) w[u] = *(gptr<const u32x4_a1>)(t.src + t.base + (uint64_t)c * 16);
    }
#pragma unroll
    for (uint32_t u = 0; u < STEPS; u++) {
        const uint32_t c = threadIdx.x + u * PT;
        if (c < body) *reinterpret_cast<u32x4*>(&sh.stage[c * 4]) = w[u];
    }
    uint8_t* bytes = reinterpret_cast<uint8_t*>(sh.stage);
    for (uint32_t k = body * 16 + threadIdx.x; k < t.staged; k += PT) bytes[k] = t.src[t.base + k];
}

// The length field at tile position p (p + 4 <= staged).
__device__ inline uint32_t length_at(const PbShared& sh, uint32_t p) {
    const uint32_t d = p >> 2;
    return __builtin_amdgcn_alignbit(sh.stage[d + 1], sh.stage[d], (p & 3) * 8);
}

// plain.go:81-88: what a record at p (inside the page) must satisfy.
__device__ inline bool record_fits(const PbShared& sh, const Tile& t, uint32_t p, uint32_t* len) {
    if (t.rem - p < 4) return false;
    *len = length_at(sh, p);
    return *len <= t.rem - p - 4;
}

// How much a position looks like the start of a record: 0 not at all (a record there would not fit the page, or the
// one after it, ... would not), 2 the record fits and so do those after it as far as they are staged (at least one
// of them), 1 the record fits but the next one lies beyond the staged bytes.
__device__ inline uint32_t looks_like_a_record(const PbShared& sh, const Tile& t, uint32_t p) {
    uint32_t len;
    if (!record_fits(sh, t, p, &len)) return 0;
    p += 4 + len;
    if (p == t.rem) return 2;            // the last record of the page
    if (p + 4 > t.staged) return 1;
    for (uint32_t s = 1; s < PSTEPS; s++) {
        if (!record_fits(sh, t, p, &len)) return 0;
        p += 4 + len;
        if (p == t.rem || p + 4 > t.staged) break;
    }
    return 2;
}

// The records that start in the lane's sub-chunk, from tile position `from`.
__device__ inline void walk_lane(PbShared& sh, const Tile& t, uint32_t lane, uint32_t from) {
    const uint32_t end = std::min((lane + 1) * PSUB, t.bytes);
    uint32_t p = from;
    unsigned long long mask = 0;
    bool failed = false;
    while (p < end) {
        uint32_t len;
        if (!record_fits(sh, t, p, &len)) { failed = true; break; }
        mask |= 1ull << (p - lane * PSUB);
        p += 4 + len;
    }
    sh.starts[lane] = mask;
    sh.err[lane] = failed ? 1 : 0;
    sh.exit_[lane] = p;
}

// Per byte of the lane's 64 (+3): is it zero?  Branch-free, out of the staged dwords.
__device__ inline void zero_bytes(const PbShared& sh, uint32_t lane, unsigned long long* lo, uint32_t* hi) {
    unsigned long long z = 0;
    uint32_t top = 0;
    uint32_t d[17]; // four 16-byte reads and one more dword: a quarter of the bank conflicts of seventeen dword reads
#pragma unroll
    for (uint32_t i = 0; i < 4; i++) {
        const u32x4 v = *reinterpret_cast<const u32x4*>(&sh.stage[lane * (PSUB / 4) + 4 * i]);
        d[4 * i] = v.x; d[4 * i + 1] = v.y; d[4 * i + 2] = v.z; d[4 * i + 3] = v.w;
    }
    d[16] = sh.stage[lane * (PSUB / 4) + 16];
#pragma unroll
    for (uint32_t i = 0; i < 17; i++) {
        const uint32_t w = d[i];
        const uint32_t y = ~(((w & 0x7f7f7f7fu) + 0x7f7f7f7fu) | w | 0x7f7f7f7fu); // 0x80 in every zero byte
        const uint32_t nib = (((y >> 7) * 0x00204081u) >> 21) & 0xfu;
        if (i < 16) z |= (unsigned long long)nib << (4 * i);
        else top = nib;
    }
    *lo = z;
    *hi = top;
}

// Where the lane's walk starts: the first position of its sub-chunk that looks like the start of a record, with
// a record after it that looks like one too; failing that, a position whose record reaches past the staged bytes,
// preferably the last of a row of candidates (the zero bytes of a length field make a row that ends at its start).
__device__ inline uint32_t guess_lane(const PbShared& sh, const Tile& t, uint32_t lane) {
    const uint32_t lo = lane * PSUB;
    if (lo >= t.bytes) return P_NONE;
    unsigned long long cand = ~0ull;
    if (t.narrow) { // the fourth byte of a length field is zero
        unsigned long long z;
        uint32_t top;
        zero_bytes(sh, lane, &z, &top);
        cand = (z >> 3) | ((unsigned long long)top << 61);
    }
    if (t.bytes - lo < PSUB) cand &= (1ull << (t.bytes - lo)) - 1;
    const unsigned long long row_ends = cand & ~(cand >> 1);
    uint32_t weak = P_NONE, weak_end = P_NONE;
    // the ends of rows first: with text for values the first of them is the record
    for (unsigned long long m = row_ends; m; m &= m - 1) {
        const uint32_t bit = (uint32_t)__builtin_ctzll(m);
        const uint32_t how = looks_like_a_record(sh, t, lo + bit);
        if (how == 2) return lo + bit;
        if (how == 1 && weak_end == P_NONE) weak_end = lo + bit;
    }
    for (unsigned long long m = cand & ~row_ends; m; m &= m - 1) {
        const uint32_t bit = (uint32_t)__builtin_ctzll(m);
        const uint32_t how = looks_like_a_record(sh, t, lo + bit);
        if (how == 2) return lo + bit;
        if (how == 1 && weak == P_NONE) weak = lo + bit;
    }
    return weak_end != P_NONE ? weak_end : weak;
}

// Finds the chain of the tile that starts at tile position `entry` (P_NONE: at the first guess).
// On return reach / starts / exit_ / err describe the lanes on the chain (starts is zero elsewhere),
// term_lane and term_type its end, count its records, last_start the last of them.
__device__ inline void resolve_tile(PbShared& sh, const Tile& t, uint32_t entry) {
    const uint32_t lane = threadIdx.x;
    if (lane == 0) { sh.first_guess = P_NONE; sh.all_ok = 1; sh.last_lane = 0; sh.count = 0; sh.last_start = 0; sh.term_type = T_NONE; sh.term_lane = 0; }
    sh.exits[2 * lane] = 0;
    sh.exits[2 * lane + 1] = 0;
    lds_sync();
    // ---- every lane guesses and walks ----
    uint32_t g = P_NONE;
    if (entry != P_NONE && entry / PSUB == lane) g = entry;
    else if (entry == P_NONE || lane > entry / PSUB) g = guess_lane(sh, t, lane);
    sh.guess[lane] = g;
    sh.starts[lane] = 0; sh.err[lane] = 0; sh.exit_[lane] = 0;
    if (g != P_NONE) {
        walk_lane(sh, t, lane, g);
        const uint32_t x = sh.exit_[lane];
        if (!sh.err[lane] && x < t.bytes) atomicOr(&sh.exits[x >> 5], 1u << (x & 31));
    }
    {
        const uint32_t least = wave_least(g), last = wave_greatest(g != P_NONE ? lane : 0u);
        if ((lane & 63) == 0) { sh.part[0][lane >> 6] = least; sh.part[1][lane >> 6] = last; }
    }
    lds_sync();
    if (lane == 0) { sh.first_guess = part_least(sh, 0); sh.last_lane = part_greatest(sh, 1); }
    lds_sync();
    const uint32_t start = entry != P_NONE ? entry : sh.first_guess;
    if (start == P_NONE) { // no record starts in this tile as far as anyone can tell
        sh.reach[lane] = 0;
        lds_sync();
        return;
    }
    // ---- the walks are the chain when every walk ends where another starts, and starts where another ends ----
    if (g != P_NONE) {
        const uint32_t x = sh.exit_[lane];
        bool ok = true;
        if (!sh.err[lane] && x < t.bytes) ok = (sh.starts[x / PSUB] >> (x % PSUB)) & 1;
        else if (lane != sh.last_lane) ok = false;              // only the last walk may leave the tile or fail
        if (g != start) ok = ok && ((sh.exits[g >> 5] >> (g & 31)) & 1);
        if (!ok) sh.all_ok = 0;
    }
    lds_sync();
    if (sh.all_ok) {
        sh.reach[lane] = g != P_NONE ? 1 : 0;
        if (lane == 0) { sh.term_lane = sh.last_lane; sh.term_type = sh.err[sh.last_lane] ? T_ERR : T_END; }
    } else {
        // ---- some guess was wrong: follow the links from the entry, pointer doubling over the lanes ----
        const uint32_t lane0 = start / PSUB;
        for (uint32_t round = 0; round <= PT; round++) {
            uint32_t hop = T_NONE;
            if (sh.guess[lane] != P_NONE) {
                const uint32_t x = sh.exit_[lane];
                if (sh.err[lane]) hop = T_ERR;
                else if (x >= t.bytes) hop = T_END;
                else hop = sh.guess[x / PSUB] == x ? x / PSUB : T_BROKEN;
            }
            sh.hop[0][lane] = (uint16_t)hop;
            if (lane == 0) { sh.term_type = T_END; sh.term_lane = lane0; } // (always overwritten: a chain has one last lane)
            sh.reach[lane] = lane == lane0 ? 1 : 0;
            lds_sync();
            for (uint32_t k = 1; k < 8; k++) {
                const uint32_t a = sh.hop[k - 1][lane];
                sh.hop[k][lane] = a < PT ? sh.hop[k - 1][a] : (uint16_t)a;
                lds_sync();
            }
            // who is on the chain: the lanes 128, then 64, ... then 1 links from one already known
            for (int k = 7; k >= 0; k--) {
                const bool on = sh.reach[lane] != 0;
                const uint32_t a = sh.hop[k][lane];
                lds_sync();
                if (on && a < PT) sh.reach[a] = 1;
                lds_sync();
            }
            if (sh.reach[lane] && sh.hop[0][lane] >= PT) { sh.term_lane = lane; sh.term_type = sh.hop[0][lane]; }
            lds_sync();
            if (sh.term_type != T_BROKEN) break;
            // the chain arrives somewhere nobody guessed: that lane walks from there, and the links are redone
            const uint32_t x = sh.exit_[sh.term_lane];
            lds_sync();
            if (lane == x / PSUB) {
                sh.guess[lane] = x;
                walk_lane(sh, t, lane, x);
            }
            lds_sync();
        }
        if (!sh.reach[lane]) sh.starts[lane] = 0;
    }
    // ---- the records of the chain: how many, and the last ----
    const unsigned long long mine = sh.reach[lane] ? sh.starts[lane] : 0;
    {
        const uint32_t records = wave_total((uint32_t)__builtin_popcountll(mine));
        const uint32_t last = wave_greatest(mine ? lane * PSUB + 63 - (uint32_t)__builtin_clzll(mine) : 0u);
        if ((lane & 63) == 0) { sh.part[2][lane >> 6] = records; sh.part[3][lane >> 6] = last; }
    }
    lds_sync();
    if (lane == 0) { sh.count = part_total(sh, 2); sh.last_start = part_greatest(sh, 3); }
    lds_sync();
}

// The same result as resolve_tile for the pages most byte-array columns make -- values shorter than 256 bytes
// with few zero bytes in them (text) -- at a fraction of the work; false when the tile is not like that, and
// resolve_tile has to be asked.  A position whose next three bytes are zero is taken for a record start (the last
// of a row of such positions: an empty value makes a row of two).  Every one of them sends a mark to where its
// record ends.  When the marked positions are the candidates themselves, all but the first, the candidates are a
// chain: each has one successor, so N candidates send at most N - 1 marks into the tile (the last one's leaves
// it); N - 1 distinct marked positions then mean every candidate but the first is the end of exactly one other,
// and following the ends from the first reaches them all.
__device__ inline bool resolve_tile_text(PbShared& sh, const Tile& t, uint32_t entry) {
    if (!t.narrow) return false;
    const uint32_t lane = threadIdx.x, lo = lane * PSUB;
    if (lane == 0) { sh.first_guess = P_NONE; sh.all_ok = 1; sh.last_lane = 0; sh.count = 0; sh.last_start = 0; sh.term_type = T_NONE; sh.term_lane = 0; }
    sh.exits[2 * lane] = 0;
    sh.exits[2 * lane + 1] = 0;
    lds_sync();
    unsigned long long mask = 0;
    uint32_t exit_at = 0;
    bool ok = true;
    if (lo < t.bytes) {
        unsigned long long z;
        uint32_t top; // zero bytes at 64 .. 67 of the sub-chunk
        zero_bytes(sh, lane, &z, &top);
        const unsigned long long cand = ((z >> 1) | (unsigned long long)(top & 1u) << 63) & ((z >> 2) | (unsigned long long)(top & 3u) << 62) &
                                        ((z >> 3) | (unsigned long long)(top & 7u) << 61);
        const unsigned long long cand_next = (cand >> 1) | ((top & 0xeu) == 0xeu ? 1ull << 63 : 0ull);
        mask = cand & ~cand_next;
        // positions of the tile whose length field lies inside the page, from the entry on
        const uint32_t lim = std::min(t.bytes, t.rem >= 4 ? t.rem - 3 : 0u);
        if (lim <= lo) mask = 0;
        else if (lim - lo < PSUB) mask &= (1ull << (lim - lo)) - 1;
        if (entry != P_NONE) {
            if (lo + PSUB <= entry) mask = 0;
            else if (lo <= entry) {
                mask &= ~((1ull << (entry - lo)) - 1);
                if (!(mask >> (entry - lo) & 1)) ok = false; // the chain enters at a position that is no candidate
            }
        }
        for (unsigned long long m = mask; m; m &= m - 1) {
            const uint32_t q = lo + (uint32_t)__builtin_ctzll(m);
            const uint32_t len = length_at(sh, q);
            if (len > t.rem - q - 4) { ok = false; break; }
            exit_at = q + 4 + len;
            if (exit_at < t.bytes) atomicOr(&sh.exits[exit_at >> 5], 1u << (exit_at & 31));
        }
    }
    const uint32_t g = mask ? lo + (uint32_t)__builtin_ctzll(mask) : P_NONE;
    sh.starts[lane] = mask;
    sh.exit_[lane] = exit_at;
    sh.err[lane] = 0;
    sh.guess[lane] = g;
    {
        const uint32_t least = wave_least(g), last = wave_greatest(mask ? lane : 0u);
        const uint32_t records = wave_total((uint32_t)__builtin_popcountll(mask));
        const uint32_t last_start = wave_greatest(mask ? lo + 63 - (uint32_t)__builtin_clzll(mask) : 0u);
        if ((lane & 63) == 0) {
            sh.part[0][lane >> 6] = least; sh.part[1][lane >> 6] = last;
            sh.part[2][lane >> 6] = records; sh.part[3][lane >> 6] = last_start;
        }
    }
    lds_sync();
    const uint32_t first_guess = part_least(sh, 0);
    const uint32_t start = entry != P_NONE ? entry : first_guess;
    if (start == P_NONE) return false; // nothing here looks like text
    {
        const unsigned long long marked = (unsigned long long)sh.exits[2 * lane] | (unsigned long long)sh.exits[2 * lane + 1] << 32;
        unsigned long long want = mask;
        if (start >= lo && start < lo + PSUB) want &= ~(1ull << (start - lo));
        if (marked != want) ok = false;
    }
    if (!ok) sh.all_ok = 0;
    lds_sync();
    const bool all_ok = sh.all_ok != 0;
    lds_sync(); // (whoever goes on to resolve_tile sets the flag again: not before everyone has read it)
    if (!all_ok) return false;
    sh.reach[lane] = mask ? 1 : 0;
    if (lane == 0) {
        sh.first_guess = first_guess;
        sh.last_lane = part_greatest(sh, 1);
        sh.term_lane = sh.last_lane;
        sh.term_type = T_END;
        sh.count = part_total(sh, 2);
        sh.last_start = part_greatest(sh, 3);
    }
    lds_sync();
    return true;
}

// Where the chain of records stands between two tiles.
struct Chain {
    uint32_t e;       // page position of the next record (of the record that failed, once one has)
    uint32_t rank;    // records before it
    uint32_t prev;    // the last of them, or P_NONE
    uint32_t status;  // 1: a record did not fit the page: the chain has ended
};

__device__ inline unsigned long long ts_load(const unsigned long long* w) { return __hip_atomic_load(w, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ inline void ts_store(unsigned long long* w, unsigned long long v) { __hip_atomic_store(w, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

__device__ inline Chain chain_of(unsigned long long p0, unsigned long long p1) {
    Chain c;
    c.e = (uint32_t)p0;
    c.rank = (uint32_t)(p0 >> 32) & 0x3fffffffu;
    c.status = (p0 & TS_BIT) ? 1u : 0u;
    c.prev = (p1 & TS_BIT) ? (uint32_t)p1 : P_NONE;
    return c;
}

__device__ inline void publish_chain(unsigned long long* tw, const Chain& c) {
    ts_store(tw + TS_P0, TS_READY | (c.status ? TS_BIT : 0ull) | (unsigned long long)c.rank << 32 | c.e);
    ts_store(tw + TS_P1, TS_READY | (c.prev != P_NONE ? TS_BIT | c.prev : 0ull));
}

// Where the chain stands in front of `tile` (one wavefront; the same in every lane).  Lane l looks at tile - 1 - l: the nearest
// tile that knows its P words is the base, the tiles between it and this one are added from their A words when the chain
// enters them where they guessed (or passes over them); a tile that guessed wrong is waited for -- it resolves again
// from its true entry and then says where the chain stands behind it.
__device__ inline Chain look_back(unsigned long long* ts, uint32_t tile, uint32_t n) {
    const uint32_t lane = threadIdx.x & 63;
    Chain S{0, 0, P_NONE, 0};
    if (tile == 0) return S;
    const uint32_t look = std::min(64u, tile);
    const unsigned long long* mine = ts + (uint64_t)(tile - 1 - (lane < look ? lane : 0)) * TS_WORDS;
    unsigned long long a0 = 0, a1 = 0, p0 = 0, p1 = 0, has_p;
    for (;;) {
        if (lane < look) {
            for (;;) {
                p0 = ts_load(mine + TS_P0); p1 = ts_load(mine + TS_P1);
                if (p0 & p1 & TS_READY) break;
                a0 = ts_load(mine + TS_A0); a1 = ts_load(mine + TS_A1);
                if (a0 & a1 & TS_READY) break;
                __builtin_amdgcn_s_sleep(2);
            }
        }
        has_p = __ballot(lane < look && (p0 & p1 & TS_READY) != 0);
        if (has_p) break;
        __builtin_amdgcn_s_sleep(8); // (the tiles further back are on their way: tile 0 knows its P words at once)
    }
    const uint32_t b = (uint32_t)__builtin_ctzll(has_p);
    S = chain_of(lane_u64(p0, b), lane_u64(p1, b));
    for (uint32_t l = b; l-- > 0;) {
        const uint32_t t = tile - 1 - l;
        const uint32_t tend = std::min(t * PTILE + PTILE, n);
        if (S.status != 0 || S.e >= tend) continue; // the chain has ended, or passes over the tile
        const unsigned long long x0 = lane_u64(a0, l), x1 = lane_u64(a1, l);
        if ((uint32_t)x0 == S.e) {
            S.rank += (uint32_t)(x0 >> 32) & 0x3fffu;
            if (x1 & TS_BIT) S.prev = t * PTILE + ((uint32_t)(x1 >> 32) & 0x3fffu);
            S.status = (x0 & TS_BIT) ? 1u : 0u;
            S.e = (uint32_t)x1;
        } else {
            const unsigned long long* tw = ts + (uint64_t)t * TS_WORDS;
            unsigned long long q0, q1;
            for (;;) {
                q0 = ts_load(tw + TS_P0); q1 = ts_load(tw + TS_P1);
                if (q0 & q1 & TS_READY) break;
                __builtin_amdgcn_s_sleep(2);
            }
            S = chain_of(lane_u64(q0, 0), lane_u64(q1, 0));
        }
    }
    return S;
}

// One pass: a workgroup takes the page's next tile, resolves it from its own guess, finds out where the chain really
// enters it, writes the offsets and copies the bytes.
__global__ __launch_bounds__(PT) __attribute__((amdgpu_waves_per_eu(6, 6))) void plain_bytes_tile_kernel(const DevPage* __restrict__ pages, const uint32_t* __restrict__ items) {
    __shared__ PbShared sh;
    const DevPage& pg = pages[items[blockIdx.x]];
    if (pg.lengths == nullptr) return; // left to the one-wavefront walker
    const uint64_t n = pg.src_len;
    const uint32_t tiles = (uint32_t)((n + PTILE - 1) / PTILE);
    uint32_t* page = reinterpret_cast<uint32_t*>(pg.lengths);
    unsigned long long* ts = reinterpret_cast<unsigned long long*>(page + PW_WORDS);
    gptr<uint8_t> dst = as_global(pg.dst);
    gptr<uint32_t> offsets = as_global(pg.offsets);
    const uint64_t dst_cap = pg.dst_cap, offsets_cap = pg.offsets_cap;
    const uint32_t lane = threadIdx.x;
    // records [0, off_room) have room for their offset and the one after it; record off_room is the first that has not
    const uint32_t off_room = offsets_cap >= 0x100000000ull ? 0xffffffffu : (uint32_t)(std::max<uint64_t>(offsets_cap, 1) - 1);
    for (;;) {
        lds_sync(); // the tile before is done with the shared memory
        if (lane == 0) sh.ticket = __hip_atomic_fetch_add(&page[PW_TICKET], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        lds_sync();
        const uint32_t tile = sh.ticket;
        if (tile >= tiles) break;
        const Tile t = make_tile(pg, tile);
        unsigned long long* tw = ts + (uint64_t)tile * TS_WORDS;
        stage_tile(sh, t);
        lds_sync();
        if (!resolve_tile_text(sh, t, tile == 0 ? 0u : P_NONE)) resolve_tile(sh, t, tile == 0 ? 0u : P_NONE);
        // ---- what the tile found, for the tiles behind it; where the chain stands in front of it ----
        const uint32_t type = sh.term_type;
        const uint32_t my_guess = type == T_NONE ? P_NONE : (uint32_t)t.base + (tile == 0 ? 0u : sh.first_guess);
        const uint32_t my_exit = type == T_NONE ? 0u : (uint32_t)t.base + sh.exit_[sh.term_lane];
        const uint32_t my_count = type == T_NONE ? 0u : sh.count, my_last = sh.last_start;
        const bool my_err = type == T_ERR;
        if (lane < 64) {
            if (lane == 0) {
                ts_store(tw + TS_A0, TS_READY | (my_err ? TS_BIT : 0ull) | (unsigned long long)my_count << 32 | my_guess);
                ts_store(tw + TS_A1, TS_READY | (my_count ? TS_BIT | (unsigned long long)my_last << 32 : 0ull) | my_exit);
            }
            const Chain S = look_back(ts, tile, (uint32_t)n);
            if (lane == 0) { sh.chain[0] = S.e; sh.chain[1] = S.rank; sh.chain[2] = S.prev; sh.chain[3] = S.status; }
        }
        lds_sync();
        const Chain S{sh.chain[0], sh.chain[1], sh.chain[2], sh.chain[3]};
        const uint32_t tend = (uint32_t)t.base + t.bytes;
        const uint32_t entry = S.status == 0 && S.e < tend ? S.e : P_NONE;
        const uint32_t rank0 = S.rank, prev0 = S.prev;
        Chain P = S;
        uint32_t vend = S.status ? S.e : 0xffffffffu; // bytes from here on belong to no record
        lds_sync(); // (sh.chain is written again for the next tile)
        if (entry == P_NONE) {
            sh.starts[lane] = 0; sh.reach[lane] = 0;
        } else {
            if (entry != my_guess) {
                if (!resolve_tile_text(sh, t, entry - (uint32_t)t.base)) resolve_tile(sh, t, entry - (uint32_t)t.base);
            }
            const uint32_t ty = sh.term_type; // (the chain enters the tile, so it has an end)
            P.e = (uint32_t)t.base + sh.exit_[sh.term_lane];
            P.rank = S.rank + sh.count;
            if (sh.count) P.prev = (uint32_t)t.base + sh.last_start;
            P.status = ty == T_ERR ? 1u : 0u;
            if (ty == T_ERR) vend = P.e;
        }
        if (lane == 0) {
            publish_chain(tw, P);
            if (tile == tiles - 1) { page[PW_STATUS] = P.status; page[PW_COUNT] = P.rank; page[PW_END] = P.e; }
        }
        // ---- records in front of every sub-chunk: a prefix sum over the lanes ----
        const unsigned long long starts = sh.reach[lane] ? sh.starts[lane] : 0;
        sh.starts[lane] = starts;
        {
            const uint32_t c = (uint32_t)__builtin_popcountll(starts);
            const uint32_t incl = wave_inclusive_scan(c);
            if ((lane & 63) == 63) sh.wave_sum[lane >> 6] = incl;
            lds_sync();
            uint32_t front = incl - c;
            for (uint32_t wv = 0; wv < (lane >> 6); wv++) front += sh.wave_sum[wv];
            sh.before[lane] = (uint16_t)front;
        }
        // ---- offsets, and the first record that does not fit the destination ----
        {
            uint32_t k = rank0 + sh.before[lane];  // the record's number in the page
            unsigned long long m = starts;
            while (m) {
                const uint32_t bit = (uint32_t)__builtin_ctzll(m);
                m &= m - 1;
                // (positions, and so output positions and lengths, fit 32 bits on this path)
                const uint32_t pos = (uint32_t)t.base + lane * PSUB + bit;
                const uint32_t next = (uint32_t)t.base + (m ? lane * PSUB + (uint32_t)__builtin_ctzll(m) : sh.exit_[lane]);
                const uint32_t out = pos - 4 * k, len = next - pos - 4;
                if (k < off_room) offsets[k] = out;
                else if (k == off_room) { page[PW_OFF_OUT] = out; page[PW_OFF_POS] = pos; }
                if (out <= dst_cap && (uint64_t)out + len > dst_cap) {
                    page[PW_DST_RANK] = k + 1; page[PW_DST_OUT] = out; page[PW_DST_LEN] = len; page[PW_DST_POS] = pos;
                }
                k++;
            }
        }
        // ---- the bytes of length fields: four from every start, some of them in the next sub-chunk ----
        lds_sync();
        {
            unsigned long long headers = starts | starts << 1 | starts << 2 | starts << 3;
            if (lane > 0) {
                const unsigned long long ps = sh.starts[lane - 1];
                headers |= ps >> 61 | ps >> 62 | ps >> 63; // starts at bytes 61, 62, 63 of the sub-chunk before
            } else if (prev0 != P_NONE && (uint64_t)prev0 + 4 > t.base) {
                headers |= (1ull << ((uint64_t)prev0 + 4 - t.base)) - 1; // the last record of the tiles before
            }
            sh.headers[lane] = headers;
        }
        lds_sync();
        // ---- the copy: a thread takes 16 bytes of the tile at a time, neighbours take neighbouring bytes ----
        // (page positions and output positions fit 32 bits on this path; so does the capacity that matters)
        const uint32_t cap32 = dst_cap > 0xffffffffull ? 0xffffffffu : (uint32_t)dst_cap;
        for (uint32_t c16 = lane; c16 * 16 < t.bytes; c16 += PT) {
            const uint32_t at = c16 * 16, sub = at / PSUB, sh_bits = at % PSUB;
            const unsigned long long st = sh.starts[sub], hd = sh.headers[sub];
            const u32x4 v = *reinterpret_cast<const u32x4*>(&sh.stage[at / 4]);
            const uint32_t st16 = (uint32_t)(st >> sh_bits) & 0xffffu, hd16 = (uint32_t)(hd >> sh_bits) & 0xffffu;
            // records that start before byte `at`
            const uint32_t k = rank0 + sh.before[sub] + (uint32_t)__builtin_popcountll(st & ((1ull << sh_bits) - 1));
            const uint32_t q = (uint32_t)t.base + at;
            const uint32_t have = std::min<uint32_t>(16, t.bytes - at);
            const bool whole = have == 16 && q + 16 <= vend;
            const uint32_t out = q - 4 * k; // (k records before `at` take 4 k bytes of fields before it: never negative)
            if (whole && hd16 == 0) {
                // sixteen bytes of one value
                if (k > 0 && out <= cap32 && cap32 - out >= 16) {
                    *(gptr<u32x4_a1>)(dst + out) = v;
                    continue;
                }
            } else if (whole) {
                // One length field, or a part of one at either end of the sixteen bytes: what is left is one
                // stretch of 12 to 15 bytes.  A whole field is a dword long, so the dwords behind it move up by one as
                // they are and only the dword it starts in is put together of two; a part of a field at the front
                // shifts everything down by its bytes; one at the back takes nothing but the count down.
                const uint32_t h0 = (uint32_t)__builtin_ctz(hd16), L = (uint32_t)__builtin_ctz(~(hd16 >> h0));
                const bool one_run = (hd16 >> h0) == (1u << L) - 1;
                const bool inside = L == 4 && st16 == (1u << h0);                  // a whole field, begun here
                const bool front = h0 == 0 && L < 4 && st16 == 0;                  // the end of a field begun before
                const bool back = h0 + L == 16 && L < 4 && st16 == (1u << h0);     // the beginning of one that goes on
                if (one_run && (inside || front || back) && (k > 0 || (h0 == 0 && !front))) {
                    u32x4 o = v;
                    uint32_t first = out; // where the stretch goes
                    if (inside) {
                        const uint32_t j = h0 >> 2, low = (1u << (8 * (h0 & 3))) - 1; // bytes of dword j in front of the field
                        o.x = j == 0 ? (v.x & low) | (v.y & ~low) : v.x;
                        o.y = j == 0 ? v.z : (j == 1 ? (v.y & low) | (v.z & ~low) : v.y);
                        o.z = j <= 1 ? v.w : (j == 2 ? (v.z & low) | (v.w & ~low) : v.z);
                    } else if (front) {
                        o.x = __builtin_amdgcn_alignbyte(v.y, v.x, L);
                        o.y = __builtin_amdgcn_alignbyte(v.z, v.y, L);
                        o.z = __builtin_amdgcn_alignbyte(v.w, v.z, L);
                        o.w = v.w >> (8 * L);
                        first = out + L; // q + L - 4 k: the field's record is among the k
                    }
                    const uint32_t n = 16 - L; // 12 when inside, else 13 .. 15
                    if (first <= cap32 && cap32 - first >= n) {
                        gptr<uint8_t> d = dst + first;
                        u32x3_a1 o3 = {o.x, o.y, o.z};
                        *(gptr<u32x3_a1>)d = o3;
                        uint32_t rest = o.w;
                        if (n & 2) { *(gptr<u16_a1>)(d + 12) = (uint16_t)rest; rest >>= 16; }
                        if (n & 1) d[12 + (n & 2)] = (uint8_t)rest;
                        continue;
                    }
                }
            }
            // anything else: the bytes between length fields, a stretch at a time, each in at most four stores
            uint32_t keep = ~hd16 & (have == 16 ? 0xffffu : (1u << have) - 1);
            if (q + have > vend) keep &= q < vend ? (1u << (vend - q)) - 1 : 0u;
            if (k == 0) keep &= st16 ? ~((1u << __builtin_ctz(st16)) - 1) : 0u; // bytes in front of the page's first record
            if (!keep) continue;
            const unsigned long long x_lo = (unsigned long long)v.x | (unsigned long long)v.y << 32;
            const unsigned long long x_hi = (unsigned long long)v.z | (unsigned long long)v.w << 32;
            while (keep) {
                const uint32_t a = (uint32_t)__builtin_ctz(keep);
                uint32_t run = (uint32_t)__builtin_ctz(~(keep >> a));
                keep &= ~(((1u << run) - 1) << a);
                const uint32_t o = q + a - 4 * (k + (uint32_t)__builtin_popcount(st16 & ((1u << a) - 1)));
                if (o >= cap32) continue;
                if (cap32 - o < run) run = cap32 - o;
                // the stretch's bytes, from bit 0 on (run <= 15: sixteen whole bytes went the first way or have a field inside)
                unsigned long long w0 = a == 0 ? x_lo : (a < 8 ? (x_lo >> (8 * a)) | (x_hi << (64 - 8 * a)) : x_hi >> (8 * (a - 8)));
                const unsigned long long w1 = a < 8 ? x_hi >> (8 * a) : 0;
                gptr<uint8_t> d = dst + o;
                if (run & 8) { *(gptr<u64_a1>)d = w0; d += 8; w0 = w1; }
                if (run & 4) { *(gptr<u32_a1>)d = (uint32_t)w0; d += 4; w0 >>= 32; }
                if (run & 2) { *(gptr<u16_a1>)d = (uint16_t)w0; d += 2; w0 >>= 16; }
                if (run & 1) *d = (uint8_t)w0;
            }
        }
    }
}

// Behind it, a workgroup per page: the page's result (plain.go:76-94 walks record by record: the first record that
// fails in any way decides), and the page's state cleared for the next launch.
__global__ __launch_bounds__(PT) void plain_bytes_finish_kernel(const DevPage* __restrict__ pages, DevResult* __restrict__ results,
                                                                const uint32_t* __restrict__ items) {
    const DevPage& pg = pages[items[blockIdx.x]];
    if (pg.lengths == nullptr) return;
    DevResult* res = &results[items[blockIdx.x]];
    const uint64_t n = pg.src_len;
    const uint64_t tiles = (n + PTILE - 1) / PTILE;
    gptr<uint32_t> page = (gptr<uint32_t>)pg.lengths;
    if (threadIdx.x == 0) {
        const uint64_t total = page[PW_COUNT], end = page[PW_END];
        const uint64_t offsets_cap = pg.offsets_cap;
        // candidates, by record number: the record the offsets have no room for, the record the destination has
        // no room for, the record that does not fit the page
        const uint64_t k_off = std::max<uint64_t>(offsets_cap, 1) - 1;
        const uint64_t k_dst = page[PW_DST_RANK] == 0 ? ~0ull : (uint64_t)page[PW_DST_RANK] - 1;
        int32_t status = PQD_OK;
        int64_t arg = 0;
        uint64_t values = total, bytes = end - 4 * total, consumed = n;
        if (k_off < total && k_off <= k_dst) {
            status = PQD_ERR_OFFSETS_TOO_SMALL; arg = (int64_t)(k_off + 2);
            values = k_off; bytes = page[PW_OFF_OUT]; consumed = page[PW_OFF_POS];
        } else if (k_dst < total) {
            status = PQD_ERR_DST_TOO_SMALL; arg = (int64_t)((uint64_t)page[PW_DST_OUT] + page[PW_DST_LEN]);
            values = k_dst; bytes = page[PW_DST_OUT]; consumed = page[PW_DST_POS];
        } else if (page[PW_STATUS] != 0) {
            status = PQD_ERR_UNEXPECTED_EOF; arg = PQD_EOF_PLAIN_BYTE_ARRAY;
            consumed = end;
        } else if (total + 1 > offsets_cap) {
            status = PQD_ERR_OFFSETS_TOO_SMALL; arg = (int64_t)(total + 1);
        } else {
            as_global(pg.offsets)[total] = (uint32_t)bytes;
        }
        res->status = status;
        res->arg = arg;
        res->out_values = values;
        res->out_bytes = bytes;
        res->consumed = consumed;
        res->total = values;
    }
    // (the wavefront's loads above are done before any of its lanes goes on: one instruction stream)
    __builtin_amdgcn_s_waitcnt(0);
    __syncthreads();
    const uint64_t quads = (PW_WORDS + tiles * TS_WORDS * 2) / 4; // (a page's state starts on a 16-byte boundary and is a whole number of them)
    for (uint64_t w = threadIdx.x; w < quads; w += PT) *(gptr<u32x4>)(page + 4 * w) = u32x4{0, 0, 0, 0};
}

} // namespace

uint64_t plain_bytes_scratch_words(uint64_t src_len) {
    // pages this path does not take: positions are kept in 32 bits
    if (src_len == 0 || src_len > 0xfff00000ull) return 0;
    return ((src_len + PTILE - 1) / PTILE) * (TS_WORDS * 2) + PW_WORDS;
}

void launch_plain_byte_array(const DevPage* pages, DevResult* results, const uint32_t* items, uint32_t n_items, uint64_t max_bytes,
                             hipStream_t stream) {
    if (n_items == 0) return;
    launch_plain_byte_array_walk(pages, results, items, n_items, stream); // the pages without scratch
    if (max_bytes == 0) return;                                           // (no page has any)
    // (tiles are handed out by the pages' counters: a workgroup goes on until its page has none left, so the grid need not
    // hold a workgroup for every tile of the longest page)
    const uint32_t gy = (uint32_t)std::min<uint64_t>(std::max<uint64_t>(1, (max_bytes + PTILE - 1) / PTILE), 16384);
    hipLaunchKernelGGL(plain_bytes_tile_kernel, dim3(n_items, gy), dim3(PT), 0, stream, pages, items);
    hipLaunchKernelGGL(plain_bytes_finish_kernel, dim3(n_items), dim3(PT), 0, stream, pages, results, items);
}

} // namespace pqd
