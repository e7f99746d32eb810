// levels.hip -- what parquet-go does with decoded definition and repetition levels, on gfx950:
//
//   countLevelsEqual / countLevelsNotEqual         level.go:11-17
//     (optionalPage.NumNulls page_optional.go:32-34, repeatedPage.NumRows page_repeated.go:33,
//      Column.decodeDataPageV1 column.go:772)
//   optionalPageValues.ReadValues                  page_optional.go:72-117
//     slot i of an optional column takes the next value when its definition level is the
//     maximum, and is null otherwise
//   repeatedPage.Slice row search                  page_repeated.go:73-98
//     row k starts at the k-th repetition level that is 0
//
// All three are a count of the levels equal to one value, per stretch of 2048 levels, an
// exclusive sum over the stretches, and a pass over the stretches that knows where each one
// starts.  The caller supplies the workspace for the per-stretch counts.
#include "pqd_internal.h"

namespace pqd {
namespace {

constexpr uint32_t LT = 256;          // threads per workgroup
constexpr uint32_t STRETCH = 2048;    // levels per wavefront
constexpr uint32_t COUNT_UNROLL = 4;  // stretches a wavefront of the counting kernel takes per turn
constexpr uint32_t SCAN_TURN = 2048;  // entries a workgroup of the scan takes per turn
constexpr uint32_t SCAN_GROUPS = 64;  // at most so many workgroups scan
constexpr uint32_t COUNT_BITS = 12;   // a stretch's count (at most STRETCH) keeps the low bits of its entry after the scan
constexpr uint64_t COUNT_MASK = (1u << COUNT_BITS) - 1;
static_assert(STRETCH <= COUNT_MASK, "a count fits its bits");

// Bytes of a dword equal to the bytes of `pattern` (the value in all four bytes).
__device__ inline uint32_t equal_bytes(uint32_t x, uint32_t pattern) {
    const uint32_t t = x ^ pattern;                                   // zero bytes where equal
    const uint32_t y = ~(((t & 0x7f7f7f7fu) + 0x7f7f7f7fu) | t | 0x7f7f7f7fu); // 0x80 in every zero byte
    return (uint32_t)__popc(y);
}

typedef uint64_t u64x2 __attribute__((ext_vector_type(2)));

// counts[w] = levels equal to `value` in stretch w.
__global__ __launch_bounds__(LT) void levels_count_kernel(const uint8_t* __restrict__ levels_, uint64_t n, uint32_t value,
                                                          uint64_t n_stretches, uint64_t* __restrict__ counts_) {
    gptr<const uint8_t> levels = as_global(levels_);
    gptr<uint64_t> counts = as_global(counts_);
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const uint32_t pattern = value * 0x01010101u;
    // COUNT_UNROLL stretches per wavefront and turn, their loads in flight together: a wavefront with two
    // loads to wait for and a scan to do leaves the memory system idle most of its life
    for (uint64_t w0 = ((uint64_t)blockIdx.x * (LT / 64) + wave) * COUNT_UNROLL; w0 < n_stretches;
         w0 += (uint64_t)gridDim.x * (LT / 64) * COUNT_UNROLL) {
        constexpr uint32_t PER = STRETCH / (64 * 16);
        u32x4_a4 v[COUNT_UNROLL * PER];
        uint32_t whole = 0; // bit j * PER + u: lane's 16 bytes of that load are all in range and dword aligned
#pragma unroll
        for (uint32_t j = 0; j < COUNT_UNROLL; j++) {
#pragma unroll
            for (uint32_t u = 0; u < PER; u++) {
                const uint64_t b = (w0 + j) * STRETCH + (uint64_t)(u * 64 + lane) * 16;
                v[j * PER + u] = u32x4_a4{0, 0, 0, 0};
                if (b + 16 <= n && ((uintptr_t)(levels_ + b) & 3) == 0) {
                    v[j * PER + u] = *(gptr<const u32x4_a4>)(levels + b);
                    whole |= 1u << (j * PER + u);
                }
            }
        }
#pragma unroll
        for (uint32_t j = 0; j < COUNT_UNROLL; j++) {
            if (w0 + j >= n_stretches) break;
            uint32_t c = 0;
#pragma unroll
            for (uint32_t u = 0; u < PER; u++) {
                const uint64_t b = (w0 + j) * STRETCH + (uint64_t)(u * 64 + lane) * 16;
                if (whole >> (j * PER + u) & 1) {
                    const u32x4_a4 x = v[j * PER + u];
                    c += equal_bytes(x.x, pattern) + equal_bytes(x.y, pattern) + equal_bytes(x.z, pattern) + equal_bytes(x.w, pattern);
                } else {
                    for (uint32_t e = 0; e < 16 && b + e < n; e++) c += levels[b + e] == value ? 1u : 0u;
                }
            }
            const uint32_t total = lane_u32(wave_inclusive_scan<uint32_t>(c), 63);
            if (lane == 0) counts[w0 + j] = total;
        }
    }
}

// The counts become (start << COUNT_BITS | count) in place, start being the sum of the counts before; the total
// goes to counts[n_stretches], as it is, and, when asked for, to `total_out`.  Workgroup g owns `chunk` entries
// (a multiple of SCAN_TURN) and first adds up the counts of every entry before its own: whether another
// workgroup has already rewritten an entry or not, its low COUNT_BITS bits are the count.  One workgroup for
// the whole array spends its time waiting on one compute unit's path to memory.
__global__ __launch_bounds__(1024) void levels_scan_kernel(uint64_t* __restrict__ counts_, uint64_t n_stretches, uint64_t chunk,
                                                           uint64_t* __restrict__ total_out) {
    __shared__ uint64_t wsum[16];
    gptr<uint64_t> counts = as_global(counts_);
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const uint64_t lo = (uint64_t)blockIdx.x * chunk;
    const uint64_t hi = lo + chunk < n_stretches ? lo + chunk : n_stretches;
    uint64_t before = 0;
#pragma unroll 8
    for (uint64_t j = (uint64_t)tid * 2; j < lo; j += 2048) { // lo is a multiple of 2048: pairs never straddle it
        const u32x4_a4 p = *(gptr<const u32x4_a4>)(counts + j);
        before += (p.x & COUNT_MASK) + (p.z & COUNT_MASK);
    }
    {
        const uint64_t x = wave_inclusive_scan<uint64_t>(before);
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        before = 0;
        for (uint32_t w = 0; w < 16; w++) before += wsum[w];
        __syncthreads();
    }
    uint64_t carry = before;
    for (uint64_t base = lo; base < hi; base += SCAN_TURN) {
        constexpr uint32_t PER = SCAN_TURN / 1024;
        const uint64_t k = base + (uint64_t)tid * PER;
        uint64_t v[PER], mine = 0;
        for (uint32_t e = 0; e < PER; e++) {
            v[e] = k + e < hi ? counts[k + e] & COUNT_MASK : 0;
            mine += v[e];
        }
        const uint64_t x = wave_inclusive_scan<uint64_t>(mine);
        if (lane == 63) wsum[wave] = x;
        __syncthreads();
        uint64_t o = carry + x - mine, all = 0;
        for (uint32_t w = 0; w < 16; w++) {
            if (w < wave) o += wsum[w];
            all += wsum[w];
        }
        for (uint32_t e = 0; e < PER; e++)
            if (k + e < hi) { counts[k + e] = o << COUNT_BITS | v[e]; o += v[e]; }
        carry += all;
        __syncthreads();
    }
    if (tid == 0 && hi == n_stretches) {
        counts[n_stretches] = carry;
        if (total_out != nullptr) *as_global(total_out) = carry;
    }
}

// SLOTS neighbouring elements in one store where the address allows it.
template <typename T, uint32_t SLOTS>
__device__ inline void store_group(gptr<T> p, const T (&v)[SLOTS], uint32_t in_bits) {
    constexpr uint32_t ALL = (1u << SLOTS) - 1;
    if (in_bits == ALL && ((uintptr_t)p & (SLOTS * sizeof(T) - 1)) == 0) {
        if (sizeof(T) == 8 && SLOTS == 2) { u64x2 q = {(uint64_t)v[0], (uint64_t)v[1]}; *(gptr<u64x2>)p = q; return; }
        if (sizeof(T) == 4 && SLOTS == 4) { u32x4 q = {(uint32_t)v[0], (uint32_t)v[1], (uint32_t)v[2], (uint32_t)v[3]}; *(gptr<u32x4>)p = q; return; }
        if (sizeof(T) == 1 && SLOTS == 4) {
            *(gptr<uint32_t>)p = (uint32_t)v[0] | (uint32_t)v[1] << 8 | (uint32_t)v[2] << 16 | (uint32_t)v[3] << 24;
            return;
        }
    }
#pragma unroll
    for (uint32_t e = 0; e < SLOTS; e++)
        if (in_bits >> e & 1) p[e] = v[e];
}

// page_optional.go:72-117 for fixed-width values.  A wavefront takes a stretch 64 * SLOTS slots
// at a time, SLOTS neighbouring slots per lane (sixteen bytes of 8- or 4-byte values): the place
// of a slot's value among the values is the stretch's start plus the present slots before it
// (SLOTS ballots and mbcnt), the values are read where they lie (neighbouring lanes read
// neighbouring values) and written in one store per lane.  Slots without a value are zero.
// OFFSETS: the values are the n_values + 1 offsets of a byte-array page and every slot, present
// or not, takes the offset at its place, so that null slots come out empty; dst then has
// n_slots + 1 entries.
template <typename T, bool OFFSETS>
__global__ __launch_bounds__(LT) void optional_expand_kernel(const uint8_t* __restrict__ def_, uint64_t n_slots, uint32_t max_def,
                                                             const T* __restrict__ values_, uint64_t n_values, T* __restrict__ dst_,
                                                             uint8_t* __restrict__ validity_, const uint64_t* __restrict__ starts_,
                                                             uint64_t n_stretches, int64_t* __restrict__ info_) {
    gptr<const uint8_t> def = as_global(def_);
    gptr<const T> values = as_global(values_);
    gptr<T> dst = as_global(dst_);
    gptr<uint8_t> validity = as_global(validity_);
    gptr<const uint64_t> starts = as_global(starts_);
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const uint64_t present = starts[n_stretches];
        gptr<int64_t> info = as_global(info_);
        info[0] = (int64_t)present;
        // more present slots than values: the page under the levels ran out (page_optional.go:95-99)
        info[1] = present > n_values ? PQD_ERR_UNEXPECTED_EOF : PQD_OK;
        if (OFFSETS) dst[n_slots] = values[present < n_values ? present : n_values];
    }
    constexpr uint32_t SLOTS = sizeof(T) >= 8 ? 2 : 4, ALL = (1u << SLOTS) - 1;
    constexpr uint32_t STEPS = STRETCH / (64 * SLOTS);
    for (uint64_t w = (uint64_t)blockIdx.x * (LT / 64) + wave; w < n_stretches; w += (uint64_t)gridDim.x * (LT / 64)) {
        uint64_t base = starts[w] >> COUNT_BITS;
        const uint64_t at = w * STRETCH;
        uint32_t lv[STEPS]; // per step: which of the lane's slots hold a value (bits 0 ..), which exist (bits 8 ..)
        // the levels of all steps first, one load per lane and step at whatever address (all of them in flight
        // together); only the last lanes of the column read byte by byte
        typedef uint32_t u32_a1 __attribute__((aligned(1)));
        typedef uint16_t u16_a1 __attribute__((aligned(1)));
#pragma unroll
        for (uint32_t u = 0; u < STEPS; u++) {
            const uint64_t s0 = at + (uint64_t)(u * 64 + lane) * SLOTS;
            const bool whole = s0 + SLOTS <= n_slots;
            const uint64_t from = whole ? s0 : 0; // (a column shorter than SLOTS has no whole group: nothing is loaded then)
            lv[u] = n_slots >= SLOTS ? (SLOTS == 2 ? (uint32_t)*(gptr<const u16_a1>)(def + from) : *(gptr<const u32_a1>)(def + from)) : 0u;
        }
#pragma unroll
        for (uint32_t u = 0; u < STEPS; u++) {
            const uint64_t s0 = at + (uint64_t)(u * 64 + lane) * SLOTS;
            const uint32_t d = lv[u];
            uint32_t bits = 0;
            if (s0 + SLOTS <= n_slots) {
#pragma unroll
                for (uint32_t e = 0; e < SLOTS; e++) bits |= ((d >> (8 * e)) & 0xff) == max_def ? 1u << e : 0u;
                bits |= ALL << 8;
            } else if (s0 < n_slots) {
                for (uint32_t e = 0; e < SLOTS && s0 + e < n_slots; e++) bits |= (def[s0 + e] == max_def ? 1u << e : 0u) | (0x100u << e);
            }
            lv[u] = bits;
        }
#pragma unroll
        for (uint32_t u = 0; u < STEPS; u++) {
            const uint64_t s0 = at + (uint64_t)(u * 64 + lane) * SLOTS;
            const uint32_t has = lv[u] & 0xff, in = lv[u] >> 8;
            uint32_t below = 0, step_total = 0;
#pragma unroll
            for (uint32_t e = 0; e < SLOTS; e++) {
                const unsigned long long b = __ballot(has >> e & 1);
                below += __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0));
                step_total += (uint32_t)__popcll(b);
            }
            uint64_t r = base + below; // the place of the lane's first value among the values
            T v[SLOTS];
#pragma unroll
            for (uint32_t e = 0; e < SLOTS; e++) {
                v[e] = T();
                if (OFFSETS) {
                    if (in >> e & 1) v[e] = values[r < n_values ? r : n_values];
                } else {
                    if ((has >> e & 1) && r < n_values) v[e] = values[r];
                }
                r += has >> e & 1;
            }
            store_group<T, SLOTS>(dst + s0, v, in);
            if (validity_ != nullptr) {
                uint8_t ok[SLOTS];
#pragma unroll
                for (uint32_t e = 0; e < SLOTS; e++) ok[e] = (uint8_t)(has >> e & 1);
                if (SLOTS == 2 && in == ALL && ((uintptr_t)(validity_ + s0) & 1) == 0) *(gptr<uint16_t>)(validity + s0) = (uint16_t)(ok[0] | ok[1] << 8);
                else store_group<uint8_t, SLOTS>(validity + s0, ok, in);
            }
            base += step_total;
        }
    }
}

// The same for values of any width: a lane per slot, bytes copied one at a time.
__global__ __launch_bounds__(LT) void optional_expand_bytes_kernel(const uint8_t* __restrict__ def_, uint64_t n_slots, uint32_t max_def,
                                                                   const uint8_t* __restrict__ values_, uint64_t n_values, uint32_t width,
                                                                   uint8_t* __restrict__ dst_, uint8_t* __restrict__ validity_,
                                                                   const uint64_t* __restrict__ starts_, uint64_t n_stretches,
                                                                   int64_t* __restrict__ info_) {
    gptr<const uint8_t> def = as_global(def_);
    gptr<const uint8_t> values = as_global(values_);
    gptr<uint8_t> dst = as_global(dst_);
    gptr<uint8_t> validity = as_global(validity_);
    gptr<const uint64_t> starts = as_global(starts_);
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const uint64_t present = starts[n_stretches];
        gptr<int64_t> info = as_global(info_);
        info[0] = (int64_t)present;
        info[1] = present > n_values ? PQD_ERR_UNEXPECTED_EOF : PQD_OK;
    }
    for (uint64_t w = (uint64_t)blockIdx.x * (LT / 64) + wave; w < n_stretches; w += (uint64_t)gridDim.x * (LT / 64)) {
        uint64_t base = starts[w] >> COUNT_BITS;
        for (uint32_t u = 0; u < STRETCH / 64; u++) {
            const uint64_t s0 = w * STRETCH + u * 64 + lane;
            const bool in = s0 < n_slots;
            const bool p = in && def[s0] == max_def;
            const unsigned long long b = __ballot(p);
            const uint64_t r = base + __builtin_amdgcn_mbcnt_hi((uint32_t)(b >> 32), __builtin_amdgcn_mbcnt_lo((uint32_t)b, 0));
            if (in) {
                const bool have = p && r < n_values;
                for (uint32_t e = 0; e < width; e++) dst[s0 * width + e] = have ? values[r * width + e] : (uint8_t)0;
                if (validity_ != nullptr) validity[s0] = p ? 1 : 0;
            }
            base += (uint64_t)__popcll(b);
        }
    }
}

// page_repeated.go:73-98: row_starts[k] = index of the k-th repetition level equal to 0.
__global__ __launch_bounds__(LT) void row_starts_kernel(const uint8_t* __restrict__ rep_, uint64_t n, uint64_t* __restrict__ rows_,
                                                        uint64_t row_cap, const uint64_t* __restrict__ starts_, uint64_t n_stretches,
                                                        int64_t* __restrict__ info_) {
    gptr<const uint8_t> rep = as_global(rep_);
    gptr<uint64_t> rows = as_global(rows_);
    gptr<const uint64_t> starts = as_global(starts_);
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        const uint64_t n_rows = starts[n_stretches];
        gptr<int64_t> info = as_global(info_);
        info[0] = (int64_t)n_rows;
        info[1] = n_rows + 1 > row_cap ? PQD_ERR_DST_TOO_SMALL : PQD_OK;
        if (n_rows < row_cap) rows[n_rows] = n; // the end of the last row
    }
    __shared__ uint16_t stage[LT / 64][64 * 16];
    for (uint64_t w = (uint64_t)blockIdx.x * (LT / 64) + wave; w < n_stretches; w += (uint64_t)gridDim.x * (LT / 64)) {
        uint64_t base = starts[w] >> COUNT_BITS;
        // sixteen levels per lane and step: one 16-byte load, the zero bytes as a 16-bit mask (four bits per
        // dword by a multiplication), a wave scan of their counts, then every lane writes the places of its own
        uint32_t m[STRETCH / (64 * 16)];
#pragma unroll
        for (uint32_t u = 0; u < STRETCH / (64 * 16); u++) {
            const uint64_t s0 = w * STRETCH + (uint64_t)(u * 64 + lane) * 16;
            uint32_t mask = 0;
            if (s0 + 16 <= n) {
                typedef uint32_t u32x4_a1 __attribute__((ext_vector_type(4), aligned(1)));
                const u32x4_a1 v = *(gptr<const u32x4_a1>)(rep + s0); // at whatever address the levels start
                const uint32_t d[4] = {v.x, v.y, v.z, v.w};
#pragma unroll
                for (uint32_t k = 0; k < 4; k++) {
                    const uint32_t t = d[k];
                    const uint32_t y = ~(((t & 0x7f7f7f7fu) + 0x7f7f7f7fu) | t | 0x7f7f7f7fu); // 0x80 in every zero byte
                    mask |= ((((y >> 7) * 0x00204081u) >> 21) & 0xfu) << (4 * k);
                }
            } else {
                for (uint32_t e = 0; e < 16 && s0 + e < n; e++) mask |= rep[s0 + e] == 0 ? 1u << e : 0u;
            }
            m[u] = mask;
        }
#pragma unroll
        for (uint32_t u = 0; u < STRETCH / (64 * 16); u++) {
            const uint64_t s0 = w * STRETCH + (uint64_t)(u * 64 + lane) * 16;
            const uint32_t cnt = (uint32_t)__popc(m[u]);
            const uint32_t incl = wave_inclusive_scan<uint32_t>(cnt);
            const uint32_t total = lane_u32(incl, 63);
            // the places within the step go through the wavefront's own corner of LDS in their order, and
            // come out 64 neighbours at a time: whole cache lines instead of a few entries per lane
            uint32_t r = incl - cnt;
            for (uint32_t left = m[u]; left; left &= left - 1, r++)
                stage[wave][r] = (uint16_t)(lane * 16 + (uint32_t)__builtin_ctz(left));
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            const uint64_t step0 = s0 - (uint64_t)lane * 16;
            for (uint32_t t = lane; t < total; t += 64)
                if (base + t < row_cap) rows[base + t] = step0 + stage[wave][t];
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            base += total;
        }
    }
}

uint64_t stretches_of(uint64_t n) { return (n + STRETCH - 1) / STRETCH; }

// The per-stretch counts of pqd_optional_expand_noalloc: a workspace that is part of the code
// object instead of one the caller brings.  Every GPU the library runs on holds its own copy at
// its own address for the life of the process; host code only ever uses the address
// hipGetSymbolAddress reported for the context's device (kept in pqd_ctx), never the array's name.
constexpr uint64_t SCRATCH_BYTES = PQD_NOALLOC_WORKSPACE_BYTES;
static __device__ __align__(256) unsigned char scratch[SCRATCH_BYTES];
static_assert((SCRATCH_BYTES / sizeof(uint64_t) - 1) * STRETCH == PQD_NOALLOC_MAX_SLOTS, "the documented limit is what the array holds");

uint32_t groups_for(uint64_t n_stretches) {
    const uint64_t g = (n_stretches + LT / 64 - 1) / (LT / 64);
    return (uint32_t)(g < 1 ? 1 : (g > 65536 ? 65536 : g));
}

uint32_t count_groups_for(uint64_t n_stretches) { return groups_for((n_stretches + COUNT_UNROLL - 1) / COUNT_UNROLL); }

bool count_and_scan(const uint8_t* levels, uint64_t n, uint8_t value, void* workspace, uint64_t workspace_bytes, uint64_t* total,
                    hipStream_t s) {
    const uint64_t ns = stretches_of(n);
    if (workspace == nullptr || workspace_bytes < (ns + 1) * sizeof(uint64_t) || ((uintptr_t)workspace & 7) != 0) return false;
    uint64_t* counts = static_cast<uint64_t*>(workspace);
    if (ns > 0) hipLaunchKernelGGL(levels_count_kernel, dim3(count_groups_for(ns)), dim3(LT), 0, s, levels, n, (uint32_t)value, ns, counts);
    // every workgroup of the scan owns a whole number of turns; an empty array still gets its total written
    uint64_t groups = (ns + SCAN_TURN - 1) / SCAN_TURN;
    if (groups < 1) groups = 1;
    if (groups > SCAN_GROUPS) groups = SCAN_GROUPS;
    const uint64_t chunk = ((ns + groups - 1) / groups + SCAN_TURN - 1) / SCAN_TURN * SCAN_TURN;
    groups = chunk ? (ns + chunk - 1) / chunk : 1;
    if (groups < 1) groups = 1;
    hipLaunchKernelGGL(levels_scan_kernel, dim3((uint32_t)groups), dim3(1024), 0, s, counts, ns, chunk ? chunk : SCAN_TURN, total);
    return true;
}

} // namespace

// Called by pqd_ctx_create with the context's device current.
int levels_scratch_address(void** out) {
    void* p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)) != hipSuccess || p == nullptr) return PQD_ERR_DEVICE;
    *out = p;
    return PQD_OK;
}

} // namespace pqd

using namespace pqd;

namespace {

int optional_expand(const uint8_t* def_levels_dev, uint64_t n_slots, uint8_t max_definition_level, const void* values_dev,
                    uint64_t n_values, uint32_t width, int32_t offsets_mode, void* dst_dev, uint8_t* validity_dev,
                    void* workspace_dev, uint64_t workspace_bytes, int64_t* info_dev, hipStream_t s) {
    if (!count_and_scan(def_levels_dev, n_slots, max_definition_level, workspace_dev, workspace_bytes, nullptr, s))
        return PQD_ERR_INVALID_ARGUMENT;
    const uint64_t ns = stretches_of(n_slots);
    const uint64_t* starts = static_cast<const uint64_t*>(workspace_dev);
    const uint32_t groups = groups_for(ns);
    const uint32_t md = max_definition_level;
    const bool aligned = ((uintptr_t)values_dev % width) == 0 && ((uintptr_t)dst_dev % width) == 0;
    if (offsets_mode)
        hipLaunchKernelGGL((optional_expand_kernel<uint32_t, true>), dim3(groups), dim3(LT), 0, s, def_levels_dev, n_slots, md,
                           static_cast<const uint32_t*>(values_dev), n_values, static_cast<uint32_t*>(dst_dev), validity_dev, starts, ns,
                           info_dev);
    else if (width == 8 && aligned)
        hipLaunchKernelGGL((optional_expand_kernel<uint64_t, false>), dim3(groups), dim3(LT), 0, s, def_levels_dev, n_slots, md,
                           static_cast<const uint64_t*>(values_dev), n_values, static_cast<uint64_t*>(dst_dev), validity_dev, starts, ns,
                           info_dev);
    else if (width == 4 && aligned)
        hipLaunchKernelGGL((optional_expand_kernel<uint32_t, false>), dim3(groups), dim3(LT), 0, s, def_levels_dev, n_slots, md,
                           static_cast<const uint32_t*>(values_dev), n_values, static_cast<uint32_t*>(dst_dev), validity_dev, starts, ns,
                           info_dev);
    else if (width == 1)
        hipLaunchKernelGGL((optional_expand_kernel<uint8_t, false>), dim3(groups), dim3(LT), 0, s, def_levels_dev, n_slots, md,
                           static_cast<const uint8_t*>(values_dev), n_values, static_cast<uint8_t*>(dst_dev), validity_dev, starts, ns,
                           info_dev);
    else
        hipLaunchKernelGGL(optional_expand_bytes_kernel, dim3(groups), dim3(LT), 0, s, def_levels_dev, n_slots, md,
                           static_cast<const uint8_t*>(values_dev), n_values, width, static_cast<uint8_t*>(dst_dev), validity_dev, starts,
                           ns, info_dev);
    return hipGetLastError() == hipSuccess ? PQD_OK : PQD_ERR_DEVICE;
}

bool optional_expand_args_ok(pqd_ctx* ctx, const uint8_t* def_levels_dev, uint64_t n_slots, const void* values_dev, uint64_t n_values,
                             uint32_t width, int32_t offsets_mode, void* dst_dev, int64_t* info_dev) {
    if (ctx == nullptr || info_dev == nullptr || width == 0 || (def_levels_dev == nullptr && n_slots > 0) ||
        (dst_dev == nullptr && n_slots > 0) || (values_dev == nullptr && n_values > 0))
        return false;
    if (offsets_mode && (width != 4 || values_dev == nullptr || dst_dev == nullptr)) return false;
    return true;
}

} // namespace

extern "C" {

uint64_t pqd_levels_workspace_size(uint64_t n_levels) { return (stretches_of(n_levels) + 1) * sizeof(uint64_t); }

int pqd_levels_count_equal_device(pqd_ctx* ctx, const uint8_t* levels_dev, uint64_t n, uint8_t value, void* workspace_dev,
                                  uint64_t workspace_bytes, uint64_t* count_dev, void* hip_stream) {
    if (ctx == nullptr || count_dev == nullptr || (levels_dev == nullptr && n > 0)) return PQD_ERR_INVALID_ARGUMENT;
    pqd::DeviceGuard guard(ctx->device);
    if (!guard.ok) return PQD_ERR_DEVICE;
    if (!count_and_scan(levels_dev, n, value, workspace_dev, workspace_bytes, count_dev, static_cast<hipStream_t>(hip_stream)))
        return PQD_ERR_INVALID_ARGUMENT;
    return hipGetLastError() == hipSuccess ? PQD_OK : PQD_ERR_DEVICE;
}

int pqd_optional_expand_device(pqd_ctx* ctx, const uint8_t* def_levels_dev, uint64_t n_slots, uint8_t max_definition_level,
                               const void* values_dev, uint64_t n_values, uint32_t width, int32_t offsets_mode, void* dst_dev,
                               uint8_t* validity_dev, void* workspace_dev, uint64_t workspace_bytes, int64_t* info_dev,
                               void* hip_stream) {
    if (!optional_expand_args_ok(ctx, def_levels_dev, n_slots, values_dev, n_values, width, offsets_mode, dst_dev, info_dev))
        return PQD_ERR_INVALID_ARGUMENT;
    pqd::DeviceGuard guard(ctx->device);
    if (!guard.ok) return PQD_ERR_DEVICE;
    return optional_expand(def_levels_dev, n_slots, max_definition_level, values_dev, n_values, width, offsets_mode, dst_dev,
                           validity_dev, workspace_dev, workspace_bytes, info_dev, static_cast<hipStream_t>(hip_stream));
}

int pqd_optional_expand_noalloc(pqd_ctx* ctx, const uint8_t* def_levels_dev, uint64_t n_slots, uint8_t max_definition_level,
                                const void* values_dev, uint64_t n_values, uint32_t width, int32_t offsets_mode, void* dst_dev,
                                uint8_t* validity_dev, int64_t* info_dev, void* hip_stream) {
    if (!optional_expand_args_ok(ctx, def_levels_dev, n_slots, values_dev, n_values, width, offsets_mode, dst_dev, info_dev))
        return PQD_ERR_INVALID_ARGUMENT;
    // the workspace is the static array: more slots than its counts can cover are refused, not truncated
    if (n_slots > PQD_NOALLOC_MAX_SLOTS) return PQD_ERR_DST_TOO_SMALL;
    if (ctx->levels_scratch == nullptr) return PQD_ERR_DEVICE; // hipGetSymbolAddress failed when the context was made
    pqd::DeviceGuard guard(ctx->device);
    if (!guard.ok) return PQD_ERR_DEVICE;
    // The same three kernels as pqd_optional_expand_device.  The counts are written for every
    // stretch before they are read, so nothing of an earlier call needs clearing.
    return optional_expand(def_levels_dev, n_slots, max_definition_level, values_dev, n_values, width, offsets_mode, dst_dev,
                           validity_dev, ctx->levels_scratch, SCRATCH_BYTES, info_dev, static_cast<hipStream_t>(hip_stream));
}

int pqd_repeated_row_starts_device(pqd_ctx* ctx, const uint8_t* rep_levels_dev, uint64_t n, uint64_t* row_starts_dev, uint64_t row_cap,
                                   void* workspace_dev, uint64_t workspace_bytes, int64_t* info_dev, void* hip_stream) {
    if (ctx == nullptr || info_dev == nullptr || (rep_levels_dev == nullptr && n > 0) || (row_starts_dev == nullptr && row_cap > 0))
        return PQD_ERR_INVALID_ARGUMENT;
    pqd::DeviceGuard guard(ctx->device);
    if (!guard.ok) return PQD_ERR_DEVICE;
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    if (!count_and_scan(rep_levels_dev, n, 0, workspace_dev, workspace_bytes, nullptr, s)) return PQD_ERR_INVALID_ARGUMENT;
    const uint64_t ns = stretches_of(n);
    hipLaunchKernelGGL(row_starts_kernel, dim3(groups_for(ns)), dim3(LT), 0, s, rep_levels_dev, n, row_starts_dev, row_cap,
                       static_cast<const uint64_t*>(workspace_dev), ns, info_dev);
    return hipGetLastError() == hipSuccess ? PQD_OK : PQD_ERR_DEVICE;
}

} // extern "C"
