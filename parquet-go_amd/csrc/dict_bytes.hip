// dict_bytes.hip -- Dictionary.Lookup for dictionaries whose values are not 4, 8 or 12 bytes wide:
// BYTE_ARRAY (byteArrayDictionary, dictionary_byte_array.go; lookupString,
// dictionary_purego.go:40-45), FIXED_LEN_BYTE_ARRAY of any size (fixedLenByteArrayDictionary,
// dictionary_purego.go:47-52) and BOOLEAN (booleanDictionary.lookup, dictionary_boolean.go:93-98).
//
// The reference hands out string headers that point into the dictionary; a column that stays on
// the device is materialised instead, in the layout DecodeByteArray produces: the values back to
// back and n + 1 offsets.  Two kernels per batch (or per call of the stand-alone entry point):
//   offsets  one workgroup per page: bounds check of every index (the reference panics
//            "index out of bounds", errors.go:75-83), length of the value it names, exclusive
//            sum -> offsets; FIXED_LEN_BYTE_ARRAY and BOOLEAN only check.
//   copy     a wavefront per 64 values: the bytes of value after value, a lane per byte.
// Fused with RLE_DICTIONARY pages, the indexes come from an RLE hybrid item of the same batch
// that decodes them (with newIndexedPage's length fix-up) into scratch.
#include <algorithm>

#include "pqd_internal.h"

namespace pqd {
namespace {

constexpr int DT = 256;

// What the two kernels work on: a page of a batch, or the arguments of a stand-alone call.
struct Lookup {
    gptr<const uint32_t> idx;   // n indexes
    uint64_t n;
    gptr<const uint8_t> dict;   // value bytes (BOOLEAN: packed bits, LSB first)
    gptr<const uint32_t> doff;  // BYTE_ARRAY: dict_len + 1 offsets into dict
    uint64_t dict_len;
    uint32_t width;             // FIXED_LEN_BYTE_ARRAY: bytes per value; BOOLEAN: 1
    int32_t kind;
    gptr<uint8_t> dst;
    uint64_t dst_cap;
    gptr<uint32_t> offsets;     // BYTE_ARRAY: n + 1 offsets out
    uint64_t offsets_cap;
};

struct Outcome {
    int32_t status;
    int64_t arg;
    uint64_t out_values, out_bytes;
};

// A barrier for hand-offs through LDS: no wait for outstanding loads and stores.
__device__ inline void lds_only_barrier() { asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); }

// Runs in one workgroup of DT threads; the outcome is valid in thread 0.
__device__ inline Outcome lookup_offsets(const Lookup& L) {
    __shared__ unsigned long long first_bad;
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    Outcome o = {PQD_OK, 0, 0, 0};
    const bool ba = L.kind == PQD_KIND_BYTE_ARRAY;
    if (ba && L.n + 1 > L.offsets_cap) { o.status = PQD_ERR_OFFSETS_TOO_SMALL; o.arg = (int64_t)(L.n + 1); return o; }
    if (!ba && (L.width == 0 || L.n > L.dst_cap / L.width)) {
        o.status = L.width == 0 ? PQD_ERR_INVALID_ARGUMENT : PQD_ERR_DST_TOO_SMALL;
        o.arg = (int64_t)(L.n * L.width);
        return o;
    }
    if (tid == 0) first_bad = ~0ull;
    __syncthreads();
    // Eight values per thread and turn, the next turn's indexes and lengths asked for before this turn's sums:
    // one workgroup walks the whole page, and index, then length, then scan, two barriers and the stores, each
    // waiting for the one before, is a long way round for 1024 values.
    constexpr uint32_t PER = 8;
    auto lengths = [&](uint64_t base, uint32_t (&v)[PER]) {
        const uint64_t k = base + (uint64_t)tid * PER;
        // (one workgroup has one compute unit's path to memory: few, wide accesses -- 16 bytes of indexes a load,
        // both ends of a dictionary entry in one 8-byte load, 16 bytes of offsets a store)
        uint32_t j[PER];
        if (k + PER <= L.n) {
#pragma unroll
            for (uint32_t e = 0; e < PER; e += 4) {
                const u32x4_a4 q = *(gptr<const u32x4_a4>)(L.idx + k + e);
                j[e] = q.x; j[e + 1] = q.y; j[e + 2] = q.z; j[e + 3] = q.w;
            }
        } else {
#pragma unroll
            for (uint32_t e = 0; e < PER; e++) j[e] = k + e < L.n ? L.idx[k + e] : 0u;
        }
        // (no branch around the loads: they are asked for together; an index out of range reads entry 0 and is reported)
        uint32_t lo[PER], hi[PER];
        const bool look = ba && L.dict_len > 0;
#pragma unroll
        for (uint32_t e = 0; e < PER; e++) {
            const uint32_t jj = (uint64_t)j[e] < L.dict_len ? j[e] : 0u;
            u32x2_a4 ends = {0, 0};
            if (look) ends = *(gptr<const u32x2_a4>)(L.doff + jj);
            lo[e] = ends.x;
            hi[e] = ends.y;
        }
        bool bad = false;
#pragma unroll
        for (uint32_t e = 0; e < PER; e++) {
            const bool in = k + e < L.n, ok = (uint64_t)j[e] < L.dict_len;
            v[e] = in && ok ? hi[e] - lo[e] : 0u;
            bad = bad || (in && !ok);
        }
        if (bad) {
            for (uint32_t e = 0; e < PER; e++)
                if (k + e < L.n && (uint64_t)j[e] >= L.dict_len) { atomicMin(&first_bad, (unsigned long long)(k + e)); break; }
        }
    };
    // The running sum is kept in 64 bits: a few indexes times a long dictionary entry pass 4 GiB, which
    // offsets of 32 bits cannot name (the entries written from there on are wrapped and never used: the
    // page ends in an error that says how many bytes it came to, and nothing is copied).
    uint64_t carry = 0;
    uint32_t v[PER], nxt[PER];
    __shared__ uint64_t wsum64[DT / 64];
    lengths(0, v);
    for (uint64_t base = 0; base < L.n; base += (uint64_t)DT * PER) {
        const uint64_t k = base + (uint64_t)tid * PER;
        const bool more = base + (uint64_t)DT * PER < L.n;
        if (more) lengths(base + (uint64_t)DT * PER, nxt);
        if (ba) { // uniform
            uint64_t mine = 0;
#pragma unroll
            for (uint32_t e = 0; e < PER; e++) mine += v[e];
            const uint64_t x = wave_inclusive_scan<uint64_t>(mine);
            if (lane == 63) wsum64[wave] = x;
            lds_only_barrier(); // (not __syncthreads: that would wait for this turn's stores to be acknowledged)
            uint64_t before = carry, all = 0;
            for (uint32_t w = 0; w < DT / 64; w++) {
                if (w < wave) before += wsum64[w];
                all += wsum64[w];
            }
            uint32_t at = (uint32_t)(before + x - mine);
            if (k + PER <= L.n) {
#pragma unroll
                for (uint32_t e = 0; e < PER; e += 4) {
                    u32x4_a4 q;
                    q.x = at; at += v[e];
                    q.y = at; at += v[e + 1];
                    q.z = at; at += v[e + 2];
                    q.w = at; at += v[e + 3];
                    *(gptr<u32x4_a4>)(L.offsets + k + e) = q;
                }
            } else {
#pragma unroll
                for (uint32_t e = 0; e < PER; e++)
                    if (k + e < L.n) { L.offsets[k + e] = at; at += v[e]; }
            }
            carry += all;
            lds_only_barrier();
        }
        if (more) {
#pragma unroll
            for (uint32_t e = 0; e < PER; e++) v[e] = nxt[e];
        }
    }
    __syncthreads();
    if (tid == 0) {
        if (first_bad != ~0ull) { o.status = PQD_ERR_INDEX_OUT_OF_BOUNDS; o.arg = (int64_t)first_bad; }
        else if (ba) {
            L.offsets[L.n] = (uint32_t)carry;
            // more bytes than 32-bit offsets reach: no capacity makes the page decodable, the caller is told its size
            if (carry > 0xffffffffull) { o.status = PQD_ERR_LENGTH_OUT_OF_BOUNDS; o.arg = (int64_t)carry; }
            else if (carry > L.dst_cap) { o.status = PQD_ERR_DST_TOO_SMALL; o.arg = (int64_t)carry; o.out_values = L.n; }
            else { o.out_values = L.n; o.out_bytes = carry; }
        } else {
            o.out_values = L.n;
            o.out_bytes = L.n * L.width;
        }
    }
    return o;
}

constexpr uint32_t IMG_CAP = 6144; // output bytes of 64 values that go through the wavefront's LDS image

typedef uint32_t u32_a1 __attribute__((aligned(1)));
typedef uint32_t u32x4_a1 __attribute__((ext_vector_type(4), aligned(1)));

// The values [k0, k0 + 64) of the page by one wavefront; `img` is the wavefront's own IMG_CAP + 32 bytes of LDS.
// The 64 values are neighbours in the output, so they are put together in LDS first -- every lane moves its own
// value out of the dictionary, sixteen bytes a load, into dword-aligned places of an image that agrees with the
// destination modulo 16 -- and leave in whole 16-byte stores.  (A lane per byte of one value after the other,
// the first form, cost two memory instructions of 46 active lanes per value: 410 us for 375 MB.)
__device__ inline void lookup_copy(const Lookup& L, uint64_t k0, uint8_t* img, uint64_t dict_bytes) {
    const uint32_t lane = threadIdx.x & 63;
    const uint64_t k = k0 + lane;
    const bool in = k < L.n;
    const uint32_t j = in ? L.idx[k] : 0u;
    if (L.kind == PQD_KIND_BOOLEAN) {
        if (in) L.dst[k] = (uint8_t)((L.dict[j >> 3] >> (j & 7)) & 1);
        return;
    }
    const bool ba = L.kind == PQD_KIND_BYTE_ARRAY;
    uint64_t from = 0, to = 0;
    uint32_t len = 0;
    if (in) {
        if (ba) { const u32x2_a4 ends = *(gptr<const u32x2_a4>)(L.doff + j); from = ends.x; len = ends.y - ends.x; to = L.offsets[k]; } // (both ends in one gather)
        else { from = (uint64_t)j * L.width; len = L.width; to = k * L.width; }
    }
    const uint32_t cnt = L.n - k0 < 64 ? (uint32_t)(L.n - k0) : 64u;
    const uint64_t to0 = lane_u64(to, 0);
    const uint64_t total = lane_u64(to, cnt - 1) + lane_u32(len, cnt - 1) - to0; // the values follow one another
    if (total > IMG_CAP) {
        // long values: a lane per byte of one value after the other
        for (uint32_t v = 0; v < cnt; v++) {
            const uint32_t n = lane_u32(len, v);
            gptr<const uint8_t> s = L.dict + lane_u64(from, v);
            gptr<uint8_t> d = L.dst + lane_u64(to, v);
            for (uint32_t b = lane; b < n; b += 64) d[b] = s[b];
        }
        return;
    }
    const uint32_t align = (uint32_t)((uintptr_t)(L.dst + to0) & 15);
    // ---- the lane's value into its place in the image ----
    // Sixteen bytes of the value a load, from wherever it lies in the dictionary, and nothing but those loads:
    // a load of 64 lanes at 64 places costs the compute unit's address unit the same whether it fetches a
    // byte or sixteen, and it is what this kernel waits for.  The bytes in front of the image's next dword, the
    // dwords behind them (shifted into place across the loaded registers) and the last bytes come out of registers.
    {
        gptr<const uint8_t> s = L.dict + from;
        auto load16 = [&](uint32_t c) {
            u32x4 w = {0, 0, 0, 0};
            if (from + 16ull * c + 16 <= dict_bytes) {
                const u32x4_a1 x = *(gptr<const u32x4_a1>)(s + 16 * c);
                w = u32x4{x.x, x.y, x.z, x.w};
            } else { // the last bytes of the dictionary: one at a time
                uint32_t d[4] = {0, 0, 0, 0};
                for (uint32_t i = 0; i < 16 && from + 16ull * c + i < dict_bytes; i++) d[i >> 2] |= (uint32_t)s[16 * c + i] << (8 * (i & 3));
                w = u32x4{d[0], d[1], d[2], d[3]};
            }
            return w;
        };
        const uint32_t q = align + (uint32_t)(to - to0);
        const uint32_t head = std::min(len, (4 - (q & 3)) & 3); // bytes up to the next dword of the image
        const uint32_t chunks = (len + 15) / 16;
        u32x4 cur = {0, 0, 0, 0};
        if (chunks > 0) cur = load16(0);
        for (uint32_t b2 = 0; b2 < head; b2++) img[q + b2] = (uint8_t)(cur.x >> (8 * b2));
        uint32_t* out = reinterpret_cast<uint32_t*>(img + q + head);
        uint32_t rem = len - head; // bytes still to place: the value's bytes from `head` on
        for (uint32_t c = 0; rem > 0; c++) {
            u32x4 nx = {0, 0, 0, 0};
            if (c + 1 < chunks) nx = load16(c + 1);
            // the value's bytes [16 c + head, 16 c + head + 16)
            const uint32_t o0 = __builtin_amdgcn_alignbyte(cur.y, cur.x, head), o1 = __builtin_amdgcn_alignbyte(cur.z, cur.y, head);
            const uint32_t o2 = __builtin_amdgcn_alignbyte(cur.w, cur.z, head), o3 = __builtin_amdgcn_alignbyte(nx.x, cur.w, head);
            const uint32_t take = rem < 16 ? rem : 16, dwords = take >> 2;
            if (dwords > 0) out[4 * c] = o0;
            if (dwords > 1) out[4 * c + 1] = o1;
            if (dwords > 2) out[4 * c + 2] = o2;
            if (dwords > 3) out[4 * c + 3] = o3;
            if (take & 3) { // the value's last bytes
                const uint32_t last = dwords == 0 ? o0 : (dwords == 1 ? o1 : (dwords == 2 ? o2 : o3));
                uint8_t* t = reinterpret_cast<uint8_t*>(out + 4 * c + dwords);
                for (uint32_t b2 = 0; b2 < (take & 3); b2++) t[b2] = (uint8_t)(last >> (8 * b2));
            }
            rem -= take;
            cur = nx;
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // one wavefront: its LDS operations execute in order
    // ---- the image goes out: it agrees with the destination modulo 16 ----
    {
        gptr<uint8_t> o = L.dst + to0;
        const uint32_t bytes = (uint32_t)total;
        const uint32_t lead = std::min(bytes, (16 - align) & 15);
        const uint32_t whole = (bytes - lead) / 16, tail0 = lead + whole * 16;
        if (lane < lead) o[lane] = img[align + lane];
        if (tail0 + lane < bytes) o[tail0 + lane] = img[align + tail0 + lane];
        for (uint32_t c = lane; c < whole; c += 64)
            *(gptr<u32x4>)(o + lead + c * 16) = *reinterpret_cast<const u32x4*>(img + align + lead + c * 16);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); // the image is read before the next 64 values overwrite it
}

// Bytes of the dictionary's values: what may be read.
__device__ inline uint64_t dictionary_bytes(const Lookup& L) {
    if (L.kind == PQD_KIND_BYTE_ARRAY) return L.dict_len ? L.doff[L.dict_len] : 0;
    return L.dict_len * L.width;
}

// ---- pages of a batch ----

__device__ inline Lookup lookup_of(const DevPage& pg, const DevResult& lr) {
    Lookup L;
    L.idx = (gptr<const uint32_t>)as_global(pg.lengths);
    L.n = lr.out_values;
    L.dict = as_global(pg.dict);
    L.doff = as_global(pg.dict_offsets);
    L.dict_len = pg.dict_len;
    L.width = pg.kind == PQD_KIND_BOOLEAN ? 1u : (uint32_t)(pg.bit_width > 0 ? pg.bit_width : 0);
    L.kind = pg.kind;
    L.dst = as_global(pg.dst);
    L.dst_cap = pg.dst_cap;
    L.offsets = as_global(pg.offsets);
    L.offsets_cap = pg.offsets_cap;
    return L;
}

__global__ __launch_bounds__(DT) void dict_bytes_offsets_kernel(const DevPage* __restrict__ pages, DevResult* __restrict__ results,
                                                                const uint32_t* __restrict__ items) {
    const uint32_t item = items[blockIdx.x];
    const DevPage& pg = pages[item];
    DevResult* res = &results[item];
    const DevResult lr = results[pg.chain]; // the item that decoded the indexes
    if (lr.status != PQD_OK) {
        if (threadIdx.x == 0) {
            int32_t st = lr.status;
            int64_t arg = lr.arg;
            if (st == PQD_ERR_DST_TOO_SMALL) { // more indexes than the caller's buffers were sized for
                const uint64_t total = (uint64_t)lr.arg / 4;
                if (pg.kind == PQD_KIND_BYTE_ARRAY) { st = PQD_ERR_OFFSETS_TOO_SMALL; arg = (int64_t)(total + 1); }
                else arg = (int64_t)(total * (uint64_t)(pg.kind == PQD_KIND_BOOLEAN ? 1 : (pg.bit_width > 0 ? pg.bit_width : 1)));
            }
            res->status = st; res->arg = arg; res->out_values = 0; res->out_bytes = 0; res->consumed = lr.consumed; res->total = 0;
        }
        return;
    }
    const Outcome o = lookup_offsets(lookup_of(pg, lr));
    if (threadIdx.x == 0) {
        res->status = o.status; res->arg = o.arg; res->out_values = o.out_values; res->out_bytes = o.out_bytes;
        res->consumed = lr.consumed; res->total = lr.out_values;
    }
}

__global__ __launch_bounds__(DT) void dict_bytes_copy_kernel(const DevPage* __restrict__ pages, const DevResult* __restrict__ results,
                                                             const uint32_t* __restrict__ items) {
    const uint32_t item = items[blockIdx.x];
    const DevPage& pg = pages[item];
    if (results[item].status != PQD_OK) return;
    __shared__ __attribute__((aligned(16))) uint8_t img[DT / 64][IMG_CAP + 32];
    const Lookup L = lookup_of(pg, results[pg.chain]);
    const uint32_t wave = threadIdx.x >> 6;
    const uint64_t dict_bytes = dictionary_bytes(L);
    for (uint64_t k0 = ((uint64_t)blockIdx.y * (DT / 64) + wave) * 64; k0 < L.n; k0 += (uint64_t)gridDim.y * DT)
        lookup_copy(L, k0, img[wave], dict_bytes);
}

// ---- a stand-alone call ----

__global__ __launch_bounds__(DT) void dict_bytes_call_offsets_kernel(Lookup L, int64_t* __restrict__ status) {
    const Outcome o = lookup_offsets(L);
    if (threadIdx.x == 0) { status[0] = o.status; status[1] = o.arg; status[2] = (int64_t)o.out_bytes; }
}

__global__ __launch_bounds__(DT) void dict_bytes_call_copy_kernel(Lookup L, const int64_t* __restrict__ status) {
    __shared__ __attribute__((aligned(16))) uint8_t img[DT / 64][IMG_CAP + 32];
    if (status[0] != PQD_OK) return;
    const uint32_t wave = threadIdx.x >> 6;
    const uint64_t dict_bytes = dictionary_bytes(L);
    for (uint64_t k0 = ((uint64_t)blockIdx.x * (DT / 64) + wave) * 64; k0 < L.n; k0 += (uint64_t)gridDim.x * DT)
        lookup_copy(L, k0, img[wave], dict_bytes);
}

uint32_t copy_groups(uint64_t n) {
    const uint64_t g = (n + DT - 1) / DT;
    return (uint32_t)(g < 1 ? 1 : (g > 16384 ? 16384 : g));
}

} // namespace

void launch_dict_bytes(const DevPage* pages, DevResult* results, const uint32_t* items, uint32_t n_items, uint64_t max_values,
                       hipStream_t stream) {
    if (n_items == 0) return;
    hipLaunchKernelGGL(dict_bytes_offsets_kernel, dim3(n_items), dim3(DT), 0, stream, pages, results, items);
    hipLaunchKernelGGL(dict_bytes_copy_kernel, dim3(n_items, copy_groups(max_values)), dim3(DT), 0, stream, pages, results, items);
}

} // namespace pqd

using namespace pqd;

extern "C" int pqd_dict_lookup_bytes_device(pqd_ctx* ctx, int32_t kind, const void* dict_dev, const uint32_t* dict_offsets_dev,
                                            uint64_t dict_len, uint32_t width, const int32_t* indexes_dev, uint64_t n, void* dst_dev,
                                            uint64_t dst_cap, uint32_t* offsets_dev, uint64_t offsets_cap, int64_t* status_dev,
                                            void* hip_stream) {
    if (ctx == nullptr || status_dev == nullptr || (indexes_dev == nullptr && n > 0) || (dst_dev == nullptr && dst_cap > 0) ||
        (dict_dev == nullptr && dict_len > 0))
        return PQD_ERR_INVALID_ARGUMENT;
    if (kind != PQD_KIND_BYTE_ARRAY && kind != PQD_KIND_FIXED_LEN_BYTE_ARRAY && kind != PQD_KIND_BOOLEAN) return PQD_ERR_NOT_SUPPORTED;
    if (kind == PQD_KIND_BYTE_ARRAY && (dict_offsets_dev == nullptr || offsets_dev == nullptr)) return PQD_ERR_INVALID_ARGUMENT;
    if (kind == PQD_KIND_FIXED_LEN_BYTE_ARRAY && width == 0) return PQD_ERR_INVALID_ARGUMENT;
    DeviceGuard guard(ctx->device);
    if (!guard.ok) return PQD_ERR_DEVICE;
    Lookup L; // (plain casts: on the host a pointer is a number to hand on)
    L.idx = (gptr<const uint32_t>)reinterpret_cast<const uint32_t*>(indexes_dev);
    L.n = n;
    L.dict = (gptr<const uint8_t>)static_cast<const uint8_t*>(dict_dev);
    L.doff = (gptr<const uint32_t>)dict_offsets_dev;
    L.dict_len = dict_len;
    L.width = kind == PQD_KIND_BOOLEAN ? 1u : width;
    L.kind = kind;
    L.dst = (gptr<uint8_t>)static_cast<uint8_t*>(dst_dev);
    L.dst_cap = dst_cap;
    L.offsets = (gptr<uint32_t>)offsets_dev;
    L.offsets_cap = offsets_cap;
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    hipLaunchKernelGGL(dict_bytes_call_offsets_kernel, dim3(1), dim3(DT), 0, s, L, status_dev);
    hipLaunchKernelGGL(dict_bytes_call_copy_kernel, dim3(copy_groups(n)), dim3(DT), 0, s, L, status_dev);
    return hipGetLastError() == hipSuccess ? PQD_OK : PQD_ERR_DEVICE;
}
