// misc.hip -- the element-parallel and byte-array decoders for gfx950:
//
//   BIT_PACKED levels          bitpacked.Encoding.DecodeLevels
//                              (encoding/bitpacked/bitpacked.go:26-29,75-119)
//   BYTE_STREAM_SPLIT          bytestreamsplit.Encoding.Decode{Int32,Int64,Float,
//                              Double,FixedLenByteArray}
//                              (encoding/bytestreamsplit/bytestreamsplit.go:35-105,
//                              bytestreamsplit_purego.go:47-83,
//                              bytestreamsplit_fixedlen.go:13-21)
//   PLAIN fixed width          plain.Encoding.Decode{Boolean,Int32,...}
//                              (encoding/plain/plain.go:65-74,96-104, plain_le.go:26-52)
//   PLAIN byte arrays          plain.Encoding.DecodeByteArray (plain.go:76-94)
//   DELTA_LENGTH_BYTE_ARRAY    delta.LengthByteArrayEncoding.DecodeByteArray
//                              (encoding/delta/length_byte_array.go:37-63,
//                              length_byte_array_purego.go:11-24)
//   DELTA_BYTE_ARRAY           delta.ByteArrayEncoding.DecodeByteArray /
//                              DecodeFixedLenByteArray (encoding/delta/byte_array.go:124-174,
//                              byte_array_purego.go:5-63)
//
// The length streams of the two DELTA byte-array encodings are ordinary
// DELTA_BINARY_PACKED int32 items of the same batch (delta.hip); the kernels
// here start from their results.
#include <algorithm>

#include "pqd_internal.h"

namespace pqd {
namespace {

constexpr int MT = 256;                 // threads per workgroup of the tiled kernels
constexpr uint32_t MAX_FLBA = 32767;    // encoding.MaxFixedLenByteArraySize (encoding/encoding.go:12-14)

typedef uint32_t u32x4_a1 __attribute__((ext_vector_type(4), aligned(1)));
typedef uint32_t u32_a1 __attribute__((aligned(1)));

__device__ inline void finish(DevResult* r, int32_t status, int64_t arg, uint64_t n_values, uint64_t n_bytes,
                              uint64_t consumed) {
    r->status = status;
    r->arg = arg;
    r->out_values = n_values;
    r->out_bytes = n_bytes;
    r->consumed = consumed;
    r->total = n_values;
}

// Go shifts of a byte by an unsigned amount: amounts of 8 and more give 0.
__device__ inline uint32_t shl8(uint32_t v, uint64_t s) { return s >= 8 ? 0u : (v << s) & 0xffu; }
__device__ inline uint32_t shr8(uint32_t v, uint64_t s) { return s >= 8 ? 0u : (v & 0xffu) >> s; }

// ------------------------------------------------------------- BIT_PACKED --

// One value of the MSB-first stream (bitpacked.go:86-113).
__device__ inline uint32_t bitpacked_value(gptr<const uint8_t> src, uint64_t n, uint64_t k, uint32_t bw) {
    const uint64_t bit = k * bw;
    const uint64_t i = bit >> 3, j = bit & 7, available = 8 - j;
    const uint32_t mask = (shl8(1, bw) - 1) & 0xffu;
    if (bw <= available) return shr8(src[i], available - bw) & mask;
    const uint64_t remaining = bw - available;
    const uint32_t top = src[i] & ((shl8(1, available) - 1) & 0xffu);
    uint32_t bottom = 0;
    if (i + 1 < n) bottom = remaining > 8 ? 0u : shr8(src[i + 1], 8 - remaining);
    return (shl8(top, remaining) | bottom) & 0xffu;
}

// grid.x = pages of the family, grid.y = tile slots; a tile is MT * 16 values.
// Sixteen fields of BW bits, the first at the top of the 128-bit number x[3] .. x[0], one byte each (the shifts
// are compile-time constants: a bit-field extract per value, a funnel shift where a field straddles two words).
template <uint32_t BW>
__device__ inline void bitpacked_fields16(const uint32_t (&x)[4], uint32_t (&w)[4]) {
#pragma unroll
    for (uint32_t q = 0; q < 4; q++) {
        uint32_t o = 0;
#pragma unroll
        for (uint32_t i = 0; i < 4; i++) {
            const uint32_t e = q * 4 + i, pos = 128 - (e + 1) * BW, k = pos >> 5, off = pos & 31;
            uint32_t f;
            if (off + BW <= 32) f = (x[k] >> off) & ((1u << BW) - 1);
            else f = __builtin_amdgcn_alignbit(x[(k + 1) & 3], x[k], off) & ((1u << BW) - 1);
            o |= f << (8 * i);
        }
        w[q] = o;
    }
}

constexpr uint32_t BP_UNROLL = 4;

__global__ __launch_bounds__(MT) void bitpacked_kernel(const DevPage* __restrict__ pages, DevResult* __restrict__ results,
                                                       const uint32_t* __restrict__ items) {
    const uint32_t item = items[blockIdx.x];
    const DevPage& pg = pages[item];
    DevResult* res = &results[item];
    gptr<const uint8_t> src = as_global(pg.src);
    gptr<uint8_t> dst = as_global(pg.dst);
    const uint64_t n = pg.src_len;
    const uint32_t bw = (uint32_t)pg.bit_width;
    const bool lead = blockIdx.y == 0 && threadIdx.x == 0;
    if (pg.bit_width < 0) { // a negative width is a caller bug; Go would index out of range
        if (lead) finish(res, PQD_ERR_INVALID_BIT_WIDTH, pg.bit_width, 0, 0, 0);
        return;
    }
    if (bw == 0 || n == 0) { // bitpacked.go:76-78: one zero level
        if (lead) {
            if (pg.dst_cap < 1) finish(res, PQD_ERR_DST_TOO_SMALL, 1, 0, 0, 0);
            else { dst[0] = 0; finish(res, PQD_OK, 0, 1, 1, n); }
        }
        return;
    }
    const uint64_t bits = 8 * n;
    const uint64_t count = bits / bw + (bits % bw ? 1 : 0);
    if (count > pg.dst_cap) {
        if (lead) finish(res, PQD_ERR_DST_TOO_SMALL, (int64_t)count, 0, 0, 0);
        return;
    }
    // A workgroup takes BP_UNROLL stretches of MT * 16 values at a time, a thread sixteen values of each: its loads
    // are issued together (a workgroup of one load and one store per thread lives mostly on its own start-up).
    const uint64_t tile = (uint64_t)MT * 16 * BP_UNROLL;
    const bool dst_ok = ((uintptr_t)pg.dst & 3) == 0;
    for (uint64_t t = blockIdx.y; t * tile < count; t += gridDim.y) {
        // Sixteen values start on a byte boundary (16 * bw bits) and take 2 * bw bytes.  Widths up to 8
        // with all sixteen bytes inside the page: one unaligned 16-byte load, read as a big-endian
        // 128-bit number whose fields are cut out from the top; 16-byte store.
        uint64_t k[BP_UNROLL];
        bool fast[BP_UNROLL];
        u32x4_a1 raw[BP_UNROLL];
#pragma unroll
        for (uint32_t u = 0; u < BP_UNROLL; u++) {
            k[u] = t * tile + (uint64_t)u * MT * 16 + (uint64_t)threadIdx.x * 16;
            const uint64_t byte0 = 2 * (k[u] / 16) * bw;
            fast[u] = bw <= 8 && k[u] + 16 <= count && byte0 + 16 <= n && dst_ok;
            raw[u] = u32x4_a1{0, 0, 0, 0};
            if (fast[u]) raw[u] = *(gptr<const u32x4_a1>)(src + byte0);
        }
#pragma unroll
        for (uint32_t u = 0; u < BP_UNROLL; u++) {
            if (k[u] >= count) continue;
            if (fast[u]) {
                // x[0] holds the last four bytes: field e lies 128 - (e + 1) * bw bits above the bottom
                const uint32_t x[4] = {__builtin_bswap32(raw[u].w), __builtin_bswap32(raw[u].z), __builtin_bswap32(raw[u].y), __builtin_bswap32(raw[u].x)};
                uint32_t w[4];
                switch (bw) {
                    case 1: bitpacked_fields16<1>(x, w); break;
                    case 2: bitpacked_fields16<2>(x, w); break;
                    case 3: bitpacked_fields16<3>(x, w); break;
                    case 4: bitpacked_fields16<4>(x, w); break;
                    case 5: bitpacked_fields16<5>(x, w); break;
                    case 6: bitpacked_fields16<6>(x, w); break;
                    case 7: bitpacked_fields16<7>(x, w); break;
                    default: bitpacked_fields16<8>(x, w); break;
                }
                u32x4_a4 o;
                o.x = w[0]; o.y = w[1]; o.z = w[2]; o.w = w[3];
                *(gptr<u32x4_a4>)(dst + k[u]) = o;
            } else {
                for (uint32_t e = 0; e < 16 && k[u] + e < count; e++) dst[k[u] + e] = (uint8_t)bitpacked_value(src, n, k[u] + e, bw);
            }
        }
    }
    if (lead) finish(res, PQD_OK, 0, count, count, n);
}

// ------------------------------------------------------ BYTE_STREAM_SPLIT --

__device__ inline int32_t fixed_width(const DevPage& pg) {
    switch (pg.kind) {
        case PQD_KIND_BOOLEAN: return 1;
        case PQD_KIND_INT32: case PQD_KIND_FLOAT: return 4;
        case PQD_KIND_INT64: case PQD_KIND_DOUBLE: return 8;
        case PQD_KIND_INT96: return 12;
        default: return pg.bit_width; // FIXED_LEN_BYTE_ARRAY: the type length
    }
}

// Checks shared by BYTE_STREAM_SPLIT and PLAIN fixed-width pages
// (bytestreamsplit.go:35-105, plain.go:65-74,96-104).  Returns the value count.
__device__ inline bool fixed_checks(const DevPage& pg, DevResult* res, bool lead, int32_t width, uint64_t* count) {
    const uint64_t n = pg.src_len;
    if (width <= 0 || (pg.kind == PQD_KIND_FIXED_LEN_BYTE_ARRAY && (uint32_t)width > MAX_FLBA)) {
        if (lead) finish(res, PQD_ERR_INVALID_ARGUMENT, width, 0, 0, 0);
        return false;
    }
    if (n % (uint64_t)width) {
        if (lead) finish(res, PQD_ERR_INVALID_ARGUMENT, (int64_t)n, 0, 0, 0);
        return false;
    }
    if (n > pg.dst_cap) {
        if (lead) finish(res, PQD_ERR_DST_TOO_SMALL, (int64_t)n, 0, 0, 0);
        return false;
    }
    *count = n / (uint64_t)width;
    return true;
}

// dst[k * W + s] = src[s * count + k].  A thread takes four consecutive
// values: one (unaligned) dword from each of the W streams, transposed in
// registers, stored as 4 * W contiguous bytes.
template <int W>
__device__ inline void bss_tile(gptr<const uint8_t> src, gptr<uint8_t> dst, uint64_t count, uint64_t k) {
    if (k + 4 <= count) {
        uint32_t in[W];
#pragma unroll
        for (int s = 0; s < W; s++) in[s] = *(gptr<const u32_a1>)(src + (uint64_t)s * count + k);
        uint32_t out[W]; // value e occupies bytes [e * W, e * W + W)
#pragma unroll
        for (int d = 0; d < W; d++) {
            // output dword d holds bytes 4d .. 4d+3 of the 4*W block: byte b = value (b / W), stream (b % W)
            uint32_t w = 0;
#pragma unroll
            for (int c = 0; c < 4; c++) {
                const int b = 4 * d + c, e = b / W, s = b % W;
                w |= ((in[s] >> (8 * e)) & 0xffu) << (8 * c);
            }
            out[d] = w;
        }
        gptr<uint8_t> o = dst + k * W;
        if (((uintptr_t)o & 3) == 0) {
#pragma unroll
            for (int d = 0; d < W; d++) *(gptr<uint32_t>)(o + 4 * d) = out[d];
        } else {
#pragma unroll
            for (int d = 0; d < W; d++) *(gptr<u32_a1>)(o + 4 * d) = out[d];
        }
    } else {
        for (uint64_t e = k; e < count; e++)
            for (int s = 0; s < W; s++) dst[e * W + s] = src[(uint64_t)s * count + e];
    }
}

constexpr uint32_t BSS_UNROLL = 4;

__global__ __launch_bounds__(MT) void bss_kernel(const DevPage* __restrict__ pages, DevResult* __restrict__ results,
                                                 const uint32_t* __restrict__ items) {
    const uint32_t item = items[blockIdx.x];
    const DevPage& pg = pages[item];
    DevResult* res = &results[item];
    const bool lead = blockIdx.y == 0 && threadIdx.x == 0;
    const int32_t width = fixed_width(pg);
    uint64_t count;
    if (!fixed_checks(pg, res, lead, width, &count)) return;
    gptr<const uint8_t> src = as_global(pg.src);
    gptr<uint8_t> dst = as_global(pg.dst);
    // a workgroup takes BSS_UNROLL stretches of MT * 4 values at a time (the compiler keeps their loads together)
    const uint64_t tile = (uint64_t)MT * 4 * BSS_UNROLL;
    for (uint64_t t = blockIdx.y; t * tile < count; t += gridDim.y) {
#pragma unroll
        for (uint32_t u = 0; u < BSS_UNROLL; u++) {
            const uint64_t k = t * tile + (uint64_t)u * MT * 4 + threadIdx.x * 4;
            if (k >= count) continue;
            if (width == 4) bss_tile<4>(src, dst, count, k);
            else if (width == 8) bss_tile<8>(src, dst, count, k);
            else {
                for (uint64_t e = k; e < k + 4 && e < count; e++)
                    for (int32_t s = 0; s < width; s++) dst[e * (uint64_t)width + s] = src[(uint64_t)s * count + e];
            }
        }
    }
    if (lead) finish(res, PQD_OK, 0, count, pg.src_len, pg.src_len);
}

// ------------------------------------------------------------------- copy --

// Copies n bytes; the caller gives every workgroup of the page the same
// arguments.  16 bytes per thread and step, destination aligned.
__device__ inline void copy_span(gptr<uint8_t> dst, gptr<const uint8_t> src, uint64_t n) {
    if (n == 0 || (uintptr_t)dst == (uintptr_t)src) return;
    const uint64_t head = ((16 - ((uintptr_t)dst & 15)) & 15) < n ? ((16 - ((uintptr_t)dst & 15)) & 15) : n;
    const uint64_t body = (n - head) / 16;
    const uint64_t tail0 = head + body * 16;
    if (blockIdx.y == 0) {
        for (uint64_t i = threadIdx.x; i < head; i += MT) dst[i] = src[i];
        for (uint64_t i = tail0 + threadIdx.x; i < n; i += MT) dst[i] = src[i];
    }
    for (uint64_t c = (uint64_t)blockIdx.y * MT + threadIdx.x; c < body; c += (uint64_t)gridDim.y * MT) {
        const u32x4_a1 w = *(gptr<const u32x4_a1>)(src + head + c * 16);
        *(gptr<u32x4>)(dst + head + c * 16) = w;
    }
}

__global__ __launch_bounds__(MT) void plain_fixed_kernel(const DevPage* __restrict__ pages, DevResult* __restrict__ results,
                                                         const uint32_t* __restrict__ items) {
    const uint32_t item = items[blockIdx.x];
    const DevPage& pg = pages[item];
    DevResult* res = &results[item];
    const bool lead = blockIdx.y == 0 && threadIdx.x == 0;
    const int32_t width = fixed_width(pg);
    uint64_t count;
    if (!fixed_checks(pg, res, lead, width, &count)) return;
    copy_span(as_global(pg.dst), as_global(pg.src), pg.src_len);
    if (lead) finish(res, PQD_OK, 0, count, pg.src_len, pg.src_len);
}

// ------------------------------------------------------ PLAIN byte arrays --

constexpr int WT = 64;                   // the walkers below run one wavefront per page

// Hand-off through LDS inside a workgroup of ONE wavefront: the wave's LDS operations execute in
// order, so all it takes is that the compiler keeps them in order and the earlier ones have
// landed.  __syncthreads() would also wait for every outstanding global store of the wave
// (vmcnt(0)), a microsecond or more per call once the stores are queued behind others.
__device__ inline void wave_sync() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); }
constexpr uint32_t PSTAGE = 16384;       // bytes of the page staged in LDS at a time
constexpr uint32_t PBATCH = 64;          // values handled per round

// Fills LDS bytes [0, n) from global memory: 16 bytes per lane and step, eight steps' loads in
// flight together (one wavefront has nobody to hide a load's latency behind), then the tail.
__device__ inline void stage_fill(uint8_t* stage, gptr<const uint8_t> src, uint64_t n, uint32_t lane) {
    const uint64_t body = n / 16;
    uint64_t c0 = 0;
    for (; c0 + 8 * WT <= body; c0 += 8 * WT) {
        u32x4_a1 w[8];
#pragma unroll
        for (uint32_t u = 0; u < 8; u++) w[u] = *(gptr<const u32x4_a1>)(src + (c0 + u * WT + lane) * 16);
#pragma unroll
        for (uint32_t u = 0; u < 8; u++) *reinterpret_cast<u32x4*>(stage + (c0 + u * WT + lane) * 16) = w[u];
    }
    for (uint64_t c = c0 + lane; c < body; c += WT) {
        const u32x4_a1 w = *(gptr<const u32x4_a1>)(src + c * 16);
        *reinterpret_cast<u32x4*>(stage + c * 16) = w;
    }
    for (uint64_t b = body * 16 + lane; b < n; b += WT) stage[b] = src[b];
}

__device__ inline uint32_t le32(const uint8_t* p) {
    return (uint32_t)p[0] | ((uint32_t)p[1] << 8) | ((uint32_t)p[2] << 16) | ((uint32_t)p[3] << 24);
}

// Four bytes at any LDS byte offset: two aligned dwords and a funnel shift.
__device__ inline uint32_t lds_u32(const uint32_t* stage32, uint32_t off) {
    const uint32_t d = off >> 2;
    return __builtin_amdgcn_alignbit(stage32[d + 1], stage32[d], (off & 3) * 8);
}

// plain.go:76-94: a walk over `u32 length | bytes` records.  Each record's position depends on
// the one before, so the lengths are found speculatively: lane l guesses that record l starts
// l * (4 + g) bytes after the first, g being the length of the record before, reads the length
// it finds there, and the guess holds for every lane whose predecessors all measured g again
// (equal lengths in a row -- ids, dates, fixed formats -- give 64 records per round, anything else
// at least one).  Rounds fill a batch of 64 records out of an LDS copy of 16 KB of the page, which
// serves batch after batch until the next record is not wholly in it; then every lane owns one
// record: offsets by a wave scan, bytes copied four at a time by all lanes at once.
__global__ __launch_bounds__(WT) void plain_byte_array_kernel(const DevPage* __restrict__ pages,
                                                              DevResult* __restrict__ results,
                                                              const uint32_t* __restrict__ items) {
    __shared__ __attribute__((aligned(16))) uint8_t stage[PSTAGE + 16];
    __shared__ uint32_t v_pos[PBATCH], v_len[PBATCH];
    const uint32_t* stage32 = reinterpret_cast<const uint32_t*>(stage);
    const uint32_t item = items[blockIdx.x];
    const DevPage& pg = pages[item];
    if (pg.lengths != nullptr) return; // a page with scratch is decoded by many workgroups (plain_bytes.hip)
    DevResult* res = &results[item];
    gptr<const uint8_t> src = as_global(pg.src);
    gptr<uint8_t> dst = as_global(pg.dst);
    gptr<uint32_t> offsets = as_global(pg.offsets);
    const uint64_t n = pg.src_len, dst_cap = pg.dst_cap, offsets_cap = pg.offsets_cap;
    const uint32_t lane = threadIdx.x;
    uint64_t i = 0, out = 0, nv = 0;
    uint32_t guess = 0;
    int32_t status = PQD_OK;
    int64_t arg = 0;
    while (i < n && status == PQD_OK) {
        // stage [i, i + PSTAGE); batches are taken out of it for as long as whole records are staged
        const uint32_t avail = n - i < PSTAGE ? (uint32_t)(n - i) : PSTAGE;
        stage_fill(stage, src + i, avail, lane);
        wave_sync();
        const uint64_t left = n - i; // bytes from the start of the stage to the end of the page
        uint32_t p = 0;              // position in the stage
        bool restage = false;
        while (!restage && status == PQD_OK) {
            uint32_t cnt = 0;        // records in the batch
            bool big = false;        // a record that does not fit the stage: handled on its own
            while (cnt < PBATCH && p < left && status == PQD_OK) {
                // ---- one speculative round ----
                const uint64_t c64 = (uint64_t)p + (uint64_t)lane * (4ull + guess);
                const bool inwin = c64 + 4 <= avail;                 // header staged
                const uint32_t c = inwin ? (uint32_t)c64 : 0;
                const uint32_t len = lds_u32(stage32, c);
                // 0 fine, 1 truncated input (plain.ErrTooShort), 2 stop here: not (all) staged
                uint32_t what = 0;
                if (c64 >= left) what = 2;                           // past the last record
                else if (left - c64 < 4) what = 1;
                else if (!inwin) what = 2;
                else if (len > n - 4 || len > left - c64 - 4) what = 1;
                else if (c64 + 4 + len > avail) what = 2;
                const unsigned long long same = __ballot(what == 0 && len == guess);
                const uint32_t lead = same == ~0ull ? 64u : (uint32_t)__builtin_ctzll(~same);
                // lanes [0, lead) measured the guess again; lane `lead` sits at a true record start too
                uint32_t valid = lead;
                uint32_t stop_what = 2;
                if (lead < 64) {
                    stop_what = lane_u32(what, lead);
                    if (stop_what == 0) valid = lead + 1;
                }
                if (valid > PBATCH - cnt) { valid = PBATCH - cnt; stop_what = 0; }
                if (lane < valid) { v_pos[cnt + lane] = c + 4; v_len[cnt + lane] = len; }
                if (valid > 0) {
                    const uint32_t l = valid - 1;
                    guess = lane_u32(len, l);
                    p = lane_u32(c, l) + 4 + guess;
                    cnt += valid;
                }
                if (valid > 0 && valid <= 2 && stop_what == 0 && cnt < PBATCH) {
                    // Lengths that differ from record to record: a round costs some sixty instructions and
                    // yields one record.  The wavefront then follows the chain itself, one uniform LDS read per
                    // record, for the rest of the batch; anything but a whole, staged, fitting record ends
                    // the chase and is left to the next round above, which knows what to call it.
                    uint32_t pp = (uint32_t)__builtin_amdgcn_readfirstlane((int)p);
                    uint32_t c = (uint32_t)__builtin_amdgcn_readfirstlane((int)cnt);
                    uint32_t my_pos = 0, my_len = 0, last = guess;
                    while (c < PBATCH && (uint64_t)pp + 4 <= avail) {
                        const uint32_t l = (uint32_t)__builtin_amdgcn_readfirstlane((int)lds_u32(stage32, pp));
                        if (l > left - pp - 4 || l > avail - pp - 4) break;
                        if (lane == c) { my_pos = pp + 4; my_len = l; }
                        pp += 4 + l;
                        last = l;
                        c++;
                    }
                    if (lane >= cnt && lane < c) { v_pos[lane] = my_pos; v_len[lane] = my_len; }
                    guess = last;
                    p = pp;
                    cnt = c;
                    continue;
                }
                if (valid == 0 || (valid <= lead && lead < 64 && stop_what != 0)) {
                    // the record after the valid ones cannot be taken this round
                    const uint32_t at = valid == 0 ? 0 : lead; // lane holding it (valid == lead here)
                    const uint32_t w = valid == 0 ? lane_u32(what, 0) : stop_what;
                    if (w == 1) { status = PQD_ERR_UNEXPECTED_EOF; arg = PQD_EOF_PLAIN_BYTE_ARRAY; }
                    else if (cnt == 0 && p == 0 && lane_u32((uint32_t)inwin, at) && lane_u32((uint32_t)(c64 < left), at)) {
                        // the stage starts with a record whose bytes it cannot hold: take it alone
                        if (lane == 0) { v_pos[0] = lane_u32(c, at) + 4; v_len[0] = lane_u32(len, at); }
                        big = true;
                        cnt = 1;
                    }
                    break;
                }
            }
            wave_sync();
            if (big) {
                // one record longer than the stage: straight from global memory, all lanes on it
                const uint32_t blen = v_len[0], bpos = v_pos[0];
                if (nv + 2 > offsets_cap) { status = PQD_ERR_OFFSETS_TOO_SMALL; arg = (int64_t)(nv + 2); }
                else if (out + blen > dst_cap) { status = PQD_ERR_DST_TOO_SMALL; arg = (int64_t)(out + blen); }
                else {
                    if (lane == 0) offsets[nv] = (uint32_t)out;
                    for (uint64_t b = lane; b < blen; b += WT) dst[out + b] = src[i + bpos + b];
                    out += blen;
                    nv += 1;
                    p = 0;
                    i += (uint64_t)bpos + blen;
                    guess = blen;
                }
                restage = true;
            } else if (cnt == 0) {
                // nothing staged could be taken: the page is through, or the next record starts a new stage
                if (p == 0 && status == PQD_OK) status = -1; // a fresh stage that yields nothing: cannot happen, but never spin
                restage = true;
            } else {
                const uint32_t pos = lane < cnt ? v_pos[lane] : 0, len = lane < cnt ? v_len[lane] : 0;
                const uint32_t incl = wave_inclusive_scan<uint32_t>(len);
                const uint64_t o = out + (incl - len);
                // capacity checks in record order (plain.go walks and appends one value at a time)
                uint32_t cap_err = 0;
                if (lane < cnt) {
                    if (nv + lane + 2 > offsets_cap) cap_err = 1;
                    else if (o + len > dst_cap) cap_err = 2;
                }
                const unsigned long long capm = __ballot(cap_err != 0);
                uint32_t good = cnt;
                if (capm) {
                    good = (uint32_t)__builtin_ctzll(capm);
                    const uint32_t e = lane_u32(cap_err, good);
                    status = e == 1 ? PQD_ERR_OFFSETS_TOO_SMALL : PQD_ERR_DST_TOO_SMALL;
                    arg = e == 1 ? (int64_t)(nv + good + 2) : (int64_t)(lane_u64(o, good) + lane_u32(len, good));
                }
                const uint32_t mylen = lane < good ? len : 0;
                if (lane < good) offsets[nv + lane] = (uint32_t)o;
                // the longest record of the batch sets the number of rounds
                uint32_t longest = mylen;
                for (int d = 32; d > 0; d >>= 1) { const uint32_t y = __shfl_xor(longest, d); longest = y > longest ? y : longest; }
                // four rounds at a time, their LDS reads issued together
                for (uint32_t r = 0; r * 4 < longest; r += 4) {
                    uint32_t w[4];
#pragma unroll
                    for (uint32_t u = 0; u < 4; u++) w[u] = (r + u) * 4 < mylen ? lds_u32(stage32, pos + (r + u) * 4) : 0u;
#pragma unroll
                    for (uint32_t u = 0; u < 4; u++) {
                        const uint32_t b = (r + u) * 4;
                        gptr<uint8_t> d = dst + o + b;
                        if (b + 4 <= mylen) *(gptr<u32_a1>)d = w[u];
                        else if (b < mylen) for (uint32_t k = 0; b + k < mylen; k++) d[k] = (uint8_t)(w[u] >> (8 * k));
                    }
                }
                if (good > 0) {
                    out = lane_u64(o, good - 1) + lane_u32(len, good - 1);
                    nv += good;
                }
                if (good < cnt) p = good > 0 ? lane_u32(pos, good - 1) + lane_u32(len, good - 1) : (cnt > 0 ? lane_u32(pos, 0) - 4 : p);
            }
            wave_sync();
        }
        if (p > 0) i += p;
        if (status == -1) { status = PQD_OK; break; }
    }
    if (lane == 0) {
        if (status == PQD_OK) {
            if (nv + 1 > offsets_cap) { status = PQD_ERR_OFFSETS_TOO_SMALL; arg = (int64_t)(nv + 1); }
            else offsets[nv] = (uint32_t)out;
        }
        finish(res, status, arg, nv, out, status == PQD_OK ? n : i);
    }
}

// ------------------------------------------- DELTA_LENGTH_BYTE_ARRAY scan --

// Maps the result of a length stream that failed.  A stream that only ran out
// of room asks for more offsets (BYTE_ARRAY) or more output (FIXED_LEN).
__device__ inline void lengths_failed(DevResult* res, const DevResult& r, const DevPage& pg) {
    int32_t st = r.status;
    int64_t arg = r.arg;
    if (st == PQD_ERR_DST_TOO_SMALL) {
        const uint64_t total = (uint64_t)r.arg / 4;
        if (pg.kind == PQD_KIND_BYTE_ARRAY) { st = PQD_ERR_OFFSETS_TOO_SMALL; arg = (int64_t)(total + 1); }
        else arg = (int64_t)(total * (uint64_t)(pg.bit_width > 0 ? pg.bit_width : 1));
    }
    finish(res, st, arg, 0, 0, r.consumed);
}

// length_byte_array_purego.go:11-24 on the lengths the chained item left in
// the offsets buffer: exclusive prefix sum in place, the first negative length
// is an error.  One workgroup per page; the copy of the bytes is a second
// kernel so that it can use the whole chip.
__global__ __launch_bounds__(MT) void dlba_offsets_kernel(const DevPage* __restrict__ pages, DevResult* __restrict__ results,
                                                          const uint32_t* __restrict__ items) {
    __shared__ uint32_t wsum[MT / 64];
    __shared__ unsigned long long first_bad;
    const uint32_t item = items[blockIdx.x];
    const DevPage& pg = pages[item];
    DevResult* res = &results[item];
    const DevResult lr = results[pg.chain];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (lr.status != PQD_OK) {
        if (tid == 0) lengths_failed(res, lr, pg);
        return;
    }
    const uint64_t nl = lr.out_values;
    if (nl + 1 > pg.offsets_cap) {
        if (tid == 0) finish(res, PQD_ERR_OFFSETS_TOO_SMALL, (int64_t)(nl + 1), 0, 0, lr.consumed);
        return;
    }
    gptr<uint32_t> off = as_global(pg.offsets);
    if (tid == 0) first_bad = ~0ull;
    __syncthreads();
    // Sixteen lengths per thread and turn, the next turn's loads issued before this turn's sums: one workgroup
    // walks the whole page, and a turn of load, scan, two barriers, store, each waiting for the one before, took
    // 1.3 us per 1024 lengths whatever the chip was doing.
    constexpr uint32_t PER = 16;
    auto load = [&](uint64_t base, u32x4_a4 (&q)[PER / 4]) {
        const uint64_t k = base + (uint64_t)tid * PER;
#pragma unroll
        for (uint32_t e = 0; e < PER / 4; e++) {
            q[e] = u32x4_a4{0, 0, 0, 0};
            if (k + 4 * e + 4 <= nl) q[e] = *(gptr<const u32x4_a4>)(off + k + 4 * e);
            else if (k + 4 * e < nl) {
                uint32_t t[4] = {0, 0, 0, 0};
                for (uint32_t i = 0; i < 4 && k + 4 * e + i < nl; i++) t[i] = off[k + 4 * e + i];
                q[e] = u32x4_a4{t[0], t[1], t[2], t[3]};
            }
        }
    };
    uint32_t carry = 0;
    u32x4_a4 cur[PER / 4], nxt[PER / 4];
    load(0, cur);
    for (uint64_t base = 0; base < nl; base += (uint64_t)MT * PER) {
        const uint64_t k = base + (uint64_t)tid * PER;
        if (base + (uint64_t)MT * PER < nl) load(base + (uint64_t)MT * PER, nxt);
        uint32_t v[PER];
#pragma unroll
        for (uint32_t e = 0; e < PER / 4; e++) { v[4 * e] = cur[e].x; v[4 * e + 1] = cur[e].y; v[4 * e + 2] = cur[e].z; v[4 * e + 3] = cur[e].w; }
        uint32_t mine = 0, neg = 0;
#pragma unroll
        for (uint32_t e = 0; e < PER; e++) { mine += v[e]; neg |= v[e]; }
        if ((int32_t)neg < 0) { // (lengths past the end are zeros)
            for (uint32_t e = 0; e < PER; e++)
                if ((int32_t)v[e] < 0) { atomicMin(&first_bad, ((unsigned long long)(k + e) << 32) | v[e]); break; }
        }
        const uint32_t x = wave_inclusive_scan<uint32_t>(mine);
        if (lane == 63) wsum[wave] = x;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory"); // LDS only: __syncthreads would wait for this turn's stores
        uint32_t before = carry, all = 0;
        for (uint32_t w = 0; w < MT / 64; w++) {
            if (w < wave) before += wsum[w];
            all += wsum[w];
        }
        uint32_t o = before + x - mine;
#pragma unroll
        for (uint32_t e = 0; e < PER / 4; e++) {
            u32x4_a4 w4;
            w4.x = o; o += v[4 * e];
            w4.y = o; o += v[4 * e + 1];
            w4.z = o; o += v[4 * e + 2];
            w4.w = o; o += v[4 * e + 3];
            if (k + 4 * e + 4 <= nl) *(gptr<u32x4_a4>)(off + k + 4 * e) = w4;
            else {
                const uint32_t t[4] = {w4.x, w4.y, w4.z, w4.w};
                for (uint32_t i = 0; i < 4 && k + 4 * e + i < nl; i++) off[k + 4 * e + i] = t[i];
            }
        }
        carry += all;
        asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
#pragma unroll
        for (uint32_t e = 0; e < PER / 4; e++) cur[e] = nxt[e];
    }
    __syncthreads();
    if (tid == 0) {
        const unsigned long long bad = first_bad;
        const uint64_t rest = pg.src_len - lr.consumed;
        if (bad != ~0ull) {
            finish(res, PQD_ERR_NEGATIVE_LENGTH, (int64_t)(int32_t)(uint32_t)bad, 0, 0, lr.consumed);
        } else {
            off[nl] = carry;
            if ((uint64_t)carry > rest) finish(res, PQD_ERR_LENGTH_OUT_OF_BOUNDS, (int64_t)carry, nl, 0, lr.consumed);
            else if ((uint64_t)carry > pg.dst_cap) finish(res, PQD_ERR_DST_TOO_SMALL, (int64_t)carry, nl, 0, lr.consumed);
            else finish(res, PQD_OK, 0, nl, carry, lr.consumed + carry);
        }
    }
}

// length_byte_array.go:58-60: the values are the bytes after the lengths.
__global__ __launch_bounds__(MT) void dlba_copy_kernel(const DevPage* __restrict__ pages, DevResult* __restrict__ results,
                                                       const uint32_t* __restrict__ items) {
    const uint32_t item = items[blockIdx.x];
    const DevPage& pg = pages[item];
    const DevResult r = results[item];
    if (r.status != PQD_OK) return;
    copy_span(as_global(pg.dst), as_global(pg.src) + (r.consumed - r.out_bytes), r.out_bytes);
}

// ------------------------------------------------------- DELTA_BYTE_ARRAY --

constexpr uint32_t LAST_CAP = 2048;  // leading bytes of the previous value kept in LDS
constexpr uint32_t DSTAGE = 8192;    // suffix bytes staged per pass
constexpr uint32_t DPREF = 2048;     // suffix bytes fetched a round ahead

// Exclusive prefix sum over the wave and the wave's total, with the DPP scan (no LDS round trips).
__device__ inline uint64_t wave_excl_scan64(uint64_t v, uint32_t lane, uint64_t* total) {
    const uint64_t x = wave_inclusive_scan<uint64_t>(v);
    *total = lane_u64(x, 63);
    return x - v;
}

// byte_array_purego.go:5-63.  Value k is the first prefix[k] bytes of value
// k-1 followed by suffix[k] bytes of the data section, so values are produced
// in order, one wavefront per page: lengths are checked and turned into source
// and output positions 64 values at a time, the suffix bytes of those values
// are staged in LDS with coalesced loads, and the running "previous value"
// lives in LDS too, where a new value only overwrites the bytes after its
// prefix.
__global__ __launch_bounds__(WT) void dba_kernel(const DevPage* __restrict__ pages, DevResult* __restrict__ results,
                                                 const uint32_t* __restrict__ items) {
    __shared__ __attribute__((aligned(16))) uint8_t stage[DSTAGE + 16];
    __shared__ __attribute__((aligned(16))) uint8_t last[LAST_CAP];
    const uint32_t item = items[blockIdx.x];
    const DevPage& pg = pages[item];
    DevResult* res = &results[item];
    const int32_t s_item = pg.chain, p_item = pages[s_item].chain;
    const DevResult rp = results[p_item], rs = results[s_item];
    const uint32_t lane = threadIdx.x;
    if (res->aux != 0) return; // decoded block by block (dba_plan_kernel, dba_expand_kernel)
    const bool with_offsets = pg.kind == PQD_KIND_BYTE_ARRAY;
    if (!with_offsets && (pg.bit_width < 0 || (uint32_t)pg.bit_width > MAX_FLBA)) { // byte_array.go:149-151
        if (lane == 0) finish(res, PQD_ERR_INVALID_ARGUMENT, pg.bit_width, 0, 0, 0);
        return;
    }
    if (rp.status != PQD_OK) { if (lane == 0) lengths_failed(res, rp, pg); return; }
    if (rs.status != PQD_OK) { if (lane == 0) lengths_failed(res, rs, pg); return; }
    const uint64_t np = rp.out_values, ns = rs.out_values;
    if (np != ns) { // byte_array.go:142-144
        if (lane == 0) finish(res, PQD_ERR_PREFIX_SUFFIX_MISMATCH, (int64_t)((np << 32) | (ns & 0xffffffffull)), 0, 0, rs.consumed);
        return;
    }
    gptr<const int32_t> prefix = (gptr<const int32_t>)as_global(pg.lengths);
    gptr<const int32_t> suffix = (gptr<const int32_t>)as_global(pg.lengths2);
    gptr<const uint8_t> data = as_global(pg.src) + rs.consumed;
    const uint64_t data_total = pg.src_len - rs.consumed;
    gptr<uint8_t> dst = as_global(pg.dst);
    gptr<uint32_t> offsets = as_global(pg.offsets);

    uint64_t src_base = 0, out_base = 0, done = 0;
    uint64_t prev_len = 0, prev_out = 0; // the previous value: its length and where it went
    int32_t status = PQD_OK;
    int64_t arg = 0;
    // Two things are fetched a round ahead, so that no round starts by waiting for memory: the
    // lengths of the next 64 values, and the DPREF suffix bytes that follow this round's.
    uint64_t staged_base = 0, staged_len = 0; // data bytes [staged_base, staged_base + staged_len) are in `stage`
    int32_t pl_next = lane < ns ? prefix[lane] : 0, sl_next = lane < ns ? suffix[lane] : 0;
    while (done < ns && status == PQD_OK) {
        const uint64_t k = done + lane;
        const bool valid = k < ns;
        const int64_t pl = valid ? pl_next : 0, sl = valid ? sl_next : 0;
        pl_next = k + WT < ns ? prefix[k + WT] : 0;
        sl_next = k + WT < ns ? suffix[k + WT] : 0;
        const uint64_t s_pos = sl > 0 ? (uint64_t)sl : 0, p_pos = pl > 0 ? (uint64_t)pl : 0;
        uint64_t s_tot, v_tot;
        const uint64_t s_off = src_base + wave_excl_scan64(s_pos, lane, &s_tot);
        const uint64_t o_off = out_base + wave_excl_scan64(p_pos + s_pos, lane, &v_tot);
        uint64_t before = dpp0<0x138, 0xf>(p_pos + s_pos); // wave_shr:1: the value one lane down
        if (lane == 0) before = prev_len;
        int32_t err = PQD_OK;
        int64_t err_arg = 0;
        if (valid) {
            if (sl < 0) { err = PQD_ERR_NEGATIVE_LENGTH; err_arg = sl; }
            else if ((uint64_t)sl > data_total - s_off) { err = PQD_ERR_LENGTH_OUT_OF_BOUNDS; err_arg = sl; }
            else if (pl < 0) { err = PQD_ERR_NEGATIVE_PREFIX; err_arg = pl; }
            else if ((uint64_t)pl > before) { err = PQD_ERR_PREFIX_OUT_OF_BOUNDS; err_arg = pl; }
            else if (with_offsets && k + 2 > pg.offsets_cap) { err = PQD_ERR_OFFSETS_TOO_SMALL; err_arg = (int64_t)(ns + 1); }
            else if (o_off + p_pos + s_pos > pg.dst_cap) { err = PQD_ERR_DST_TOO_SMALL; err_arg = -1; }
        }
        const unsigned long long bad = __ballot(err != PQD_OK);
        const unsigned long long live = __ballot(valid);
        uint32_t m = (uint32_t)__popcll(live);
        if (bad) {
            const uint32_t first = (uint32_t)__builtin_ctzll(bad);
            m = first;
            status = (int32_t)lane_u32((uint32_t)err, first);
            arg = (int64_t)lane_u64((uint64_t)err_arg, first);
        }
        // the suffix bytes after this round's, requested now and put into `stage` once the round is through
        const uint64_t next0 = src_base + s_tot;
        u32x4_a1 pf[DPREF / (16 * WT)];
        uint64_t pf_len = 0;
        if (!bad && next0 < data_total) {
            const uint64_t room = (data_total - next0) & ~15ull;
            pf_len = room < DPREF ? room : DPREF;
        }
#pragma unroll
        for (uint32_t u = 0; u < DPREF / (16 * WT); u++) {
            const uint64_t at = (uint64_t)(u * WT + lane) * 16;
            pf[u] = (u32x4_a1)(0u);
            if (at < pf_len) pf[u] = *(gptr<const u32x4_a1>)(data + next0 + at);
        }
        // values [done, done + m) are good; the staged spans may cut the round into several passes
        uint32_t taken = 0;
        while (taken < m) {
            const uint64_t span0 = lane_u64(s_off, taken);
            // how many values from `taken` on fit one pass: suffix bytes in `stage`, no value longer
            // than the wave can hold in registers (4 bytes per lane)
            const bool fits = lane >= taken && lane < m && (s_off + s_pos - span0) <= DSTAGE && p_pos + s_pos <= 4 * WT;
            const unsigned long long fmask = __ballot(fits) >> taken;
            const uint32_t cnt = fmask == ~0ull ? 64u : (uint32_t)__builtin_ctzll(~fmask);
            if (cnt > 0) {
                // ---- fast pass.  The running value lives in registers, lane b holding its bytes
                // 4b..4b+3: a new value keeps the bytes below its prefix length and takes the rest
                // from the staged suffix bytes, so the only thing carried from value to value is a
                // register, and the LDS reads of the suffixes do not wait for one another. ----
                const uint64_t span1 = lane_u64(s_off + s_pos, taken + cnt - 1);
                if (span0 < staged_base || span1 > staged_base + staged_len) { // not what the last round fetched ahead
                    stage_fill(stage, data + span0, span1 - span0, lane);
                    staged_base = span0;
                    staged_len = span1 - span0;
                    wave_sync();
                }
                const uint32_t* stage32 = reinterpret_cast<const uint32_t*>(stage);
                const uint32_t so32 = (uint32_t)(s_off - staged_base), p32 = (uint32_t)p_pos, s32 = (uint32_t)s_pos;
                if (with_offsets && lane >= taken && lane < taken + cnt) offsets[done + lane] = (uint32_t)o_off;
                uint32_t cur = reinterpret_cast<const uint32_t*>(last)[lane];
                // four values per trip: their suffix dwords are read from LDS together, then the
                // register is updated and stored value by value
                for (uint32_t v0 = taken; v0 < taken + cnt; v0 += 4) {
                    uint32_t p_v[4], len_v[4], suf[4];
                    int32_t kk[4];
#pragma unroll
                    for (uint32_t u = 0; u < 4; u++) {
                        const uint32_t v = v0 + u < taken + cnt ? v0 + u : taken + cnt - 1; // the last one again: harmless
                        p_v[u] = lane_u32(p32, v);
                        len_v[u] = p_v[u] + lane_u32(s32, v);
                        kk[u] = (int32_t)p_v[u] - (int32_t)(4 * lane);   // bytes of this lane's dword below the prefix length
                        suf[u] = lds_u32(stage32, lane_u32(so32, v) + (kk[u] < 0 ? (uint32_t)(-kk[u]) : 0u));
                    }
#pragma unroll
                    for (uint32_t u = 0; u < 4; u++) {
                        if (v0 + u >= taken + cnt) break;
                        const uint64_t oo = lane_u64(o_off, v0 + u);
                        const int32_t k = kk[u];
                        if (k <= 0) cur = suf[u];
                        else if (k < 4) cur = (cur & ((1u << (8 * k)) - 1)) | (suf[u] << (8 * k));
                        const uint32_t len = len_v[u];
                        gptr<uint8_t> o = dst + oo + 4 * lane;
                        if (4 * lane + 4 <= len) *(gptr<u32_a1>)(o) = cur;
                        else if (4 * lane < len) {
                            const uint32_t rest = len - 4 * lane; // 1..3 bytes
                            if (rest >= 2) *(gptr<uint16_t>)(o) = (uint16_t)cur;
                            if (rest & 1) o[rest - 1] = (uint8_t)(cur >> (8 * (rest - 1)));
                        }
                    }
                }
                const uint32_t last_len = lane_u32(p32 + s32, taken + cnt - 1);
                reinterpret_cast<uint32_t*>(last)[lane] = cur; // the slow pass and the next fast pass start from here
                prev_len = last_len;
                prev_out = lane_u64(o_off, taken + cnt - 1);
                wave_sync();
                taken += cnt;
                continue;
            }
            // ---- slow pass: one value that is long, or follows a long one ----
            {
                const uint32_t v = taken;
                const uint64_t p_v = lane_u64(p_pos, v), s_v = lane_u64(s_pos, v);
                const uint64_t so = lane_u64(s_off, v), oo = lane_u64(o_off, v);
                if (with_offsets && lane == 0) offsets[done + v] = (uint32_t)oo;
                const uint64_t len = p_v + s_v;
                // prefix bytes: from the LDS copy of the previous value, beyond it from where that value went
                if (p_v > LAST_CAP) {
                    __threadfence(); // the previous value's stores must be readable
                    for (uint64_t b = LAST_CAP + lane; b < p_v; b += WT) dst[oo + b] = dst[prev_out + b];
                }
                // suffix bytes into the LDS copy (as far as it reaches) ...
                for (uint64_t b = lane; b < s_v && p_v + b < LAST_CAP; b += WT) last[p_v + b] = data[so + b];
                wave_sync();
                // ... the value's first bytes from the LDS copy ...
                const uint64_t head = len < LAST_CAP ? len : LAST_CAP;
                for (uint64_t b = lane; b < head; b += WT) dst[oo + b] = last[b];
                // ... and the suffix bytes that did not pass through it
                if (len > LAST_CAP) {
                    const uint64_t from = p_v > LAST_CAP ? 0 : LAST_CAP - p_v; // first suffix byte not in `last`
                    for (uint64_t b = from + lane; b < s_v; b += WT) dst[oo + p_v + b] = data[so + b];
                }
                prev_len = len;
                prev_out = oo;
                wave_sync();
                taken += 1;
            }
        }
        // every pass of the round has read what it wanted from `stage`: in go the bytes fetched ahead
        if (pf_len > 0) {
#pragma unroll
            for (uint32_t u = 0; u < DPREF / (16 * WT); u++) {
                const uint64_t at = (uint64_t)(u * WT + lane) * 16;
                if (at < pf_len) *reinterpret_cast<u32x4*>(stage + at) = pf[u];
            }
            staged_base = next0;
            staged_len = pf_len;
            wave_sync();
        }
        if (m > 0) {
            src_base = lane_u64(s_off + s_pos, m - 1);
            out_base = lane_u64(o_off + p_pos + s_pos, m - 1);
        }
        done += m;
        if (status != PQD_OK) break;
    }
    if (status == PQD_ERR_DST_TOO_SMALL) {
        // report the whole size so that one retry suffices
        uint64_t need = 0;
        for (uint64_t k = lane; k < ns; k += WT) {
            const int32_t p = prefix[k], s = suffix[k];
            need += (uint64_t)(p > 0 ? p : 0) + (uint64_t)(s > 0 ? s : 0);
        }
        for (int d = 32; d > 0; d >>= 1) need += __shfl_down(need, d);
        arg = (int64_t)__shfl(need, 0);
    }
    if (lane == 0) {
        if (status == PQD_OK && with_offsets) {
            if (done + 1 > pg.offsets_cap) { status = PQD_ERR_OFFSETS_TOO_SMALL; arg = (int64_t)(done + 1); }
            else offsets[done] = (uint32_t)out_base;
        }
        finish(res, status, arg, done, out_base, status == PQD_OK ? rs.consumed + src_base : rs.consumed);
    }
}

// ---------------------------------------- DELTA_BYTE_ARRAY, block-parallel --
//
// The same decode (byte_array_purego.go:5-34) cut into blocks of 64 values that are decoded
// independently of one another.  Byte k of value i is a suffix byte of the latest value j <= i
// whose prefix length is <= k, so "what does a stretch of values do to the running value" is
// associative: it keeps the bytes below the least prefix length of the stretch and replaces the
// rest.  Three kernels:
//   dba_block_kernel   per block: suffix bytes, output bytes, least prefix length, validity;
//   dba_plan_kernel    per page: exclusive sums over the blocks, the page-level checks, and the
//                      decision whether the page takes this path (anything unusual -- an error,
//                      very long values -- is left to dba_kernel, which
//                      reports errors exactly as the sequential decoder does);
//   dba_expand_kernel  a wavefront per run of consecutive blocks, a lane per value: the bytes the
//                      first block of the run inherits from earlier values are looked up by
//                      walking the least prefix lengths of the blocks before it, those of the
//                      other blocks are the last value the wavefront has just built; the block's values are put together in an LDS image
//                      of its output bytes, dword column by dword column, the inherited bytes of
//                      a column travelling from lane to lane in a segmented wave scan; the image
//                      is copied out with aligned 16-byte stores.

constexpr uint32_t DBLK = 64;      // values per block
constexpr uint32_t DIMG = 8192;    // output bytes a block may produce on this path
constexpr uint32_t DCARRY = 1024;  // longest prefix length on this path
// Pages whose blocks all stay below these take the kernel instantiation with the small LDS image
// (more wavefronts per CU); one block above them and the page takes the large one.
constexpr uint32_t DIMG_SMALL = 5120, DCARRY_SMALL = 256;
constexpr uint32_t DBAD = 0x80000000u, DBIG = 0x40000000u;

// (by ds_bpermute, not DPP: the kernels that use it are bound by issuing vector instructions, and a
// permute goes down the LDS pipe instead)
__device__ inline uint32_t wave_min_u32(uint32_t v) {
    for (int d = 32; d > 0; d >>= 1) {
        const uint32_t o = (uint32_t)__shfl_xor((int)v, d);
        v = o < v ? o : v;
    }
    return v;
}

// (FIXED_LEN_BYTE_ARRAY pages are the same decode without offsets: byte_array_purego.go:36-63 checks what
// decodeByteArray checks, and nothing about the type length but its range, byte_array.go:149-151)
__device__ inline bool dba_parallel_candidate(const DevPage& pg, const DevResult& rp, const DevResult& rs) {
    const bool with_offsets = pg.kind == PQD_KIND_BYTE_ARRAY;
    const bool fixed = pg.kind == PQD_KIND_FIXED_LEN_BYTE_ARRAY && pg.bit_width >= 0 && (uint32_t)pg.bit_width <= MAX_FLBA;
    return (with_offsets || fixed) && pg.tile_run != nullptr && rp.status == PQD_OK && rs.status == PQD_OK &&
           rp.out_values == rs.out_values && rs.out_values > 0 && rs.out_values <= pg.lengths_cap &&
           (!with_offsets || rs.out_values + 1 <= pg.offsets_cap) && rs.out_values < 0x7fffffffull;
}

// grid.x = pages, grid.y = slots; a wavefront takes a block of 64 values at a time.
__global__ __launch_bounds__(MT) void dba_block_kernel(const DevPage* __restrict__ pages, DevResult* __restrict__ results,
                                                       const uint32_t* __restrict__ items) {
    const uint32_t item = items[blockIdx.x];
    const DevPage& pg = pages[item];
    const int32_t s_item = pg.chain, p_item = pages[s_item].chain;
    const DevResult rp = results[p_item], rs = results[s_item];
    if (!dba_parallel_candidate(pg, rp, rs)) return;
    const uint64_t ns = rs.out_values, nblk = (ns + DBLK - 1) / DBLK;
    gptr<const int32_t> prefix = (gptr<const int32_t>)as_global(pg.lengths);
    gptr<const int32_t> suffix = (gptr<const int32_t>)as_global(pg.lengths2);
    gptr<uint32_t> bstat = as_global(pg.tile_run);
    const uint32_t lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // four blocks per wavefront and trip, their loads in flight together
    constexpr uint32_t UN = 4;
    for (uint64_t blk0 = ((uint64_t)blockIdx.y * (MT / 64) + wave) * UN; blk0 < nblk; blk0 += (uint64_t)gridDim.y * (MT / 64) * UN) {
        int32_t pv[UN], sv[UN];
        uint32_t bv[UN];
#pragma unroll
        for (uint32_t u = 0; u < UN; u++) {
            const uint64_t k = (blk0 + u) * DBLK + lane;
            const bool valid = k < ns;
            pv[u] = valid ? prefix[k] : 0;
            sv[u] = valid ? suffix[k] : 0;
            bv[u] = 0;
            if (lane == 0 && valid && k > 0) bv[u] = (uint32_t)prefix[k - 1] + (uint32_t)suffix[k - 1];
        }
#pragma unroll
        for (uint32_t u = 0; u < UN; u++) {
            const uint64_t blk = blk0 + u;
            if (blk >= nblk) break;
            const uint64_t k = blk * DBLK + lane;
            const bool valid = k < ns;
            const int32_t p = pv[u], s = sv[u];
            bool bad = p < 0 || s < 0 || (uint32_t)p > DCARRY || (uint32_t)s > DIMG;
            const uint32_t len = (uint32_t)p + (uint32_t)s;
            uint32_t before = dpp0<0x138, 0xf>(len); // wave_shr:1
            if (lane == 0) before = bv[u];
            bad = bad || (valid && (uint32_t)p > before); // byte_array_purego.go:22-24
            const uint32_t sum_s = lane_u32(wave_inclusive_scan<uint32_t>(bad ? 0u : (uint32_t)s), 63);
            const uint32_t sum_l = lane_u32(wave_inclusive_scan<uint32_t>(bad ? 0u : len), 63);
            const uint32_t least = wave_min_u32(valid ? (uint32_t)p : 0xffffffffu);
            const bool any_bad = __ballot(bad) != 0 || sum_l > DIMG;
            const bool big = sum_l + 16 > DIMG_SMALL || __ballot((uint32_t)p > DCARRY_SMALL) != 0;
            if (lane == 0) {
                bstat[3 * blk] = sum_s;
                bstat[3 * blk + 1] = sum_l;
                bstat[3 * blk + 2] = any_bad ? DBAD : (least | (big ? DBIG : 0u));
            }
        }
    }
}

// One workgroup per page: block sums become block starts, and the page is either finished here
// (aux = 1 or 2: dba_expand_kernel with the small or the large image writes its bytes) or left to
// dba_kernel (aux = 0).
__global__ __launch_bounds__(MT) void dba_plan_kernel(const DevPage* __restrict__ pages, DevResult* __restrict__ results,
                                                      const uint32_t* __restrict__ items) {
    __shared__ uint64_t ws[MT / 64], wo[MT / 64];
    const uint32_t item = items[blockIdx.x];
    const DevPage& pg = pages[item];
    DevResult* res = &results[item];
    const int32_t s_item = pg.chain, p_item = pages[s_item].chain;
    const DevResult rp = results[p_item], rs = results[s_item];
    const uint32_t tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (!dba_parallel_candidate(pg, rp, rs)) {
        if (tid == 0) res->aux = 0;
        return;
    }
    const uint64_t ns = rs.out_values, nblk = (ns + DBLK - 1) / DBLK;
    gptr<uint32_t> bstat = as_global(pg.tile_run);
    uint64_t carry_s = 0, carry_o = 0;
    bool bad = false, big = false;
    for (uint64_t base = 0; base < nblk; base += MT) {
        const uint64_t b = base + tid;
        const bool v = b < nblk;
        const uint64_t s = v ? bstat[3 * b] : 0, o = v ? bstat[3 * b + 1] : 0;
        const uint32_t least = v ? bstat[3 * b + 2] : 0;
        bad = bad || (least & DBAD) != 0;
        big = big || (least & DBIG) != 0;
        const uint64_t xs = wave_inclusive_scan<uint64_t>(s), xo = wave_inclusive_scan<uint64_t>(o);
        if (lane == 63) { ws[wave] = xs; wo[wave] = xo; }
        __syncthreads();
        uint64_t before_s = carry_s, before_o = carry_o, all_s = 0, all_o = 0;
        for (uint32_t w = 0; w < MT / 64; w++) {
            if (w < wave) { before_s += ws[w]; before_o += wo[w]; }
            all_s += ws[w];
            all_o += wo[w];
        }
        if (v) {
            bstat[3 * b] = (uint32_t)(before_s + xs - s);
            bstat[3 * b + 1] = (uint32_t)(before_o + xo - o);
        }
        carry_s += all_s;
        carry_o += all_o;
        __syncthreads();
    }
    const bool any_bad = __syncthreads_or(bad ? 1 : 0) != 0;
    const bool any_big = __syncthreads_or(big ? 1 : 0) != 0;
    if (tid == 0) {
        const uint64_t data_total = pg.src_len - rs.consumed;
        const bool ok = !any_bad && carry_s <= data_total && carry_o <= pg.dst_cap && carry_o < 0xffffffffull;
        res->aux = ok ? (any_big ? 2 : 1) : 0;
        if (ok) finish(res, PQD_OK, 0, ns, carry_o, rs.consumed + carry_s);
    }
}

// Four bytes of the data section at any offset; the last bytes of the section one at a time.
__device__ inline uint32_t data_u32(gptr<const uint8_t> data, uint64_t at, uint64_t total) {
    if (at + 4 <= total) return *(gptr<const u32_a1>)(data + at);
    uint32_t w = 0;
    for (uint32_t e = 0; at + e < total; e++) w |= (uint32_t)data[at + e] << (8 * e);
    return w;
}

template <uint32_t IMG, uint32_t CARRY, uint32_t CLASS>
__global__ __launch_bounds__(WT) void dba_expand_kernel(const DevPage* __restrict__ pages, DevResult* __restrict__ results,
                                                        const uint32_t* __restrict__ items) {
    __shared__ __attribute__((aligned(16))) uint8_t img[IMG + 48];
    __shared__ __attribute__((aligned(16))) uint8_t carry[CARRY + 16];
    const uint32_t item = items[blockIdx.x];
    if (results[item].aux != CLASS) return;
    const DevPage& pg = pages[item];
    const int32_t s_item = pg.chain;
    const DevResult rs = results[s_item];
    const uint32_t lane = threadIdx.x;
    const uint64_t ns = rs.out_values, nblk = (ns + DBLK - 1) / DBLK;
    gptr<const int32_t> prefix = (gptr<const int32_t>)as_global(pg.lengths);
    gptr<const int32_t> suffix = (gptr<const int32_t>)as_global(pg.lengths2);
    gptr<const uint32_t> bstat = (gptr<const uint32_t>)as_global(pg.tile_run);
    gptr<const uint8_t> data = as_global(pg.src) + rs.consumed;
    const uint64_t data_total = pg.src_len - rs.consumed;
    gptr<uint8_t> dst = as_global(pg.dst);
    gptr<uint32_t> offsets = as_global(pg.offsets);
    const bool with_offsets = pg.kind == PQD_KIND_BYTE_ARRAY;
    uint32_t* img32 = reinterpret_cast<uint32_t*>(img);
    const uint32_t* carry32 = reinterpret_cast<const uint32_t*>(carry);

    // A wavefront takes blocks that follow one another: only for the first does it look the inherited
    // bytes up in the blocks before; after that they are the last value of the image it has just built.
    const uint64_t per = (nblk + gridDim.y - 1) / gridDim.y;
    const uint64_t b_lo = blockIdx.y * per, b_hi = b_lo + per < nblk ? b_lo + per : nblk;
    // lengths and sums of the block to come, asked for a block ahead
    uint32_t p_n = 0, s_n = 0, bs_n = 0, bo_n = 0;
    if (b_lo < b_hi) {
        const uint64_t k0 = b_lo * DBLK + lane;
        if (k0 < ns) { p_n = (uint32_t)prefix[k0]; s_n = (uint32_t)suffix[k0]; }
        bs_n = bstat[3 * b_lo];
        bo_n = bstat[3 * b_lo + 1];
    }
    for (uint64_t blk = b_lo; blk < b_hi; blk++) {
        const bool inherited = blk != b_lo; // carry[] holds the value before this block
        const uint64_t k = blk * DBLK + lane;
        const bool valid = k < ns;
        const uint32_t p = p_n, s = s_n;
        const uint32_t len = p + s;
        const uint32_t base_s = bs_n, base_o = bo_n;
        // Asked for now, used by the look-up of the inherited bytes below, so that it does not start
        // with two more trips to memory: the least prefix lengths of the 64 blocks before this one,
        // and the lengths of the block just before it (last value first).
        const uint32_t bm0 = !inherited && (uint64_t)lane < blk ? bstat[3 * (blk - 1 - lane) + 2] & ~DBIG : 0xffffffffu;
        uint32_t pj0 = 0, sj0 = 0;
        if (blk > 0 && !inherited) {
            pj0 = (uint32_t)prefix[(blk - 1) * DBLK + (63 - lane)];
            sj0 = (uint32_t)suffix[(blk - 1) * DBLK + (63 - lane)];
        }
        const uint32_t xs = wave_inclusive_scan<uint32_t>(s), xo = wave_inclusive_scan<uint32_t>(len);
        const uint32_t block_bytes = lane_u32(xo, 63);
        const uint32_t s_at = base_s + xs - s;  // where this value's suffix bytes start in the data section
        const uint32_t o_rel = xo - len;        // where the value starts in the block's output
        if (with_offsets) {
            if (valid) offsets[k] = base_o + o_rel;
            if (k + 1 == ns) offsets[ns] = base_o + xo;
        }
        const uint32_t align = (uint32_t)((uintptr_t)(dst + base_o) & 15);
        const uint32_t q = align + o_rel;       // the value's place in the image
        // the first twelve suffix bytes of every value, in flight while the inherited bytes are looked up
        uint32_t w_first[3];
#pragma unroll
        for (uint32_t u = 0; u < 3; u++) w_first[u] = 4 * u < s ? data_u32(data, (uint64_t)s_at + 4 * u, data_total) : 0u;
        if (blk + 1 < b_hi) {
            const uint64_t kn = k + DBLK;
            p_n = kn < ns ? (uint32_t)prefix[kn] : 0u;
            s_n = kn < ns ? (uint32_t)suffix[kn] : 0u;
            bs_n = bstat[3 * (blk + 1)];
            bo_n = bstat[3 * (blk + 1) + 1];
        }

        // ---- what the block inherits: bytes [0, need) of the value before it ----
        const uint32_t need = lane_u32(p, 0);
        {
            uint32_t r = inherited ? 0u : need; // bytes [0, r) are still to be found
            uint64_t cur = blk;    // among the blocks before `cur`
            while (r > 0 && cur > 0) {
                const bool have = (uint64_t)lane < cur;
                uint32_t bm = bm0;
                if (cur != blk) bm = have ? bstat[3 * (cur - 1 - lane) + 2] & ~DBIG : 0xffffffffu;
                const unsigned long long hit = __ballot(have && bm < r);
                if (hit == 0) { cur = cur > 64 ? cur - 64 : 0; continue; }
                const uint32_t first = (uint32_t)__builtin_ctzll(hit);
                const uint64_t b2 = cur - 1 - first;
                // lanes take the values of that block last to first
                uint32_t pj = pj0, sj = sj0, next_s = base_s;
                if (b2 + 1 != blk) {
                    const uint64_t kk = b2 * DBLK + (63 - lane);
                    pj = (uint32_t)prefix[kk];
                    sj = (uint32_t)suffix[kk];
                    next_s = bstat[3 * (b2 + 1)];
                }
                uint32_t incl = pj; // least prefix length among this value and the later ones
                for (int d = 1; d < 64; d <<= 1) {
                    const uint32_t o = (uint32_t)__shfl_up((int)incl, d);
                    if ((int)lane >= d && o < incl) incl = o;
                }
                uint32_t later = (uint32_t)__shfl_up((int)incl, 1);
                if (lane == 0) later = 0xffffffffu;
                const uint32_t rj = later < r ? later : r;   // the value supplies bytes [pj, rj)
                const uint32_t sj_at = next_s - wave_inclusive_scan<uint32_t>(sj);
                unsigned long long cm = __ballot(pj < rj);
                while (cm) {
                    const uint32_t j = (uint32_t)__builtin_ctzll(cm);
                    cm &= cm - 1;
                    const uint32_t lo = lane_u32(pj, j), hi = lane_u32(rj, j), at = lane_u32(sj_at, j);
                    for (uint32_t kb = lo + lane; kb < hi; kb += WT) carry[kb] = data[(uint64_t)at + (kb - lo)];
                }
                const uint32_t least = lane_u32(bm, first);
                r = least < r ? least : r;
                cur = b2;
            }
        }

        // ---- suffix bytes into their places in the image ----
#pragma unroll
        for (uint32_t u = 0; u < 3; u++) {
            const uint32_t t = 4 * u;
            if (t < s) {
                const uint32_t n = s - t < 4 ? s - t : 4;
                for (uint32_t e = 0; e < n; e++) img[q + p + t + e] = (uint8_t)(w_first[u] >> (8 * e));
            }
        }
        for (uint32_t t = 12; __ballot(t < s) != 0; t += 4) {
            if (t < s) {
                const uint32_t w = data_u32(data, (uint64_t)s_at + t, data_total);
                const uint32_t n = s - t < 4 ? s - t : 4;
                for (uint32_t e = 0; e < n; e++) img[q + p + t + e] = (uint8_t)(w >> (8 * e));
            }
        }
        wave_sync();

        // ---- inherited bytes, a dword column at a time ----
        uint32_t longest = p;
        for (int d = 32; d > 0; d >>= 1) {
            const uint32_t o = (uint32_t)__shfl_xor((int)longest, d);
            longest = o > longest ? o : longest;
        }
        // columns that lie below every value's prefix length hold the inherited bytes as they are
        const uint32_t plain = wave_min_u32(valid ? p : 0xffffffffu) / 4;
        for (uint32_t c = 0; c < plain; c++) {
            const uint32_t cw = carry32[c];
            if (valid) {
                img[q + 4 * c] = (uint8_t)cw;
                img[q + 4 * c + 1] = (uint8_t)(cw >> 8);
                img[q + 4 * c + 2] = (uint8_t)(cw >> 16);
                img[q + 4 * c + 3] = (uint8_t)(cw >> 24);
            }
        }
        for (uint32_t c = plain; 4 * c < longest; c++) {
            uint32_t at = q + 4 * c;
            at = at < IMG + 32 ? at : IMG + 32;
            uint32_t v = lds_u32(img32, at);
            const int32_t rel = (int32_t)p - (int32_t)(4 * c); // bytes of this dword below the prefix length
            uint32_t m = rel <= 0 ? 0xffffffffu : (rel >= 4 ? 0u : 0xffffffffu << (8 * rel));
            // inclusive scan: a lane takes from the lanes before it the bytes it does not define itself
#define PQD_DBA_STEP(CTRL, ROWS)                                   \
            {                                                          \
                const uint32_t vl = dpp0<CTRL, ROWS>(v), ml = dpp0<CTRL, ROWS>(m); \
                v = (v & m) | (vl & ~m);                               \
                m |= ml;                                               \
            }
            PQD_DBA_STEP(0x111, 0xf)
            PQD_DBA_STEP(0x112, 0xf)
            PQD_DBA_STEP(0x114, 0xf)
            PQD_DBA_STEP(0x118, 0xf)
            PQD_DBA_STEP(0x142, 0xa)
            PQD_DBA_STEP(0x143, 0xc)
#undef PQD_DBA_STEP
            const uint32_t cw = carry32[c];
            const uint32_t w = (v & m) | (cw & ~m);
            {
                // the bytes of this dword below the prefix length; the others go to a spare corner of the image (four
                // writes without a branch: four branches cost more than the writes)
                const uint32_t n = rel <= 0 ? 0u : (rel < 4 ? (uint32_t)rel : 4u);
                const uint32_t to = q + 4 * c, spare = IMG + 44;
#pragma unroll
                for (uint32_t e = 0; e < 4; e++) img[e < n ? to + e : spare] = (uint8_t)(w >> (8 * e));
            }
        }
        wave_sync();

        // ---- the image goes out: image and destination agree modulo 16 ----
        {
            gptr<uint8_t> out = dst + base_o;
            const uint32_t head = ((16 - align) & 15) < block_bytes ? ((16 - align) & 15) : block_bytes;
            const uint32_t body = (block_bytes - head) / 16;
            const uint32_t tail0 = head + body * 16;
            if (lane < head) out[lane] = img[align + lane];
            if (tail0 + lane < block_bytes) out[tail0 + lane] = img[align + tail0 + lane];
            for (uint32_t c = lane; c < body; c += WT)
                *(gptr<u32x4>)(out + head + c * 16) = *reinterpret_cast<const u32x4*>(img + align + head + c * 16);
        }
        // the block's last value, for the block after it
        if (blk + 1 < b_hi) {
            const uint32_t q_last = lane_u32(q, 63), len_last = lane_u32(len, 63);
            const uint32_t nb = len_last < CARRY ? len_last : CARRY;
            for (uint32_t c = lane * 4; c < nb; c += WT * 4) *reinterpret_cast<uint32_t*>(carry + c) = lds_u32(img32, q_last + c);
        }
        wave_sync();
    }
}

// Wrapping int32 sum of a decoded length stream (delta.go:28-33), for
// EstimateDecodeByteArraySize (length_byte_array.go:65-70, byte_array.go:176-183).
__global__ __launch_bounds__(MT) void sum_lengths_kernel(const int32_t* __restrict__ v, uint64_t n, uint32_t* __restrict__ out) {
    __shared__ uint32_t part[MT / 64];
    uint32_t s = 0;
    for (uint64_t k = threadIdx.x; k < n; k += MT) s += (uint32_t)v[k];
    for (int d = 32; d > 0; d >>= 1) s += __shfl_down(s, d);
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t t = 0;
        for (int w = 0; w < MT / 64; w++) t += part[w];
        *out += t;
    }
}

uint32_t tile_slots(uint64_t max_units, uint64_t per_tile) {
    uint64_t t = (max_units + per_tile - 1) / per_tile;
    if (t < 1) t = 1;
    return (uint32_t)(t > 4096 ? 4096 : t);
}

} // namespace

void launch_bitpacked(const DevPage* pages, DevResult* results, const uint32_t* items, uint32_t n_items, uint64_t max_values,
                      hipStream_t stream) {
    if (n_items == 0) return;
    hipLaunchKernelGGL(bitpacked_kernel, dim3(n_items, tile_slots(max_values, MT * 16 * BP_UNROLL)), dim3(MT), 0, stream, pages, results, items);
}

void launch_bss(const DevPage* pages, DevResult* results, const uint32_t* items, uint32_t n_items, uint64_t max_bytes,
                hipStream_t stream) {
    if (n_items == 0) return;
    hipLaunchKernelGGL(bss_kernel, dim3(n_items, tile_slots(max_bytes / 4, MT * 4 * BSS_UNROLL)), dim3(MT), 0, stream, pages, results, items);
}

void launch_plain_fixed(const DevPage* pages, DevResult* results, const uint32_t* items, uint32_t n_items, uint64_t max_bytes,
                        hipStream_t stream) {
    if (n_items == 0) return;
    hipLaunchKernelGGL(plain_fixed_kernel, dim3(n_items, tile_slots(max_bytes, MT * 64)), dim3(MT), 0, stream, pages, results, items);
}

void launch_plain_byte_array_walk(const DevPage* pages, DevResult* results, const uint32_t* items, uint32_t n_items,
                                  hipStream_t stream) {
    if (n_items == 0) return;
    hipLaunchKernelGGL(plain_byte_array_kernel, dim3(n_items), dim3(WT), 0, stream, pages, results, items);
}

void launch_dlba(const DevPage* pages, DevResult* results, const uint32_t* items, uint32_t n_items, uint64_t max_bytes,
                 hipStream_t stream) {
    if (n_items == 0) return;
    hipLaunchKernelGGL(dlba_offsets_kernel, dim3(n_items), dim3(MT), 0, stream, pages, results, items);
    hipLaunchKernelGGL(dlba_copy_kernel, dim3(n_items, tile_slots(max_bytes, MT * 64)), dim3(MT), 0, stream, pages, results, items);
}

void launch_dba(const DevPage* pages, DevResult* results, const uint32_t* items, uint32_t n_items, uint64_t max_values,
                hipStream_t stream) {
    if (n_items == 0) return;
    const uint64_t blocks = (max_values + DBLK - 1) / DBLK;
    // enough workgroups to fill the chip several times over, not one per block: the workgroups of
    // the instantiation a page does not use only look at its result and leave
    auto slots = [&](uint64_t units, uint64_t want_groups) {
        uint64_t per_page = std::max<uint64_t>(1, want_groups / n_items);
        return (uint32_t)std::min<uint64_t>(std::max<uint64_t>(units, 1), std::min<uint64_t>(per_page, 4096));
    };
    hipLaunchKernelGGL(dba_block_kernel, dim3(n_items, slots((blocks + 15) / 16, 4096)), dim3(MT), 0, stream, pages, results, items);
    hipLaunchKernelGGL(dba_plan_kernel, dim3(n_items), dim3(MT), 0, stream, pages, results, items);
    hipLaunchKernelGGL((dba_expand_kernel<DIMG_SMALL, DCARRY_SMALL, 1>), dim3(n_items, slots(blocks, 16384)), dim3(WT), 0, stream, pages,
                       results, items);
    hipLaunchKernelGGL((dba_expand_kernel<DIMG, DCARRY, 2>), dim3(n_items, slots(blocks, 16384)), dim3(WT), 0, stream, pages, results,
                       items);
    hipLaunchKernelGGL(dba_kernel, dim3(n_items), dim3(WT), 0, stream, pages, results, items);
}

void launch_sum_lengths(const int32_t* v, uint64_t n, uint32_t* out, hipStream_t stream) {
    hipLaunchKernelGGL(sum_lengths_kernel, dim3(1), dim3(MT), 0, stream, v, n, out);
}

} // namespace pqd
