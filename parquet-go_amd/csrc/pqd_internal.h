// pqd_internal.h -- device-side descriptors shared by the kernels and the host API.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <mutex>
#include <vector>

#include "../../include/pqd.h"

namespace pqd { struct Workspace; }

#define PQD_HIP(call)                                   \
    do {                                                \
        hipError_t e_ = (call);                         \
        if (e_ != hipSuccess) return PQD_ERR_DEVICE;    \
    } while (0)

// Opaque handles of the C ABI.
struct pqd_ctx {
    int device;
    hipStream_t side[3]; // small batches run their kernel families side by side on these (null: not created)
    uint32_t n_cus;       // compute units of the device
    int32_t dict_lds;     // dictionary pages through an LDS-resident dictionary: 0 never, 1 when it pays, 2 whenever possible
    uint64_t dict_lds_max_bytes; // dictionaries above this size never take that path
    uint64_t dict_lds_cap_bytes; // at most this much of a dictionary is kept in LDS (0: all that fits)
    uint32_t dict_slices;        // most workgroups one dictionary is sliced over (below 2: never)
    uint64_t dict_slice_bytes;   // bytes of a slice (0: the default)
    int32_t plain_tiled;         // PLAIN byte arrays by many workgroups per page (0: one wavefront per page)
    int32_t rle_page;            // level pages by a workgroup each, index and expansion in one launch (0: never)
    uint32_t rle_page_threads;    // threads of a level page's workgroup: 256 or 1024; 0: sized to the batch
    uint32_t rle_page_wave_steps; // such pages of at most this many steps are taken by a single wavefront
    std::mutex mu;                          // guards the list below
    std::vector<struct pqd::Workspace*> idle; // workspaces of the host entry points that no call is using
    void* levels_scratch; // this device's copy of the static workspace of pqd_optional_expand_noalloc (null: look-up failed)
    // page-locked host memory the caller has registered (pqd_host_register) or taken from the library (pqd_host_alloc):
    // the host entry points copy straight out of and into it
    struct HostRange { uintptr_t first; uint64_t second; bool owned; }; // first byte, bytes, allocated by pqd_host_alloc
    std::mutex reg_mu;
    std::vector<HostRange> registered;
    bool host_registered(const void* p, uint64_t bytes) {
        if (p == nullptr || bytes == 0) return false;
        std::lock_guard<std::mutex> lock(reg_mu);
        const uintptr_t a = reinterpret_cast<uintptr_t>(p);
        for (const auto& r : registered)
            if (a >= r.first && bytes <= r.second && a - r.first <= r.second - bytes) return true;
        return false;
    }
};

struct pqd_dict {
    pqd_ctx* ctx;
    void* values;   // device copy of the dictionary values
    uint64_t n;
    uint32_t width;
};

namespace pqd {

// Makes a context's device the calling thread's current device for the length of an entry
// point and puts the caller's back afterwards: a process that drives several GPUs from
// threads that move between them must not launch on, or allocate from, whichever device
// the thread happened to use last.
struct DeviceGuard {
    int prev = -1;
    bool ok = false;
    explicit DeviceGuard(int device) {
        if (hipGetDevice(&prev) != hipSuccess) prev = -1;
        ok = prev == device || hipSetDevice(device) == hipSuccess;
        if (prev == device) prev = -1; // nothing to restore
    }
    ~DeviceGuard() {
        if (prev >= 0) (void)hipSetDevice(prev);
    }
    DeviceGuard(const DeviceGuard&) = delete;
    DeviceGuard& operator=(const DeviceGuard&) = delete;
};

// One entry of a run table built by the RLE index kernel.
struct Run {
    uint32_t out_start;  // first output value of the run
    uint32_t in_off;     // byte offset in src of the packed data (bit-packed runs)
    uint32_t value;      // repeated value (RLE runs)
    uint32_t count_kind; // count << 1 | bitpacked
};

// Device copy of a work item plus the scratch the batch assigned to it.
struct DevPage {
    const uint8_t* src;
    uint64_t src_len;
    uint8_t* dst;
    uint64_t dst_cap;
    uint32_t* offsets;
    uint64_t offsets_cap;
    const uint8_t* dict;
    uint64_t dict_len;
    const uint32_t* dict_offsets;
    int64_t num_values;
    int32_t encoding;
    int32_t kind;
    int32_t bit_width;
    int32_t dst_stride;
    // scratch
    Run* runs;          // RLE run table
    uint64_t runs_cap;
    int32_t* lengths;   // DELTA length streams (prefix / lengths)
    int32_t* lengths2;  // DELTA_BYTE_ARRAY suffix lengths
    uint64_t lengths_cap; // entries each
    uint64_t tile_base;   // first tile of this page in its family's tile grid
    uint32_t* tile_run;   // RLE: per step of RLE_TILE values, the run-table entry holding its first value;
                          // DELTA_BYTE_ARRAY: three words per block of 64 values (suffix bytes, output bytes, least prefix)
    uint32_t n_steps;     // RLE: steps of RLE_TILE values this page may produce (bound known on the host)
    uint32_t group;       // RLE: steps one expand workgroup takes (several for 1-byte elements)
    int32_t elem_width;   // output element width in bytes
    int32_t family;       // kernel family
    int32_t chain;        // index of the item whose `consumed` gives this item's src offset, or -1
    int32_t fused;        // RLE family: indices go straight through the dictionary gather
    int32_t ck_index;     // RLE family: position in the list of pages indexed chunk by chunk, or -1
    // DELTA families, batches of few pages: where each block of 128 values starts (flat_cap + 1
    // entries) and one sum per tile of 1024 values, for the tile-parallel kernels; null otherwise
    uint32_t* flat_pos;
    uint64_t* flat_sum;
    uint64_t flat_cap;    // blocks
    int32_t flat_off;     // nonzero: never on the tile-parallel path (length streams of DELTA_BYTE_ARRAY: widths change all the time)
    int32_t reserved;
};

struct DevResult {
    int32_t status;
    int32_t bw;        // RLE: effective bit width (first byte of a dictionary-index page)
    int64_t arg;
    uint64_t out_values;
    uint64_t out_bytes;
    uint64_t consumed;
    uint64_t n_runs;   // RLE: runs in the table
    uint64_t total;    // RLE: values the stream holds; DELTA: header total
    uint64_t aux;      // family specific
    uint32_t first[4]; // RLE: the first entry of the run table (a Run), zero when the table is empty
    // dictionary pages through an LDS-resident dictionary (their index and their decode run in one launch and
    // write disjoint fields):
    uint64_t need_bytes; // destination bytes the page asks for when dst_cap is too small, else 0
    unsigned long long bad; // launch number << 32 | ~(first position whose index lies outside the dictionary)
};

enum Family : int32_t {
    FAM_NONE = 0,
    FAM_DELTA32 = 1,
    FAM_DELTA64 = 2,
    FAM_RLE = 3,        // RLE hybrid to 4-byte values or through a dictionary of 12-byte values
    FAM_RLE_BYTES = 4,  // RLE hybrid to one byte per element: levels, booleans
    FAM_BITPACKED = 5,
    FAM_BSS = 6,
    FAM_PLAIN_FIXED = 7,
    FAM_PLAIN_BYTE_ARRAY = 8,
    FAM_DLBA = 9,
    FAM_DBA = 10,
    FAM_ERROR = 11,     // rejected on the host; status already set
    FAM_DELTA32_CHAINED = 12, // a length stream that starts where another item's stream ended
    FAM_RLE_DICT8 = 13, // RLE hybrid through a dictionary of 8-byte values
    FAM_RLE_DICT4 = 14, // ... of 4-byte values
    FAM_RLE_LDS8 = 15,  // RLE hybrid through a dictionary of 8-byte values held in LDS, dense destination
    FAM_RLE_LDS4 = 16,  // ... of 4-byte values
    FAM_DICT_BYTES = 17, // look-up in a BYTE_ARRAY / FIXED_LEN_BYTE_ARRAY / BOOLEAN dictionary of indexes an RLE item decoded
    FAM_RLE_PAGE = 18,  // RLE hybrid levels, a workgroup per page: run headers followed and runs expanded in one launch
    N_FAMILIES = 20
};

// Pointers read out of a descriptor are generic to the compiler, which then
// emits FLAT loads and stores: those count on both the vector-memory and the
// LDS counter, so every LDS wait would also wait for outstanding global stores.
// Kernels cast descriptor pointers to the global address space before use.
template <typename T> using gptr = __attribute__((address_space(1))) T*;
template <typename T> __device__ inline gptr<T> as_global(T* p) { return (gptr<T>)p; }
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint64_t u64x2_a8 __attribute__((ext_vector_type(2), aligned(8)));
typedef uint32_t u32x4_a4 __attribute__((ext_vector_type(4), aligned(4)));
typedef uint32_t u32x2_a4 __attribute__((ext_vector_type(2), aligned(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));

// Cross-lane moves of the wave scan, done with DPP (no LDS round trip).  Lanes
// whose source lies outside the row, or whose row is masked off, read 0.
template <int CTRL, int ROW_MASK>
__device__ inline uint32_t dpp0(uint32_t v) {
    // all rows enabled: bound_ctrl makes out-of-row sources read 0, no `old` value to set up;
    // masked rows keep `old`, which then has to be 0
    if (ROW_MASK == 0xf) return (uint32_t)__builtin_amdgcn_update_dpp((int)v, (int)v, CTRL, 0xf, 0xf, true);
    return (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, CTRL, ROW_MASK, 0xf, false);
}

template <int CTRL, int ROW_MASK>
__device__ inline uint64_t dpp0(uint64_t v) {
    uint32_t lo = dpp0<CTRL, ROW_MASK>((uint32_t)v);
    uint32_t hi = dpp0<CTRL, ROW_MASK>((uint32_t)(v >> 32));
    return ((uint64_t)hi << 32) | lo;
}

// Inclusive prefix sum over the 64 lanes of a wave: Kogge-Stone inside each row
// of 16 lanes (row_shr 1,2,4,8), then the row totals are passed on with
// row_bcast:15 (rows 1 and 3) and row_bcast:31 (rows 2 and 3).
template <typename U>
__device__ inline U wave_inclusive_scan(U x) {
    x += dpp0<0x111, 0xf>(x);
    x += dpp0<0x112, 0xf>(x);
    x += dpp0<0x114, 0xf>(x);
    x += dpp0<0x118, 0xf>(x);
    x += dpp0<0x142, 0xa>(x);
    x += dpp0<0x143, 0xc>(x);
    return x;
}

// Greatest / least value over the 64 lanes, the same in every lane.  Where all lanes would otherwise send an
// atomic to one LDS or global address: the compiler turns that into a loop over the active lanes, half a dozen
// scalar instructions for each, on the one scalar unit the compute unit's wavefronts share.
__device__ inline uint32_t wave_greatest_u32(uint32_t x) {
    uint32_t o;
    o = dpp0<0x111, 0xf>(x); x = o > x ? o : x;
    o = dpp0<0x112, 0xf>(x); x = o > x ? o : x;
    o = dpp0<0x114, 0xf>(x); x = o > x ? o : x;
    o = dpp0<0x118, 0xf>(x); x = o > x ? o : x;
    o = dpp0<0x142, 0xa>(x); x = o > x ? o : x;
    o = dpp0<0x143, 0xc>(x); x = o > x ? o : x;
    return (uint32_t)__builtin_amdgcn_readlane((int)x, 63);
}
__device__ inline uint32_t wave_least_u32(uint32_t x) { return ~wave_greatest_u32(~x); }

// Value of a register in the lane a wave-uniform index names (v_readlane: no LDS round trip).
__device__ inline uint32_t lane_u32(uint32_t v, uint32_t lane) { return (uint32_t)__builtin_amdgcn_readlane((int)v, (int)lane); }
__device__ inline uint64_t lane_u64(uint64_t v, uint32_t lane) {
    return ((uint64_t)lane_u32((uint32_t)(v >> 32), lane) << 32) | lane_u32((uint32_t)v, lane);
}

// Failure bookkeeping: the first failing condition in stream order wins; a
// kernel records one with this helper from a single thread.
__device__ inline void set_status(DevResult* r, int32_t status, int64_t arg) {
    r->status = status;
    r->arg = arg;
}

// Device and pinned host memory one caller of the host entry points keeps from call to call,
// with a stream of its own: nothing is allocated per call once the buffers have grown to the
// sizes the caller's pages need.
struct Workspace {
    hipStream_t stream = nullptr;
    void* arena = nullptr;  size_t arena_cap = 0;   // the batch's arena
    void* pinned = nullptr; size_t pinned_cap = 0;  // host image of the arena's uploaded part; results read back
    void* d_io = nullptr;   size_t d_io_cap = 0;    // device staging of a page: source, destination, offsets
    void* h_io = nullptr;   size_t h_io_cap = 0;    // the same in pinned host memory
    static bool grow(void** p, size_t* cap, size_t need, bool host) {
        if (need <= *cap) return true;
        const size_t want = std::max(need + need / 2, (size_t)1 << 16);
        void* q = nullptr;
        if ((host ? hipHostMalloc(&q, want, hipHostMallocDefault) : hipMalloc(&q, want)) != hipSuccess) return false;
        if (*p) (void)(host ? hipHostFree(*p) : hipFree(*p));
        *p = q;
        *cap = want;
        return true;
    }
    bool reserve_arena(size_t n) { return grow(&arena, &arena_cap, n, false); }
    bool reserve_pinned(size_t n) { return grow(&pinned, &pinned_cap, n, true); }
    bool reserve_io(size_t n) { return grow(&d_io, &d_io_cap, n, false) && grow(&h_io, &h_io_cap, n, true); }
    void release() {
        if (arena) (void)hipFree(arena);
        if (d_io) (void)hipFree(d_io);
        if (pinned) (void)hipHostFree(pinned);
        if (h_io) (void)hipHostFree(h_io);
        if (stream) (void)hipStreamDestroy(stream);
        arena = d_io = pinned = h_io = nullptr;
        stream = nullptr;
    }
};
int batch_create_ws(pqd_ctx* ctx, const pqd_page* pages, uint64_t n_pages, pqd_batch** out, Workspace* ws);
int batch_finish_ws(pqd_batch* b, Workspace* ws, pqd_page* pages);
void batch_report(pqd_batch* b, const struct DevResult* results, pqd_page* pages);
Workspace* workspace_acquire(pqd_ctx* ctx); // a free workspace of the context (a new one when all are in use); nullptr on failure
void workspace_release(pqd_ctx* ctx, Workspace* ws);

// Kernel launchers (defined in the .hip files).
void launch_delta(const DevPage* pages, DevResult* results, const uint32_t* items, uint32_t n_items, bool is64,
                  uint64_t flat_max_blocks, hipStream_t stream);
constexpr size_t DELTA_FLAT_MAX_PAGES = 256; // families with more pages than this fill the chip with one workgroup per page
void launch_rle_index(const DevPage* pages, DevResult* results, const uint32_t* items, uint32_t n_items, const uint32_t* pk_flag,
                      hipStream_t stream);

// Device arrays of the chunked RLE index (long pages cut into chunks, one workgroup each).
// Workgroups take (page, chunk) pairs in the order of the `page` / `chunk` lists, which go
// round the pages chunk by chunk, so that all pages advance together; the per-chunk arrays
// are indexed by prefix[page] + chunk.
struct RleChunks {
    uint32_t n_chunks = 0;
    const uint32_t* items = nullptr;   // per chunked page: its index in the batch's item array
    const uint32_t* prefix = nullptr;  // per chunked page: its first place in the per-chunk arrays
    const uint32_t* page = nullptr;    // per work item: its page in the list of chunked pages
    const uint32_t* chunk = nullptr;   // per work item: its chunk of that page
    uint32_t* entry = nullptr;         // per chunk: where the chain of run headers enters it
    uint32_t* exit = nullptr;          // per chunk: where it leaves it
    uint16_t* links = nullptr;         // per chunk and byte: where the header after one starting there starts
    // cleared before every launch, one block of memory:
    void* state = nullptr;
    size_t state_bytes = 0;
    uint32_t* counters = nullptr;      // two: the order in which the workgroups of each kernel start
    uint32_t* pk_flag = nullptr;       // per chunked page: nonzero = left to the one-wavefront kernel
    uint32_t* page_pos = nullptr;      // per chunked page: how far the chain of run headers is known
    unsigned long long* runs = nullptr; // per chunk: ready bit | run-table entries it adds
    unsigned long long* vals = nullptr; // per chunk: ready bit | values it adds
};
uint32_t rle_chunk_bytes();
void launch_rle_chunked(const DevPage* pages, DevResult* results, const RleChunks& c, hipStream_t stream);
void launch_rle_expand(const DevPage* pages, DevResult* results, const uint32_t* items, uint32_t n_items,
                       const uint64_t* tile_prefix, const uint32_t* tile_item, uint64_t n_tiles, int32_t family, hipStream_t stream);
void launch_rle_pages(const DevPage* pages, DevResult* results, const uint32_t* items, uint32_t n_items, uint32_t n_small, uint32_t threads,
                      uint32_t* retry_word,
                      uint32_t launch, hipStream_t stream);

// Dictionary pages through an LDS-resident dictionary (persistent workgroups).
bool rle_dict_lds_init();            // false: the kernels cannot get their LDS; the path is not used
uint32_t rle_dict_lds_capacity();    // bytes of dictionary a workgroup can hold
struct LdsPlan {                     // launch shape of a family, fixed when the batch is made
    uint32_t grid = 0;               // persistent workgroups
    uint32_t lds_bytes = 0, dict_bytes = 0, tab_cap = 0;
    bool part = false;               // some dictionary of the family is larger than what LDS holds of it
    // slices > 1: the workgroups stand in `groups` groups of `slices` (grid = groups * slices), every workgroup of a
    // group holds slice_bytes of the dictionary and all of them walk the group's tiles; otherwise groups == grid
    uint32_t slices = 1, groups = 0, slice_bytes = 0;
};
// Dictionaries of more than *above and at most *upto bytes are sliced over a group of workgroups (*upto 0: none
// is); max_slices: most workgroups in a group (below 2: no slicing); slice_bytes_opt: bytes per slice (0: the default)
void rle_dict_slice_range(uint64_t cap_bytes, uint32_t n_cus, uint32_t max_slices, uint64_t slice_bytes_opt, uint64_t* above, uint64_t* upto);
// max_dict_bytes: the family's largest dictionary; sliced_dict_bytes: its largest inside the sliced range (0: none);
// all_sliced: every dictionary of the family lies in that range (only then is the family sliced)
LdsPlan rle_dict_lds_plan(uint64_t n_tiles, uint64_t max_dict_bytes, uint64_t sliced_dict_bytes, bool all_sliced, uint64_t cap_bytes,
                          uint32_t n_cus, uint64_t slice_bytes_opt);
// wg_item[2b], wg_item[2b + 1]: positions in the family list of the items that hold group b's first and last
// tile (tiles n_tiles * b / groups and n_tiles * (b + 1) / groups - 1); seg_last[i]: the last position from i on whose
// item shares item i's dictionary; pk_flag: the chunked index's per-page flags; launch: how many times the batch
// has been launched, this launch included
void launch_rle_dict_lds(const DevPage* pages, DevResult* results, const uint32_t* items, uint32_t n_items,
                         const uint64_t* tile_prefix, uint64_t n_tiles, int32_t elem, const LdsPlan& plan, const uint32_t* wg_item,
                         const uint32_t* seg_last, const uint32_t* pk_flag, uint32_t launch, hipStream_t stream);

void launch_bitpacked(const DevPage* pages, DevResult* results, const uint32_t* items, uint32_t n_items, uint64_t max_values,
                      hipStream_t stream);
void launch_bss(const DevPage* pages, DevResult* results, const uint32_t* items, uint32_t n_items, uint64_t max_bytes,
                hipStream_t stream);
void launch_plain_fixed(const DevPage* pages, DevResult* results, const uint32_t* items, uint32_t n_items, uint64_t max_bytes,
                        hipStream_t stream);
// PLAIN byte arrays: pages that were given scratch (plain_bytes_scratch_words(src_len) 32-bit words in
// DevPage::lengths) are decoded by many workgroups each, the others by one wavefront each; max_bytes is the
// longest page with scratch (0: none has any)
uint64_t plain_bytes_scratch_words(uint64_t src_len);
void launch_plain_byte_array(const DevPage* pages, DevResult* results, const uint32_t* items, uint32_t n_items, uint64_t max_bytes,
                             hipStream_t stream);
void launch_plain_byte_array_walk(const DevPage* pages, DevResult* results, const uint32_t* items, uint32_t n_items,
                                  hipStream_t stream);
void launch_dlba(const DevPage* pages, DevResult* results, const uint32_t* items, uint32_t n_items, uint64_t max_bytes,
                 hipStream_t stream);
void launch_dba(const DevPage* pages, DevResult* results, const uint32_t* items, uint32_t n_items, uint64_t max_values,
                hipStream_t stream);
void launch_dict_bytes(const DevPage* pages, DevResult* results, const uint32_t* items, uint32_t n_items, uint64_t max_values,
                       hipStream_t stream);
void launch_sum_lengths(const int32_t* v, uint64_t n, uint32_t* out, hipStream_t stream);

int levels_scratch_address(void** out); // device address of the static workspace, for the current device

constexpr uint32_t RLE_TILE = 2048; // output values per expand workgroup

} // namespace pqd
