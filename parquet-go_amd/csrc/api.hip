// api.hip -- host side of the C ABI declared in include/pqd.h: contexts,
// batches (classification of pages into kernel families, descriptor upload,
// scratch carving, launch order) and the per-method host-pointer wrappers.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "pqd_internal.h"

using namespace pqd;

namespace {

// One user page maps to `primary` (the item whose result is reported) plus,
// for the DELTA byte-array encodings, the length-stream items before it.
struct PageMap {
    int32_t primary;    // item index, or -1 when rejected on the host
    int32_t host_status;
    int64_t host_arg;
};

struct FamilyList {
    std::vector<uint32_t> items; // item indexes
    uint32_t* dev = nullptr;
};

} // namespace

struct pqd_batch {
    pqd_ctx* ctx = nullptr;
    uint64_t n_pages = 0;
    std::vector<PageMap> map;
    std::vector<DevPage> items;
    void* arena = nullptr;
    size_t arena_bytes = 0;
    bool own_arena = true; // false: the arena belongs to a caller's workspace
    bool plain_tiled = false; // PLAIN byte-array pages were given scratch for the many-workgroups-per-page kernels
    DevPage* d_items = nullptr;
    DevResult* d_results = nullptr;
    FamilyList fam[N_FAMILIES];
    // RLE expand grids, one per RLE hybrid family
    uint64_t* d_rle_tile_prefix[7] = {};
    uint32_t* d_rle_tile_item[7] = {}; // grid tile -> position in the family list (null: the kernel searches the prefix)
    uint64_t rle_tiles[7] = {};
    uint64_t lds_dict_bytes[7] = {}; // LDS families: the largest dictionary of the family, in bytes
    uint64_t lds_sliced_bytes[7] = {}; // ... the largest one that is to be sliced over a group of workgroups
    bool lds_unsliced[7] = {};       // ... some dictionary of the family is not to be sliced
    uint32_t launches = 0;           // times the batch has been launched
    LdsPlan lds_plan[7];             // ... their launch shape
    uint32_t* d_lds_wg_item[7] = {}; // ... the item each workgroup starts in
    uint32_t* d_lds_seg_last[7] = {}; // ... per item, the last one from there on with the same dictionary
    uint64_t rle_steps = 0;
    RleChunks ck; // long RLE pages indexed chunk by chunk
    // a small batch forks its independent kernel families onto the context's side streams
    bool side_by_side = false;
    hipEvent_t fork = nullptr, join[3] = {nullptr, nullptr, nullptr};
    // generic tile grids for the element-parallel families
    uint64_t max_src[N_FAMILIES] = {0}; // longest src of each family, sizes the tile grids
    uint32_t rle_page_small = 0;        // FAM_RLE_PAGE: the first so many items of the list are short pages (a wavefront each)
    uint32_t* d_page_retry = nullptr;   // FAM_RLE_PAGE: a word the page kernel raises to the launch's number when it leaves a page to the retry kernel
    uint64_t dict_bytes_max_values = 0; // most values a page looked up in a byte dictionary may hold
    uint64_t dba_max_values = 0; // most values a DELTA_BYTE_ARRAY page of the batch is expected to hold
    uint64_t flat_blocks[N_FAMILIES] = {0}; // DELTA families decoded tile by tile: most blocks a page may hold (0: not used)
    std::vector<DevResult> h_results;
    ~pqd_batch() {
        if (arena && own_arena) (void)hipFree(arena);
        if (fork) (void)hipEventDestroy(fork);
        for (hipEvent_t e : join)
            if (e) (void)hipEventDestroy(e);
    }
};

namespace {

inline bool hip_ok(hipError_t e) { return e == hipSuccess; }

inline size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

// Entries of an RLE page's run table: a run takes at least a byte of the page, and the run that holds value p
// is at most the p-th, so neither more than the page has bytes nor more than the values that can be written.
uint64_t run_table_entries(const DevPage& d) {
    const uint64_t stride = d.dst_stride > 0 ? (uint64_t)d.dst_stride : (uint64_t)(d.elem_width > 0 ? d.elem_width : 1);
    uint64_t values = d.dst_cap / stride + 1;
    if (d.num_values >= 0 && (uint64_t)d.num_values < values) values = (uint64_t)d.num_values;
    return std::min<uint64_t>(d.src_len, values) + 1;
}

int32_t elem_width_of(int32_t kind, int32_t bit_width) {
    switch (kind) {
        case PQD_KIND_LEVELS: return 1;
        case PQD_KIND_BOOLEAN: return 1;
        case PQD_KIND_INT32: return 4;
        case PQD_KIND_INT64: return 8;
        case PQD_KIND_INT96: return 12;
        case PQD_KIND_FLOAT: return 4;
        case PQD_KIND_DOUBLE: return 8;
        case PQD_KIND_BYTE_ARRAY: return 1;
        case PQD_KIND_FIXED_LEN_BYTE_ARRAY: return bit_width;
        default: return 0;
    }
}

// Which kernel family decodes (encoding, kind); FAM_NONE = not supported, as
// encoding.NotSupported would answer (encoding/notsupported.go:163-197).
// The RLE hybrid families share the index kernel; each has an expand kernel and a tile grid of its own.
constexpr size_t SIDE_BY_SIDE_MAX_ITEMS = 1024;     // batches up to this size fork their families onto side streams
// Level pages of at most this many steps of RLE_TILE values are decoded by a workgroup each (one workgroup writes
// about a kilobyte per 0.2 us: beyond, the tiles of the index + expand pair, spread over the chip, are quicker)
constexpr uint64_t RLE_PAGE_MAX_STEPS = 128;
// ... and longer ones whose bytes are few (long repeated runs: eight windows of the page kernel at most); what is slow
// about a long page there is its expansion by one compute unit, and a batch of few pages shares that out
constexpr uint64_t RLE_PAGE_MAX_BYTES = 32768;
constexpr uint64_t RLE_PAGE_MAX_STEPS_FEW_BYTES = 1024; // (two million levels: a hundred million nulls in one run are the tiles')
constexpr int RLE_GRIDS = 7;
static_assert(RLE_GRIDS == sizeof(pqd_batch::rle_tiles) / sizeof(uint64_t), "one tile grid per RLE hybrid family");
constexpr size_t RLE_CHUNKED_MAX_PAGES = 512;      // batches with more RLE pages than this index them one wavefront each
constexpr uint64_t RLE_CHUNKED_MIN_BYTES = 16384;   // shorter pages are not worth cutting up ...
constexpr uint64_t RLE_CHUNKED_MIN_BYTES_FEW = 4096; // ... unless the batch holds so few that wavefronts would idle
constexpr size_t RLE_CHUNKED_FEW_PAGES = 128;
constexpr int32_t RLE_FAMILY[RLE_GRIDS] = {FAM_RLE, FAM_RLE_BYTES, FAM_RLE_DICT8, FAM_RLE_DICT4, FAM_RLE_LDS8, FAM_RLE_LDS4, FAM_RLE_PAGE};
inline bool is_lds_family(int32_t family) { return family == FAM_RLE_LDS8 || family == FAM_RLE_LDS4; }
// Largest dictionary that goes through the persistent kernels by default (beyond what LDS holds, the first
// part of a dictionary is kept there and the rest read from memory: measured faster than the tile-per-workgroup
// kernel up to 8 MB dictionaries at least)
constexpr uint64_t LDS_DICT_DEFAULT_MAX_BYTES = 1ull << 30;
inline int rle_grid(int32_t family) {
    for (int g = 0; g < RLE_GRIDS; g++) if (RLE_FAMILY[g] == family) return g;
    return -1;
}
inline bool is_rle(int32_t family) { return rle_grid(family) >= 0; }

int32_t classify(const pqd_page& p) {
    const int32_t e = p.encoding, k = p.kind;
    if (k < PQD_KIND_LEVELS || k > PQD_KIND_FIXED_LEN_BYTE_ARRAY) return FAM_NONE;
    switch (e) {
        case PQD_ENC_DELTA_BINARY_PACKED:
            if (k == PQD_KIND_INT32) return FAM_DELTA32;
            if (k == PQD_KIND_INT64) return FAM_DELTA64;
            return FAM_NONE;
        case PQD_ENC_RLE:
            if (k == PQD_KIND_INT32) return FAM_RLE;
            if (k == PQD_KIND_LEVELS || k == PQD_KIND_BOOLEAN) return FAM_RLE_BYTES;
            return FAM_NONE;
        case PQD_ENC_RLE_DICTIONARY:
        case PQD_ENC_PLAIN_DICTIONARY:
            if (p.dict != nullptr || p.dict_len > 0) {
                if (k == PQD_KIND_INT32 || k == PQD_KIND_INT64 || k == PQD_KIND_FLOAT || k == PQD_KIND_DOUBLE ||
                    k == PQD_KIND_INT96)
                    return FAM_RLE;
                if (k == PQD_KIND_BYTE_ARRAY || k == PQD_KIND_FIXED_LEN_BYTE_ARRAY || k == PQD_KIND_BOOLEAN) return FAM_DICT_BYTES;
                return FAM_NONE;
            }
            // Without a dictionary the item stands for the codec object's own DecodeInt32:
            // rle.DictionaryEncoding reads a width byte and hybrid runs (rle/dictionary.go:30-37),
            // plain.DictionaryEncoding is a PLAIN decode of raw int32 (plain/dictionary.go:25-27; the
            // rewrite of PLAIN_DICTIONARY data pages to RLE_DICTIONARY is the caller's, column.go:850-857).
            if (k != PQD_KIND_INT32) return FAM_NONE;
            return e == PQD_ENC_PLAIN_DICTIONARY ? FAM_PLAIN_FIXED : FAM_RLE;
        case PQD_ENC_BIT_PACKED:
            return k == PQD_KIND_LEVELS ? FAM_BITPACKED : FAM_NONE;
        case PQD_ENC_BYTE_STREAM_SPLIT:
            if (k == PQD_KIND_INT32 || k == PQD_KIND_INT64 || k == PQD_KIND_FLOAT || k == PQD_KIND_DOUBLE ||
                k == PQD_KIND_FIXED_LEN_BYTE_ARRAY)
                return FAM_BSS;
            return FAM_NONE;
        case PQD_ENC_PLAIN:
            if (k == PQD_KIND_BYTE_ARRAY) return FAM_PLAIN_BYTE_ARRAY;
            if (k == PQD_KIND_LEVELS) return FAM_NONE;
            return FAM_PLAIN_FIXED;
        case PQD_ENC_DELTA_LENGTH_BYTE_ARRAY:
            return k == PQD_KIND_BYTE_ARRAY ? FAM_DLBA : FAM_NONE;
        case PQD_ENC_DELTA_BYTE_ARRAY:
            return (k == PQD_KIND_BYTE_ARRAY || k == PQD_KIND_FIXED_LEN_BYTE_ARRAY) ? FAM_DBA : FAM_NONE;
        default:
            return FAM_NONE;
    }
}

DevPage make_item(const pqd_page& p, int32_t family) {
    DevPage d;
    std::memset(&d, 0, sizeof d);
    d.src = static_cast<const uint8_t*>(p.src);
    d.src_len = p.src_len;
    d.dst = static_cast<uint8_t*>(p.dst);
    d.dst_cap = p.dst_cap;
    d.offsets = p.offsets;
    d.offsets_cap = p.offsets_cap;
    d.dict = static_cast<const uint8_t*>(p.dict);
    d.dict_len = p.dict_len;
    d.dict_offsets = p.dict_offsets;
    d.num_values = p.num_values;
    d.encoding = p.encoding;
    d.kind = p.kind;
    d.bit_width = p.bit_width;
    d.dst_stride = p.dst_stride;
    d.elem_width = elem_width_of(p.kind, p.bit_width);
    d.family = family;
    d.chain = -1;
    d.fused = 0;
    d.ck_index = -1;
    return d;
}

} // namespace

Workspace* pqd::workspace_acquire(pqd_ctx* ctx) {
    {
        std::lock_guard<std::mutex> lock(ctx->mu);
        if (!ctx->idle.empty()) {
            Workspace* ws = ctx->idle.back();
            ctx->idle.pop_back();
            return ws;
        }
    }
    Workspace* ws = new (std::nothrow) Workspace;
    if (ws == nullptr) return nullptr;
    if (hipStreamCreateWithFlags(&ws->stream, hipStreamNonBlocking) != hipSuccess) { delete ws; return nullptr; }
    return ws;
}

void pqd::workspace_release(pqd_ctx* ctx, Workspace* ws) {
    std::lock_guard<std::mutex> lock(ctx->mu);
    ctx->idle.push_back(ws);
}

extern "C" {

int pqd_abi_version(void) { return PQD_ABI_VERSION; }

const char* pqd_status_name(int status) {
    switch (status) {
        case PQD_OK: return "OK";
        case PQD_ERR_NOT_SUPPORTED: return "NOT_SUPPORTED";
        case PQD_ERR_INVALID_ARGUMENT: return "INVALID_ARGUMENT";
        case PQD_ERR_INVALID_BIT_WIDTH: return "INVALID_BIT_WIDTH";
        case PQD_ERR_UNEXPECTED_EOF: return "UNEXPECTED_EOF";
        case PQD_ERR_VARINT_OVERFLOW: return "VARINT_OVERFLOW";
        case PQD_ERR_RUN_TOO_LARGE: return "RUN_TOO_LARGE";
        case PQD_ERR_BAD_DELTA_HEADER: return "BAD_DELTA_HEADER";
        case PQD_ERR_FIRST_VALUE_RANGE: return "FIRST_VALUE_RANGE";
        case PQD_ERR_MISSING_VALUES: return "MISSING_VALUES";
        case PQD_ERR_NEGATIVE_LENGTH: return "NEGATIVE_LENGTH";
        case PQD_ERR_LENGTH_OUT_OF_BOUNDS: return "LENGTH_OUT_OF_BOUNDS";
        case PQD_ERR_PREFIX_SUFFIX_MISMATCH: return "PREFIX_SUFFIX_MISMATCH";
        case PQD_ERR_NEGATIVE_PREFIX: return "NEGATIVE_PREFIX";
        case PQD_ERR_PREFIX_OUT_OF_BOUNDS: return "PREFIX_OUT_OF_BOUNDS";
        case PQD_ERR_INDEX_OUT_OF_BOUNDS: return "INDEX_OUT_OF_BOUNDS";
        case PQD_ERR_DST_TOO_SMALL: return "DST_TOO_SMALL";
        case PQD_ERR_DEVICE: return "DEVICE";
        case PQD_ERR_OFFSETS_TOO_SMALL: return "OFFSETS_TOO_SMALL";
        case PQD_ERR_CHECKSUM: return "CHECKSUM";
        default: return "UNKNOWN";
    }
}

int pqd_ctx_create(int device, pqd_ctx** out) {
    if (out == nullptr) return PQD_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    int count = 0;
    if (hipGetDeviceCount(&count) != hipSuccess || device < 0 || device >= count) return PQD_ERR_DEVICE;
    DeviceGuard guard(device);
    if (!guard.ok) return PQD_ERR_DEVICE;
    pqd_ctx* ctx = new (std::nothrow) pqd_ctx;
    if (ctx == nullptr) return PQD_ERR_DEVICE;
    ctx->device = device;
    hipDeviceProp_t prop;
    ctx->n_cus = hipGetDeviceProperties(&prop, device) == hipSuccess && prop.multiProcessorCount > 0 ? (uint32_t)prop.multiProcessorCount : 256;
    // PQD_DICT_LDS: 0 = dictionary pages never go through an LDS-resident dictionary, 2 = whenever the page
    // qualifies, otherwise when it is expected to pay.  PQD_DICT_LDS_MAX_BYTES / PQD_DICT_LDS_CAP_BYTES: the
    // largest dictionary that takes the path, and how much of a dictionary LDS may hold (both for experiments and tests).
    const char* lds_opt = getenv("PQD_DICT_LDS");
    ctx->dict_lds = lds_opt != nullptr && lds_opt[0] >= '0' && lds_opt[0] <= '2' ? lds_opt[0] - '0' : 1;
    if (ctx->dict_lds != 0 && !rle_dict_lds_init()) ctx->dict_lds = 0;
    const char* lds_max = getenv("PQD_DICT_LDS_MAX_BYTES");
    ctx->dict_lds_max_bytes = lds_max != nullptr ? strtoull(lds_max, nullptr, 10) : LDS_DICT_DEFAULT_MAX_BYTES;
    const char* lds_cap = getenv("PQD_DICT_LDS_CAP_BYTES");
    ctx->dict_lds_cap_bytes = lds_cap != nullptr ? strtoull(lds_cap, nullptr, 10) : 0;
    // PQD_DICT_SLICES: over how many workgroups a dictionary larger than LDS may be sliced (0 or 1: never);
    // PQD_DICT_SLICE_BYTES: bytes per slice (tests slice small dictionaries)
    const char* sl = getenv("PQD_DICT_SLICES");
    ctx->dict_slices = sl != nullptr ? (uint32_t)strtoul(sl, nullptr, 10) : 2;
    const char* slb = getenv("PQD_DICT_SLICE_BYTES");
    ctx->dict_slice_bytes = slb != nullptr ? strtoull(slb, nullptr, 10) : 0;
    // PLAIN byte-array pages: one wavefront per page, or many workgroups per page when the batch has too few
    // pages to fill the chip with wavefronts (PQD_PLAIN_TILED=0 / 1: never / always, for comparisons and tests)
    const char* tiled = getenv("PQD_PLAIN_TILED");
    // PQD_RLE_PAGE: 0 = level pages go through the index + expand kernels like every other RLE page, 2 = all of
    // them take the page-per-workgroup kernel, otherwise those of up to RLE_PAGE_MAX_STEPS steps do
    const char* rp = getenv("PQD_RLE_PAGE");
    ctx->rle_page = rp != nullptr && (rp[0] == '0' || rp[0] == '2') ? rp[0] - '0' : 1;
    // level pages of at most this many steps of 2048 values are decoded by one wavefront each, longer ones by four (0: all by four)
    const char* rpw = getenv("PQD_RLE_PAGE_WAVE_STEPS");
    ctx->rle_page_wave_steps = rpw != nullptr ? (uint32_t)strtoul(rpw, nullptr, 10) : 0; // (measured: four wavefronts are quicker at every size)
    // PQD_RLE_PAGE_THREADS: 256 or 1024 threads to a level page whatever the batch (tests, measurements); otherwise
    // few pages get wide workgroups and many pages narrow ones
    const char* rpt = getenv("PQD_RLE_PAGE_THREADS");
    const uint32_t rpt_v = rpt != nullptr ? (uint32_t)strtoul(rpt, nullptr, 10) : 0;
    ctx->rle_page_threads = rpt_v == 256 || rpt_v == 1024 ? rpt_v : 0;
    ctx->plain_tiled = tiled != nullptr && (tiled[0] == '0' || tiled[0] == '1') ? tiled[0] - '0' : 2;
    // with this context's device current: where this GPU keeps the static workspace
    ctx->levels_scratch = nullptr;
    if (levels_scratch_address(&ctx->levels_scratch) != PQD_OK) ctx->levels_scratch = nullptr;
    // Side streams for the kernel families of a small batch are an option (PQD_SIDE_BY_SIDE=1): they
    // gain a single-stream caller a few per cent, but every batch of the context goes through the same
    // three, which ties together batches that a caller launched on streams of its own -- and those
    // overlap far better when left alone.
    const char* opt = getenv("PQD_SIDE_BY_SIDE");
    const bool side_by_side = opt != nullptr && opt[0] == '1';
    for (hipStream_t& st : ctx->side) {
        st = nullptr;
        if (side_by_side && hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) st = nullptr;
    }
    *out = ctx;
    return PQD_OK;
}

void pqd_ctx_destroy(pqd_ctx* ctx) {
    if (ctx == nullptr) return;
    DeviceGuard guard(ctx->device);
    for (Workspace* ws : ctx->idle) { ws->release(); delete ws; }
    for (const auto& r : ctx->registered) { // what the caller left registered or allocated
        if (r.owned) (void)hipHostFree(reinterpret_cast<void*>(r.first));
        else (void)hipHostUnregister(reinterpret_cast<void*>(r.first));
    }
    for (hipStream_t st : ctx->side)
        if (st != nullptr) (void)hipStreamDestroy(st);
    delete ctx;
}

int pqd_batch_create(pqd_ctx* ctx, const pqd_page* pages, uint64_t n_pages, pqd_batch** out) {
    return pqd::batch_create_ws(ctx, pages, n_pages, out, nullptr);
}

} // extern "C"

// pqd_batch_create; with a workspace, the arena and the host image of its uploaded part are the
// workspace's (kept from call to call) and the upload is one asynchronous copy on its stream.
int pqd::batch_create_ws(pqd_ctx* ctx, const pqd_page* pages, uint64_t n_pages, pqd_batch** out, Workspace* ws) {
    if (ctx == nullptr || out == nullptr || (pages == nullptr && n_pages > 0)) return PQD_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    DeviceGuard guard(ctx->device);
    if (!guard.ok) return PQD_ERR_DEVICE;
    pqd_batch* b = new (std::nothrow) pqd_batch;
    if (b == nullptr) return PQD_ERR_DEVICE;
    b->ctx = ctx;
    b->n_pages = n_pages;
    b->map.resize(n_pages);

    // ---- classify ----
    size_t runs_entries = 0, lengths_entries = 0, plain_state_entries = 0;
    for (uint64_t i = 0; i < n_pages; i++) {
        const pqd_page& p = pages[i];
        PageMap m = {-1, PQD_OK, 0};
        int32_t family = classify(p);
        if (family == FAM_NONE) {
            m.host_status = PQD_ERR_NOT_SUPPORTED;
        } else if ((p.src == nullptr && p.src_len > 0) || (p.dst == nullptr && p.dst_cap > 0) ||
                   (p.dict == nullptr && p.dict_len > 0)) {
            m.host_status = PQD_ERR_INVALID_ARGUMENT;
        } else if (family == FAM_DELTA32 || family == FAM_DELTA64) {
            m.primary = (int32_t)b->items.size();
            b->items.push_back(make_item(p, family));
        } else if (is_rle(family)) {
            DevPage d = make_item(p, family);
            const bool dict_enc = p.encoding == PQD_ENC_RLE_DICTIONARY || p.encoding == PQD_ENC_PLAIN_DICTIONARY;
            d.fused = (dict_enc && p.dict != nullptr) ? 1 : 0;
            if (!d.fused && dict_enc) d.elem_width = 4;
            if (d.fused && d.elem_width == 8) d.family = FAM_RLE_DICT8;
            if (d.fused && d.elem_width == 4) d.family = FAM_RLE_DICT4;
            d.runs_cap = run_table_entries(d);
            runs_entries += d.runs_cap;
            m.primary = (int32_t)b->items.size();
            b->items.push_back(d);
        } else if (family == FAM_DICT_BYTES) {
            // the indexes are decoded (and cut or zero-extended to num_values) into scratch by an RLE hybrid item;
            // as many as the caller's buffers could take values
            const bool ba = p.kind == PQD_KIND_BYTE_ARRAY;
            const uint64_t width = p.kind == PQD_KIND_BOOLEAN ? 1 : (p.bit_width > 0 ? (uint64_t)p.bit_width : 0);
            if ((ba && (p.dict_offsets == nullptr || (p.offsets == nullptr && p.offsets_cap > 0))) || (!ba && width == 0)) {
                m.host_status = PQD_ERR_INVALID_ARGUMENT;
                m.host_arg = ba ? 0 : (int64_t)p.bit_width;
            } else {
                uint64_t cap = ba ? (p.offsets_cap > 0 ? p.offsets_cap - 1 : 0) : p.dst_cap / width;
                if (p.num_values >= 0) cap = (uint64_t)p.num_values;
                if (cap > 0x7fffffffull) cap = 0x7fffffffull;
                DevPage ix = make_item(p, FAM_RLE);
                ix.kind = PQD_KIND_INT32;
                ix.elem_width = 4;
                ix.dict = nullptr; ix.dict_len = 0; ix.fused = 0;
                ix.dst_stride = 0;
                ix.dst_cap = cap * 4;
                // (dst is pointed into the arena after the carve)
                ix.runs_cap = run_table_entries(ix);
                runs_entries += ix.runs_cap;
                const int32_t ix_idx = (int32_t)b->items.size();
                b->items.push_back(ix);
                DevPage d = make_item(p, family);
                d.chain = ix_idx;
                d.lengths_cap = cap;
                lengths_entries += cap + 8;
                b->dict_bytes_max_values = std::max<uint64_t>(b->dict_bytes_max_values, cap);
                m.primary = (int32_t)b->items.size();
                b->items.push_back(d);
            }
        } else if (family == FAM_DLBA) {
            // the lengths are decoded straight into the caller's offsets buffer and scanned there
            if (p.offsets == nullptr && p.offsets_cap > 0) {
                m.host_status = PQD_ERR_INVALID_ARGUMENT;
            } else {
                DevPage l = make_item(p, FAM_DELTA32);
                l.kind = PQD_KIND_INT32;
                l.elem_width = 4;
                l.dst = reinterpret_cast<uint8_t*>(p.offsets);
                l.dst_cap = p.offsets_cap * 4;
                const int32_t l_idx = (int32_t)b->items.size();
                b->items.push_back(l);
                DevPage d = make_item(p, family);
                d.chain = l_idx;
                m.primary = (int32_t)b->items.size();
                b->items.push_back(d);
            }
        } else if (family == FAM_DBA) {
            // prefix and suffix lengths go to scratch; as many entries as the caller could take values
            const bool ba = p.kind == PQD_KIND_BYTE_ARRAY;
            if (ba && p.offsets == nullptr && p.offsets_cap > 0) {
                m.host_status = PQD_ERR_INVALID_ARGUMENT;
            } else {
                const uint64_t size = p.bit_width > 0 ? (uint64_t)p.bit_width : 1;
                const uint64_t cap = ba ? p.offsets_cap : p.dst_cap / size + 1;
                DevPage pre = make_item(p, FAM_DELTA32);
                pre.kind = PQD_KIND_INT32;
                pre.elem_width = 4;
                pre.lengths_cap = cap;
                pre.dst_cap = cap * 4;
                pre.flat_off = 1;
                const int32_t p_idx = (int32_t)b->items.size();
                b->items.push_back(pre);
                DevPage suf = pre;
                suf.family = FAM_DELTA32_CHAINED;
                suf.chain = p_idx;
                const int32_t s_idx = (int32_t)b->items.size();
                b->items.push_back(suf);
                DevPage d = make_item(p, family);
                d.chain = s_idx;
                d.lengths_cap = cap;
                // [prefix | suffix | three words per block of 64 values for the block-parallel kernels]
                lengths_entries += 2 * cap + 8 + 3 * (cap / 64 + 2);
                uint64_t expect = cap;
                if (p.num_values >= 0 && (uint64_t)p.num_values < expect) expect = (uint64_t)p.num_values;
                b->dba_max_values = std::max<uint64_t>(b->dba_max_values, expect);
                m.primary = (int32_t)b->items.size();
                b->items.push_back(d);
            }
        } else { // FAM_BITPACKED, FAM_BSS, FAM_PLAIN_FIXED, FAM_PLAIN_BYTE_ARRAY: one item, no scratch
            if (family == FAM_PLAIN_BYTE_ARRAY && p.offsets == nullptr && p.offsets_cap > 0) {
                m.host_status = PQD_ERR_INVALID_ARGUMENT;
            } else {
                m.primary = (int32_t)b->items.size();
                b->items.push_back(make_item(p, family));
            }
        }
        b->map[i] = m;
    }

    // ---- PLAIN byte arrays: one wavefront per page walks record by record; many workgroups per page take it tile by
    // tile in one pass.  Measured on short strings (49 bytes a record): 256 pages of 1.5 MB 1390 us walked, 207 us
    // tiled; 2048 pages of 190 KB 403 / 211 us; 16384 pages of 24 KB 433 / 252 us; 32768 pages of 6 KB 382 / 231 us;
    // 32768 pages of 1.5 KB 155 / 179 us.  The average page decides ----
    {
        size_t n_plain = 0;
        uint64_t plain_bytes = 0;
        for (const DevPage& d : b->items)
            if (d.family == FAM_PLAIN_BYTE_ARRAY) { n_plain++; plain_bytes += d.src_len; }
        const bool tiled = ctx->plain_tiled == 1 || (ctx->plain_tiled == 2 && n_plain > 0 && plain_bytes / n_plain >= (4u << 10));
        b->plain_tiled = tiled;
        if (tiled) {
            for (DevPage& d : b->items) {
                if (d.family != FAM_PLAIN_BYTE_ARRAY) continue;
                d.lengths_cap = plain_bytes_scratch_words(d.src_len);
                plain_state_entries += d.lengths_cap;
            }
        }
    }

    // ---- dictionary pages with dense 4- or 8-byte values: through an LDS-resident dictionary when that pays ----
    auto rle_steps_of = [](const DevPage& d) {
        const uint64_t stride = d.dst_stride > 0 ? (uint64_t)d.dst_stride : (uint64_t)d.elem_width;
        uint64_t bound = stride ? d.dst_cap / stride + 1 : 0;
        if (d.num_values >= 0 && (uint64_t)d.num_values < bound) bound = (uint64_t)d.num_values;
        return std::max<uint64_t>(1, (bound + RLE_TILE - 1) / RLE_TILE);
    };
    if (ctx->dict_lds != 0) {
        for (int32_t from : {FAM_RLE_DICT8, FAM_RLE_DICT4}) {
            const int32_t to = from == FAM_RLE_DICT8 ? FAM_RLE_LDS8 : FAM_RLE_LDS4;
            // A page well shorter than its values bit-packed in one run is full of run headers: its tiles meet several
            // runs each, which the persistent kernel decodes a wavefront per tile and the tile-per-workgroup kernel
            // three times as fast (unless the choice is forced)
            auto full_of_runs = [&](const DevPage& d) {
                if (ctx->dict_lds == 2 || d.num_values <= 0 || d.dict_len == 0) return false;
                uint64_t width = 0;
                for (uint64_t m = d.dict_len - 1; m; m >>= 1) width++;
                const uint64_t packed = ((uint64_t)d.num_values * width + 7) / 8;
                return d.src_len >= RLE_CHUNKED_MIN_BYTES_FEW && d.src_len * 10 < packed * 9;
            };
            auto candidate = [&](const DevPage& d) {
                return d.family == from && d.fused && (d.dst_stride == 0 || d.dst_stride == d.elem_width) &&
                       d.dict_len * (uint64_t)d.elem_width <= ctx->dict_lds_max_bytes && !full_of_runs(d);
            };
            // What the path costs: every persistent workgroup copies the dictionary of every stretch of pages
            // it meets into LDS.  It pays while those copies stay below the bytes decoded.
            uint64_t tiles = 0;
            for (const DevPage& d : b->items)
                if (candidate(d)) tiles += rle_steps_of(d);
            if (tiles == 0) continue;
            if (ctx->dict_lds == 1) {
                const uint64_t grid = std::max<uint64_t>(1, std::min<uint64_t>(ctx->n_cus, (tiles + 7) / 8));
                const uint64_t cap = rle_dict_lds_capacity();
                uint64_t copied = 0, at = 0, wg_seen = ~0ull;
                const void* last_dict = nullptr;
                for (const DevPage& d : b->items) {
                    if (!candidate(d)) continue;
                    const uint64_t steps = rle_steps_of(d), bytes = std::min<uint64_t>(d.dict_len * (uint64_t)d.elem_width, cap);
                    // workgroups whose stretch [tiles * w / grid, tiles * (w + 1) / grid) meets this page's tiles [at, at + steps)
                    const uint64_t w0 = at * grid / tiles, w1 = ((at + steps) * grid - 1) / tiles;
                    uint64_t n = w1 - w0 + 1;
                    if (d.dict == last_dict && w0 == wg_seen) n--; // the page before left this dictionary in the first of them
                    copied += n * bytes;
                    last_dict = d.dict;
                    wg_seen = w1;
                    at += steps;
                }
                if (copied > tiles * RLE_TILE * (uint64_t)(from == FAM_RLE_DICT8 ? 8 : 4)) continue;
            }
            uint64_t slice_above = 0, slice_upto = 0;
            rle_dict_slice_range(ctx->dict_lds_cap_bytes, ctx->n_cus, ctx->dict_slices, ctx->dict_slice_bytes, &slice_above, &slice_upto);
            for (DevPage& d : b->items)
                if (candidate(d)) {
                    d.family = to;
                    const uint64_t bytes = d.dict_len * (uint64_t)d.elem_width;
                    b->lds_dict_bytes[rle_grid(to)] = std::max<uint64_t>(b->lds_dict_bytes[rle_grid(to)], bytes);
                    if (bytes > slice_above && bytes <= slice_upto)
                        b->lds_sliced_bytes[rle_grid(to)] = std::max<uint64_t>(b->lds_sliced_bytes[rle_grid(to)], bytes);
                    else
                        b->lds_unsliced[rle_grid(to)] = true;
                }
        }
    }

    // ---- level pages: a workgroup per page, index and expansion in one launch ----
    if (ctx->rle_page != 0) {
        for (DevPage& d : b->items)
            if (d.family == FAM_RLE_BYTES && d.kind == PQD_KIND_LEVELS && d.encoding == PQD_ENC_RLE && d.dst_stride <= 1 && !d.fused &&
                (ctx->rle_page == 2 || rle_steps_of(d) <= RLE_PAGE_MAX_STEPS ||
                 (d.src_len <= RLE_PAGE_MAX_BYTES && rle_steps_of(d) <= RLE_PAGE_MAX_STEPS_FEW_BYTES)))
                d.family = FAM_RLE_PAGE;
    }

    // ---- family lists and the RLE tile grids ----
    std::vector<uint64_t> rle_prefix[RLE_GRIDS];
    for (uint32_t k = 0; k < b->items.size(); k++) {
        DevPage& d = b->items[k];
        b->fam[d.family].items.push_back(k);
        b->max_src[d.family] = std::max<uint64_t>(b->max_src[d.family], d.src_len);
        if (is_rle(d.family)) {
            const int g = rle_grid(d.family);
            const uint64_t steps = rle_steps_of(d);
            if (steps > 0x7fffffffull) { delete b; return PQD_ERR_INVALID_ARGUMENT; }
            d.group = d.family == FAM_RLE_BYTES || d.family == FAM_RLE_PAGE ? 8 : 1; // 1-byte elements: 2 KB per step is too little work for a workgroup
            d.n_steps = (uint32_t)steps;
            const uint64_t tiles = (steps + d.group - 1) / d.group;
            rle_prefix[g].push_back(b->rle_tiles[g]);
            d.tile_base = b->rle_tiles[g];
            d.lengths_cap = b->rle_steps; // offset of this page in the tile_run array (pointer set after the carve)
            b->rle_tiles[g] += tiles;
            b->rle_steps += steps;
        }
    }
    for (int g = 0; g < RLE_GRIDS; g++) {
        rle_prefix[g].push_back(b->rle_tiles[g]);
        if (b->rle_tiles[g] > 0x7fffffffull) { delete b; return PQD_ERR_INVALID_ARGUMENT; }
    }
    {
        // level pages a wavefront takes by itself come first in their family's list, those four wavefronts share behind them
        // (nothing else goes by a page's place in this list: the page kernels use no tile prefix)
        std::vector<uint32_t>& lp = b->fam[FAM_RLE_PAGE].items;
        const uint32_t small = ctx->rle_page_wave_steps;
        b->rle_page_small = (uint32_t)(std::stable_partition(lp.begin(), lp.end(), [&](uint32_t k) { return b->items[k].n_steps <= small; }) - lp.begin());
    }

    // ---- long RLE pages of a batch with few of them: indexed chunk by chunk, by many workgroups each ----
    // (a batch with many RLE pages keeps the chip busy with one wavefront per page; a page that is
    // nearly as long as its values bit-packed in one run has few run headers and is quickest that way too)
    std::vector<uint32_t> ck_items, ck_prefix, ck_page, ck_chunk;
    {
        size_t n_rle = 0;
        for (int g = 0; g < RLE_GRIDS; g++) n_rle += b->fam[RLE_FAMILY[g]].items.size();
        const uint64_t chunk = rle_chunk_bytes();
        uint64_t n_chunks = 0;
        std::vector<uint32_t> per_page;
        if (n_rle <= RLE_CHUNKED_MAX_PAGES) {
            const uint64_t min_bytes = n_rle <= RLE_CHUNKED_FEW_PAGES ? RLE_CHUNKED_MIN_BYTES_FEW : RLE_CHUNKED_MIN_BYTES;
            for (uint32_t k = 0; k < b->items.size(); k++) {
                DevPage& d = b->items[k];
                if (!is_rle(d.family) || d.family == FAM_RLE_PAGE || d.src_len < min_bytes || d.src_len > 0xfff00000ull) continue;
                if (d.num_values >= 0) {
                    // widest the values can be: the declared width, or what the dictionary's size allows
                    uint64_t width = d.bit_width > 0 ? (uint64_t)d.bit_width : 32;
                    // (the indexes of a byte-array dictionary page are an item of their own, right in front of the page's)
                    uint64_t entries = d.fused ? d.dict_len : 0;
                    if (!d.fused && k + 1 < b->items.size() && b->items[k + 1].family == FAM_DICT_BYTES && b->items[k + 1].chain == (int32_t)k)
                        entries = b->items[k + 1].dict_len;
                    if (entries > 0) {
                        width = 0;
                        for (uint64_t m = entries - 1; m; m >>= 1) width++;
                    } else if (d.kind == PQD_KIND_BOOLEAN) {
                        width = 1;
                    }
                    const uint64_t packed = ((uint64_t)d.num_values * width + 7) / 8;
                    if (d.src_len * 10 >= packed * 9) continue;
                }
                const uint64_t span = (reinterpret_cast<uintptr_t>(d.src) & 15) + d.src_len;
                const uint64_t n = (span + chunk - 1) / chunk;
                if (n_chunks + n > 0x7fffffffull) break;
                d.ck_index = (int32_t)ck_items.size();
                ck_items.push_back(k);
                ck_prefix.push_back((uint32_t)n_chunks);
                per_page.push_back((uint32_t)n);
                n_chunks += n;
            }
        }
        ck_prefix.push_back((uint32_t)n_chunks);
        for (uint32_t c = 0; ck_page.size() < n_chunks; c++)
            for (uint32_t k = 0; k < per_page.size(); k++)
                if (c < per_page[k]) { ck_page.push_back(k); ck_chunk.push_back(c); }
        b->ck.n_chunks = (uint32_t)n_chunks;
    }

    // ---- a small batch cannot fill the chip with any one family: its families run side by side ----
    {
        const bool rle = [&] { for (int g = 0; g < RLE_GRIDS; g++) if (!b->fam[RLE_FAMILY[g]].items.empty()) return true; return false; }();
        const bool delta = !b->fam[FAM_DELTA32].items.empty() || !b->fam[FAM_DELTA64].items.empty();
        const bool flat = !b->fam[FAM_BITPACKED].items.empty() || !b->fam[FAM_BSS].items.empty() || !b->fam[FAM_PLAIN_FIXED].items.empty();
        const bool walk = !b->fam[FAM_PLAIN_BYTE_ARRAY].items.empty();
        const int groups = (int)rle + (int)delta + (int)flat + (int)walk;
        bool streams = true;
        for (hipStream_t st : ctx->side) streams = streams && st != nullptr;
        if (groups >= 2 && streams && b->items.size() <= SIDE_BY_SIDE_MAX_ITEMS) {
            bool ev = hipEventCreateWithFlags(&b->fork, hipEventDisableTiming) == hipSuccess;
            for (hipEvent_t& e : b->join) ev = ev && hipEventCreateWithFlags(&e, hipEventDisableTiming) == hipSuccess;
            b->side_by_side = ev;
        }
    }

    // ---- carve one arena ----
    const size_t n_items = b->items.size();
    size_t off = 0;
    auto carve = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    const size_t o_items = carve(n_items * sizeof(DevPage));
    const size_t o_results = carve(n_items * sizeof(DevResult));
    const size_t o_page_retry = carve(sizeof(uint32_t)); // the last launch in which a level page was left to the retry kernel (uploaded as 0)
    size_t o_fam[N_FAMILIES];
    for (int f = 0; f < N_FAMILIES; f++) o_fam[f] = carve(b->fam[f].items.size() * sizeof(uint32_t));
    size_t o_rle_prefix[RLE_GRIDS], o_rle_tile_item[RLE_GRIDS];
    bool with_tile_item[RLE_GRIDS];
    for (int g = 0; g < RLE_GRIDS; g++) {
        o_rle_prefix[g] = carve(rle_prefix[g].size() * sizeof(uint64_t));
        with_tile_item[g] = b->rle_tiles[g] <= (1ull << 24) && !is_lds_family(RLE_FAMILY[g]) && RLE_FAMILY[g] != FAM_RLE_PAGE; // the persistent kernel walks the prefix
        o_rle_tile_item[g] = carve(with_tile_item[g] ? b->rle_tiles[g] * sizeof(uint32_t) : 0);
    }
    size_t o_lds_wg[RLE_GRIDS] = {}, o_lds_seg[RLE_GRIDS] = {};
    std::vector<uint32_t> lds_wg[RLE_GRIDS], lds_seg[RLE_GRIDS];
    for (int g = 0; g < RLE_GRIDS; g++) {
        if (!is_lds_family(RLE_FAMILY[g]) || b->rle_tiles[g] == 0) continue;
        const std::vector<uint32_t>& list = b->fam[RLE_FAMILY[g]].items;
        const LdsPlan plan = rle_dict_lds_plan(b->rle_tiles[g], b->lds_dict_bytes[g], b->lds_sliced_bytes[g], !b->lds_unsliced[g], ctx->dict_lds_cap_bytes, ctx->n_cus,
                                               ctx->dict_slice_bytes);
        b->lds_plan[g] = plan;
        lds_wg[g].resize(2 * (size_t)plan.groups);
        for (uint32_t w = 0, k = 0; w < plan.groups; w++) {
            const uint64_t t0 = b->rle_tiles[g] * w / plan.groups, t1 = b->rle_tiles[g] * (w + 1) / plan.groups;
            while (k + 1 < list.size() && rle_prefix[g][k + 1] <= t0) k++;
            lds_wg[g][2 * w] = k;
            uint32_t k1 = k;
            while (t1 > 0 && k1 + 1 < list.size() && rle_prefix[g][k1 + 1] <= t1 - 1) k1++;
            lds_wg[g][2 * w + 1] = k1;
        }
        lds_seg[g].resize(list.size());
        for (size_t k = list.size(); k-- > 0;) {
            const DevPage& d = b->items[list[k]];
            const bool same = k + 1 < list.size() && b->items[list[k + 1]].dict == d.dict && b->items[list[k + 1]].dict_len == d.dict_len;
            lds_seg[g][k] = same ? lds_seg[g][k + 1] : (uint32_t)k;
        }
        o_lds_wg[g] = carve(lds_wg[g].size() * sizeof(uint32_t));
        o_lds_seg[g] = carve(lds_seg[g].size() * sizeof(uint32_t));
    }
    const size_t nck = b->ck.n_chunks, npk = ck_items.size();
    const size_t o_ck_items = carve(npk * sizeof(uint32_t));
    const size_t o_ck_prefix = carve(ck_prefix.size() * sizeof(uint32_t));
    const size_t o_ck_page = carve(nck * sizeof(uint32_t));
    const size_t o_ck_chunk = carve(nck * sizeof(uint32_t));
    const size_t upload_end = off; // everything up to here comes from the host; the rest is scratch
    const size_t o_rle_tile_run = carve(b->rle_steps * sizeof(uint32_t));
    const size_t o_ck_entry = carve(nck * sizeof(uint32_t));
    const size_t o_ck_exit = carve(nck * sizeof(uint32_t));
    const size_t o_ck_links = carve(nck * (size_t)rle_chunk_bytes() * sizeof(uint16_t));
    const size_t o_ck_state = carve(0); // the block cleared before every launch starts here
    const size_t o_ck_counters = carve(2 * sizeof(uint32_t));
    const size_t o_ck_flag = carve(npk * sizeof(uint32_t));
    const size_t o_ck_pos = carve(npk * sizeof(uint32_t));
    const size_t o_ck_runs = carve(nck * sizeof(unsigned long long));
    const size_t o_ck_vals = carve(nck * sizeof(unsigned long long));
    const size_t o_ck_state_end = carve(0);
    const size_t o_runs = carve(runs_entries * sizeof(Run));
    const size_t o_lengths = carve(lengths_entries * sizeof(int32_t));
    const size_t o_plain_state = carve(plain_state_entries * sizeof(uint32_t)); // zero whenever a launch starts
    // few DELTA pages: block positions and tile sums for the tile-parallel kernels
    size_t flat_bytes = 0;
    for (int f : {FAM_DELTA32, FAM_DELTA32_CHAINED, FAM_DELTA64}) {
        if (b->fam[f].items.empty() || b->fam[f].items.size() > DELTA_FLAT_MAX_PAGES) continue;
        for (uint32_t k : b->fam[f].items) {
            DevPage& d = b->items[k];
            if (d.flat_off) continue;
            uint64_t values = d.dst_cap / (uint64_t)(d.elem_width > 0 ? d.elem_width : 4) + 1;
            if (d.num_values >= 0 && (uint64_t)d.num_values < values) values = (uint64_t)d.num_values;
            if (values > 0x7fffffffull) values = 0x7fffffffull;
            d.flat_cap = values / 128 + 2;
            flat_bytes += align_up((d.flat_cap + 2) * sizeof(uint32_t), 16) + (d.flat_cap / 8 + 3) * sizeof(uint64_t);
            b->flat_blocks[f] = std::max<uint64_t>(b->flat_blocks[f], d.flat_cap);
        }
    }
    const size_t o_flat = carve(flat_bytes);
    b->arena_bytes = std::max<size_t>(off, 256);
    if (ws != nullptr) {
        if (!ws->reserve_arena(b->arena_bytes)) { delete b; return PQD_ERR_DEVICE; }
        b->arena = ws->arena;
        b->own_arena = false;
    } else if (hipMalloc(&b->arena, b->arena_bytes) != hipSuccess) { delete b; return PQD_ERR_DEVICE; }
    uint8_t* a = static_cast<uint8_t*>(b->arena);
    b->d_items = reinterpret_cast<DevPage*>(a + o_items);
    b->d_results = reinterpret_cast<DevResult*>(a + o_results);
    b->d_page_retry = reinterpret_cast<uint32_t*>(a + o_page_retry);
    for (int g = 0; g < RLE_GRIDS; g++) {
        b->d_rle_tile_prefix[g] = reinterpret_cast<uint64_t*>(a + o_rle_prefix[g]);
        b->d_rle_tile_item[g] = with_tile_item[g] ? reinterpret_cast<uint32_t*>(a + o_rle_tile_item[g]) : nullptr;
    }
    b->ck.items = reinterpret_cast<uint32_t*>(a + o_ck_items);
    b->ck.prefix = reinterpret_cast<uint32_t*>(a + o_ck_prefix);
    b->ck.page = reinterpret_cast<uint32_t*>(a + o_ck_page);
    b->ck.chunk = reinterpret_cast<uint32_t*>(a + o_ck_chunk);
    b->ck.entry = reinterpret_cast<uint32_t*>(a + o_ck_entry);
    b->ck.exit = reinterpret_cast<uint32_t*>(a + o_ck_exit);
    b->ck.links = reinterpret_cast<uint16_t*>(a + o_ck_links);
    b->ck.page_pos = reinterpret_cast<uint32_t*>(a + o_ck_pos);
    b->ck.state = a + o_ck_state;
    b->ck.state_bytes = o_ck_state_end - o_ck_state;
    b->ck.counters = reinterpret_cast<uint32_t*>(a + o_ck_counters);
    b->ck.pk_flag = reinterpret_cast<uint32_t*>(a + o_ck_flag);
    b->ck.runs = reinterpret_cast<unsigned long long*>(a + o_ck_runs);
    b->ck.vals = reinterpret_cast<unsigned long long*>(a + o_ck_vals);
    Run* runs = reinterpret_cast<Run*>(a + o_runs);
    int32_t* lengths = reinterpret_cast<int32_t*>(a + o_lengths);
    int32_t* plain_state = reinterpret_cast<int32_t*>(a + o_plain_state);
    uint8_t* flat = a + o_flat;
    for (DevPage& d : b->items) {
        if (d.flat_cap > 0) {
            d.flat_pos = reinterpret_cast<uint32_t*>(flat);
            flat += align_up((d.flat_cap + 2) * sizeof(uint32_t), 16);
            d.flat_sum = reinterpret_cast<uint64_t*>(flat);
            flat += (d.flat_cap / 8 + 3) * sizeof(uint64_t);
        }
        if (is_rle(d.family)) {
            d.runs = runs; runs += d.runs_cap;
            d.tile_run = reinterpret_cast<uint32_t*>(a + o_rle_tile_run) + d.lengths_cap;
            d.lengths_cap = 0;
        }
        if (d.family == FAM_PLAIN_BYTE_ARRAY && d.lengths_cap > 0) {
            d.lengths = plain_state;
            plain_state += d.lengths_cap;
        }
        if (d.family == FAM_DICT_BYTES) {
            DevPage& ix = b->items[d.chain];
            d.lengths = lengths;
            ix.dst = reinterpret_cast<uint8_t*>(lengths);
            ix.lengths_cap = 0;
            lengths += d.lengths_cap + 8;
        }
        if (d.family == FAM_DBA) {
            // [prefix | suffix], each lengths_cap entries (+4 of slack so that vector stores of the decoder stay inside)
            DevPage& suf = b->items[d.chain];
            DevPage& pre = b->items[suf.chain];
            d.lengths = lengths;
            d.lengths2 = lengths + d.lengths_cap + 4;
            pre.dst = reinterpret_cast<uint8_t*>(d.lengths);
            suf.dst = reinterpret_cast<uint8_t*>(d.lengths2);
            d.tile_run = reinterpret_cast<uint32_t*>(lengths + 2 * d.lengths_cap + 8);
            lengths += 2 * d.lengths_cap + 8 + 3 * (d.lengths_cap / 64 + 2);
        }
    }

    // ---- upload: a host image of the arena's front part, copied in one go ----
    std::vector<uint8_t> local;
    uint8_t* image;
    if (ws != nullptr) {
        if (!ws->reserve_pinned(upload_end + n_items * sizeof(DevResult))) { delete b; return PQD_ERR_DEVICE; }
        image = static_cast<uint8_t*>(ws->pinned);
        std::memset(image, 0, upload_end);
    } else {
        local.assign(upload_end, 0);
        image = local.data();
    }
    auto put = [&](size_t at, const void* p, size_t n) { if (n) std::memcpy(image + at, p, n); };
    put(o_items, b->items.data(), n_items * sizeof(DevPage)); // (the results behind them start out as zeros)
    for (int f = 0; f < N_FAMILIES; f++) {
        b->fam[f].dev = reinterpret_cast<uint32_t*>(a + o_fam[f]);
        put(o_fam[f], b->fam[f].items.data(), b->fam[f].items.size() * sizeof(uint32_t));
    }
    for (int g = 0; g < RLE_GRIDS; g++) {
        put(o_rle_prefix[g], rle_prefix[g].data(), rle_prefix[g].size() * sizeof(uint64_t));
        if (with_tile_item[g] && b->rle_tiles[g] > 0) {
            uint32_t* tile_item = reinterpret_cast<uint32_t*>(image + o_rle_tile_item[g]);
            for (size_t k = 0; k + 1 < rle_prefix[g].size(); k++)
                std::fill(tile_item + rle_prefix[g][k], tile_item + rle_prefix[g][k + 1], (uint32_t)k);
        }
        if (lds_wg[g].empty()) continue;
        b->d_lds_wg_item[g] = reinterpret_cast<uint32_t*>(a + o_lds_wg[g]);
        b->d_lds_seg_last[g] = reinterpret_cast<uint32_t*>(a + o_lds_seg[g]);
        put(o_lds_wg[g], lds_wg[g].data(), lds_wg[g].size() * sizeof(uint32_t));
        put(o_lds_seg[g], lds_seg[g].data(), lds_seg[g].size() * sizeof(uint32_t));
    }
    put(o_ck_items, ck_items.data(), npk * sizeof(uint32_t));
    put(o_ck_prefix, ck_prefix.data(), ck_prefix.size() * sizeof(uint32_t));
    put(o_ck_page, ck_page.data(), nck * sizeof(uint32_t));
    put(o_ck_chunk, ck_chunk.data(), nck * sizeof(uint32_t));
    bool ok = ws != nullptr ? hip_ok(hipMemcpyAsync(a, image, upload_end, hipMemcpyHostToDevice, ws->stream))
                            : hip_ok(hipMemcpy(a, image, upload_end, hipMemcpyHostToDevice));
    // the tile states of PLAIN byte-array pages start out as zeros (every launch leaves them so again)
    if (ok && plain_state_entries > 0)
        ok = ws != nullptr ? hip_ok(hipMemsetAsync(a + o_plain_state, 0, plain_state_entries * sizeof(uint32_t), ws->stream))
                           : hip_ok(hipMemset(a + o_plain_state, 0, plain_state_entries * sizeof(uint32_t))) && hip_ok(hipStreamSynchronize(nullptr));
    if (!ok) { delete b; return PQD_ERR_DEVICE; }
    b->h_results.resize(n_items);
    *out = b;
    return PQD_OK;
}

extern "C" {

int pqd_batch_launch(pqd_batch* b, void* hip_stream) {
    if (b == nullptr) return PQD_ERR_INVALID_ARGUMENT;
    DeviceGuard guard(b->ctx->device);
    if (!guard.ok) return PQD_ERR_DEVICE;
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    // The kernel families of a batch are independent of one another, except that the DELTA byte-array
    // families start from length streams the DELTA kernels decode.  A small batch runs them side by
    // side: the caller's stream forks into the context's side streams and joins them again, so to the
    // caller everything still happens in its stream (the pattern is legal under stream capture too).
    hipStream_t s_rle = s, s_flat = s, s_walk = s;
    if (b->side_by_side) {
        PQD_HIP(hipEventRecord(b->fork, s));
        s_rle = b->ctx->side[0]; s_flat = b->ctx->side[1]; s_walk = b->ctx->side[2];
        for (hipStream_t st : b->ctx->side) PQD_HIP(hipStreamWaitEvent(st, b->fork, 0));
    }
    launch_delta(b->d_items, b->d_results, b->fam[FAM_DELTA32].dev, (uint32_t)b->fam[FAM_DELTA32].items.size(), false, b->flat_blocks[FAM_DELTA32], s);
    launch_delta(b->d_items, b->d_results, b->fam[FAM_DELTA32_CHAINED].dev, (uint32_t)b->fam[FAM_DELTA32_CHAINED].items.size(), false,
                 b->flat_blocks[FAM_DELTA32_CHAINED], s);
    launch_delta(b->d_items, b->d_results, b->fam[FAM_DELTA64].dev, (uint32_t)b->fam[FAM_DELTA64].items.size(), true, b->flat_blocks[FAM_DELTA64], s);
    launch_rle_chunked(b->d_items, b->d_results, b->ck, s_rle);
    b->launches++;
    for (int g = 0; g < RLE_GRIDS; g++) {
        if (is_lds_family(RLE_FAMILY[g]) || RLE_FAMILY[g] == FAM_RLE_PAGE) continue; // these index their pages in the launch that decodes them
        const FamilyList& f = b->fam[RLE_FAMILY[g]];
        launch_rle_index(b->d_items, b->d_results, f.dev, (uint32_t)f.items.size(), b->ck.pk_flag, s_rle);
    }
    for (int g = 0; g < RLE_GRIDS; g++) {
        const FamilyList& f = b->fam[RLE_FAMILY[g]];
        if (is_lds_family(RLE_FAMILY[g]))
            launch_rle_dict_lds(b->d_items, b->d_results, f.dev, (uint32_t)f.items.size(), b->d_rle_tile_prefix[g], b->rle_tiles[g],
                                RLE_FAMILY[g] == FAM_RLE_LDS8 ? 8 : 4, b->lds_plan[g], b->d_lds_wg_item[g], b->d_lds_seg_last[g], b->ck.pk_flag,
                                b->launches, s_rle);
        else if (RLE_FAMILY[g] == FAM_RLE_PAGE)
            launch_rle_pages(b->d_items, b->d_results, f.dev, (uint32_t)f.items.size(), b->rle_page_small, b->ctx->rle_page_threads, b->d_page_retry, b->launches, s_rle);
        else
            launch_rle_expand(b->d_items, b->d_results, f.dev, (uint32_t)f.items.size(), b->d_rle_tile_prefix[g], b->d_rle_tile_item[g],
                              b->rle_tiles[g], RLE_FAMILY[g], s_rle);
    }
    auto fam_n = [&](int f) { return (uint32_t)b->fam[f].items.size(); };
    launch_bitpacked(b->d_items, b->d_results, b->fam[FAM_BITPACKED].dev, fam_n(FAM_BITPACKED), b->max_src[FAM_BITPACKED] * 8, s_flat);
    launch_bss(b->d_items, b->d_results, b->fam[FAM_BSS].dev, fam_n(FAM_BSS), b->max_src[FAM_BSS], s_flat);
    launch_plain_fixed(b->d_items, b->d_results, b->fam[FAM_PLAIN_FIXED].dev, fam_n(FAM_PLAIN_FIXED), b->max_src[FAM_PLAIN_FIXED], s_flat);
    launch_plain_byte_array(b->d_items, b->d_results, b->fam[FAM_PLAIN_BYTE_ARRAY].dev, fam_n(FAM_PLAIN_BYTE_ARRAY),
                            b->plain_tiled ? b->max_src[FAM_PLAIN_BYTE_ARRAY] : 0, s_walk);
    launch_dict_bytes(b->d_items, b->d_results, b->fam[FAM_DICT_BYTES].dev, fam_n(FAM_DICT_BYTES), b->dict_bytes_max_values, s_rle);
    launch_dlba(b->d_items, b->d_results, b->fam[FAM_DLBA].dev, fam_n(FAM_DLBA), b->max_src[FAM_DLBA], s);
    launch_dba(b->d_items, b->d_results, b->fam[FAM_DBA].dev, fam_n(FAM_DBA), b->dba_max_values, s);
    if (b->side_by_side) {
        for (int i = 0; i < 3; i++) {
            PQD_HIP(hipEventRecord(b->join[i], b->ctx->side[i]));
            PQD_HIP(hipStreamWaitEvent(s, b->join[i], 0));
        }
    }
    return hipGetLastError() == hipSuccess ? PQD_OK : PQD_ERR_DEVICE;
}

int pqd_batch_finish(pqd_batch* b, void* hip_stream, pqd_page* pages, uint64_t n_pages) {
    if (b == nullptr || (pages == nullptr && n_pages > 0) || n_pages != b->n_pages) return PQD_ERR_INVALID_ARGUMENT;
    DeviceGuard guard(b->ctx->device);
    if (!guard.ok) return PQD_ERR_DEVICE;
    hipStream_t s = static_cast<hipStream_t>(hip_stream);
    PQD_HIP(hipStreamSynchronize(s));
    if (!b->items.empty())
        PQD_HIP(hipMemcpy(b->h_results.data(), b->d_results, b->items.size() * sizeof(DevResult), hipMemcpyDeviceToHost));
    pqd::batch_report(b, b->h_results.data(), pages);
    return PQD_OK;
}

} // extern "C"

// pqd_batch_finish on a workspace: the results come back by an asynchronous copy into the
// workspace's pinned memory, one synchronisation in all.
int pqd::batch_finish_ws(pqd_batch* b, Workspace* ws, pqd_page* pages) {
    const size_t bytes = b->items.size() * sizeof(DevResult);
    if (!ws->reserve_pinned(bytes + 64)) return PQD_ERR_DEVICE;
    if (bytes) PQD_HIP(hipMemcpyAsync(ws->pinned, b->d_results, bytes, hipMemcpyDeviceToHost, ws->stream));
    PQD_HIP(hipStreamSynchronize(ws->stream));
    batch_report(b, static_cast<const DevResult*>(ws->pinned), pages);
    return PQD_OK;
}

// The result fields of the caller's pages out of the items' device results.
void pqd::batch_report(pqd_batch* b, const DevResult* results, pqd_page* pages) {
    for (uint64_t i = 0; i < b->n_pages; i++) {
        const PageMap& m = b->map[i];
        pqd_page& p = pages[i];
        p.reserved = 0;
        if (m.primary < 0) {
            p.status = m.host_status;
            p.status_arg = m.host_arg;
            p.out_values = p.out_bytes = p.consumed = 0;
            continue;
        }
        const DevResult& r = results[m.primary];
        const DevPage& d = b->items[m.primary];
        p.status = r.status;
        p.status_arg = r.arg;
        p.out_values = r.out_values;
        p.out_bytes = r.out_bytes;
        p.consumed = r.consumed;
        if (is_lds_family(d.family)) {
            // index and decode ran in one launch and wrote apart: capacity first, then the first bad index of THIS launch
            if (r.status == PQD_OK && r.need_bytes > 0) {
                p.status = PQD_ERR_DST_TOO_SMALL;
                p.status_arg = (int64_t)r.need_bytes;
            } else if (r.status == PQD_OK && r.bad != 0 && (uint32_t)(r.bad >> 32) == b->launches) {
                p.status = PQD_ERR_INDEX_OUT_OF_BOUNDS;
                p.status_arg = (int64_t)(0xffffffffull - (r.bad & 0xffffffffull));
            }
        } else if (is_rle(d.family) && d.fused && r.status == PQD_OK && r.aux != ~0ull) {
            p.status = PQD_ERR_INDEX_OUT_OF_BOUNDS; // the reference panics here (errors.go:75-83)
            p.status_arg = (int64_t)r.aux;
        }
    }
}

extern "C" {

void pqd_batch_destroy(pqd_batch* b) {
    if (b == nullptr) return;
    DeviceGuard guard(b->ctx->device);
    delete b;
}

int pqd_decode_batch(pqd_ctx* ctx, pqd_page* pages, uint64_t n_pages, void* hip_stream) {
    pqd_batch* b = nullptr;
    int rc = pqd_batch_create(ctx, pages, n_pages, &b);
    if (rc != PQD_OK) return rc;
    rc = pqd_batch_launch(b, hip_stream);
    if (rc == PQD_OK) rc = pqd_batch_finish(b, hip_stream, pages, n_pages);
    pqd_batch_destroy(b);
    return rc;
}

} // extern "C"
