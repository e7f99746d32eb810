// host_api.hip -- per-method host-pointer wrappers of include/pqd.h (the surface a
// cgo shim binds, one call per Decode method of encoding.Encoding) and the
// dictionary handle.  Each wrapper stages the page in device memory, runs the
// same single-item batch the device-resident path runs, and copies the result
// back; there is no CPU decode path.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstring>
#include <new>

#include "pqd_internal.h"

namespace pqd {

// ---- stand-alone dictionary gather (Dictionary.Lookup on decoded indexes) ----
// rows[i*stride .. +W) = dict[idx[i]]; reference: dictionaryLookup32/64,
// dictionary_amd64.s:578-711 (bounds check + gather + strided scatter).
template <int W>
__global__ __launch_bounds__(256) void dict_gather_kernel(const uint8_t* __restrict__ dict, uint64_t dict_len,
                                                         const int32_t* __restrict__ idx, uint64_t n,
                                                         uint8_t* __restrict__ rows, uint64_t stride,
                                                         unsigned long long* __restrict__ status) {
    for (uint64_t i = blockIdx.x * 256ull + threadIdx.x; i < n; i += (uint64_t)gridDim.x * 256ull) {
        const uint32_t j = (uint32_t)idx[i];
        const bool ok = (uint64_t)j < dict_len;
        if (!ok) atomicMin(&status[1], (unsigned long long)i);
        const uint32_t* d = reinterpret_cast<const uint32_t*>(dict + (uint64_t)(ok ? j : 0) * W);
        uint32_t* r = reinterpret_cast<uint32_t*>(rows + i * stride);
        if (W == 8 && (stride & 7) == 0) {
            uint64_t v = ok ? *reinterpret_cast<const uint64_t*>(d) : 0ull;
            *reinterpret_cast<uint64_t*>(r) = v;
        } else {
#pragma unroll
            for (int k = 0; k < W / 4; k++) r[k] = ok ? d[k] : 0u;
        }
    }
}

__global__ void dict_status_init(unsigned long long* status) {
    status[0] = 0;
    status[1] = ~0ull;
}

__global__ void dict_status_fini(unsigned long long* status) {
    if (status[1] != ~0ull) status[0] = PQD_ERR_INDEX_OUT_OF_BOUNDS;
    else status[1] = 0;
}

int launch_dict_gather(const void* dict, uint64_t dict_len, uint32_t width, const int32_t* idx, uint64_t n, void* rows,
                       uint64_t stride, int64_t* status_dev, hipStream_t s) {
    if (width != 4 && width != 8 && width != 12) return PQD_ERR_INVALID_ARGUMENT;
    if (stride == 0) stride = width;
    if (stride < width || (stride & 3)) return PQD_ERR_INVALID_ARGUMENT;
    unsigned long long* st = reinterpret_cast<unsigned long long*>(status_dev);
    hipLaunchKernelGGL(dict_status_init, dim3(1), dim3(1), 0, s, st);
    if (n > 0) {
        uint64_t blocks = (n + 255) / 256;
        if (blocks > 256 * 8 * 4) blocks = 256 * 8 * 4;
        const uint8_t* d = static_cast<const uint8_t*>(dict);
        uint8_t* r = static_cast<uint8_t*>(rows);
        if (width == 4) hipLaunchKernelGGL(dict_gather_kernel<4>, dim3((uint32_t)blocks), dim3(256), 0, s, d, dict_len, idx, n, r, stride, st);
        else if (width == 8) hipLaunchKernelGGL(dict_gather_kernel<8>, dim3((uint32_t)blocks), dim3(256), 0, s, d, dict_len, idx, n, r, stride, st);
        else hipLaunchKernelGGL(dict_gather_kernel<12>, dim3((uint32_t)blocks), dim3(256), 0, s, d, dict_len, idx, n, r, stride, st);
    }
    hipLaunchKernelGGL(dict_status_fini, dim3(1), dim3(1), 0, s, st);
    return hipGetLastError() == hipSuccess ? PQD_OK : PQD_ERR_DEVICE;
}

namespace {

// A workspace of the context for the length of one host call.
struct Lease {
    pqd_ctx* ctx;
    Workspace* ws;
    explicit Lease(pqd_ctx* c) : ctx(c), ws(workspace_acquire(c)) {}
    ~Lease() { if (ws) workspace_release(ctx, ws); }
    Lease(const Lease&) = delete;
    Lease& operator=(const Lease&) = delete;
};

inline size_t pad256(size_t n) { return (n + 255) & ~(size_t)255; }

// One page, host pointers in and out, through the batched device path.  Staging memory, the
// batch's arena and the stream are the leased workspace's: a call allocates nothing once the
// workspace has seen a page of this size, and calls of different threads run on different streams.
int decode_host(pqd_ctx* ctx, int encoding, int kind, int bit_width, const uint8_t* src, uint64_t src_len, void* dst,
                uint64_t dst_cap, uint32_t* offsets, uint64_t offsets_cap, uint64_t* out_bytes, uint64_t* out_offsets,
                int64_t* status_arg) {
    if (out_bytes) *out_bytes = 0;
    if (out_offsets) *out_offsets = 0;
    if (status_arg) *status_arg = 0;
    if (ctx == nullptr || (src == nullptr && src_len > 0) || (dst == nullptr && dst_cap > 0)) return PQD_ERR_INVALID_ARGUMENT;
    DeviceGuard guard(ctx->device);
    if (!guard.ok) return PQD_ERR_DEVICE;
    Lease lease(ctx);
    Workspace* ws = lease.ws;
    if (ws == nullptr) return PQD_ERR_DEVICE;
    // 32 spare bytes each so the kernels' aligned 16-byte loads of the last bytes stay inside the allocation
    const size_t o_src = 0, o_dst = pad256(src_len + 32), o_off = o_dst + pad256(dst_cap + 32);
    const size_t total = o_off + pad256(offsets_cap * 4 + 32);
    if (!ws->reserve_io(total)) return PQD_ERR_DEVICE;
    uint8_t* d_io = static_cast<uint8_t*>(ws->d_io);
    uint8_t* h_io = static_cast<uint8_t*>(ws->h_io);
    // Memory the caller has registered (pqd_host_register) is reached by the copy engines where it lies: no
    // memcpy through the workspace's pinned staging, which for a page of 1 MiB is most of what a call costs the host.
    const bool src_direct = ctx->host_registered(src, src_len);
    const bool dst_direct = ctx->host_registered(dst, dst_cap);
    if (src_len) {
        if (!src_direct) std::memcpy(h_io + o_src, src, src_len);
        PQD_HIP(hipMemcpyAsync(d_io + o_src, src_direct ? src : h_io + o_src, src_len, hipMemcpyHostToDevice, ws->stream));
    }
    pqd_page p;
    std::memset(&p, 0, sizeof p);
    p.src = d_io + o_src; p.src_len = src_len;
    p.dst = d_io + o_dst; p.dst_cap = dst_cap;
    p.offsets = reinterpret_cast<uint32_t*>(d_io + o_off); p.offsets_cap = offsets_cap;
    p.num_values = -1;
    p.encoding = encoding; p.kind = kind; p.bit_width = bit_width;
    pqd_batch* b = nullptr;
    int rc = batch_create_ws(ctx, &p, 1, &b, ws);
    if (rc != PQD_OK) return rc;
    rc = pqd_batch_launch(b, ws->stream);
    // Most pages fill the destination the caller sized for them: the values start back together with
    // the results, and only a page that came out shorter or failed is looked at again.
    const bool with_offsets = offsets != nullptr && kind == PQD_KIND_BYTE_ARRAY;
    const bool off_direct = with_offsets && ctx->host_registered(offsets, offsets_cap * 4);
    if (rc == PQD_OK && dst_cap)
        rc = hipMemcpyAsync(dst_direct ? dst : h_io + o_dst, d_io + o_dst, dst_cap, hipMemcpyDeviceToHost, ws->stream) == hipSuccess ? PQD_OK : PQD_ERR_DEVICE;
    if (rc == PQD_OK && with_offsets && offsets_cap)
        rc = hipMemcpyAsync(off_direct ? (void*)offsets : (void*)(h_io + o_off), d_io + o_off, offsets_cap * 4, hipMemcpyDeviceToHost, ws->stream) == hipSuccess ? PQD_OK : PQD_ERR_DEVICE;
    if (rc == PQD_OK) rc = batch_finish_ws(b, ws, &p);
    else (void)hipStreamSynchronize(ws->stream);
    pqd_batch_destroy(b);
    if (rc != PQD_OK) return rc;
    if (status_arg) *status_arg = p.status_arg;
    const uint64_t nb = p.out_bytes < dst_cap ? p.out_bytes : dst_cap;
    if (nb && !dst_direct) std::memcpy(dst, h_io + o_dst, nb);
    if (out_bytes) *out_bytes = nb;
    if (with_offsets) {
        uint64_t no = p.status == PQD_OK ? p.out_values + 1 : p.out_values;
        if (no > offsets_cap) no = offsets_cap;
        if (no && !off_direct) std::memcpy(offsets, h_io + o_off, no * 4);
        if (out_offsets) *out_offsets = no;
    }
    return p.status;
}

// ---- EstimateDecodeByteArraySize on the host: a walk over the DELTA_BINARY_PACKED framing ----
struct Reader {
    const uint8_t* p;
    const uint8_t* end;
    bool uvarint(uint64_t* v) {
        uint64_t x = 0;
        for (int shift = 0; shift < 70 && p < end; shift += 7) {
            const uint8_t c = *p++;
            if (shift == 63 && c > 1) return false;
            x |= (uint64_t)(c & 0x7f) << shift;
            if (!(c & 0x80)) { *v = x; return true; }
        }
        return false;
    }
    bool svarint(int64_t* v) {
        uint64_t u;
        if (!uvarint(&u)) return false;
        *v = (int64_t)(u >> 1) ^ -(int64_t)(u & 1);
        return true;
    }
};

// The wrapping int32 sum of what int32Buffer.decode leaves in its buffer for one
// DELTA_BINARY_PACKED int32 stream (encoding/delta/delta.go:22-33), and how far the decoder
// got.  The walk follows decodeInt32 (encoding/delta/binary_packed.go:279-344) step by step,
// including what it leaves behind when it stops early, because the estimate's callers ignore the
// error and sum the buffer (length_byte_array.go:65-70, byte_array.go:176-183): the buffer has
// the header's count of values once the header is accepted, values never reached are zero,
// a block cut short by a bad width keeps its unpacked deltas without the running sum, and
// packed data that ends early reads as zeros.
uint32_t sum_delta_int32(const uint8_t* src, uint64_t len, uint64_t* consumed) {
    Reader r{src, src + len};
    *consumed = 0;
    uint64_t u[3];
    int64_t first;
    if (!r.uvarint(&u[0]) || !r.uvarint(&u[1]) || !r.uvarint(&u[2]) || !r.svarint(&first)) return 0;
    *consumed = (uint64_t)(r.p - src);
    const int64_t block_size = (int64_t)u[0], n_mini = (int64_t)u[1], total = (int64_t)u[2];
    if (n_mini <= 0 || block_size <= 0 || block_size % 128 != 0 || block_size > 65536) return 0;
    if ((block_size / n_mini) % 32 != 0 || total <= 0 || total > INT32_MAX) return 0;
    if (first < INT32_MIN || first > INT32_MAX) return 0;
    const uint64_t mini = (uint64_t)(block_size / n_mini);
    uint32_t last = (uint32_t)first, sum = (uint32_t)first;
    uint64_t left = (uint64_t)total - 1;
    while (left > 0 && r.p < r.end) {
        *consumed = (uint64_t)(r.p - src);
        int64_t min_delta;
        if (!r.svarint(&min_delta)) return sum;
        const uint64_t n_widths = std::min<uint64_t>((uint64_t)n_mini, (uint64_t)(r.end - r.p));
        const uint8_t* widths = r.p;
        r.p += n_widths;
        uint32_t raw = 0, run = last, summed = 0; // this block: deltas as unpacked; running value; values
        for (uint64_t m = 0; m < n_widths; m++) {
            const uint32_t w = widths[m];
            const uint64_t n = std::min(mini, left);
            if (w != 0) {
                if (w > 32 && n > 0) { *consumed = (uint64_t)(r.p - src); return sum + raw; }
                const uint64_t bytes = mini * w / 8;
                const uint64_t have = std::min<uint64_t>(bytes, (uint64_t)(r.end - r.p));
                const uint8_t* q = r.p;
                const uint8_t* q_end = r.p + have;
                const uint64_t mask = w >= 32 ? 0xffffffffull : ((1ull << w) - 1);
                uint64_t acc = 0;
                uint32_t bits = 0;
                for (uint64_t i = 0; i < n; i++) {
                    while (bits < w) { acc |= (uint64_t)(q < q_end ? *q : 0) << bits; q++; bits += 8; }
                    const uint32_t d = (uint32_t)(acc & mask);
                    acc >>= w;
                    bits -= w;
                    raw += d;
                    run += d + (uint32_t)min_delta;
                    summed += run;
                }
                r.p += have;
            } else {
                // n deltas of zero: run, run + m, ... with m = min_delta
                for (uint64_t i = 0; i < n; i++) { run += (uint32_t)min_delta; summed += run; }
            }
            left -= n;
            if (left == 0) break;
        }
        last = run;
        sum += summed;
    }
    *consumed = (uint64_t)(r.p - src);
    return sum;
}

} // namespace
} // namespace pqd

using namespace pqd;

extern "C" {

int pqd_decode_levels(pqd_ctx* ctx, int encoding, int bit_width, const uint8_t* src, uint64_t src_len, uint8_t* dst,
                      uint64_t dst_cap, uint64_t* out_bytes, int64_t* status_arg) {
    return decode_host(ctx, encoding, PQD_KIND_LEVELS, bit_width, src, src_len, dst, dst_cap, nullptr, 0, out_bytes, nullptr, status_arg);
}

int pqd_decode_boolean(pqd_ctx* ctx, int encoding, const uint8_t* src, uint64_t src_len, uint8_t* dst, uint64_t dst_cap,
                       uint64_t* out_bytes, int64_t* status_arg) {
    return decode_host(ctx, encoding, PQD_KIND_BOOLEAN, 1, src, src_len, dst, dst_cap, nullptr, 0, out_bytes, nullptr, status_arg);
}

int pqd_decode_int32(pqd_ctx* ctx, int encoding, int bit_width, const uint8_t* src, uint64_t src_len, int32_t* dst,
                     uint64_t dst_cap, uint64_t* out_bytes, int64_t* status_arg) {
    return decode_host(ctx, encoding, PQD_KIND_INT32, bit_width, src, src_len, dst, dst_cap, nullptr, 0, out_bytes, nullptr, status_arg);
}

int pqd_decode_int64(pqd_ctx* ctx, int encoding, const uint8_t* src, uint64_t src_len, int64_t* dst, uint64_t dst_cap,
                     uint64_t* out_bytes, int64_t* status_arg) {
    return decode_host(ctx, encoding, PQD_KIND_INT64, 0, src, src_len, dst, dst_cap, nullptr, 0, out_bytes, nullptr, status_arg);
}

int pqd_decode_int96(pqd_ctx* ctx, int encoding, const uint8_t* src, uint64_t src_len, uint8_t* dst, uint64_t dst_cap,
                     uint64_t* out_bytes, int64_t* status_arg) {
    return decode_host(ctx, encoding, PQD_KIND_INT96, 0, src, src_len, dst, dst_cap, nullptr, 0, out_bytes, nullptr, status_arg);
}

int pqd_decode_float(pqd_ctx* ctx, int encoding, const uint8_t* src, uint64_t src_len, float* dst, uint64_t dst_cap,
                     uint64_t* out_bytes, int64_t* status_arg) {
    return decode_host(ctx, encoding, PQD_KIND_FLOAT, 0, src, src_len, dst, dst_cap, nullptr, 0, out_bytes, nullptr, status_arg);
}

int pqd_decode_double(pqd_ctx* ctx, int encoding, const uint8_t* src, uint64_t src_len, double* dst, uint64_t dst_cap,
                      uint64_t* out_bytes, int64_t* status_arg) {
    return decode_host(ctx, encoding, PQD_KIND_DOUBLE, 0, src, src_len, dst, dst_cap, nullptr, 0, out_bytes, nullptr, status_arg);
}

int pqd_decode_byte_array(pqd_ctx* ctx, int encoding, const uint8_t* src, uint64_t src_len, uint8_t* dst, uint64_t dst_cap,
                          uint32_t* offsets, uint64_t offsets_cap, uint64_t* out_bytes, uint64_t* out_offsets,
                          int64_t* status_arg) {
    return decode_host(ctx, encoding, PQD_KIND_BYTE_ARRAY, 0, src, src_len, dst, dst_cap, offsets, offsets_cap, out_bytes, out_offsets, status_arg);
}

int pqd_decode_fixed_len_byte_array(pqd_ctx* ctx, int encoding, int size, const uint8_t* src, uint64_t src_len, uint8_t* dst,
                                    uint64_t dst_cap, uint64_t* out_bytes, int64_t* status_arg) {
    return decode_host(ctx, encoding, PQD_KIND_FIXED_LEN_BYTE_ARRAY, size, src, src_len, dst, dst_cap, nullptr, 0, out_bytes, nullptr, status_arg);
}

int64_t pqd_estimate_decode_byte_array_size(pqd_ctx* ctx, int encoding, const uint8_t* src, uint64_t src_len) {
    // plain.go:106-108: PLAIN answers len(src).  The DELTA encodings sum their length
    // streams with wrapping int32 arithmetic (delta.go:28-33; length_byte_array.go:65-70,
    // byte_array.go:176-183).  The streams are a few bytes per value and the page is in host
    // memory: they are walked where they lie, with no trip to the device.
    (void)ctx;
    if (encoding == PQD_ENC_PLAIN) return (int64_t)src_len;
    if (encoding != PQD_ENC_DELTA_LENGTH_BYTE_ARRAY && encoding != PQD_ENC_DELTA_BYTE_ARRAY) return 0;
    if (src == nullptr && src_len > 0) return 0;
    const int streams = encoding == PQD_ENC_DELTA_BYTE_ARRAY ? 2 : 1;
    int64_t total = 0;
    uint64_t at = 0;
    for (int k = 0; k < streams; k++) {
        uint64_t used = 0;
        total += (int64_t)(int32_t)sum_delta_int32(src + at, src_len - at, &used);
        at += used < src_len - at ? used : src_len - at;
    }
    return total;
}

int pqd_encode_values(pqd_ctx* ctx, int encoding, int kind, int bit_width, const void* src, uint64_t n_values, uint8_t* dst,
                      uint64_t dst_cap, uint64_t* out_bytes, int64_t* status_arg) {
    if (out_bytes) *out_bytes = 0;
    if (status_arg) *status_arg = 0;
    if (ctx == nullptr || (src == nullptr && n_values > 0) || (dst == nullptr && dst_cap > 0)) return PQD_ERR_INVALID_ARGUMENT;
    uint64_t width = 0;
    switch (kind) {
        case PQD_KIND_INT32: case PQD_KIND_FLOAT: width = 4; break;
        case PQD_KIND_INT64: case PQD_KIND_DOUBLE: width = 8; break;
        case PQD_KIND_FIXED_LEN_BYTE_ARRAY: width = bit_width > 0 ? (uint64_t)bit_width : 0; break;
        case PQD_KIND_LEVELS: case PQD_KIND_BOOLEAN: width = 1; break; // a level a byte; booleans: n_values bytes of packed bits
        default: break;
    }
    if (kind == PQD_KIND_FIXED_LEN_BYTE_ARRAY && encoding == PQD_ENC_DELTA_BYTE_ARRAY) // EncodeFixedLenByteArray (byte_array.go:76-122)
        return pqd_encode_byte_array(ctx, encoding, kind, bit_width, src, n_values * width, nullptr, n_values, dst, dst_cap, out_bytes, status_arg);
    DeviceGuard guard(ctx->device);
    if (!guard.ok) return PQD_ERR_DEVICE;
    Lease lease(ctx);
    Workspace* ws = lease.ws;
    if (ws == nullptr) return PQD_ERR_DEVICE;
    pqd_encode_page p;
    std::memset(&p, 0, sizeof p);
    p.n_values = n_values; p.dst_cap = dst_cap;
    p.encoding = encoding; p.kind = kind; p.bit_width = bit_width;
    const uint64_t src_bytes = n_values * width, scratch = pqd_encode_workspace_size(&p, 1);
    const size_t o_src = 0, o_dst = pad256(src_bytes + 32), o_ws = o_dst + pad256(dst_cap + 32);
    if (!ws->reserve_io(o_ws + scratch + 256)) return PQD_ERR_DEVICE;
    uint8_t* d_io = static_cast<uint8_t*>(ws->d_io);
    uint8_t* h_io = static_cast<uint8_t*>(ws->h_io);
    if (src_bytes) {
        std::memcpy(h_io + o_src, src, src_bytes);
        PQD_HIP(hipMemcpyAsync(d_io + o_src, h_io + o_src, src_bytes, hipMemcpyHostToDevice, ws->stream));
    }
    p.src = d_io + o_src;
    p.dst = d_io + o_dst;
    int rc = pqd_encode_batch(ctx, &p, 1, d_io + o_ws, scratch + 256, ws->stream);
    if (rc != PQD_OK) return rc;
    if (status_arg) *status_arg = p.status_arg;
    if (p.status == PQD_OK && p.out_bytes) {
        PQD_HIP(hipMemcpyAsync(h_io + o_dst, d_io + o_dst, p.out_bytes, hipMemcpyDeviceToHost, ws->stream));
        PQD_HIP(hipStreamSynchronize(ws->stream));
        std::memcpy(dst, h_io + o_dst, p.out_bytes);
    }
    if (out_bytes) *out_bytes = p.status == PQD_OK ? p.out_bytes : 0;
    return p.status;
}

int pqd_encode_byte_array(pqd_ctx* ctx, int encoding, int kind, int type_length, const void* src, uint64_t src_bytes,
                          const uint32_t* offsets, uint64_t n_values, uint8_t* dst, uint64_t dst_cap, uint64_t* out_bytes,
                          int64_t* status_arg) {
    if (out_bytes) *out_bytes = 0;
    if (status_arg) *status_arg = 0;
    if (ctx == nullptr || (src == nullptr && src_bytes > 0) || (dst == nullptr && dst_cap > 0)) return PQD_ERR_INVALID_ARGUMENT;
    DeviceGuard guard(ctx->device);
    if (!guard.ok) return PQD_ERR_DEVICE;
    Lease lease(ctx);
    Workspace* ws = lease.ws;
    if (ws == nullptr) return PQD_ERR_DEVICE;
    pqd_encode_page p;
    std::memset(&p, 0, sizeof p);
    p.n_values = n_values; p.dst_cap = dst_cap; p.src_bytes = src_bytes;
    p.encoding = encoding; p.kind = kind; p.bit_width = type_length;
    const uint64_t off_bytes = offsets ? (n_values + 1) * 4 : 0;
    p.offsets = offsets ? reinterpret_cast<const uint32_t*>(8) : nullptr; // (a stand-in while the workspace is sized: only its being there counts)
    const uint64_t scratch = pqd_encode_workspace_size(&p, 1);
    const size_t o_src = 0, o_off = pad256(src_bytes + 32), o_dst = o_off + pad256(off_bytes + 32), o_ws = o_dst + pad256(dst_cap + 32);
    if (!ws->reserve_io(o_ws + scratch + 256)) return PQD_ERR_DEVICE;
    uint8_t* d_io = static_cast<uint8_t*>(ws->d_io);
    uint8_t* h_io = static_cast<uint8_t*>(ws->h_io);
    if (src_bytes) std::memcpy(h_io + o_src, src, src_bytes);
    if (off_bytes) std::memcpy(h_io + o_off, offsets, off_bytes);
    if (src_bytes) PQD_HIP(hipMemcpyAsync(d_io + o_src, h_io + o_src, src_bytes, hipMemcpyHostToDevice, ws->stream));
    if (off_bytes) PQD_HIP(hipMemcpyAsync(d_io + o_off, h_io + o_off, off_bytes, hipMemcpyHostToDevice, ws->stream));
    p.src = d_io + o_src;
    p.offsets = offsets ? reinterpret_cast<const uint32_t*>(d_io + o_off) : nullptr;
    p.dst = d_io + o_dst;
    int rc = pqd_encode_batch(ctx, &p, 1, d_io + o_ws, scratch + 256, ws->stream);
    if (rc != PQD_OK) return rc;
    if (status_arg) *status_arg = p.status_arg;
    if (p.status == PQD_OK && p.out_bytes) {
        PQD_HIP(hipMemcpyAsync(h_io + o_dst, d_io + o_dst, p.out_bytes, hipMemcpyDeviceToHost, ws->stream));
        PQD_HIP(hipStreamSynchronize(ws->stream));
        std::memcpy(dst, h_io + o_dst, p.out_bytes);
    }
    if (out_bytes) *out_bytes = p.status == PQD_OK ? p.out_bytes : 0;
    return p.status;
}

int pqd_host_register(pqd_ctx* ctx, void* ptr, uint64_t bytes) {
    if (ctx == nullptr || ptr == nullptr || bytes == 0) return PQD_ERR_INVALID_ARGUMENT;
    DeviceGuard guard(ctx->device);
    if (!guard.ok) return PQD_ERR_DEVICE;
    if (hipHostRegister(ptr, bytes, hipHostRegisterDefault) != hipSuccess) {
        (void)hipGetLastError();
        return PQD_ERR_DEVICE;
    }
    std::lock_guard<std::mutex> lock(ctx->reg_mu);
    ctx->registered.push_back({reinterpret_cast<uintptr_t>(ptr), bytes, false});
    return PQD_OK;
}

int pqd_host_alloc(pqd_ctx* ctx, uint64_t bytes, void** out) {
    if (ctx == nullptr || out == nullptr || bytes == 0) return PQD_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    DeviceGuard guard(ctx->device);
    if (!guard.ok) return PQD_ERR_DEVICE;
    void* p = nullptr;
    if (hipHostMalloc(&p, bytes, hipHostMallocDefault) != hipSuccess) {
        (void)hipGetLastError();
        return PQD_ERR_DEVICE;
    }
    std::lock_guard<std::mutex> lock(ctx->reg_mu);
    ctx->registered.push_back({reinterpret_cast<uintptr_t>(p), bytes, true});
    *out = p;
    return PQD_OK;
}

// Takes the range that starts at ptr off the list; `owned` says which of the two calls put it there.
static bool drop_host_range(pqd_ctx* ctx, void* ptr, bool owned) {
    std::lock_guard<std::mutex> lock(ctx->reg_mu);
    auto it = ctx->registered.begin();
    while (it != ctx->registered.end() && !(it->first == reinterpret_cast<uintptr_t>(ptr) && it->owned == owned)) ++it;
    if (it == ctx->registered.end()) return false;
    ctx->registered.erase(it);
    return true;
}

int pqd_host_free(pqd_ctx* ctx, void* ptr) {
    if (ctx == nullptr || ptr == nullptr) return PQD_ERR_INVALID_ARGUMENT;
    if (!drop_host_range(ctx, ptr, true)) return PQD_ERR_INVALID_ARGUMENT;
    DeviceGuard guard(ctx->device);
    return guard.ok && hipHostFree(ptr) == hipSuccess ? PQD_OK : PQD_ERR_DEVICE;
}

int pqd_host_unregister(pqd_ctx* ctx, void* ptr) {
    if (ctx == nullptr || ptr == nullptr) return PQD_ERR_INVALID_ARGUMENT;
    if (!drop_host_range(ctx, ptr, false)) return PQD_ERR_INVALID_ARGUMENT;
    DeviceGuard guard(ctx->device);
    return guard.ok && hipHostUnregister(ptr) == hipSuccess ? PQD_OK : PQD_ERR_DEVICE;
}

int pqd_can_decode_in_place(int encoding) {
    return encoding == PQD_ENC_PLAIN || encoding == PQD_ENC_DELTA_LENGTH_BYTE_ARRAY;
}

int pqd_dict_create(pqd_ctx* ctx, const void* values, uint64_t n, uint32_t width, pqd_dict** out) {
    if (ctx == nullptr || out == nullptr || (values == nullptr && n > 0)) return PQD_ERR_INVALID_ARGUMENT;
    if (width != 4 && width != 8 && width != 12) return PQD_ERR_INVALID_ARGUMENT;
    *out = nullptr;
    DeviceGuard guard(ctx->device);
    if (!guard.ok) return PQD_ERR_DEVICE;
    pqd_dict* d = new (std::nothrow) pqd_dict;
    if (d == nullptr) return PQD_ERR_DEVICE;
    d->ctx = ctx; d->n = n; d->width = width; d->values = nullptr;
    if (hipMalloc(&d->values, n * width + 16) != hipSuccess) { delete d; return PQD_ERR_DEVICE; }
    if (n && hipMemcpy(d->values, values, n * width, hipMemcpyHostToDevice) != hipSuccess) {
        (void)hipFree(d->values); delete d; return PQD_ERR_DEVICE;
    }
    *out = d;
    return PQD_OK;
}

void pqd_dict_destroy(pqd_dict* d) {
    if (d == nullptr) return;
    DeviceGuard guard(d->ctx->device);
    if (d->values) (void)hipFree(d->values);
    delete d;
}

int pqd_dict_lookup_device(pqd_ctx* ctx, const void* dict_dev, uint64_t dict_len, uint32_t width, const int32_t* indexes_dev,
                           uint64_t n, void* rows_dev, uint64_t stride, int64_t* status_dev, void* hip_stream) {
    if (ctx == nullptr || status_dev == nullptr) return PQD_ERR_INVALID_ARGUMENT;
    DeviceGuard guard(ctx->device);
    if (!guard.ok) return PQD_ERR_DEVICE;
    return launch_dict_gather(dict_dev, dict_len, width, indexes_dev, n, rows_dev, stride, status_dev,
                              static_cast<hipStream_t>(hip_stream));
}

int pqd_dict_lookup(pqd_dict* d, const int32_t* indexes, uint64_t n, void* rows, uint64_t stride, int64_t* status_arg) {
    if (status_arg) *status_arg = 0;
    if (d == nullptr || (indexes == nullptr && n > 0) || (rows == nullptr && n > 0)) return PQD_ERR_INVALID_ARGUMENT;
    if (stride == 0) stride = d->width;
    if (stride < d->width) return PQD_ERR_INVALID_ARGUMENT;
    DeviceGuard guard(d->ctx->device);
    if (!guard.ok) return PQD_ERR_DEVICE;
    Lease lease(d->ctx);
    Workspace* ws = lease.ws;
    if (ws == nullptr) return PQD_ERR_DEVICE;
    const size_t o_idx = 0, o_rows = pad256(n * 4), o_st = o_rows + pad256(n * d->width);
    if (!ws->reserve_io(o_st + 256)) return PQD_ERR_DEVICE;
    uint8_t* d_io = static_cast<uint8_t*>(ws->d_io);
    uint8_t* h_io = static_cast<uint8_t*>(ws->h_io);
    if (n) {
        std::memcpy(h_io + o_idx, indexes, n * 4);
        PQD_HIP(hipMemcpyAsync(d_io + o_idx, h_io + o_idx, n * 4, hipMemcpyHostToDevice, ws->stream));
    }
    int rc = launch_dict_gather(d->values, d->n, d->width, reinterpret_cast<const int32_t*>(d_io + o_idx), n, d_io + o_rows,
                                d->width, reinterpret_cast<int64_t*>(d_io + o_st), ws->stream);
    if (rc != PQD_OK) { (void)hipStreamSynchronize(ws->stream); return rc; }
    // dense gather on the device, strided placement on the host (rows live in host memory here;
    // the device-pointer variant scatters in the kernel)
    PQD_HIP(hipMemcpyAsync(h_io + o_rows, d_io + o_rows, n * d->width + (o_st - o_rows - n * d->width) + 16, hipMemcpyDeviceToHost, ws->stream));
    PQD_HIP(hipStreamSynchronize(ws->stream));
    int64_t st[2];
    std::memcpy(st, h_io + o_st, 16);
    if (st[0] != PQD_OK) { if (status_arg) *status_arg = st[1]; return (int)st[0]; }
    uint8_t* r = static_cast<uint8_t*>(rows);
    const uint8_t* dense = h_io + o_rows;
    if (stride == d->width) std::memcpy(r, dense, n * d->width);
    else for (uint64_t i = 0; i < n; i++) std::memcpy(r + i * stride, dense + i * d->width, d->width);
    return PQD_OK;
}

} // extern "C"
