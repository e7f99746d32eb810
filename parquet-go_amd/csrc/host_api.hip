// host_api.hip -- per-method host-pointer wrappers of include/pqd.h (the surface a
// cgo shim binds, one call per Decode method of encoding.Encoding) and the
// dictionary handle.  Each wrapper stages the page in device memory, runs the
// same single-item batch the device-resident path runs, and copies the result
// back; there is no CPU decode path.
#include <hip/hip_runtime.h>

#include <cstring>
#include <new>
#include <vector>

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

struct DevMem {
    void* p = nullptr;
    ~DevMem() { if (p) (void)hipFree(p); }
    bool alloc(size_t n) { return hipMalloc(&p, n ? n : 16) == hipSuccess; }
};

// One page, host pointers in and out, through the batched device path.
int decode_host(pqd_ctx* ctx, int encoding, int kind, int bit_width, const uint8_t* src, uint64_t src_len, void* dst,
                uint64_t dst_cap, uint32_t* offsets, uint64_t offsets_cap, uint64_t* out_bytes, uint64_t* out_offsets,
                int64_t* status_arg) {
    if (out_bytes) *out_bytes = 0;
    if (out_offsets) *out_offsets = 0;
    if (status_arg) *status_arg = 0;
    if (ctx == nullptr || (src == nullptr && src_len > 0) || (dst == nullptr && dst_cap > 0)) return PQD_ERR_INVALID_ARGUMENT;
    DeviceGuard guard(ctx->device);
    if (!guard.ok) return PQD_ERR_DEVICE;
    DevMem dsrc, ddst, doff;
    // 16 spare bytes so the kernels' aligned 16-byte loads of the last bytes stay inside the allocation
    if (!dsrc.alloc(src_len + 32) || !ddst.alloc(dst_cap + 32) || !doff.alloc(offsets_cap * 4 + 32)) return PQD_ERR_DEVICE;
    if (src_len && hipMemcpy(dsrc.p, src, src_len, hipMemcpyHostToDevice) != hipSuccess) return PQD_ERR_DEVICE;
    pqd_page p;
    std::memset(&p, 0, sizeof p);
    p.src = dsrc.p; p.src_len = src_len;
    p.dst = ddst.p; p.dst_cap = dst_cap;
    p.offsets = static_cast<uint32_t*>(doff.p); p.offsets_cap = offsets_cap;
    p.num_values = -1;
    p.encoding = encoding; p.kind = kind; p.bit_width = bit_width;
    int rc = pqd_decode_batch(ctx, &p, 1, nullptr);
    if (rc != PQD_OK) return rc;
    if (status_arg) *status_arg = p.status_arg;
    uint64_t nb = p.out_bytes < dst_cap ? p.out_bytes : dst_cap;
    if (nb && hipMemcpy(dst, ddst.p, nb, hipMemcpyDeviceToHost) != hipSuccess) return PQD_ERR_DEVICE;
    if (out_bytes) *out_bytes = nb;
    if (offsets != nullptr && kind == PQD_KIND_BYTE_ARRAY) {
        uint64_t no = p.status == PQD_OK ? p.out_values + 1 : p.out_values;
        if (no > offsets_cap) no = offsets_cap;
        if (no && hipMemcpy(offsets, doff.p, no * 4, hipMemcpyDeviceToHost) != hipSuccess) return PQD_ERR_DEVICE;
        if (out_offsets) *out_offsets = no;
    }
    return p.status;
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
    // byte_array.go:176-183): the streams are decoded and summed on the device.
    if (encoding == PQD_ENC_PLAIN) return (int64_t)src_len;
    if (encoding != PQD_ENC_DELTA_LENGTH_BYTE_ARRAY && encoding != PQD_ENC_DELTA_BYTE_ARRAY) return 0;
    if (ctx == nullptr || (src == nullptr && src_len > 0)) return 0;
    DeviceGuard guard(ctx->device);
    if (!guard.ok) return 0;
    DevMem dsrc, dsum;
    if (!dsrc.alloc(src_len + 32) || !dsum.alloc(16)) return 0;
    if (src_len && hipMemcpy(dsrc.p, src, src_len, hipMemcpyHostToDevice) != hipSuccess) return 0;
    const int streams = encoding == PQD_ENC_DELTA_BYTE_ARRAY ? 2 : 1;
    int64_t total = 0;
    uint64_t at = 0;
    for (int k = 0; k < streams; k++) {
        uint64_t cap = 4096;
        pqd_page p;
        DevMem dlen;
        for (int attempt = 0; attempt < 2; attempt++) {
            if (dlen.p) { (void)hipFree(dlen.p); dlen.p = nullptr; }
            if (!dlen.alloc(cap + 32)) return 0;
            std::memset(&p, 0, sizeof p);
            p.src = static_cast<uint8_t*>(dsrc.p) + at; p.src_len = src_len - at;
            p.dst = dlen.p; p.dst_cap = cap;
            p.num_values = -1;
            p.encoding = PQD_ENC_DELTA_BINARY_PACKED; p.kind = PQD_KIND_INT32;
            if (pqd_decode_batch(ctx, &p, 1, nullptr) != PQD_OK) return 0;
            if (p.status != PQD_ERR_DST_TOO_SMALL) break;
            cap = (uint64_t)p.status_arg;
        }
        if (p.status == PQD_ERR_DST_TOO_SMALL) return 0;
        uint32_t sum = 0;
        if (hipMemset(dsum.p, 0, 4) != hipSuccess) return 0;
        if (p.out_values > 0) {
            launch_sum_lengths(static_cast<const int32_t*>(dlen.p), p.out_values, static_cast<uint32_t*>(dsum.p), nullptr);
            if (hipMemcpy(&sum, dsum.p, 4, hipMemcpyDeviceToHost) != hipSuccess) return 0;
        }
        total += (int64_t)(int32_t)sum;
        at += p.consumed < src_len - at ? p.consumed : src_len - at;
    }
    return total;
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
    DevMem didx, drows, dst;
    if (!didx.alloc(n * 4) || !drows.alloc(n * d->width) || !dst.alloc(16)) return PQD_ERR_DEVICE;
    if (n && hipMemcpy(didx.p, indexes, n * 4, hipMemcpyHostToDevice) != hipSuccess) return PQD_ERR_DEVICE;
    int rc = launch_dict_gather(d->values, d->n, d->width, static_cast<const int32_t*>(didx.p), n, drows.p, d->width,
                                static_cast<int64_t*>(dst.p), nullptr);
    if (rc != PQD_OK) return rc;
    int64_t st[2] = {0, 0};
    if (hipMemcpy(st, dst.p, 16, hipMemcpyDeviceToHost) != hipSuccess) return PQD_ERR_DEVICE;
    if (st[0] != PQD_OK) { if (status_arg) *status_arg = st[1]; return (int)st[0]; }
    // dense gather on the device, strided placement on the host (rows live in
    // host memory here; the device-pointer variant scatters in the kernel)
    std::vector<uint8_t> dense(n * d->width);
    if (n && hipMemcpy(dense.data(), drows.p, n * d->width, hipMemcpyDeviceToHost) != hipSuccess) return PQD_ERR_DEVICE;
    uint8_t* r = static_cast<uint8_t*>(rows);
    for (uint64_t i = 0; i < n; i++) std::memcpy(r + i * stride, dense.data() + i * d->width, d->width);
    return PQD_OK;
}

} // extern "C"
