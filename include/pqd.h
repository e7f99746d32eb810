/*
 * pqd.h -- C ABI of the MI355X (gfx950) Parquet page-value decode library.
 *
 * This is the boundary a Go shim binds with cgo to stand behind parquet-go's
 * codec interface: plain pointers and sizes, no C++ or torch types.  Each
 * entry point names the reference interface it replaces (paths relative to
 * the parquet-go source tree).
 *
 * Two levels:
 *   1. a batched, device-resident interface (pqd_batch_*): many pages per
 *      launch, all pointers are device pointers, nothing is copied or
 *      allocated between create and finish.  This is the measured path.
 *   2. per-method host wrappers (pqd_decode_*): host pointers in, host
 *      pointers out, one call per reference Decode method; they stage through
 *      device memory and run the same kernels.
 *
 * Every decode kernel is hand-written HIP for gfx950; there is no CPU
 * fallback: without a usable GPU every call returns PQD_ERR_DEVICE.
 */
#ifndef PQD_H
#define PQD_H

#include <stddef.h>
#include <stdint.h>
#include "pqd_status.h"

#ifdef __cplusplus
extern "C" {
#endif

#define PQD_ABI_VERSION 2

typedef struct pqd_ctx pqd_ctx;     /* one per process and device */
typedef struct pqd_batch pqd_batch; /* a prepared set of pages */
typedef struct pqd_dict pqd_dict;   /* a device-resident dictionary */

/* ---- context ----------------------------------------------------------- */

int pqd_abi_version(void);
const char* pqd_status_name(int status);

/* Binds the library to a HIP device.  Safe to call from several threads;
 * contexts are independent ("Encoding instances must be safe to use
 * concurrently", encoding/encoding.go:19).  With PQD_SIDE_BY_SIDE=1 in the
 * environment the context owns three side streams, and a batch of at most 1024
 * items forks its kernel families onto them: worth it for a caller that
 * launches small batches on one stream, a loss for one that already deals
 * them to several streams (all batches of a context share the side streams). */
int pqd_ctx_create(int device, pqd_ctx** out);
void pqd_ctx_destroy(pqd_ctx* ctx);

/* ---- batched, device-resident decode ------------------------------------ */

/*
 * One page of one column chunk: the unit Column.decodeDataPage hands to
 * Type.Decode (column.go:845-911, :881), Column.decodeLevels hands to
 * Encoding.DecodeLevels (column.go:931-948) or Column.decodeDictionary hands
 * to the PLAIN decoder (column.go:963-993).
 *
 * `encoding` is the format.Encoding code, `kind` selects which Decode method
 * of encoding.Encoding the item stands for (encoding/encoding.go:48-56).
 * `bit_width` is Encoding.BitWidth for RLE / BIT_PACKED
 * (encoding/rle/rle.go:30-33, encoding/bitpacked/bitpacked.go:8-11) and the
 * type length for FIXED_LEN_BYTE_ARRAY.
 *
 * With a non-null `dict` an RLE_DICTIONARY / PLAIN_DICTIONARY page is decoded
 * straight to values: index unpack, the newIndexedPage length fix-up
 * (dictionary.go:136-162: cut to num_values or zero-extend) and
 * Dictionary.Lookup (dictionary.go:64-69, dictionary_amd64.go:36-58) fused in
 * one kernel; `kind` then names the value type (INT32/INT64/FLOAT/DOUBLE/INT96)
 * and `dst_stride` the distance between rows in bytes (0 = dense;
 * sparse.Array's stride, sparse/array.go:33-37).  Dictionaries of the other
 * types are looked up in the same pass and come out in the layout the Decode
 * methods produce (the reference's lookupString hands out string headers that
 * point into the dictionary, dictionary_purego.go:40-52): kind BYTE_ARRAY with
 * `dict` = the dictionary's value bytes and `dict_offsets` = its dict_len + 1
 * offsets (what PLAIN's DecodeByteArray leaves of the dictionary page) yields
 * values in `dst` and num_values + 1 `offsets`; FIXED_LEN_BYTE_ARRAY of any
 * `bit_width` yields dense values; BOOLEAN with `dict` = the dictionary's
 * packed bits yields one byte, 0 or 1, per value (booleanDictionary.lookup,
 * dictionary_boolean.go:93-98).
 *
 * `num_values` >= 0 cuts the output to that many values the way the callers
 * do (column.go:940-945, dictionary.go:136-162); -1 returns everything the
 * stream holds, as the Decode methods themselves do.
 */
typedef struct pqd_page {
    const void* src;      /* encoded page bytes */
    uint64_t    src_len;
    void*       dst;      /* decoded values */
    uint64_t    dst_cap;  /* bytes */
    uint32_t*   offsets;  /* BYTE_ARRAY: n+1 value offsets */
    uint64_t    offsets_cap; /* entries */
    const void* dict;     /* optional dictionary values (device) */
    uint64_t    dict_len; /* entries */
    const uint32_t* dict_offsets; /* BYTE_ARRAY dictionaries: dict_len + 1 offsets into dict (device) */
    int64_t     num_values;
    int32_t     encoding; /* pqd_encoding */
    int32_t     kind;     /* pqd_kind */
    int32_t     bit_width;
    int32_t     dst_stride;
    /* results, filled by pqd_batch_finish */
    int32_t     status;      /* pqd_status */
    int32_t     reserved;
    int64_t     status_arg;
    uint64_t    out_values;  /* values decoded (offsets written - 1 for BYTE_ARRAY) */
    uint64_t    out_bytes;   /* bytes written to dst */
    uint64_t    consumed;    /* source bytes consumed (DELTA streams) */
} pqd_page;

/* Validates the items, groups them by kernel family, uploads the descriptors
 * and reserves scratch (run tables, length buffers).  All device pointers
 * must stay valid until the batch is destroyed. */
int pqd_batch_create(pqd_ctx* ctx, const pqd_page* pages, uint64_t n_pages, pqd_batch** out);

/* Enqueues every kernel of the batch on `hip_stream` (a hipStream_t, NULL =
 * the default stream).  Does not allocate, copy or synchronise; may be called
 * any number of times. */
int pqd_batch_launch(pqd_batch* batch, void* hip_stream);

/* Waits for the stream and writes the result fields of pages[0..n). */
int pqd_batch_finish(pqd_batch* batch, void* hip_stream, pqd_page* pages, uint64_t n_pages);

void pqd_batch_destroy(pqd_batch* batch);

/* create + launch + finish + destroy. */
int pqd_decode_batch(pqd_ctx* ctx, pqd_page* pages, uint64_t n_pages, void* hip_stream);

/* ---- per-method host wrappers (the cgo surface) ---------------------------- */

/*
 * Each wrapper replaces one Decode method of encoding.Encoding
 * (encoding/encoding.go:48-56) for the encoding code given: host src, host
 * dst of dst_cap bytes; *out_bytes bytes were produced.  On
 * PQD_ERR_DST_TOO_SMALL *status_arg holds the capacity to retry with, which is
 * how the shim renders Go's "may grow dst" contract (encoding/rle/rle.go:510-522).
 */
int pqd_decode_levels(pqd_ctx* ctx, int encoding, int bit_width, const uint8_t* src, uint64_t src_len,
                      uint8_t* dst, uint64_t dst_cap, uint64_t* out_bytes, int64_t* status_arg);
int pqd_decode_boolean(pqd_ctx* ctx, int encoding, const uint8_t* src, uint64_t src_len,
                       uint8_t* dst, uint64_t dst_cap, uint64_t* out_bytes, int64_t* status_arg);
int pqd_decode_int32(pqd_ctx* ctx, int encoding, int bit_width, const uint8_t* src, uint64_t src_len,
                     int32_t* dst, uint64_t dst_cap, uint64_t* out_bytes, int64_t* status_arg);
int pqd_decode_int64(pqd_ctx* ctx, int encoding, const uint8_t* src, uint64_t src_len,
                     int64_t* dst, uint64_t dst_cap, uint64_t* out_bytes, int64_t* status_arg);
int pqd_decode_int96(pqd_ctx* ctx, int encoding, const uint8_t* src, uint64_t src_len,
                     uint8_t* dst, uint64_t dst_cap, uint64_t* out_bytes, int64_t* status_arg);
int pqd_decode_float(pqd_ctx* ctx, int encoding, const uint8_t* src, uint64_t src_len,
                     float* dst, uint64_t dst_cap, uint64_t* out_bytes, int64_t* status_arg);
int pqd_decode_double(pqd_ctx* ctx, int encoding, const uint8_t* src, uint64_t src_len,
                      double* dst, uint64_t dst_cap, uint64_t* out_bytes, int64_t* status_arg);
int pqd_decode_byte_array(pqd_ctx* ctx, int encoding, const uint8_t* src, uint64_t src_len,
                          uint8_t* dst, uint64_t dst_cap, uint32_t* offsets, uint64_t offsets_cap,
                          uint64_t* out_bytes, uint64_t* out_offsets, int64_t* status_arg);
int pqd_decode_fixed_len_byte_array(pqd_ctx* ctx, int encoding, int size, const uint8_t* src, uint64_t src_len,
                                    uint8_t* dst, uint64_t dst_cap, uint64_t* out_bytes, int64_t* status_arg);

/* Encoding.EstimateDecodeByteArraySize (encoding/encoding.go:58-67;
 * encoding/delta/length_byte_array.go:65-70, byte_array.go:176-183,
 * encoding/plain/plain.go:106-108). Never fails: 0 on any error. */
int64_t pqd_estimate_decode_byte_array_size(pqd_ctx* ctx, int encoding, const uint8_t* src, uint64_t src_len);

/* Encoding.CanDecodeInPlace (encoding/encoding.go:69-71): PLAIN and
 * DELTA_LENGTH_BYTE_ARRAY (plain.go:110-112, length_byte_array.go:72-74). */
int pqd_can_decode_in_place(int encoding);

/* ---- dictionary surface ---------------------------------------------------- */

/* A dictionary of `n` fixed-width values (width 4, 8 or 12 bytes) copied to
 * the device: what Type.NewDictionary keeps after decodeDictionary
 * (column.go:992, dictionary_int64.go:18-26). */
int pqd_dict_create(pqd_ctx* ctx, const void* values, uint64_t n, uint32_t width, pqd_dict** out);
void pqd_dict_destroy(pqd_dict* dict);

/* Dictionary.Lookup (dictionary.go:64-69; dictionaryLookup32/64,
 * dictionary_amd64.go:30-46, dictionary_amd64.s:578-711):
 * rows[i*stride .. +width) = dict[indexes[i]], host pointers.  An index outside
 * [0, n) yields PQD_ERR_INDEX_OUT_OF_BOUNDS (the reference panics). */
int pqd_dict_lookup(pqd_dict* dict, const int32_t* indexes, uint64_t n, void* rows, uint64_t stride,
                    int64_t* status_arg);

/* The same on device pointers, enqueued on `hip_stream`; `status_dev` points to
 * two int64 in device memory: {status, arg}, written by the kernel. */
int pqd_dict_lookup_device(pqd_ctx* ctx, const void* dict_dev, uint64_t dict_len, uint32_t width,
                           const int32_t* indexes_dev, uint64_t n, void* rows_dev, uint64_t stride,
                           int64_t* status_dev, void* hip_stream);

/* Dictionary.Lookup for BYTE_ARRAY, FIXED_LEN_BYTE_ARRAY (any `width`) and
 * BOOLEAN dictionaries (byteArrayDictionary / fixedLenByteArrayDictionary
 * lookupString, dictionary_purego.go:40-52; booleanDictionary.lookup,
 * dictionary_boolean.go:93-98) on device pointers, enqueued on `hip_stream`.
 * The values come out as the Decode methods lay them out: BYTE_ARRAY bytes
 * back to back in dst with n + 1 offsets in offsets_dev, FIXED_LEN_BYTE_ARRAY
 * dense, BOOLEAN (dictionary = packed bits) one byte per value.  `status_dev`
 * points to three int64 in device memory: {status, arg, bytes written}:
 * PQD_ERR_INDEX_OUT_OF_BOUNDS with the position of the first bad index,
 * PQD_ERR_DST_TOO_SMALL / PQD_ERR_OFFSETS_TOO_SMALL with the size to come back
 * with (nothing is written to dst then). */
int pqd_dict_lookup_bytes_device(pqd_ctx* ctx, int32_t kind, const void* dict_dev, const uint32_t* dict_offsets_dev,
                                 uint64_t dict_len, uint32_t width, const int32_t* indexes_dev, uint64_t n, void* dst_dev,
                                 uint64_t dst_cap, uint32_t* offsets_dev, uint64_t offsets_cap, int64_t* status_dev,
                                 void* hip_stream);

/* ---- encoders and page statistics, device-resident ------------------------ */

/*
 * One page to encode: `n_values` values at `src` (int32 / int64 / float / double, or
 * n_values * bit_width bytes of FIXED_LEN_BYTE_ARRAY values) become the page bytes the
 * reference's Encode methods produce (encoding/encoding.go:29-41), byte for byte:
 *   PQD_ENC_DELTA_BINARY_PACKED, kind INT32 / INT64: blocks of 128 values in 4 miniblocks
 *     (delta.BinaryPackedEncoding.EncodeInt32 / EncodeInt64, encoding/delta/binary_packed.go:26-34, 77-277);
 *   PQD_ENC_RLE_DICTIONARY, kind INT32: a bit-width byte and the RLE / bit-packed hybrid of the
 *     dictionary indexes (rle.DictionaryEncoding.EncodeInt32, encoding/rle/dictionary.go:23-28, 52-59,
 *     encoding/rle/rle.go:203-249);
 *   PQD_ENC_BYTE_STREAM_SPLIT, kind INT32 / INT64 / FLOAT / DOUBLE / FIXED_LEN_BYTE_ARRAY
 *     (encoding/bytestreamsplit/bytestreamsplit.go:19-33, bytestreamsplit_amd64.s:56-208);
 *   PQD_ENC_RLE with bit_width: kind LEVELS, n_values bytes of levels (rle.Encoding.EncodeLevels, encodeBytes,
 *     encoding/rle/rle.go:43-46, 141-201); kind BOOLEAN, n_values BYTES of packed bits, the 4-byte length
 *     in front (EncodeBoolean, encodeBits, rle.go:48-56, 97-139); kind INT32 (EncodeInt32, rle.go:58-61, 203-249);
 *     a width the encoder refuses (above 8 / 32, or 0 with a value that is not 0): PQD_ERR_INVALID_BIT_WIDTH;
 *   PQD_ENC_PLAIN, kind BYTE_ARRAY: every value behind its 4-byte length (plain.Encoding.EncodeByteArray,
 *     encoding/plain/plain.go:44-57);
 *   PQD_ENC_DELTA_LENGTH_BYTE_ARRAY, kind BYTE_ARRAY: the lengths as DELTA_BINARY_PACKED int32, then all of
 *     src (delta.LengthByteArrayEncoding.EncodeByteArray, encoding/delta/length_byte_array.go:20-35);
 *   PQD_ENC_DELTA_BYTE_ARRAY, kind BYTE_ARRAY / FIXED_LEN_BYTE_ARRAY: prefix lengths, suffix lengths, suffixes
 *     (delta.ByteArrayEncoding.EncodeByteArray / EncodeFixedLenByteArray, encoding/delta/byte_array.go:26-122).
 * Other combinations answer PQD_ERR_NOT_SUPPORTED, as encoding.NotSupported does.
 * All pointers are device pointers; `src` is aligned to the value width (4 or 8 bytes) except
 * for BYTE_STREAM_SPLIT, `dst` may have any alignment.  PQD_ERR_DST_TOO_SMALL: status_arg = bytes
 * the page needs.
 */
typedef struct pqd_encode_page {
    const void* src;
    uint64_t    n_values;
    void*       dst;
    uint64_t    dst_cap;
    int32_t     encoding;  /* pqd_encoding */
    int32_t     kind;      /* pqd_kind */
    int32_t     bit_width; /* FIXED_LEN_BYTE_ARRAY: the type length */
    /* results */
    int32_t     status;
    int64_t     status_arg;
    uint64_t    out_bytes;
    /* BYTE_ARRAY pages: value k is src[offsets[k] .. offsets[k + 1]) (n_values + 1 device offsets, as
     * encoding.Values holds them, encoding/values.go:47-52) and src holds src_bytes bytes;
     * FIXED_LEN_BYTE_ARRAY through DELTA_BYTE_ARRAY: offsets is NULL, src_bytes = n_values * bit_width */
    const uint32_t* offsets;
    uint64_t    src_bytes;
} pqd_encode_page;

/* The largest page `n_values` values can encode to (0: not supported, or a byte-array page). */
uint64_t pqd_encode_bound(int encoding, int kind, int bit_width, uint64_t n_values);
/* The same for BYTE_ARRAY / FIXED_LEN_BYTE_ARRAY pages of `src_bytes` bytes of values in all
 * (PLAIN, DELTA_LENGTH_BYTE_ARRAY, DELTA_BYTE_ARRAY). */
uint64_t pqd_encode_bound_byte_array(int encoding, uint64_t n_values, uint64_t src_bytes);

/* Bytes of device scratch pqd_encode_batch needs for these pages (src, dst are not looked at). */
uint64_t pqd_encode_workspace_size(const pqd_encode_page* pages, uint64_t n_pages);

/* Encodes the pages on `hip_stream`, waits, and fills their result fields.  `workspace_dev` is
 * caller-owned device scratch of at least pqd_encode_workspace_size bytes; nothing is allocated. */
int pqd_encode_batch(pqd_ctx* ctx, pqd_encode_page* pages, uint64_t n_pages, void* workspace_dev,
                     uint64_t workspace_bytes, void* hip_stream);

/* One page, host pointers in and out: the Encode method of encoding.Encoding that `encoding`
 * and `kind` name.  Go's "may grow dst" is PQD_ERR_DST_TOO_SMALL with the size in *status_arg. */
int pqd_encode_values(pqd_ctx* ctx, int encoding, int kind, int bit_width, const void* src, uint64_t n_values,
                      uint8_t* dst, uint64_t dst_cap, uint64_t* out_bytes, int64_t* status_arg);

/*
 * Host memory the copy engines may reach directly.  The per-method entry points above take plain host
 * pointers, as the reference's codecs take Go slices (encoding/encoding.go:43-56), and move the bytes
 * through pinned staging of their own -- for a page of 1 MiB those two memcpys are most of what a call
 * costs.  A reader whose page buffers come from a pool (buffer.go:476-561, column.go:863-871) makes the
 * pool's memory known once: calls whose src, dst or offsets lie inside a known range are copied from and to
 * where they lie.  Two ways: pqd_host_register page-locks memory the caller already has (it returns
 * PQD_ERR_DEVICE where the driver does not lock user memory, as under some virtual machines), and
 * pqd_host_alloc hands out page-locked memory of the library's (for cgo: memory outside the Go heap, which
 * the reference's buffer pool can wrap as it wraps its own slabs).  The memory stays locked until
 * pqd_host_unregister(ptr) / pqd_host_free(ptr) (the first byte of the range) or pqd_ctx_destroy; it must
 * outlive every call that uses it.  With a known dst the whole of dst[0 .. dst_cap) may be overwritten, not
 * only the out_bytes the call reports.
 */
int pqd_host_register(pqd_ctx* ctx, void* ptr, uint64_t bytes);
int pqd_host_unregister(pqd_ctx* ctx, void* ptr);
int pqd_host_alloc(pqd_ctx* ctx, uint64_t bytes, void** out);
int pqd_host_free(pqd_ctx* ctx, void* ptr);

/* One BYTE_ARRAY page, host pointers in and out: EncodeByteArray(dst, src, offsets) of plain.Encoding,
 * delta.LengthByteArrayEncoding and delta.ByteArrayEncoding (encoding/plain/plain.go:44-57,
 * encoding/delta/length_byte_array.go:20-35, encoding/delta/byte_array.go:26-74); `offsets` holds
 * n_values + 1 entries, or is NULL for no values at all.  With kind FIXED_LEN_BYTE_ARRAY, type_length and
 * offsets NULL: ByteArrayEncoding.EncodeFixedLenByteArray (byte_array.go:76-122). */
int pqd_encode_byte_array(pqd_ctx* ctx, int encoding, int kind, int type_length, const void* src, uint64_t src_bytes,
                          const uint32_t* offsets, uint64_t n_values, uint8_t* dst, uint64_t dst_cap, uint64_t* out_bytes,
                          int64_t* status_arg);

/* The smallest and the largest value of a page, its min / max statistics: boundsInt32 / Int64 / Uint32 /
 * Uint64 / Float32 / Float64 (page_bounds_purego.go:9-107, page_bounds_amd64.s; the typed pages' Bounds,
 * page_int32.go / page_int64.go / ...), with the loop's comparisons: of values that compare equal the first
 * stays, a NaN is only ever the answer when it is the first value.  Arguments as pqd_dict_bounds_device
 * without the dictionary: values_dev holds n values of `kind` (INT32, INT64, FLOAT, DOUBLE; is_unsigned for
 * the uint32 / uint64 pages), bounds_dev gets {min, max} (zeros when n = 0), status_dev {PQD_OK, 0}. */
int pqd_page_bounds_device(pqd_ctx* ctx, int32_t kind, int32_t is_unsigned, const void* values_dev, uint64_t n, void* bounds_dev,
                           void* workspace_dev, uint64_t workspace_bytes, int64_t* status_dev, void* hip_stream);

/* Dictionary.Bounds (dictionary.go:71-72): the smallest and the largest of dict[indexes[i]], the
 * min / max statistics of a dictionary-encoded page (dictionaryBoundsInt32 / Int64 / Float32 /
 * Float64 / Uint32 / Uint64, dictionary_amd64.go:13-28, dictionary_amd64.s:8-575), with the
 * comparisons of dictionary_purego.go:68-170: of values that compare equal the first stays, a NaN
 * is only ever the answer when it is the first value.  `kind` is INT32, INT64, FLOAT or DOUBLE,
 * `is_unsigned` selects the uint32 / uint64 dictionaries.  bounds_dev gets two values {min, max}
 * (zeros when n = 0); status_dev two int64 {status, arg}: PQD_ERR_INDEX_OUT_OF_BOUNDS with the
 * position of the first bad index (errno indexOutOfBounds).  workspace_dev: device scratch of
 * PQD_BOUNDS_WORKSPACE_BYTES, 16-byte aligned.  All pointers are device pointers; the kernels are
 * enqueued on `hip_stream`, nothing is allocated. */
#define PQD_BOUNDS_WORKSPACE_BYTES (64u << 10)
int pqd_dict_bounds_device(pqd_ctx* ctx, int32_t kind, int32_t is_unsigned, const void* dict_dev, uint64_t dict_len,
                           const int32_t* indexes_dev, uint64_t n, void* bounds_dev, void* workspace_dev,
                           uint64_t workspace_bytes, int64_t* status_dev, void* hip_stream);

/* ---- definition and repetition levels, device-resident ------------------ */

/*
 * What the reader does with the levels DecodeLevels produced, for callers that
 * keep a column on the device: all pointers are device pointers, the kernels
 * are enqueued on `hip_stream`, nothing is allocated.  `workspace_dev` is
 * caller-owned scratch of at least pqd_levels_workspace_size(n) bytes, 8-byte
 * aligned; calls that share a workspace must not overlap.
 */
uint64_t pqd_levels_workspace_size(uint64_t n_levels);

/* countLevelsEqual (level.go:11-13): *count_dev = levels equal to `value`.
 * countLevelsNotEqual (level.go:15-17) is n minus that: optionalPage.NumNulls
 * (page_optional.go:32-34) with value = the maximum definition level,
 * repeatedPage.NumRows (page_repeated.go:33) with value = 0, the null count of
 * Column.decodeDataPageV1 (column.go:772). */
int pqd_levels_count_equal_device(pqd_ctx* ctx, const uint8_t* levels_dev, uint64_t n, uint8_t value,
                                  void* workspace_dev, uint64_t workspace_bytes, uint64_t* count_dev,
                                  void* hip_stream);

/* optionalPageValues.ReadValues (page_optional.go:72-117) for a whole page:
 * slot i takes the next of the page's `n_values` values when
 * def_levels[i] == max_definition_level and is null otherwise.
 * dst[i*width .. +width) = that value, or zeros for a null slot;
 * validity_dev (may be null) gets 1 or 0 per slot.
 * offsets_mode != 0: `values_dev` holds the n_values + 1 uint32 offsets of a
 * byte-array page (width must be 4) and dst gets n_slots + 1 offsets, null
 * slots empty.
 * info_dev[0] = slots with a value, info_dev[1] = PQD_OK, or
 * PQD_ERR_UNEXPECTED_EOF when the levels ask for more values than the page has
 * (the io.EOF of page_optional.go:95-99); the slots past the last value are then
 * zero. */
int pqd_optional_expand_device(pqd_ctx* ctx, const uint8_t* def_levels_dev, uint64_t n_slots,
                               uint8_t max_definition_level, const void* values_dev, uint64_t n_values,
                               uint32_t width, int32_t offsets_mode, void* dst_dev, uint8_t* validity_dev,
                               void* workspace_dev, uint64_t workspace_bytes, int64_t* info_dev, void* hip_stream);

/*
 * pqd_optional_expand_device without a workspace argument, for callers that
 * cannot allocate one: code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc.  It has no counterpart in the reference.
 * The per-stretch counts live in a static device array of
 * PQD_NOALLOC_WORKSPACE_BYTES that every GPU the library runs on holds until
 * the process ends; the call allocates and frees nothing and launches the same
 * kernels as pqd_optional_expand_device, so the results are identical.
 * n_slots > PQD_NOALLOC_MAX_SLOTS is refused with PQD_ERR_DST_TOO_SMALL (use the
 * workspace variant).  All calls on one GPU share that array: two calls whose
 * kernels may overlap (different streams, different threads) race; serialise
 * them on one stream or with events.
 */
#define PQD_NOALLOC_WORKSPACE_BYTES (1ull << 20)
#define PQD_NOALLOC_MAX_SLOTS ((PQD_NOALLOC_WORKSPACE_BYTES / 8 - 1) * 2048ull)
int pqd_optional_expand_noalloc(pqd_ctx* ctx, const uint8_t* def_levels_dev, uint64_t n_slots,
                                uint8_t max_definition_level, const void* values_dev, uint64_t n_values,
                                uint32_t width, int32_t offsets_mode, void* dst_dev, uint8_t* validity_dev,
                                int64_t* info_dev, void* hip_stream);

/* The row search of repeatedPage.Slice (page_repeated.go:73-98) for every row
 * at once: row_starts_dev[k] = index of the k-th repetition level equal to 0,
 * and row_starts_dev[n_rows] = n.  info_dev[0] = n_rows (repeatedPage.NumRows),
 * info_dev[1] = PQD_OK or PQD_ERR_DST_TOO_SMALL when n_rows + 1 > row_cap (the
 * first row_cap entries are still written). */
int pqd_repeated_row_starts_device(pqd_ctx* ctx, const uint8_t* rep_levels_dev, uint64_t n, uint64_t* row_starts_dev,
                                   uint64_t row_cap, void* workspace_dev, uint64_t workspace_bytes, int64_t* info_dev,
                                   void* hip_stream);

/* ---- column chunks: page headers and the parts of a page (host) --------- */

/* format.PageType (format/parquet.go) */
enum { PQD_PAGE_DATA = 0, PQD_PAGE_INDEX = 1, PQD_PAGE_DICTIONARY = 2, PQD_PAGE_DATA_V2 = 3 };

/* format.PageHeader with the header of its page type folded in
 * (format/parquet.go:752-821, 910-953). */
typedef struct pqd_page_header {
    int32_t  type;                         /* PQD_PAGE_* */
    int32_t  uncompressed_page_size;
    int32_t  compressed_page_size;
    int32_t  has_crc;
    int32_t  crc;
    int32_t  num_values;                   /* of the data or dictionary page */
    int32_t  encoding;                     /* of the values */
    int32_t  definition_level_encoding;    /* data page v1; RLE for v2 */
    int32_t  repetition_level_encoding;
    int32_t  num_nulls;                    /* v2, otherwise -1 */
    int32_t  num_rows;                     /* v2, otherwise -1 */
    int32_t  definition_levels_byte_length; /* v2 */
    int32_t  repetition_levels_byte_length; /* v2 */
    int32_t  is_compressed;                /* v2 flag; 1 when absent */
    int32_t  is_sorted;                    /* dictionary page */
    int32_t  reserved;
    uint64_t header_offset;                /* of the thrift header in the chunk */
    uint64_t data_offset;                  /* the page is chunk[data_offset .. + compressed_page_size) */
} pqd_page_header;

/* The header loop of FilePages.ReadPage (file.go:1193-1259) over a column
 * chunk held in host memory: every page header, decoded with the thrift
 * compact protocol (encoding/thrift/compact.go:132-170, 196-245) and checked as
 * validatePageHeader does (file.go:1414-1445).  `num_values` >= 0 ends the walk
 * once the data pages seen hold that many values (ColumnMetaData.num_values),
 * -1 walks to the end of the bytes.  Writes up to `cap` headers and the number
 * found.  PQD_ERR_DST_TOO_SMALL when there are more than `cap`;
 * PQD_ERR_UNEXPECTED_EOF for a header or page cut short,
 * PQD_ERR_INVALID_ARGUMENT for negative sizes or counts (ErrCorrupted in the
 * reference), PQD_ERR_CHECKSUM when a header carries a CRC other than zero and
 * the IEEE CRC-32 of the page bytes differs (FilePages.readPage,
 * file.go:1455-1475); *status_arg = offset of the failing header. */
int pqd_chunk_scan_pages(const uint8_t* chunk, uint64_t chunk_len, int64_t num_values, pqd_page_header* out,
                         uint64_t cap, uint64_t* n_pages, int64_t* status_arg);

/* What Column.decodeDataPage needs to know about the column (column.go:845-911). */
typedef struct pqd_column_info {
    int32_t kind;                   /* pqd_kind of the values */
    int32_t type_length;            /* FIXED_LEN_BYTE_ARRAY */
    int32_t max_definition_level;
    int32_t max_repetition_level;
} pqd_column_info;

enum { PQD_PART_REPETITION_LEVELS = 0, PQD_PART_DEFINITION_LEVELS = 1, PQD_PART_VALUES = 2, PQD_PART_DICTIONARY = 3 };

typedef struct pqd_page_part {
    uint64_t page;        /* index into the headers */
    int32_t  role;        /* PQD_PART_* */
    int32_t  reserved;
    uint64_t src_offset;  /* of the part in the chunk */
} pqd_page_part;

/* How Column.decodeDataPageV1 / decodeDataPageV2 / decodeDictionary
 * (column.go:733-843, 963-993) cut the pages of an UNCOMPRESSED chunk into
 * work items: per data page the repetition levels (if the column has any), the
 * definition levels, the values; per dictionary page its values.  Fills
 * src (= chunk_dev + offset), src_len, encoding, kind, bit_width and
 * num_values of each item: level widths are bits.Len8(max level)
 * (encoding.go:124-134), PLAIN_DICTIONARY becomes RLE_DICTIONARY in data pages
 * and PLAIN in dictionary pages (column.go:850-857, 976-979), the values of a
 * v2 page are num_values - num_nulls (column.go:841), those of a v1 page with
 * definition levels are not known before the levels are counted (num_values
 * = -1).  The caller sets dst, offsets and dict; a dictionary-encoded page decoded
 * without `dict` yields its indexes and needs kind = PQD_KIND_INT32
 * (indexedPageType.Decode, dictionary.go:217-219).  `chunk` is the host copy
 * (the level lengths of v1 pages are read from it, column.go:913-924). */
int pqd_chunk_plan_pages(const pqd_page_header* headers, uint64_t n_headers, const uint8_t* chunk, uint64_t chunk_len,
                         const void* chunk_dev, const pqd_column_info* column, pqd_page* items, pqd_page_part* parts,
                         uint64_t cap, uint64_t* n_items, int64_t* status_arg);

/* ---- file footer: schema, row groups and chunk ranges (host) -------------- */

/* format.CompressionCodec (format/parquet.go) */
enum { PQD_CODEC_UNCOMPRESSED = 0, PQD_CODEC_SNAPPY = 1, PQD_CODEC_GZIP = 2, PQD_CODEC_LZO = 3, PQD_CODEC_BROTLI = 4,
       PQD_CODEC_LZ4 = 5, PQD_CODEC_ZSTD = 6, PQD_CODEC_LZ4_RAW = 7 };

typedef struct pqd_file_info {
    int32_t  version;            /* FileMetaData.version */
    int32_t  is_encrypted;       /* an encryption_algorithm is present (plaintext footer of an encrypted file) */
    int64_t  num_rows;
    uint64_t n_schema;           /* schema elements, the root included */
    uint64_t n_columns;          /* leaves of the schema */
    uint64_t n_row_groups;
    uint64_t n_chunks;           /* n_row_groups * n_columns */
    uint64_t names_bytes;        /* what the column paths take in `names` */
    uint64_t footer_bytes_read;  /* the thrift structure ends here; more bytes are a signature (file.go:177-197) */
} pqd_file_info;

/* A leaf of the schema, depth first: what Column holds (column.go:26-57) and what
 * pqd_column_info needs. */
typedef struct pqd_file_column {
    int32_t  kind;               /* PQD_KIND_* (= format.Type + 1), as pqd_column_info takes it */
    int32_t  type_length;        /* FIXED_LEN_BYTE_ARRAY */
    int32_t  repetition;         /* format.FieldRepetitionType of the leaf: 0 required, 1 optional, 2 repeated */
    int32_t  converted_type;     /* deprecated.ConvertedType, -1 when absent */
    uint8_t  max_definition_level; /* Column.setLevels (column.go:234-280) */
    uint8_t  max_repetition_level;
    uint8_t  depth;
    uint8_t  reserved0;
    uint32_t schema_index;       /* of the leaf in FileMetaData.schema */
    uint32_t name_offset;        /* Column.Path, its components separated by NUL bytes (a field name may hold '.'), names[name_offset .. + name_len) */
    uint32_t name_len;
} pqd_file_column;

typedef struct pqd_file_row_group {
    int64_t num_rows;
    int64_t total_byte_size;
    int64_t file_offset;           /* 0 when absent */
    int64_t total_compressed_size; /* 0 when absent */
} pqd_file_row_group;

/* ColumnChunk.meta_data (format/parquet.go:987-1050).  The pages of the chunk are
 * file[offset .. offset + total_compressed_size) (file.go:1119-1128): that range, read or
 * decompressed page by page, is what pqd_chunk_scan_pages takes. */
typedef struct pqd_file_chunk {
    int64_t  num_values;
    int64_t  offset;             /* dictionary_page_offset when there is one, else data_page_offset */
    int64_t  total_compressed_size;
    int64_t  total_uncompressed_size;
    int64_t  data_page_offset;
    int64_t  dictionary_page_offset; /* 0 when absent */
    int32_t  kind;               /* PQD_KIND_* of ColumnMetaData.type */
    int32_t  codec;              /* PQD_CODEC_* */
    uint32_t encodings;          /* bit e set: format.Encoding e is listed */
    int32_t  has_metadata;       /* 0: the chunk carries no meta_data */
    int32_t  is_encrypted;       /* encrypted_column_metadata is present */
    int32_t  reserved;
} pqd_file_chunk;

/* The tail of OpenFile (file.go:108-131): `tail` = the last `tail_len` >= 8 bytes of a file
 * of `file_size` bytes; gives where the footer (the thrift FileMetaData) lies.
 * PQD_ERR_INVALID_ARGUMENT when the magic is not "PAR1", PQD_ERR_NOT_SUPPORTED for "PARE"
 * (encrypted footer), PQD_ERR_UNEXPECTED_EOF when the length does not fit the file. */
int pqd_file_footer_range(const uint8_t* tail, uint64_t tail_len, uint64_t file_size, uint64_t* footer_offset,
                          uint64_t* footer_len);

/* format.FileMetaData (format/parquet.go:1386-1436) into flat tables: the leaf columns as
 * columnLoader.open walks the schema (column.go:369-521), the row groups, and one chunk per
 * row group and column, chunks[row_group * n_columns + column].  `info` is always filled;
 * PQD_ERR_DST_TOO_SMALL when a table does not hold its rows (call with capacities 0 to size
 * them), PQD_ERR_UNEXPECTED_EOF for malformed thrift (*status_arg = offset),
 * PQD_ERR_INVALID_ARGUMENT for a schema whose groups claim more children than there are
 * elements or that nests or repeats beyond 255 (*status_arg = schema index), or a row group
 * whose chunks are not one per leaf (*status_arg = row group; column.go:223-228). */
int pqd_file_parse_footer(const uint8_t* footer, uint64_t footer_len, pqd_file_info* info, pqd_file_column* columns,
                          uint64_t column_cap, pqd_file_row_group* row_groups, uint64_t row_group_cap,
                          pqd_file_chunk* chunks, uint64_t chunk_cap, char* names, uint64_t names_cap,
                          int64_t* status_arg);

#ifdef __cplusplus
}
#endif
#endif /* PQD_H */
