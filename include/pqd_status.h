/*
 * pqd_status.h -- status codes, encoding codes and value kinds shared by the
 * MI355X page-value decode library (pqd.h) and its CPU checker (oracle/).
 *
 * Every code maps onto one error the reference decoders return, so a Go shim
 * can rebuild errors that satisfy errors.Is against encoding.ErrNotSupported,
 * encoding.ErrInvalidArgument and io.ErrUnexpectedEOF
 * (reference: encoding/notsupported.go:11-27, encoding/rle/rle.go:255-401,
 * encoding/delta/binary_packed.go:402-488, encoding/delta/delta.go:81-99).
 */
#ifndef PQD_STATUS_H
#define PQD_STATUS_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef enum pqd_status {
    PQD_OK = 0,
    /* encoding.ErrNotSupported (encoding/notsupported.go:18,207-209) */
    PQD_ERR_NOT_SUPPORTED = 1,
    /* encoding.ErrInvalidArgument: input size not a multiple of the value
     * width, FIXED_LEN_BYTE_ARRAY size out of range
     * (encoding/notsupported.go:53-55, encoding/bytestreamsplit/bytestreamsplit.go:96-101).
     * arg = offending size */
    PQD_ERR_INVALID_ARGUMENT = 2,
    /* "cannot decode INT8/INT32 with invalid bit-width" (encoding/rle/rle.go:296-298,353-355);
     * also a DELTA miniblock width above the value width. arg = bit width */
    PQD_ERR_INVALID_BIT_WIDTH = 3,
    /* io.ErrUnexpectedEOF while reading a header, a bit-packed run, a run
     * value or a length prefix. arg = pqd_eof_where */
    PQD_ERR_UNEXPECTED_EOF = 4,
    /* varint longer than 10 bytes (binary.Uvarint n<0; encoding/rle/rle.go:257-259,
     * encoding/delta/binary_packed.go:473-475). arg = byte offset of the varint */
    PQD_ERR_VARINT_OVERFLOW = 5,
    /* "decoded run-length block cannot have more than 2147483647 values"
     * (encoding/rle/rle.go:266-268). arg = count */
    PQD_ERR_RUN_TOO_LARGE = 6,
    /* DELTA_BINARY_PACKED header validation (encoding/delta/binary_packed.go:437-449).
     * arg = pqd_delta_header_fault */
    PQD_ERR_BAD_DELTA_HEADER = 7,
    /* "first value out of range" for INT32 (encoding/delta/binary_packed.go:287-289). arg = first value */
    PQD_ERR_FIRST_VALUE_RANGE = 8,
    /* "%d missing values: unexpected EOF" (encoding/delta/binary_packed.go:339-341,402-404).
     * arg = number of missing values; wraps io.ErrUnexpectedEOF */
    PQD_ERR_MISSING_VALUES = 9,
    /* "invalid negative value length" (encoding/delta/delta.go:85-87). arg = length */
    PQD_ERR_NEGATIVE_LENGTH = 10,
    /* "value length is larger than the input size" (encoding/delta/delta.go:93-95). arg = length */
    PQD_ERR_LENGTH_OUT_OF_BOUNDS = 11,
    /* "length of prefix and suffix mismatch" (encoding/delta/delta.go:81-83). arg = (prefix<<32)|suffix counts */
    PQD_ERR_PREFIX_SUFFIX_MISMATCH = 12,
    /* "invalid negative prefix length" (encoding/delta/delta.go:89-91). arg = length */
    PQD_ERR_NEGATIVE_PREFIX = 13,
    /* "prefix length N is larger than the last value of size M" (encoding/delta/delta.go:97-99). arg = prefix */
    PQD_ERR_PREFIX_OUT_OF_BOUNDS = 14,
    /* dictionary index out of bounds: the reference panics "index out of bounds"
     * (errors.go:75-83, dictionary_amd64.s:612-615). arg = position of the first bad index */
    PQD_ERR_INDEX_OUT_OF_BOUNDS = 15,
    /* destination capacity too small; nothing the reference has (Go slices
     * grow, encoding/rle/rle.go:510-522): the shim grows dst to arg bytes
     * (or arg entries for offsets) and calls again. */
    PQD_ERR_DST_TOO_SMALL = 16,
    /* HIP runtime failure. arg = hipError_t */
    PQD_ERR_DEVICE = 17,
    /* offsets capacity too small. arg = entries needed */
    PQD_ERR_OFFSETS_TOO_SMALL = 18,
    /* "crc32 checksum mismatch in page" (file.go:1455-1475; wraps ErrCorrupted).
     * arg = offset of the page header in the chunk */
    PQD_ERR_CHECKSUM = 19
} pqd_status;

/* Where an UNEXPECTED_EOF happened (status arg). */
typedef enum pqd_eof_where {
    PQD_EOF_RUN_HEADER = 1,      /* "decoding run-length block header" */
    PQD_EOF_BITPACKED_RUN = 2,   /* "decoding bit-packed block of %d values" */
    PQD_EOF_RLE_VALUE = 3,       /* "decoding run-length block of %d values" */
    PQD_EOF_LENGTH_PREFIX = 4,   /* boolean 4-byte prefix (encoding/rle/rle.go:72-79) */
    PQD_EOF_DELTA_HEADER = 5,    /* block size / mini-blocks / total / first value */
    PQD_EOF_DELTA_MIN_DELTA = 6, /* "decoding min delta" */
    PQD_EOF_PLAIN_BYTE_ARRAY = 7 /* plain.ErrTooShort (encoding/plain/plain.go:240-242) */
} pqd_eof_where;

typedef enum pqd_delta_header_fault {
    PQD_DELTA_BAD_NUM_MINIBLOCKS = 1, /* == 0 */
    PQD_DELTA_BAD_BLOCK_SIZE = 2,     /* <= 0 or not a multiple of 128 */
    PQD_DELTA_BLOCK_TOO_LARGE = 3,    /* > 65536 */
    PQD_DELTA_BAD_MINIBLOCK_SIZE = 4, /* not a multiple of 32 */
    PQD_DELTA_NEGATIVE_TOTAL = 5,
    PQD_DELTA_TOO_MANY_VALUES = 6     /* > MaxInt32 */
} pqd_delta_header_fault;

/* format.Encoding codes (reference: format/parquet.go:589-630). */
typedef enum pqd_encoding {
    PQD_ENC_PLAIN = 0,
    PQD_ENC_PLAIN_DICTIONARY = 2,
    PQD_ENC_RLE = 3,
    PQD_ENC_BIT_PACKED = 4,
    PQD_ENC_DELTA_BINARY_PACKED = 5,
    PQD_ENC_DELTA_LENGTH_BYTE_ARRAY = 6,
    PQD_ENC_DELTA_BYTE_ARRAY = 7,
    PQD_ENC_RLE_DICTIONARY = 8,
    PQD_ENC_BYTE_STREAM_SPLIT = 9
} pqd_encoding;

/* Which Decode method of encoding.Encoding a work item stands for
 * (reference: encoding/encoding.go:48-56). */
typedef enum pqd_kind {
    PQD_KIND_LEVELS = 0,   /* DecodeLevels  -> uint8 */
    PQD_KIND_BOOLEAN = 1,  /* DecodeBoolean -> packed bits */
    PQD_KIND_INT32 = 2,
    PQD_KIND_INT64 = 3,
    PQD_KIND_INT96 = 4,
    PQD_KIND_FLOAT = 5,
    PQD_KIND_DOUBLE = 6,
    PQD_KIND_BYTE_ARRAY = 7,
    PQD_KIND_FIXED_LEN_BYTE_ARRAY = 8
} pqd_kind;

#ifdef __cplusplus
}
#endif
#endif /* PQD_STATUS_H */
