package pqd

/*
#include "pqd.h"
*/
import "C"

import (
	"unsafe"

	"github.com/parquet-go/parquet-go/encoding"
)

// kinds of include/pqd_status.h: which Decode method of encoding.Encoding a call stands for.
const (
	kindLevels            = int(C.PQD_KIND_LEVELS)
	kindBoolean           = int(C.PQD_KIND_BOOLEAN)
	kindInt32             = int(C.PQD_KIND_INT32)
	kindInt64             = int(C.PQD_KIND_INT64)
	kindInt96             = int(C.PQD_KIND_INT96)
	kindFloat             = int(C.PQD_KIND_FLOAT)
	kindDouble            = int(C.PQD_KIND_DOUBLE)
	kindByteArray         = int(C.PQD_KIND_BYTE_ARRAY)
	kindFixedLenByteArray = int(C.PQD_KIND_FIXED_LEN_BYTE_ARRAY)
)

func (c *Context) handle() *C.pqd_ctx {
	if c == nil {
		return Default().h
	}
	return c.h
}

// call makes the one library call that stands for a Decode method: host bytes in, host values
// out.  src and dst may be the same memory for the encodings that report CanDecodeInPlace: the
// library copies the page to its staging memory before anything is written back.
func call(ctx *C.pqd_ctx, kind int, code C.int, width C.int, src []byte, dst unsafe.Pointer, dstCap int) (n int, st C.int, arg int64) {
	var outBytes C.uint64_t
	var statusArg C.int64_t
	s := (*C.uint8_t)(unsafe.Pointer(unsafe.SliceData(src)))
	sn := C.uint64_t(len(src))
	dn := C.uint64_t(dstCap)
	switch kind {
	case kindLevels:
		st = C.pqd_decode_levels(ctx, code, width, s, sn, (*C.uint8_t)(dst), dn, &outBytes, &statusArg)
	case kindBoolean:
		st = C.pqd_decode_boolean(ctx, code, s, sn, (*C.uint8_t)(dst), dn, &outBytes, &statusArg)
	case kindInt32:
		st = C.pqd_decode_int32(ctx, code, width, s, sn, (*C.int32_t)(dst), dn, &outBytes, &statusArg)
	case kindInt64:
		st = C.pqd_decode_int64(ctx, code, s, sn, (*C.int64_t)(dst), dn, &outBytes, &statusArg)
	case kindInt96:
		st = C.pqd_decode_int96(ctx, code, s, sn, (*C.uint8_t)(dst), dn, &outBytes, &statusArg)
	case kindFloat:
		st = C.pqd_decode_float(ctx, code, s, sn, (*C.float)(dst), dn, &outBytes, &statusArg)
	case kindDouble:
		st = C.pqd_decode_double(ctx, code, s, sn, (*C.double)(dst), dn, &outBytes, &statusArg)
	case kindFixedLenByteArray:
		st = C.pqd_decode_fixed_len_byte_array(ctx, code, width, s, sn, (*C.uint8_t)(dst), dn, &outBytes, &statusArg)
	default:
		return 0, C.PQD_ERR_NOT_SUPPORTED, 0
	}
	return int(outBytes), st, int64(statusArg)
}

// decodeFixed renders the contract of the Decode methods (encoding/encoding.go:43-47: "deserialize
// from the source buffer into the destination slice, potentially growing it") over a library that
// never allocates the caller's memory: the values are written from index 0 of dst's capacity, and
// when the library answers PQD_ERR_DST_TOO_SMALL with the size it needs, a slice of that size is
// made and the call repeated (the counterpart of the append in encoding/rle/rle.go:510-522).  On
// an error the values decoded before it are still returned, as the reference's decoders do.
func decodeFixed[T any](e encoding.Encoding, ctx *Context, kind, width int, dst []T, src []byte) ([]T, error) {
	var zero T
	size := int(unsafe.Sizeof(zero))
	dst = dst[:cap(dst)]
	for attempt := 0; ; attempt++ {
		n, st, arg := call(ctx.handle(), kind, C.int(e.Encoding()), C.int(width), src, unsafe.Pointer(unsafe.SliceData(dst)), len(dst)*size)
		if st == C.PQD_ERR_DST_TOO_SMALL && attempt < 3 && arg > int64(len(dst)*size) {
			dst = make([]T, (int(arg)+size-1)/size)
			continue
		}
		return dst[:n/size], goError(e, st, arg)
	}
}

// decodeByteArray is DecodeByteArray (encoding/plain/plain.go:76-94,
// encoding/delta/length_byte_array.go:37-63, encoding/delta/byte_array.go:124-146): the value
// bytes back to back in dst and len(values)+1 offsets, both grown on demand.
func decodeByteArray(e encoding.Encoding, ctx *Context, dst, src []byte, offsets []uint32) ([]byte, []uint32, error) {
	dst = dst[:cap(dst)]
	offsets = offsets[:cap(offsets)]
	for attempt := 0; ; attempt++ {
		var outBytes, outOffsets C.uint64_t
		var statusArg C.int64_t
		st := C.pqd_decode_byte_array(ctx.handle(), C.int(e.Encoding()),
			(*C.uint8_t)(unsafe.Pointer(unsafe.SliceData(src))), C.uint64_t(len(src)),
			(*C.uint8_t)(unsafe.Pointer(unsafe.SliceData(dst))), C.uint64_t(len(dst)),
			(*C.uint32_t)(unsafe.Pointer(unsafe.SliceData(offsets))), C.uint64_t(len(offsets)),
			&outBytes, &outOffsets, &statusArg)
		arg := int64(statusArg)
		if st == C.PQD_ERR_DST_TOO_SMALL && attempt < 4 && arg > int64(len(dst)) {
			dst = make([]byte, arg)
			continue
		}
		if st == C.PQD_ERR_OFFSETS_TOO_SMALL && attempt < 4 && arg > int64(len(offsets)) {
			offsets = make([]uint32, arg)
			continue
		}
		return dst[:outBytes], offsets[:outOffsets], goError(e, st, arg)
	}
}

// estimateDecodeByteArraySize is EstimateDecodeByteArraySize (encoding/encoding.go:58-67): the
// library walks the page's length streams on the host, no GPU work.
func estimateDecodeByteArraySize(e encoding.Encoding, ctx *Context, src []byte) int {
	return int(C.pqd_estimate_decode_byte_array_size(ctx.handle(), C.int(e.Encoding()),
		(*C.uint8_t)(unsafe.Pointer(unsafe.SliceData(src))), C.uint64_t(len(src))))
}

func canDecodeInPlace(e encoding.Encoding) bool {
	return C.pqd_can_decode_in_place(C.int(e.Encoding())) != 0
}

// encodeValues renders the Encode methods (encoding/encoding.go:29-41): the values are encoded on
// the GPU into dst's capacity, byte for byte what the reference's encoder writes, and dst is grown
// when the library asks for more room (the reference appends, encoding/delta/binary_packed.go:77-91).
// width is the FIXED_LEN_BYTE_ARRAY length, 0 otherwise; n counts values.
func encodeValues(e encoding.Encoding, ctx *Context, kind, width int, dst []byte, src unsafe.Pointer, n int) ([]byte, error) {
	dst = dst[:cap(dst)]
	if bound := int(C.pqd_encode_bound(C.int(e.Encoding()), C.int(kind), C.int(width), C.uint64_t(n))); len(dst) < bound {
		dst = make([]byte, bound)
	}
	for attempt := 0; ; attempt++ {
		var outBytes C.uint64_t
		var statusArg C.int64_t
		st := C.pqd_encode_values(ctx.handle(), C.int(e.Encoding()), C.int(kind), C.int(width), src, C.uint64_t(n),
			(*C.uint8_t)(unsafe.Pointer(unsafe.SliceData(dst))), C.uint64_t(len(dst)), &outBytes, &statusArg)
		if st == C.PQD_ERR_DST_TOO_SMALL && attempt < 3 && int64(statusArg) > int64(len(dst)) {
			dst = make([]byte, int64(statusArg))
			continue
		}
		return dst[:outBytes], goError(e, st, int64(statusArg))
	}
}

// encodeByteArray renders EncodeByteArray(dst, src, offsets) (encoding/encoding.go:38-40) for the
// PLAIN, DELTA_LENGTH_BYTE_ARRAY and DELTA_BYTE_ARRAY codecs: value k is src[offsets[k]:offsets[k+1]].
// With kind FIXED_LEN_BYTE_ARRAY and no offsets it is ByteArrayEncoding.EncodeFixedLenByteArray
// (encoding/delta/byte_array.go:76-122): n values of `size` bytes.
func encodeByteArray(e encoding.Encoding, ctx *Context, kind, size int, dst []byte, src []byte, offsets []uint32, n int) ([]byte, error) {
	dst = dst[:cap(dst)]
	if bound := int(C.pqd_encode_bound_byte_array(C.int(e.Encoding()), C.uint64_t(n), C.uint64_t(len(src)))); len(dst) < bound {
		dst = make([]byte, bound)
	}
	for attempt := 0; ; attempt++ {
		var outBytes C.uint64_t
		var statusArg C.int64_t
		st := C.pqd_encode_byte_array(ctx.handle(), C.int(e.Encoding()), C.int(kind), C.int(size),
			unsafe.Pointer(unsafe.SliceData(src)), C.uint64_t(len(src)),
			(*C.uint32_t)(unsafe.Pointer(unsafe.SliceData(offsets))), C.uint64_t(n),
			(*C.uint8_t)(unsafe.Pointer(unsafe.SliceData(dst))), C.uint64_t(len(dst)), &outBytes, &statusArg)
		if st == C.PQD_ERR_DST_TOO_SMALL && attempt < 3 && int64(statusArg) > int64(len(dst)) {
			dst = make([]byte, int64(statusArg))
			continue
		}
		return dst[:outBytes], goError(e, st, int64(statusArg))
	}
}

// valueCount is the number of values n + 1 offsets stand for (none for no offsets at all).
func valueCount(offsets []uint32) int {
	if len(offsets) == 0 {
		return 0
	}
	return len(offsets) - 1
}
