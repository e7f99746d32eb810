package pqd

/*
#include "pqd.h"
*/
import "C"

import (
	"fmt"
	"io"
	"math"

	"github.com/parquet-go/parquet-go/encoding"
)

// ErrDevice is returned when the HIP runtime fails underneath a decode.
var ErrDevice = fmt.Errorf("pqd: GPU runtime error")

// ErrIndexOutOfBounds stands for the "index out of bounds" panic of the reference's dictionary
// lookups (errors.go:75-83); Dictionary.Lookup panics with it, the batch API returns it.
var ErrIndexOutOfBounds = fmt.Errorf("index out of bounds")

// ErrChecksum is the page CRC mismatch of FilePages.readPage (file.go:1455-1475).
var ErrChecksum = fmt.Errorf("crc32 checksum mismatch in page")

// goError turns a status of include/pqd_status.h into the error the reference's decoder returns
// for the same input, wrapped with the encoding's name the way encoding.Error does
// (encoding/notsupported.go:31-39), so that errors.Is against encoding.ErrNotSupported,
// encoding.ErrInvalidArgument and io.ErrUnexpectedEOF answers as it does with the CPU codecs.
func goError(e encoding.Encoding, st C.int, arg int64) error {
	switch st {
	case C.PQD_OK:
		return nil
	case C.PQD_ERR_NOT_SUPPORTED:
		return encoding.Error(e, encoding.ErrNotSupported)
	case C.PQD_ERR_INVALID_ARGUMENT:
		// errInvalidInputSize (encoding/notsupported.go:53-55)
		return encoding.Errorf(e, "cannot decode from input of size %d: %w", arg, encoding.ErrInvalidArgument)
	case C.PQD_ERR_INVALID_BIT_WIDTH:
		// errInvalidBitWidth (encoding/rle/rle.go:429-431)
		return encoding.Errorf(e, "cannot decode with invalid bit-width=%d", arg)
	case C.PQD_ERR_UNEXPECTED_EOF:
		return encoding.Errorf(e, "%s: %w", eofWhere(arg), io.ErrUnexpectedEOF)
	case C.PQD_ERR_VARINT_OVERFLOW:
		// encoding/rle/rle.go:257-259, encoding/delta/binary_packed.go:473-475
		return encoding.Errorf(e, "overflow decoding the variable-length integer at byte %d", arg)
	case C.PQD_ERR_RUN_TOO_LARGE:
		// encoding/rle/rle.go:266-268
		return encoding.Errorf(e, "decoded run-length block cannot have more than %d values", math.MaxInt32)
	case C.PQD_ERR_BAD_DELTA_HEADER:
		return encoding.Errorf(e, "%s", deltaHeaderFault(arg))
	case C.PQD_ERR_FIRST_VALUE_RANGE:
		// encoding/delta/binary_packed.go:287-289
		return encoding.Errorf(e, "first value out of range: %d", arg)
	case C.PQD_ERR_MISSING_VALUES:
		// encoding/delta/binary_packed.go:339-341, 402-404
		return encoding.Errorf(e, "%d missing values: %w", arg, io.ErrUnexpectedEOF)
	case C.PQD_ERR_NEGATIVE_LENGTH:
		// encoding/delta/delta.go:85-87
		return encoding.Errorf(e, "invalid negative value length: %d", arg)
	case C.PQD_ERR_LENGTH_OUT_OF_BOUNDS:
		// encoding/delta/delta.go:93-95 (the input size is not part of the status)
		return encoding.Errorf(e, "value length is larger than the input size: %d", arg)
	case C.PQD_ERR_PREFIX_SUFFIX_MISMATCH:
		// encoding/delta/delta.go:81-83
		return encoding.Errorf(e, "length of prefix and suffix mismatch: %d != %d", arg>>32, arg&0xffffffff)
	case C.PQD_ERR_NEGATIVE_PREFIX:
		// encoding/delta/delta.go:89-91
		return encoding.Errorf(e, "invalid negative prefix length: %d", arg)
	case C.PQD_ERR_PREFIX_OUT_OF_BOUNDS:
		// encoding/delta/delta.go:97-99
		return encoding.Errorf(e, "prefix length %d is larger than the last value", arg)
	case C.PQD_ERR_INDEX_OUT_OF_BOUNDS:
		return encoding.Errorf(e, "%w at position %d", ErrIndexOutOfBounds, arg)
	case C.PQD_ERR_CHECKSUM:
		return encoding.Errorf(e, "%w (header at byte %d)", ErrChecksum, arg)
	case C.PQD_ERR_DEVICE:
		return encoding.Errorf(e, "%w (%d)", ErrDevice, arg)
	default:
		return encoding.Errorf(e, "%s (%d)", statusName(st), arg)
	}
}

// eofWhere names the place an input ended early with the words the reference uses there.
func eofWhere(where int64) string {
	switch where {
	case C.PQD_EOF_RUN_HEADER:
		return "decoding run-length block header" // encoding/rle/rle.go:255
	case C.PQD_EOF_BITPACKED_RUN:
		return "decoding bit-packed block" // encoding/rle/rle.go:274
	case C.PQD_EOF_RLE_VALUE:
		return "decoding run-length block" // encoding/rle/rle.go:333
	case C.PQD_EOF_LENGTH_PREFIX:
		return "input shorter than length prefix" // encoding/rle/rle.go:72-79
	case C.PQD_EOF_DELTA_HEADER:
		return "decoding block header" // encoding/delta/binary_packed.go:409-435
	case C.PQD_EOF_DELTA_MIN_DELTA:
		return "decoding min delta" // encoding/delta/binary_packed.go:454-466
	case C.PQD_EOF_PLAIN_BYTE_ARRAY:
		return "input is too short to contain a PLAIN encoded byte array value" // encoding/plain/plain.go:240-242
	}
	return "decoding"
}

// deltaHeaderFault spells the header checks of encoding/delta/binary_packed.go:437-449.
func deltaHeaderFault(fault int64) string {
	switch fault {
	case C.PQD_DELTA_BAD_NUM_MINIBLOCKS:
		return "invalid number of mini block"
	case C.PQD_DELTA_BAD_BLOCK_SIZE:
		return "invalid block size is not a multiple of 128"
	case C.PQD_DELTA_BLOCK_TOO_LARGE:
		return "invalid block size is too large"
	case C.PQD_DELTA_BAD_MINIBLOCK_SIZE:
		return "invalid mini block size is not a multiple of 32"
	case C.PQD_DELTA_NEGATIVE_TOTAL:
		return "invalid total number of values is negative"
	case C.PQD_DELTA_TOO_MANY_VALUES:
		return "too many values"
	}
	return "invalid DELTA_BINARY_PACKED header"
}
