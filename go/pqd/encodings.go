package pqd

import (
	"unsafe"

	"github.com/parquet-go/parquet-go/deprecated"
	"github.com/parquet-go/parquet-go/encoding"
	"github.com/parquet-go/parquet-go/format"
)

// The nine codecs of the reference's `encodings` table (encoding.go:50-60).  Like the reference's
// codec types (encoding/plain/plain.go:24-26, encoding/rle/rle.go:30-33, ...) each embeds
// encoding.NotSupported, which answers every method the codec does not define with
// encoding.ErrNotSupported (encoding/notsupported.go:116-205).  DeltaBinaryPacked, RLEDictionary
// and ByteStreamSplit also encode on the GPU; the other Encode methods are left to
// encoding.NotSupported, and patches/encodings.diff pairs every codec with the reference's CPU
// encoder.  The zero value is ready to use and decodes on the Default context;
// set Context to pin a codec to another GPU.

// Plain stands in for plain.Encoding (encoding/plain/plain.go:24-112).
type Plain struct {
	encoding.NotSupported
	Context *Context
}

func (e *Plain) String() string            { return "PLAIN" }
func (e *Plain) Encoding() format.Encoding { return format.Plain }

func (e *Plain) DecodeBoolean(dst []byte, src []byte) ([]byte, error) {
	return decodeFixed(e, e.Context, kindBoolean, 1, dst, src)
}

func (e *Plain) DecodeInt32(dst []int32, src []byte) ([]int32, error) {
	return decodeFixed(e, e.Context, kindInt32, 0, dst, src)
}

func (e *Plain) DecodeInt64(dst []int64, src []byte) ([]int64, error) {
	return decodeFixed(e, e.Context, kindInt64, 0, dst, src)
}

func (e *Plain) DecodeInt96(dst []deprecated.Int96, src []byte) ([]deprecated.Int96, error) {
	return decodeFixed(e, e.Context, kindInt96, 0, dst, src)
}

func (e *Plain) DecodeFloat(dst []float32, src []byte) ([]float32, error) {
	return decodeFixed(e, e.Context, kindFloat, 0, dst, src)
}

func (e *Plain) DecodeDouble(dst []float64, src []byte) ([]float64, error) {
	return decodeFixed(e, e.Context, kindDouble, 0, dst, src)
}

func (e *Plain) DecodeByteArray(dst []byte, src []byte, offsets []uint32) ([]byte, []uint32, error) {
	return decodeByteArray(e, e.Context, dst, src, offsets)
}

func (e *Plain) DecodeFixedLenByteArray(dst []byte, src []byte, size int) ([]byte, error) {
	return decodeFixed(e, e.Context, kindFixedLenByteArray, size, dst, src)
}

func (e *Plain) EstimateDecodeByteArraySize(src []byte) int {
	return estimateDecodeByteArraySize(e, e.Context, src)
}

func (e *Plain) CanDecodeInPlace() bool { return canDecodeInPlace(e) }

// EncodeByteArray writes every value behind its 4-byte length (encoding/plain/plain.go:44-57).
func (e *Plain) EncodeByteArray(dst []byte, src []byte, offsets []uint32) ([]byte, error) {
	return encodeByteArray(e, e.Context, kindByteArray, 0, dst, src, offsets, valueCount(offsets))
}

// PlainDictionary stands in for plain.DictionaryEncoding (encoding/plain/dictionary.go:9-27):
// DecodeInt32 is a PLAIN decode of raw int32 values.  Data pages marked PLAIN_DICTIONARY are
// decoded with RLEDictionary; that rewrite is the caller's (column.go:850-857).
type PlainDictionary struct {
	encoding.NotSupported
	Context *Context
}

func (e *PlainDictionary) String() string            { return "PLAIN_DICTIONARY" }
func (e *PlainDictionary) Encoding() format.Encoding { return format.PlainDictionary }

func (e *PlainDictionary) DecodeInt32(dst []int32, src []byte) ([]int32, error) {
	return decodeFixed(e, e.Context, kindInt32, 0, dst, src)
}

// RLE stands in for rle.Encoding (encoding/rle/rle.go:30-94).  BitWidth is the width of the
// packed values, as in the reference: the level tables hold one codec per width.
type RLE struct {
	encoding.NotSupported
	BitWidth int
	Context  *Context
}

func (e *RLE) String() string            { return "RLE" }
func (e *RLE) Encoding() format.Encoding { return format.RLE }

func (e *RLE) DecodeLevels(dst []uint8, src []byte) ([]uint8, error) {
	return decodeFixed(e, e.Context, kindLevels, e.BitWidth, dst, src)
}

func (e *RLE) DecodeBoolean(dst []byte, src []byte) ([]byte, error) {
	return decodeFixed(e, e.Context, kindBoolean, 1, dst, src)
}

func (e *RLE) DecodeInt32(dst []int32, src []byte) ([]int32, error) {
	return decodeFixed(e, e.Context, kindInt32, e.BitWidth, dst, src)
}

// EncodeLevels, EncodeBoolean and EncodeInt32 (encoding/rle/rle.go:43-61): encodeBytes, the 4-byte
// length and encodeBits, encodeInt32, each with the codec's BitWidth.
func (e *RLE) EncodeLevels(dst []byte, src []uint8) ([]byte, error) {
	return encodeValues(e, e.Context, kindLevels, e.BitWidth, dst, unsafe.Pointer(unsafe.SliceData(src)), len(src))
}

func (e *RLE) EncodeBoolean(dst []byte, src []byte) ([]byte, error) {
	return encodeValues(e, e.Context, kindBoolean, 0, dst, unsafe.Pointer(unsafe.SliceData(src)), len(src))
}

func (e *RLE) EncodeInt32(dst []byte, src []int32) ([]byte, error) {
	return encodeValues(e, e.Context, kindInt32, e.BitWidth, dst, unsafe.Pointer(unsafe.SliceData(src)), len(src))
}

// RLEDictionary stands in for rle.DictionaryEncoding (encoding/rle/dictionary.go:11-59): the
// width of the indexes is the first byte of the page.
type RLEDictionary struct {
	encoding.NotSupported
	Context *Context
}

func (e *RLEDictionary) String() string            { return "RLE_DICTIONARY" }
func (e *RLEDictionary) Encoding() format.Encoding { return format.RLEDictionary }

func (e *RLEDictionary) DecodeInt32(dst []int32, src []byte) ([]int32, error) {
	return decodeFixed(e, e.Context, kindInt32, 0, dst, src)
}

// EncodeInt32 writes the bit width of the widest index and the RLE / bit-packed hybrid of the
// indexes (encoding/rle/dictionary.go:23-28, 52-59).
func (e *RLEDictionary) EncodeInt32(dst []byte, src []int32) ([]byte, error) {
	return encodeValues(e, e.Context, kindInt32, 0, dst, unsafe.Pointer(unsafe.SliceData(src)), len(src))
}

// BitPacked stands in for bitpacked.Encoding (encoding/bitpacked/bitpacked.go:8-33).
type BitPacked struct {
	encoding.NotSupported
	BitWidth int
	Context  *Context
}

func (e *BitPacked) String() string            { return "BIT_PACKED" }
func (e *BitPacked) Encoding() format.Encoding { return format.BitPacked }

func (e *BitPacked) DecodeLevels(dst []uint8, src []byte) ([]uint8, error) {
	return decodeFixed(e, e.Context, kindLevels, e.BitWidth, dst, src)
}

// DeltaBinaryPacked stands in for delta.BinaryPackedEncoding (encoding/delta/binary_packed.go:16-46).
type DeltaBinaryPacked struct {
	encoding.NotSupported
	Context *Context
}

func (e *DeltaBinaryPacked) String() string            { return "DELTA_BINARY_PACKED" }
func (e *DeltaBinaryPacked) Encoding() format.Encoding { return format.DeltaBinaryPacked }

func (e *DeltaBinaryPacked) DecodeInt32(dst []int32, src []byte) ([]int32, error) {
	return decodeFixed(e, e.Context, kindInt32, 0, dst, src)
}

func (e *DeltaBinaryPacked) DecodeInt64(dst []int64, src []byte) ([]int64, error) {
	return decodeFixed(e, e.Context, kindInt64, 0, dst, src)
}

// EncodeInt32 and EncodeInt64 write blocks of 128 values in four miniblocks
// (encoding/delta/binary_packed.go:26-34, 77-277).
func (e *DeltaBinaryPacked) EncodeInt32(dst []byte, src []int32) ([]byte, error) {
	return encodeValues(e, e.Context, kindInt32, 0, dst, unsafe.Pointer(unsafe.SliceData(src)), len(src))
}

func (e *DeltaBinaryPacked) EncodeInt64(dst []byte, src []int64) ([]byte, error) {
	return encodeValues(e, e.Context, kindInt64, 0, dst, unsafe.Pointer(unsafe.SliceData(src)), len(src))
}

// DeltaLengthByteArray stands in for delta.LengthByteArrayEncoding
// (encoding/delta/length_byte_array.go:8-74).
type DeltaLengthByteArray struct {
	encoding.NotSupported
	Context *Context
}

func (e *DeltaLengthByteArray) String() string            { return "DELTA_LENGTH_BYTE_ARRAY" }
func (e *DeltaLengthByteArray) Encoding() format.Encoding { return format.DeltaLengthByteArray }

func (e *DeltaLengthByteArray) DecodeByteArray(dst []byte, src []byte, offsets []uint32) ([]byte, []uint32, error) {
	return decodeByteArray(e, e.Context, dst, src, offsets)
}

func (e *DeltaLengthByteArray) EstimateDecodeByteArraySize(src []byte) int {
	return estimateDecodeByteArraySize(e, e.Context, src)
}

func (e *DeltaLengthByteArray) CanDecodeInPlace() bool { return canDecodeInPlace(e) }

// EncodeByteArray writes the lengths as DELTA_BINARY_PACKED int32 and then all of src
// (encoding/delta/length_byte_array.go:20-35).
func (e *DeltaLengthByteArray) EncodeByteArray(dst []byte, src []byte, offsets []uint32) ([]byte, error) {
	return encodeByteArray(e, e.Context, kindByteArray, 0, dst, src, offsets, valueCount(offsets))
}

// DeltaByteArray stands in for delta.ByteArrayEncoding (encoding/delta/byte_array.go:14-183).
type DeltaByteArray struct {
	encoding.NotSupported
	Context *Context
}

func (e *DeltaByteArray) String() string            { return "DELTA_BYTE_ARRAY" }
func (e *DeltaByteArray) Encoding() format.Encoding { return format.DeltaByteArray }

func (e *DeltaByteArray) DecodeByteArray(dst []byte, src []byte, offsets []uint32) ([]byte, []uint32, error) {
	return decodeByteArray(e, e.Context, dst, src, offsets)
}

// EncodeByteArray and EncodeFixedLenByteArray write prefix lengths, suffix lengths and suffixes
// (encoding/delta/byte_array.go:26-122).
func (e *DeltaByteArray) EncodeByteArray(dst []byte, src []byte, offsets []uint32) ([]byte, error) {
	return encodeByteArray(e, e.Context, kindByteArray, 0, dst, src, offsets, valueCount(offsets))
}

func (e *DeltaByteArray) EncodeFixedLenByteArray(dst []byte, src []byte, size int) ([]byte, error) {
	if size < 0 || size > encoding.MaxFixedLenByteArraySize {
		return dst[:0], encoding.Error(e, encoding.ErrInvalidArgument)
	}
	if size == 0 || len(src)%size != 0 {
		return dst[:0], encoding.ErrEncodeInvalidInputSize(e, "FIXED_LEN_BYTE_ARRAY", len(src))
	}
	return encodeByteArray(e, e.Context, kindFixedLenByteArray, size, dst, src, nil, len(src)/size)
}

func (e *DeltaByteArray) DecodeFixedLenByteArray(dst []byte, src []byte, size int) ([]byte, error) {
	return decodeFixed(e, e.Context, kindFixedLenByteArray, size, dst, src)
}

func (e *DeltaByteArray) EstimateDecodeByteArraySize(src []byte) int {
	return estimateDecodeByteArraySize(e, e.Context, src)
}

// ByteStreamSplit stands in for bytestreamsplit.Encoding
// (encoding/bytestreamsplit/bytestreamsplit.go:11-105).
type ByteStreamSplit struct {
	encoding.NotSupported
	Context *Context
}

func (e *ByteStreamSplit) String() string            { return "BYTE_STREAM_SPLIT" }
func (e *ByteStreamSplit) Encoding() format.Encoding { return format.ByteStreamSplit }

func (e *ByteStreamSplit) DecodeInt32(dst []int32, src []byte) ([]int32, error) {
	return decodeFixed(e, e.Context, kindInt32, 0, dst, src)
}

func (e *ByteStreamSplit) DecodeInt64(dst []int64, src []byte) ([]int64, error) {
	return decodeFixed(e, e.Context, kindInt64, 0, dst, src)
}

func (e *ByteStreamSplit) DecodeFloat(dst []float32, src []byte) ([]float32, error) {
	return decodeFixed(e, e.Context, kindFloat, 0, dst, src)
}

func (e *ByteStreamSplit) DecodeDouble(dst []float64, src []byte) ([]float64, error) {
	return decodeFixed(e, e.Context, kindDouble, 0, dst, src)
}

func (e *ByteStreamSplit) DecodeFixedLenByteArray(dst []byte, src []byte, size int) ([]byte, error) {
	return decodeFixed(e, e.Context, kindFixedLenByteArray, size, dst, src)
}

// The Encode methods scatter byte k of every value to stream k
// (encoding/bytestreamsplit/bytestreamsplit.go:19-33, bytestreamsplit_amd64.s:56-208).
func (e *ByteStreamSplit) EncodeInt32(dst []byte, src []int32) ([]byte, error) {
	return encodeValues(e, e.Context, kindInt32, 0, dst, unsafe.Pointer(unsafe.SliceData(src)), len(src))
}

func (e *ByteStreamSplit) EncodeInt64(dst []byte, src []int64) ([]byte, error) {
	return encodeValues(e, e.Context, kindInt64, 0, dst, unsafe.Pointer(unsafe.SliceData(src)), len(src))
}

func (e *ByteStreamSplit) EncodeFloat(dst []byte, src []float32) ([]byte, error) {
	return encodeValues(e, e.Context, kindFloat, 0, dst, unsafe.Pointer(unsafe.SliceData(src)), len(src))
}

func (e *ByteStreamSplit) EncodeDouble(dst []byte, src []float64) ([]byte, error) {
	return encodeValues(e, e.Context, kindDouble, 0, dst, unsafe.Pointer(unsafe.SliceData(src)), len(src))
}

func (e *ByteStreamSplit) EncodeFixedLenByteArray(dst []byte, src []byte, size int) ([]byte, error) {
	if size <= 0 || size > encoding.MaxFixedLenByteArraySize {
		return dst[:0], encoding.Error(e, encoding.ErrInvalidArgument)
	}
	if len(src)%size != 0 {
		return dst[:0], encoding.ErrEncodeInvalidInputSize(e, "FIXED_LEN_BYTE_ARRAY", len(src))
	}
	return encodeValues(e, e.Context, kindFixedLenByteArray, size, dst, unsafe.Pointer(unsafe.SliceData(src)), len(src)/size)
}

// Encodings is the reference's `encodings` table (encoding.go:50-60) over the GPU codecs, indexed
// by format.Encoding; entries the format does not define are nil, as there.
var Encodings = [...]encoding.Encoding{
	format.Plain:                &Plain{},
	format.PlainDictionary:      &PlainDictionary{},
	format.BitPacked:            &BitPacked{},
	format.RLE:                  &RLE{},
	format.RLEDictionary:        &RLEDictionary{},
	format.DeltaBinaryPacked:    &DeltaBinaryPacked{},
	format.DeltaLengthByteArray: &DeltaLengthByteArray{},
	format.DeltaByteArray:       &DeltaByteArray{},
	format.ByteStreamSplit:      &ByteStreamSplit{},
}

// LevelEncodingsRLE and LevelEncodingsBitPacked are the level codec tables of encoding.go:64-85:
// entry i decodes levels of bit width i+1, and lookupLevelEncoding (encoding.go:110-121) indexes
// them with bits.Len8(maxLevel)-1.
var LevelEncodingsRLE = [...]RLE{
	0: {BitWidth: 1},
	1: {BitWidth: 2},
	2: {BitWidth: 3},
	3: {BitWidth: 4},
	4: {BitWidth: 5},
	5: {BitWidth: 6},
	6: {BitWidth: 7},
	7: {BitWidth: 8},
}

var LevelEncodingsBitPacked = [...]BitPacked{
	0: {BitWidth: 1},
	1: {BitWidth: 2},
	2: {BitWidth: 3},
	3: {BitWidth: 4},
	4: {BitWidth: 5},
	5: {BitWidth: 6},
	6: {BitWidth: 7},
	7: {BitWidth: 8},
}

// Every codec satisfies the reference's interface.
var (
	_ encoding.Encoding = (*Plain)(nil)
	_ encoding.Encoding = (*PlainDictionary)(nil)
	_ encoding.Encoding = (*RLE)(nil)
	_ encoding.Encoding = (*RLEDictionary)(nil)
	_ encoding.Encoding = (*BitPacked)(nil)
	_ encoding.Encoding = (*DeltaBinaryPacked)(nil)
	_ encoding.Encoding = (*DeltaLengthByteArray)(nil)
	_ encoding.Encoding = (*DeltaByteArray)(nil)
	_ encoding.Encoding = (*ByteStreamSplit)(nil)
)
