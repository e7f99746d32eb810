package pqd

/*
#include "pqd.h"
*/
import "C"

import (
	"fmt"
	"runtime"
	"unsafe"

	"github.com/parquet-go/parquet-go/sparse"
)

// Dictionary holds the values of a dictionary page of a fixed-width type (INT32, INT64, FLOAT,
// DOUBLE: 4 or 8 bytes; INT96: 12) in GPU memory: what Type.NewDictionary keeps after
// Column.decodeDictionary (column.go:963-993, dictionary_int64.go:18-26).  Its Lookup methods
// are the gather of dictionaryLookup32 / dictionaryLookup64 (dictionary_amd64.go:30-58,
// dictionary_amd64.s:578-711); patches/encodings.diff shows the two call sites that change.
type Dictionary struct {
	h     *C.pqd_dict
	n     int
	width int
}

// NewDictionary copies n = len(values)/width values to the device.
func NewDictionary(ctx *Context, values []byte, width int) (*Dictionary, error) {
	if width != 4 && width != 8 && width != 12 {
		return nil, fmt.Errorf("pqd: dictionary values of %d bytes: fixed widths of 4, 8 and 12 bytes only", width)
	}
	if len(values)%width != 0 {
		return nil, fmt.Errorf("pqd: %d bytes of dictionary values is not a multiple of %d", len(values), width)
	}
	var h *C.pqd_dict
	n := len(values) / width
	st := C.pqd_dict_create(ctx.handle(), unsafe.Pointer(unsafe.SliceData(values)), C.uint64_t(n), C.uint32_t(width), &h)
	if st != C.PQD_OK {
		return nil, fmt.Errorf("pqd: copying a dictionary of %d values to the GPU: %s", n, statusName(st))
	}
	d := &Dictionary{h: h, n: n, width: width}
	runtime.SetFinalizer(d, (*Dictionary).Close)
	return d, nil
}

// Len returns the number of values (Dictionary.Len, dictionary.go:47-48).
func (d *Dictionary) Len() int { return d.n }

// Close frees the device copy.
func (d *Dictionary) Close() {
	if d.h != nil {
		C.pqd_dict_destroy(d.h)
		d.h = nil
		runtime.SetFinalizer(d, nil)
	}
}

// Lookup writes dict[indexes[i]] to rows.Index(i) for every index: the body of
// int32Dictionary.lookup / int64Dictionary.lookup (dictionary_amd64.go:36-58).  rows is the
// strided view over []Value or over a column of Go structs that the reference passes
// (sparse/array.go:33-37).  Like the reference it panics with "index out of bounds" when an index
// is negative or not below Len (errors.go:75-83), and when rows is shorter than indexes
// (checkLookupIndexBounds, dictionary.go:101-105).
func (d *Dictionary) Lookup(indexes []int32, rows sparse.Array) {
	if rows.Len() < len(indexes) {
		panic("dictionary lookup with more indexes than values")
	}
	if len(indexes) == 0 {
		return
	}
	stride := uintptr(d.width)
	if len(indexes) > 1 {
		stride = uintptr(rows.Index(1)) - uintptr(rows.Index(0))
	}
	// Rows that are not dense lie inside Go values that may hold Go pointers ([]Value, a column of
	// structs), which cgo does not let C write into: the library gathers into a dense buffer
	// and the values are placed here.  Dense rows (a typed slice) are written directly.
	var dense []byte
	out := rows.Index(0)
	if stride != uintptr(d.width) {
		dense = make([]byte, len(indexes)*d.width)
		out = unsafe.Pointer(unsafe.SliceData(dense))
	}
	var arg C.int64_t
	st := C.pqd_dict_lookup(d.h, (*C.int32_t)(unsafe.Pointer(unsafe.SliceData(indexes))), C.uint64_t(len(indexes)),
		out, C.uint64_t(d.width), &arg)
	if st == C.PQD_OK && dense != nil {
		for i := range indexes {
			copy(unsafe.Slice((*byte)(rows.Index(i)), d.width), dense[i*d.width:(i+1)*d.width])
		}
	}
	runtime.KeepAlive(d)
	switch st {
	case C.PQD_OK:
	case C.PQD_ERR_INDEX_OUT_OF_BOUNDS:
		panic("index out of bounds")
	default:
		panic(fmt.Sprintf("pqd: dictionary lookup: %s (%d)", statusName(st), int64(arg)))
	}
}
