package pqd

/*
#include <stdlib.h>
#include <string.h>
#include "pqd.h"
*/
import "C"

import (
	"fmt"
	"unsafe"

	"github.com/parquet-go/parquet-go/encoding"
	"github.com/parquet-go/parquet-go/format"
)

// Kind selects which Decode method of encoding.Encoding a batch item stands for
// (encoding/encoding.go:48-56; pqd_kind of include/pqd_status.h).
type Kind int32

const (
	Levels            Kind = Kind(C.PQD_KIND_LEVELS)
	Boolean           Kind = Kind(C.PQD_KIND_BOOLEAN)
	Int32             Kind = Kind(C.PQD_KIND_INT32)
	Int64             Kind = Kind(C.PQD_KIND_INT64)
	Int96             Kind = Kind(C.PQD_KIND_INT96)
	Float             Kind = Kind(C.PQD_KIND_FLOAT)
	Double            Kind = Kind(C.PQD_KIND_DOUBLE)
	ByteArray         Kind = Kind(C.PQD_KIND_BYTE_ARRAY)
	FixedLenByteArray Kind = Kind(C.PQD_KIND_FIXED_LEN_BYTE_ARRAY)
)

// Item is one page part to decode on the device (pqd_page of include/pqd.h): the unit
// Column.decodeDataPage hands to Type.Decode (column.go:845-911), Column.decodeLevels to
// DecodeLevels (column.go:931-948) or Column.decodeDictionary to the PLAIN decoder
// (column.go:963-993).  All addresses are GPU addresses.
type Item struct {
	Src         DevicePtr
	SrcLen      int
	Dst         DevicePtr
	DstCap      int
	Offsets     DevicePtr // BYTE_ARRAY: NumValues+1 uint32 offsets
	OffsetsCap  int       // entries
	Dict        DevicePtr // decoded dictionary values: the page is looked up as it is unpacked
	DictLen     int
	DictOffsets DevicePtr // BYTE_ARRAY dictionaries: DictLen+1 offsets
	NumValues   int64     // -1: everything the stream holds
	Encoding    format.Encoding
	Kind        Kind
	BitWidth    int // RLE / BIT_PACKED width; FIXED_LEN_BYTE_ARRAY length
	DstStride   int

	// results
	Status    int
	StatusArg int64
	OutValues int
	OutBytes  int
	Consumed  int
}

// Err returns the item's status as the error the reference's decoder gives for the same page.
func (it *Item) Err() error {
	if it.Status == int(C.PQD_OK) {
		return nil
	}
	var e encoding.Encoding = encoding.NotSupported{}
	if code := int(it.Encoding); code >= 0 && code < len(Encodings) && Encodings[code] != nil {
		e = Encodings[code]
	}
	return goError(e, C.int(it.Status), it.StatusArg)
}

// TooSmall reports whether the item failed only because Dst or Offsets was too small, and the
// capacity (bytes, or entries for offsets) to come back with.
func (it *Item) TooSmall() (dstBytes, offsetEntries int, ok bool) {
	switch it.Status {
	case int(C.PQD_ERR_DST_TOO_SMALL):
		return int(it.StatusArg), 0, true
	case int(C.PQD_ERR_OFFSETS_TOO_SMALL):
		return 0, int(it.StatusArg), true
	}
	return 0, 0, false
}

// Batch is a prepared set of items (pqd_batch): created once, launched any number of times.
type Batch struct {
	h     *C.pqd_batch
	pages *C.pqd_page // C memory: the structs hold device addresses, never Go pointers
	n     int
}

// NewBatch validates and groups the items and uploads their descriptors (pqd_batch_create).  The
// device memory the items name must stay valid until the batch is closed.
func (c *Context) NewBatch(items []Item) (*Batch, error) {
	b := &Batch{n: len(items)}
	if len(items) > 0 {
		b.pages = (*C.pqd_page)(C.calloc(C.size_t(len(items)), C.size_t(unsafe.Sizeof(C.pqd_page{}))))
		if b.pages == nil {
			return nil, fmt.Errorf("pqd: out of memory for %d page descriptors", len(items))
		}
		pages := unsafe.Slice(b.pages, len(items))
		for i := range items {
			it, p := &items[i], &pages[i]
			p.src = unsafe.Pointer(uintptr(it.Src))
			p.src_len = C.uint64_t(it.SrcLen)
			p.dst = unsafe.Pointer(uintptr(it.Dst))
			p.dst_cap = C.uint64_t(it.DstCap)
			p.offsets = (*C.uint32_t)(unsafe.Pointer(uintptr(it.Offsets)))
			p.offsets_cap = C.uint64_t(it.OffsetsCap)
			p.dict = unsafe.Pointer(uintptr(it.Dict))
			p.dict_len = C.uint64_t(it.DictLen)
			p.dict_offsets = (*C.uint32_t)(unsafe.Pointer(uintptr(it.DictOffsets)))
			p.num_values = C.int64_t(it.NumValues)
			p.encoding = C.int32_t(it.Encoding)
			p.kind = C.int32_t(it.Kind)
			p.bit_width = C.int32_t(it.BitWidth)
			p.dst_stride = C.int32_t(it.DstStride)
		}
	}
	if st := C.pqd_batch_create(c.handle(), b.pages, C.uint64_t(b.n), &b.h); st != C.PQD_OK {
		C.free(unsafe.Pointer(b.pages))
		return nil, fmt.Errorf("pqd: preparing a batch of %d items: %s", len(items), statusName(st))
	}
	return b, nil
}

// Launch enqueues every kernel of the batch on the stream (pqd_batch_launch): no allocation, no
// copy, no wait.
func (b *Batch) Launch(s *Stream) error {
	if st := C.pqd_batch_launch(b.h, s.raw()); st != C.PQD_OK {
		return fmt.Errorf("pqd: launching a batch: %s: %w", statusName(st), ErrDevice)
	}
	return nil
}

// Finish waits for the stream and fills the result fields of the items (pqd_batch_finish).
func (b *Batch) Finish(s *Stream, items []Item) error {
	if len(items) != b.n {
		return fmt.Errorf("pqd: a batch of %d items finished with %d", b.n, len(items))
	}
	if st := C.pqd_batch_finish(b.h, s.raw(), b.pages, C.uint64_t(b.n)); st != C.PQD_OK {
		return fmt.Errorf("pqd: finishing a batch: %s: %w", statusName(st), ErrDevice)
	}
	if b.n == 0 {
		return nil
	}
	pages := unsafe.Slice(b.pages, b.n)
	for i := range items {
		it, p := &items[i], &pages[i]
		it.Status = int(p.status)
		it.StatusArg = int64(p.status_arg)
		it.OutValues = int(p.out_values)
		it.OutBytes = int(p.out_bytes)
		it.Consumed = int(p.consumed)
	}
	return nil
}

// Close releases the batch (pqd_batch_destroy).
func (b *Batch) Close() {
	if b.h != nil {
		C.pqd_batch_destroy(b.h)
		b.h = nil
	}
	if b.pages != nil {
		C.free(unsafe.Pointer(b.pages))
		b.pages = nil
	}
}

// Decode is NewBatch + Launch + Finish + Close for items decoded once.
func (c *Context) Decode(items []Item, s *Stream) error {
	b, err := c.NewBatch(items)
	if err != nil {
		return err
	}
	defer b.Close()
	if err := b.Launch(s); err != nil {
		return err
	}
	return b.Finish(s, items)
}
