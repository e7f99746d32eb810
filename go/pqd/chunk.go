package pqd

/*
#include <stdlib.h>
#include "pqd.h"
*/
import "C"

import (
	"fmt"
	"io"
	"unsafe"

	"github.com/parquet-go/parquet-go/format"
)

// PageHeader is format.PageHeader with the header of its page type folded in
// (format/parquet.go:752-821, 910-953; pqd_page_header of include/pqd.h).
type PageHeader struct {
	Type                       format.PageType
	UncompressedPageSize       int32
	CompressedPageSize         int32
	HasCRC                     bool
	CRC                        int32
	NumValues                  int32
	Encoding                   format.Encoding
	DefinitionLevelEncoding    format.Encoding
	RepetitionLevelEncoding    format.Encoding
	NumNulls                   int32 // v2, otherwise -1
	NumRows                    int32 // v2, otherwise -1
	DefinitionLevelsByteLength int32
	RepetitionLevelsByteLength int32
	IsCompressed               bool
	IsSorted                   bool
	HeaderOffset               int // of the thrift header in the chunk
	DataOffset                 int // the page is chunk[DataOffset : DataOffset+CompressedPageSize]

	raw C.pqd_page_header
}

// ErrCorrupted stands for the reference's ErrCorrupted (errors.go) in what the header scan rejects.
var ErrCorrupted = fmt.Errorf("corrupted page header")

// ScanPages is the header loop of FilePages.ReadPage (file.go:1193-1259) over a column chunk held
// in memory: every thrift page header, checked as validatePageHeader does (file.go:1414-1445),
// and the page CRC when the header carries one (file.go:1455-1475).  numValues is
// ColumnMetaData.NumValues, or -1 to walk to the end of the bytes.
func ScanPages(chunk []byte, numValues int64) ([]PageHeader, error) {
	room := 64
	for {
		raw := make([]C.pqd_page_header, room)
		var n C.uint64_t
		var arg C.int64_t
		st := C.pqd_chunk_scan_pages((*C.uint8_t)(unsafe.Pointer(unsafe.SliceData(chunk))), C.uint64_t(len(chunk)),
			C.int64_t(numValues), &raw[0], C.uint64_t(room), &n, &arg)
		switch st {
		case C.PQD_OK:
			headers := make([]PageHeader, int(n))
			for i := range headers {
				r := &raw[i]
				headers[i] = PageHeader{
					Type:                       format.PageType(r._type),
					UncompressedPageSize:       int32(r.uncompressed_page_size),
					CompressedPageSize:         int32(r.compressed_page_size),
					HasCRC:                     r.has_crc != 0,
					CRC:                        int32(r.crc),
					NumValues:                  int32(r.num_values),
					Encoding:                   format.Encoding(r.encoding),
					DefinitionLevelEncoding:    format.Encoding(r.definition_level_encoding),
					RepetitionLevelEncoding:    format.Encoding(r.repetition_level_encoding),
					NumNulls:                   int32(r.num_nulls),
					NumRows:                    int32(r.num_rows),
					DefinitionLevelsByteLength: int32(r.definition_levels_byte_length),
					RepetitionLevelsByteLength: int32(r.repetition_levels_byte_length),
					IsCompressed:               r.is_compressed != 0,
					IsSorted:                   r.is_sorted != 0,
					HeaderOffset:               int(r.header_offset),
					DataOffset:                 int(r.data_offset),
					raw:                        *r,
				}
			}
			return headers, nil
		case C.PQD_ERR_DST_TOO_SMALL:
			room *= 4
		case C.PQD_ERR_UNEXPECTED_EOF:
			return nil, fmt.Errorf("page header at byte %d: %w", int64(arg), io.ErrUnexpectedEOF)
		case C.PQD_ERR_CHECKSUM:
			return nil, fmt.Errorf("page header at byte %d: %w", int64(arg), ErrChecksum)
		default:
			return nil, fmt.Errorf("page header at byte %d: %s: %w", int64(arg), statusName(st), ErrCorrupted)
		}
	}
}

// ColumnInfo is what Column.decodeDataPage needs to know about the column (column.go:845-911).
type ColumnInfo struct {
	Kind               Kind
	TypeLength         int
	MaxDefinitionLevel int
	MaxRepetitionLevel int
}

// Role says which part of a page an item decodes.
type Role int32

const (
	RepetitionLevels Role = Role(C.PQD_PART_REPETITION_LEVELS)
	DefinitionLevels Role = Role(C.PQD_PART_DEFINITION_LEVELS)
	Values           Role = Role(C.PQD_PART_VALUES)
	DictionaryValues Role = Role(C.PQD_PART_DICTIONARY)
)

// Part ties an item of PlanPages to its page.
type Part struct {
	Page      int // index into the headers
	Role      Role
	SrcOffset int // of the part in the chunk
}

// PlanPages cuts the pages of an uncompressed chunk into batch items the way
// Column.decodeDataPageV1 / decodeDataPageV2 / decodeDictionary do (column.go:733-843, 963-993):
// per data page the repetition levels (if the column has any), the definition levels and the
// values; per dictionary page its values.  Src, SrcLen, Encoding, Kind, BitWidth and NumValues of
// every item are filled in (Src = chunkDev + the part's offset); the caller adds Dst, Offsets and
// Dict.  chunk is the host copy of the same bytes: the level lengths of v1 pages are read from
// it (column.go:913-924).
func PlanPages(headers []PageHeader, chunk []byte, chunkDev DevicePtr, col ColumnInfo) ([]Item, []Part, error) {
	if len(headers) == 0 {
		return nil, nil, nil
	}
	raw := make([]C.pqd_page_header, len(headers))
	for i := range headers {
		raw[i] = headers[i].raw
	}
	info := C.pqd_column_info{
		kind:                 C.int32_t(col.Kind),
		type_length:          C.int32_t(col.TypeLength),
		max_definition_level: C.int32_t(col.MaxDefinitionLevel),
		max_repetition_level: C.int32_t(col.MaxRepetitionLevel),
	}
	room := 3 * len(headers)
	// the item structs hold device addresses only, but they are pointer-typed fields: they are
	// kept in C memory so that the Go runtime never looks at them
	pages := (*C.pqd_page)(C.calloc(C.size_t(room), C.size_t(unsafe.Sizeof(C.pqd_page{}))))
	parts := make([]C.pqd_page_part, room)
	if pages == nil {
		return nil, nil, fmt.Errorf("pqd: out of memory for %d page descriptors", room)
	}
	defer C.free(unsafe.Pointer(pages))
	var n C.uint64_t
	var arg C.int64_t
	st := C.pqd_chunk_plan_pages(&raw[0], C.uint64_t(len(raw)), (*C.uint8_t)(unsafe.Pointer(unsafe.SliceData(chunk))),
		C.uint64_t(len(chunk)), unsafe.Pointer(uintptr(chunkDev)), &info, pages, &parts[0], C.uint64_t(room), &n, &arg)
	switch st {
	case C.PQD_OK:
	case C.PQD_ERR_UNEXPECTED_EOF:
		return nil, nil, fmt.Errorf("page %d: %w", int64(arg), io.ErrUnexpectedEOF)
	case C.PQD_ERR_NOT_SUPPORTED:
		return nil, nil, fmt.Errorf("page %d: cannot read values of this page type or encoding", int64(arg))
	default:
		return nil, nil, fmt.Errorf("page %d: %s: %w", int64(arg), statusName(st), ErrCorrupted)
	}
	items := make([]Item, int(n))
	outParts := make([]Part, int(n))
	planned := unsafe.Slice(pages, int(n))
	for i := range items {
		p := &planned[i]
		items[i] = Item{
			Src:       DevicePtr(uintptr(p.src)),
			SrcLen:    int(p.src_len),
			NumValues: int64(p.num_values),
			Encoding:  format.Encoding(p.encoding),
			Kind:      Kind(p.kind),
			BitWidth:  int(p.bit_width),
		}
		outParts[i] = Part{Page: int(parts[i].page), Role: Role(parts[i].role), SrcOffset: int(parts[i].src_offset)}
	}
	return items, outParts, nil
}
