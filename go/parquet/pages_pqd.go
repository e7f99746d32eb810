// This file is added to the root package of a parquet-go fork (next to page.go).  It reads the
// pages of a column chunk the way FilePages does (file.go:1193-1259, column.go:733-911), but a
// window of pages at a time on the GPU: the chunk is read once into page-locked memory, its page
// headers are walked by pqd.ScanPages, every part of every page (repetition levels, definition
// levels, values, dictionary) becomes one item of a pqd batch, and while the consumer works on the
// pages of one window the next window is being decoded.  It takes the place of AsyncPages
// (page.go:121-293) for chunks that are neither compressed nor encrypted; the others keep the
// reference's page loop, whose Decode calls reach the same kernels one page at a time through
// the codecs of the `encodings` table.

package parquet

import (
	"fmt"
	"io"
	"runtime"

	"github.com/parquet-go/parquet-go/format"
	"github.com/parquet-go/parquet-go/internal/unsafecast"
	"github.com/parquet-go/parquet-go/pqd"
)

// GPUPagesConfig tunes GPUPages.
type GPUPagesConfig struct {
	// Context is the GPU to decode on; nil is pqd.Default().
	Context *pqd.Context
	// Window is the number of data pages decoded per batch; 0 means 64.  A window is the unit
	// of overlap: window k+1 is on the GPU while the pages of window k are being read.
	Window int
	// Materialize makes dictionary-encoded pages come back as plain pages of values, looked up
	// on the GPU in the same kernel that unpacks the indexes.  When false they come back as the
	// reference's indexed pages (dictionary.go:121-162), which keep the dictionary for writers
	// that copy pages.
	Materialize bool
}

// GPUPages returns a reader of the chunk's pages that decodes them in batches on the GPU.
func (c *FileColumnChunk) GPUPages(config GPUPagesConfig) Pages {
	if isCompressed(c.column.compression) || c.decryptionKey != nil {
		return c.Pages()
	}
	if config.Context == nil {
		config.Context = pqd.Default()
	}
	if config.Window <= 0 {
		config.Window = 64
	}
	return &gpuPages{chunk: c, config: config}
}

type gpuPage struct {
	page     Page
	err      error
	skipRows int64 // rows of a seek that end inside this page
}

type gpuPages struct {
	chunk  *FileColumnChunk
	config GPUPagesConfig
	read   chan gpuPage  // from the producer; closed at the end of the chunk
	done   chan struct{} // closed to stop the producer
	skip   int64         // rows still to drop in front of the next page
	closed bool
}

func (p *gpuPages) start(rowIndex int64) {
	p.stop()
	p.read = make(chan gpuPage, p.config.Window)
	p.done = make(chan struct{})
	p.skip = 0
	go (&gpuProducer{pages: p, read: p.read, done: p.done, fromRow: rowIndex}).run()
}

func (p *gpuPages) stop() {
	if p.done != nil {
		close(p.done)
		for q := range p.read {
			Release(q.page)
		}
		p.done, p.read = nil, nil
	}
}

// ReadPage returns the next page (Pages.ReadPage, page.go:59-66).
func (p *gpuPages) ReadPage() (Page, error) {
	if p.closed {
		return nil, io.EOF
	}
	if p.read == nil {
		p.start(0)
	}
	for {
		q, ok := <-p.read
		if !ok {
			return nil, io.EOF
		}
		if q.err != nil {
			return nil, q.err
		}
		if q.skipRows > 0 { // the producer dropped whole pages; what is left of the seek is inside this one
			p.skip = q.skipRows
		}
		if p.skip > 0 {
			numRows := q.page.NumRows()
			if p.skip >= numRows {
				p.skip -= numRows
				Release(q.page)
				continue
			}
			tail := q.page.Slice(p.skip, numRows)
			Release(q.page)
			p.skip = 0
			return tail, nil
		}
		return q.page, nil
	}
}

// SeekToRow positions the reader at the given row of the chunk (RowSeeker, page.go).
func (p *gpuPages) SeekToRow(rowIndex int64) error {
	if p.closed {
		return io.ErrClosedPipe
	}
	if rowIndex < 0 {
		return fmt.Errorf("%w: %d", ErrSeekOutOfRange, rowIndex)
	}
	p.start(rowIndex)
	return nil
}

// Close stops the producer and releases the pages it had ready.
func (p *gpuPages) Close() error {
	p.stop()
	p.closed = true
	return nil
}

// gpuProducer is the goroutine that owns the GPU buffers of one pass over the chunk.
type gpuProducer struct {
	pages   *gpuPages
	read    chan<- gpuPage
	done    <-chan struct{}
	fromRow int64

	column  *Column
	ctx     *pqd.Context
	stream  *pqd.Stream
	host    pqd.PinnedBuffer // the chunk
	dev     pqd.DeviceBuffer
	dictDev pqd.DeviceBuffer // decoded dictionary: values, then offsets
	out     [2]gpuWindowBuffers

	headers    []pqd.PageHeader
	items      []pqd.Item
	parts      []pqd.Part
	dictionary Dictionary
	dictValues int // entries
	dictBytes  int // of the values
	dictOffsetsAt int // where the offsets of a BYTE_ARRAY dictionary start in dictDev
}

type gpuWindowBuffers struct {
	dev  pqd.DeviceBuffer
	host pqd.PinnedBuffer
}

// gpuWindow is a launched batch: items [first, last) of the plan, decoded into buffers `buf`.
type gpuWindow struct {
	first, last int
	buf         *gpuWindowBuffers
	batch       *pqd.Batch
	at          []int // offset of every item's Dst in the window's buffers
	offsetsAt   []int
	total       int
}

func (g *gpuProducer) emit(page Page, skipRows int64, err error) bool {
	select {
	case g.read <- gpuPage{page: page, err: err, skipRows: skipRows}:
		return true
	case <-g.done:
		Release(page)
		return false
	}
}

// fail hands an error to the consumer; the producer stops after it.
func (g *gpuProducer) fail(err error) bool {
	g.emit(nil, 0, err)
	return false
}

func (g *gpuProducer) run() {
	// the HIP device binding is per OS thread
	runtime.LockOSThread()
	defer runtime.UnlockOSThread()
	defer close(g.read)
	defer g.free()
	if err := g.load(); err != nil {
		g.emit(nil, 0, fmt.Errorf("reading pages of column %q on the GPU: %w", g.column.Name(), err))
		return
	}
	if err := g.decodeAll(); err != nil {
		g.emit(nil, 0, fmt.Errorf("decoding pages of column %q on the GPU: %w", g.column.Name(), err))
	}
}

func (g *gpuProducer) free() {
	g.host.Free()
	g.dev.Free()
	g.dictDev.Free()
	for i := range g.out {
		g.out[i].dev.Free()
		g.out[i].host.Free()
	}
	if g.stream != nil {
		g.stream.Close()
	}
}

// pqdKind maps a column's physical type to the value kind of a batch item.
func pqdKind(kind Kind) pqd.Kind { return pqd.Kind(int32(kind) + 1) }

// load reads the chunk, finds its pages and copies it to the GPU: the work of FilePages.init and
// of the header loop of FilePages.ReadPage (file.go:1118-1150, 1193-1259) for the whole chunk.
func (g *gpuProducer) load() (err error) {
	c := g.pages.chunk
	g.column = c.column
	g.ctx = g.pages.config.Context
	if err = g.ctx.Bind(); err != nil {
		return err
	}
	if g.stream, err = pqd.NewStream(); err != nil {
		return err
	}
	meta := &c.chunk.MetaData
	base := meta.DataPageOffset
	if meta.DictionaryPageOffset != 0 {
		base = meta.DictionaryPageOffset
	}
	size := int(meta.TotalCompressedSize)
	// 64 spare bytes: the kernels read whole 16-byte words up to the end of a page
	if err = g.host.Reserve(size + 64); err != nil {
		return err
	}
	chunk := g.host.Bytes()[:size]
	// io.EOF is what a ReaderAt may return with a full read that ends at the end of the file; fewer bytes
	// than the footer promised would leave the buffer's tail as it was and send it off as page data
	n, err := c.file.reader.ReadAt(chunk, base)
	if err != nil && err != io.EOF {
		return err
	}
	if n != size {
		return fmt.Errorf("column chunk at %d: read %d of %d bytes: %w", base, n, size, io.ErrUnexpectedEOF)
	}
	if g.headers, err = pqd.ScanPages(chunk, meta.NumValues); err != nil {
		return err
	}
	if err = g.dev.Reserve(size + 64); err != nil {
		return err
	}
	if err = g.stream.Upload(g.dev.Ptr(0), chunk); err != nil {
		return err
	}
	typ := c.column.Type()
	info := pqd.ColumnInfo{
		Kind:               pqdKind(typ.Kind()),
		TypeLength:         typ.Length(),
		MaxDefinitionLevel: int(c.column.maxDefinitionLevel),
		MaxRepetitionLevel: int(c.column.maxRepetitionLevel),
	}
	g.items, g.parts, err = pqd.PlanPages(g.headers, chunk, g.dev.Ptr(0), info)
	return err
}

func align256(n int) int { return (n + 255) &^ 255 }

// valueBytes bounds the decoded size of a values item of n values.
func (g *gpuProducer) valueBytes(it *pqd.Item, n int, grow int) int {
	typ := g.column.Type()
	switch {
	case it.Dict != 0 && it.Kind == pqd.ByteArray:
		// looked-up strings: the average dictionary value for every index, more when asked for
		avg := 16
		if g.dictValues > 0 {
			avg = g.dictBytes/g.dictValues + 1
		}
		return max(n*avg*2, grow)
	case it.Kind == pqd.ByteArray:
		enc := LookupEncoding(it.Encoding)
		host := g.host.Bytes()
		at := int(it.Src - g.dev.Ptr(0))
		return max(enc.EstimateDecodeByteArraySize(host[at:at+it.SrcLen]), it.SrcLen, grow)
	case it.Kind == pqd.Boolean && it.Dict == 0:
		return (n+7)/8 + 8
	case it.Kind == pqd.Boolean:
		return n
	case it.Kind == pqd.FixedLenByteArray:
		return n * typ.Length()
	case it.Kind == pqd.Int96:
		return n * 12
	case it.Kind == pqd.Int64 || it.Kind == pqd.Double:
		return n * 8
	default:
		return n * 4
	}
}

// decodeDictionary decodes the chunk's dictionary page into g.dictDev (column.go:963-993) and,
// for indexed pages, builds the reference's Dictionary from a host copy of the values.
func (g *gpuProducer) decodeDictionary(i int) error {
	it := &g.items[i]
	header := &g.headers[g.parts[i].Page]
	n := int(header.NumValues)
	typ := g.column.Type()
	need := g.valueBytes(it, n, 0)
	offsetsAt := align256(need + 64)
	if err := g.dictDev.Reserve(offsetsAt + 4*(n+1) + 64); err != nil {
		return err
	}
	it.Dst, it.DstCap = g.dictDev.Ptr(0), need
	if it.Kind == pqd.ByteArray {
		it.Offsets, it.OffsetsCap = g.dictDev.Ptr(offsetsAt), n+1
	}
	if err := g.ctx.Decode(g.items[i:i+1], g.stream); err != nil {
		return err
	}
	if err := it.Err(); err != nil {
		return err
	}
	if it.OutValues < n && it.Kind != pqd.Boolean {
		return fmt.Errorf("dictionary page holds %d values, its header says %d", it.OutValues, n)
	}
	g.dictValues, g.dictBytes, g.dictOffsetsAt = n, it.OutBytes, offsetsAt
	if g.pages.config.Materialize {
		return nil
	}
	// indexed pages carry the dictionary as an object of the reference's own type
	values := make([]byte, it.OutBytes)
	var offsets []uint32
	stage := &g.out[0].host
	if err := stage.Reserve(offsetsAt + 4*(n+1)); err != nil {
		return err
	}
	if err := g.stream.Download(stage.Bytes()[:it.OutBytes], it.Dst); err != nil {
		return err
	}
	if it.Kind == pqd.ByteArray {
		if err := g.stream.Download(stage.Bytes()[offsetsAt:offsetsAt+4*(n+1)], it.Offsets); err != nil {
			return err
		}
	}
	if err := g.stream.Synchronize(); err != nil {
		return err
	}
	copy(values, stage.Bytes()[:it.OutBytes])
	if it.Kind == pqd.ByteArray {
		offsets = make([]uint32, n+1)
		copy(unsafecastBytes(offsets), stage.Bytes()[offsetsAt:offsetsAt+4*(n+1)])
	}
	g.dictionary = typ.NewDictionary(int(g.column.index), n, typ.NewValues(values, offsets))
	return nil
}

// launch places the items [first, last) in the window's buffers, enqueues their kernels and the
// copy of the results back to page-locked memory, and returns without waiting.
func (g *gpuProducer) launch(first, last int, buf *gpuWindowBuffers, grow map[int]int) (*gpuWindow, error) {
	w := &gpuWindow{first: first, last: last, buf: buf, at: make([]int, last-first), offsetsAt: make([]int, last-first)}
	materialize := g.pages.config.Materialize
	for i := first; i < last; i++ {
		it := &g.items[i]
		header := &g.headers[g.parts[i].Page]
		n := int(header.NumValues)
		need := n // levels: one byte each
		if g.parts[i].Role == pqd.Values {
			if isDictionaryFormat(header.Encoding) {
				if materialize {
					it.Dict, it.DictLen = g.dictDev.Ptr(0), g.dictValues
					if it.Kind == pqd.ByteArray {
						it.DictOffsets = g.dictDev.Ptr(g.dictOffsetsAt)
					}
				} else {
					it.Kind = pqd.Int32 // indexedPageType.Decode (dictionary.go:217-219)
				}
			}
			need = g.valueBytes(it, n, grow[i])
		}
		w.at[i-first] = w.total
		w.total += align256(need + 64)
		if g.parts[i].Role == pqd.Values && it.Kind == pqd.ByteArray {
			w.offsetsAt[i-first] = w.total
			w.total += align256(4*(n+1) + 64)
		}
		it.DstCap = need
	}
	if err := buf.dev.Reserve(w.total); err != nil {
		return nil, err
	}
	if err := buf.host.Reserve(w.total); err != nil {
		return nil, err
	}
	for i := first; i < last; i++ {
		it := &g.items[i]
		it.Dst = buf.dev.Ptr(w.at[i-first])
		if g.parts[i].Role == pqd.Values && it.Kind == pqd.ByteArray {
			it.Offsets = buf.dev.Ptr(w.offsetsAt[i-first])
			it.OffsetsCap = int(g.headers[g.parts[i].Page].NumValues) + 1
		}
	}
	var err error
	if w.batch, err = g.ctx.NewBatch(g.items[first:last]); err != nil {
		return nil, err
	}
	if err = w.batch.Launch(g.stream); err != nil {
		w.batch.Close()
		return nil, err
	}
	if err = g.stream.Download(buf.host.Bytes()[:w.total], buf.dev.Ptr(0)); err != nil {
		w.batch.Close()
		return nil, err
	}
	return w, nil
}

// finish waits for the window and reports the items whose destination was too small.
func (g *gpuProducer) finish(w *gpuWindow) (grow map[int]int, err error) {
	defer w.batch.Close()
	if err = w.batch.Finish(g.stream, g.items[w.first:w.last]); err != nil {
		return nil, err
	}
	for i := w.first; i < w.last; i++ {
		if dstBytes, _, small := g.items[i].TooSmall(); small && dstBytes > 0 {
			if grow == nil {
				grow = make(map[int]int)
			}
			grow[i] = dstBytes
		}
	}
	return grow, nil
}

// decodeAll walks the plan: the dictionary first, then the data pages a window at a time, with
// the pages of one window built on the host while the next window is on the GPU.
func (g *gpuProducer) decodeAll() error {
	first := 0
	if len(g.items) > 0 && g.parts[0].Role == pqd.DictionaryValues {
		if err := g.decodeDictionary(0); err != nil {
			return err
		}
		first = 1
	}
	// whole pages in front of the row sought are never decoded when a page's rows can be told
	// from its header: always for data pages v2, and for v1 when the column is not repeated
	skip := g.fromRow
	for first < len(g.items) {
		header := &g.headers[g.parts[first].Page]
		rows := int64(header.NumRows)
		if rows < 0 {
			if g.column.maxRepetitionLevel > 0 {
				break
			}
			rows = int64(header.NumValues)
		}
		if skip < rows {
			break
		}
		skip -= rows
		for page := g.parts[first].Page; first < len(g.items) && g.parts[first].Page == page; first++ {
		}
	}
	var prev *gpuWindow
	for k := 0; first < len(g.items) || prev != nil; k++ {
		var next *gpuWindow
		if first < len(g.items) {
			last, pages := first, 0
			for last < len(g.items) && pages < g.pages.config.Window {
				for page := g.parts[last].Page; last < len(g.items) && g.parts[last].Page == page; last++ {
				}
				pages++
			}
			var err error
			if next, err = g.launch(first, last, &g.out[k%2], nil); err != nil {
				return err
			}
			first = last
		}
		if prev != nil {
			if !g.build(prev, &skip) {
				if next != nil {
					g.finish(next)
				}
				return nil
			}
		}
		if next != nil {
			grow, err := g.finish(next)
			if err != nil {
				return err
			}
			if grow != nil { // decode the window once more with the sizes the library asked for
				if next, err = g.launch(next.first, next.last, next.buf, grow); err != nil {
					return err
				}
				if grow, err = g.finish(next); err != nil {
					return err
				}
				if grow != nil {
					return fmt.Errorf("a page asked for a larger destination twice")
				}
			}
		}
		prev = next
	}
	return nil
}

// build turns the decoded parts of a window into pages, as Column.decodeDataPage does
// (column.go:845-911), and hands them to the consumer.  It reports false when the consumer is gone.
func (g *gpuProducer) build(w *gpuWindow, skip *int64) bool {
	host := w.buf.host.Bytes()
	typ := g.column.Type()
	for i := w.first; i < w.last; {
		page := g.parts[i].Page
		header := &g.headers[page]
		numValues := int(header.NumValues)
		var repetitionLevels, definitionLevels, values []byte
		var offsets []uint32
		var valuesItem *pqd.Item
		for ; i < w.last && g.parts[i].Page == page; i++ {
			it := &g.items[i]
			if err := it.Err(); err != nil {
				return g.fail(fmt.Errorf("decoding page %d: %w", page, err))
			}
			data := append([]byte(nil), host[w.at[i-w.first]:w.at[i-w.first]+it.OutBytes]...)
			switch g.parts[i].Role {
			case pqd.RepetitionLevels, pqd.DefinitionLevels:
				// decodeLevels (column.go:931-948)
				if len(data) < numValues {
					return g.fail(fmt.Errorf("decoding level expected %d values but got only %d", numValues, len(data)))
				}
				if g.parts[i].Role == pqd.RepetitionLevels {
					repetitionLevels = data
				} else {
					definitionLevels = data
				}
			case pqd.Values:
				values, valuesItem = data, it
				if it.Offsets != 0 {
					n := it.OutValues + 1
					offsets = make([]uint32, n)
					copy(unsafecastBytes(offsets), host[w.offsetsAt[i-w.first]:w.offsetsAt[i-w.first]+4*n])
				}
			}
		}
		if header.NumNulls >= 0 {
			numValues -= int(header.NumNulls) // column.go:841
		} else if definitionLevels != nil {
			numValues -= countLevelsNotEqual(definitionLevels, g.column.maxDefinitionLevel) // column.go:772
		}
		if header.Type == format.DataPageV2 && len(repetitionLevels) != 0 && repetitionLevels[0] != 0 {
			return g.fail(fmt.Errorf("%w: first repetition level for column %d (%s) is %d instead of zero",
				ErrMalformedRepetitionLevel, g.column.Index(), g.column.Name(), repetitionLevels[0]))
		}
		pageType := typ
		if isDictionaryFormat(header.Encoding) && !g.pages.config.Materialize {
			pageType = indexedPageType{newIndexedType(typ, g.dictionary)}
		} else if valuesItem != nil && valuesItem.OutValues < numValues {
			return g.fail(fmt.Errorf("page %d holds %d values, its header says %d", page, valuesItem.OutValues, numValues))
		}
		newPage := pageType.NewPage(g.column.Index(), numValues, pageType.NewValues(values, offsets))
		switch {
		case g.column.maxRepetitionLevel > 0:
			newPage = newRepeatedPage(newPage, g.column.maxRepetitionLevel, g.column.maxDefinitionLevel, repetitionLevels, definitionLevels)
		case g.column.maxDefinitionLevel > 0:
			newPage = newOptionalPage(newPage, g.column.maxDefinitionLevel, definitionLevels)
		}
		if !g.emit(newPage, *skip, nil) {
			return false
		}
		*skip = 0
	}
	return true
}

// unsafecastBytes views a []uint32 as bytes.
func unsafecastBytes(v []uint32) []byte {
	return unsafecast.Slice[byte](v)
}
