// Package pqd puts the MI355X page-value decoders of libpqd.so behind parquet-go's codec
// interface (encoding.Encoding, encoding/encoding.go:20-72) and its dictionary lookup
// (Dictionary.Lookup, dictionary.go:64-69).
//
// The package lives inside a fork of parquet-go as github.com/parquet-go/parquet-go/pqd and
// imports only the encoding, format, deprecated and sparse sub-packages, so the root package can
// import it (patches/encodings.diff repoints the `encodings` table of encoding.go:50-85 at the
// codecs defined here).  Every Decode method is one call into the C ABI declared in
// include/pqd.h; the values are decoded by HIP kernels on the GPU and there is no CPU decode
// path: without a usable device the first decode panics with the library's error.
//
// Build: CGO_CFLAGS="-I<repo>/include -I/opt/rocm/include"
// CGO_LDFLAGS="-L<repo>/parquet-go_amd -L/opt/rocm/lib" go build ./pqd
package pqd

/*
#cgo CFLAGS: -D__HIP_PLATFORM_AMD__
#cgo LDFLAGS: -lpqd -lamdhip64
#include <stdlib.h>
#include "pqd.h"
*/
import "C"

import (
	"fmt"
	"os"
	"strconv"
	"sync"
)

// Context binds the library to one GPU (pqd_ctx_create).  A context is safe for concurrent use:
// every host entry point leases a stream and staging buffers of its own from the context for the
// length of the call ("Encoding instances must be safe to use concurrently from multiple
// goroutines", encoding/encoding.go:19).
type Context struct {
	h      *C.pqd_ctx
	device int
}

// NewContext opens the given HIP device.
func NewContext(device int) (*Context, error) {
	if got := int(C.pqd_abi_version()); got != int(C.PQD_ABI_VERSION) {
		return nil, fmt.Errorf("pqd: libpqd.so has ABI version %d, this package was written for %d", got, int(C.PQD_ABI_VERSION))
	}
	var h *C.pqd_ctx
	if st := C.pqd_ctx_create(C.int(device), &h); st != C.PQD_OK {
		return nil, fmt.Errorf("pqd: opening GPU %d: %s", device, statusName(st))
	}
	return &Context{h: h, device: device}, nil
}

// Device returns the HIP device ordinal of the context.
func (c *Context) Device() int { return c.device }

// Close releases the context; no call may be in flight.
func (c *Context) Close() {
	if c.h != nil {
		C.pqd_ctx_destroy(c.h)
		c.h = nil
	}
}

var defaultContext struct {
	once sync.Once
	ctx  *Context
	err  error
}

// Default returns the process-wide context the zero-value codecs use: device PQD_DEVICE
// (default 0), opened on first use.  It panics when the device cannot be opened, because a codec
// has no other way to decode.
func Default() *Context {
	defaultContext.once.Do(func() {
		device := 0
		if s := os.Getenv("PQD_DEVICE"); s != "" {
			if v, err := strconv.Atoi(s); err == nil {
				device = v
			}
		}
		defaultContext.ctx, defaultContext.err = NewContext(device)
	})
	if defaultContext.err != nil {
		panic(defaultContext.err)
	}
	return defaultContext.ctx
}

func statusName(st C.int) string { return C.GoString(C.pqd_status_name(st)) }
