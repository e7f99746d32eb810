package pqd

/*
#include <hip/hip_runtime_api.h>
#include "pqd.h"
*/
import "C"

import (
	"fmt"
	"unsafe"
)

// DevicePtr is an address in GPU memory.  It is an integer on purpose: the Go garbage collector
// must not take it for a pointer into the Go heap.
type DevicePtr uintptr

// DeviceBuffer is a block of GPU memory (hipMalloc) that a batching reader reuses from one window
// of pages to the next.
type DeviceBuffer struct {
	ptr  DevicePtr
	size int
}

func hipError(what string, e C.hipError_t) error {
	if e == C.hipSuccess {
		return nil
	}
	return fmt.Errorf("pqd: %s: %s: %w", what, C.GoString(C.hipGetErrorString(e)), ErrDevice)
}

// Bind makes the context's device current for the calling OS thread.  The library entry points
// do this themselves; direct HIP calls (the allocations and copies below) need it from the
// caller, who must also have locked the goroutine to its thread (runtime.LockOSThread).
func (c *Context) Bind() error {
	return hipError("hipSetDevice", C.hipSetDevice(C.int(c.device)))
}

// Reserve makes the buffer at least size bytes large; the contents are lost when it grows.
func (b *DeviceBuffer) Reserve(size int) error {
	if size <= b.size {
		return nil
	}
	b.Free()
	var p unsafe.Pointer
	if err := hipError("hipMalloc", C.hipMalloc(&p, C.size_t(size))); err != nil {
		return err
	}
	b.ptr, b.size = DevicePtr(uintptr(p)), size
	return nil
}

// Free releases the buffer.
func (b *DeviceBuffer) Free() {
	if b.ptr != 0 {
		C.hipFree(unsafe.Pointer(uintptr(b.ptr)))
		b.ptr, b.size = 0, 0
	}
}

// Ptr returns the address of byte `offset` of the buffer.
func (b *DeviceBuffer) Ptr(offset int) DevicePtr { return b.ptr + DevicePtr(offset) }

// Size returns the buffer's capacity in bytes.
func (b *DeviceBuffer) Size() int { return b.size }

// PinnedBuffer is page-locked host memory (hipHostMalloc): copies to and from it run at the full
// PCIe rate and asynchronously to the host.  The memory lies outside the Go heap.
type PinnedBuffer struct {
	ptr  unsafe.Pointer
	size int
}

// Reserve makes the buffer at least size bytes large; the contents are lost when it grows.
func (b *PinnedBuffer) Reserve(size int) error {
	if size <= b.size {
		return nil
	}
	b.Free()
	var p unsafe.Pointer
	if err := hipError("hipHostMalloc", C.hipHostMalloc(&p, C.size_t(size), C.hipHostMallocDefault)); err != nil {
		return err
	}
	b.ptr, b.size = p, size
	return nil
}

// Free releases the buffer.
func (b *PinnedBuffer) Free() {
	if b.ptr != nil {
		C.hipHostFree(b.ptr)
		b.ptr, b.size = nil, 0
	}
}

// Bytes returns the buffer as a slice.  It is valid until the buffer grows or is freed.
func (b *PinnedBuffer) Bytes() []byte {
	if b.ptr == nil {
		return nil
	}
	return unsafe.Slice((*byte)(b.ptr), b.size)
}

// Stream is a HIP stream: the batches of one reader are ordered on one stream, different readers
// use different streams and overlap on the device.
type Stream struct{ h C.hipStream_t }

// NewStream creates a stream on the current device.
func NewStream() (*Stream, error) {
	s := new(Stream)
	if err := hipError("hipStreamCreateWithFlags", C.hipStreamCreateWithFlags(&s.h, C.hipStreamNonBlocking)); err != nil {
		return nil, err
	}
	return s, nil
}

// Close destroys the stream.
func (s *Stream) Close() {
	if s.h != nil {
		C.hipStreamDestroy(s.h)
		s.h = nil
	}
}

// Synchronize waits for everything enqueued on the stream.
func (s *Stream) Synchronize() error {
	return hipError("hipStreamSynchronize", C.hipStreamSynchronize(s.h))
}

func (s *Stream) raw() unsafe.Pointer {
	if s == nil {
		return nil
	}
	return unsafe.Pointer(s.h)
}

// Upload enqueues a copy of src (pinned host memory for the copy to be asynchronous) to dst.
func (s *Stream) Upload(dst DevicePtr, src []byte) error {
	if len(src) == 0 {
		return nil
	}
	return hipError("hipMemcpyAsync", C.hipMemcpyAsync(unsafe.Pointer(uintptr(dst)), unsafe.Pointer(unsafe.SliceData(src)),
		C.size_t(len(src)), C.hipMemcpyHostToDevice, s.h))
}

// Download enqueues a copy of len(dst) bytes at src to dst (pinned host memory).
func (s *Stream) Download(dst []byte, src DevicePtr) error {
	if len(dst) == 0 {
		return nil
	}
	return hipError("hipMemcpyAsync", C.hipMemcpyAsync(unsafe.Pointer(unsafe.SliceData(dst)), unsafe.Pointer(uintptr(src)),
		C.size_t(len(dst)), C.hipMemcpyDeviceToHost, s.h))
}

// HostSlab is page-locked host memory that the context knows about: the per-page codec methods
// (DecodeInt64, DecodeByteArray, ...) copy straight out of and into slices that lie inside it and
// skip their own staging copies.  A reader's buffer pool cuts its page and value buffers from one.
type HostSlab struct {
	ctx        *Context
	ptr        unsafe.Pointer
	size       int
	registered bool // page-locked in place (RegisterHostMemory) rather than allocated by the library
}

// NewHostSlab allocates size bytes of page-locked memory outside the Go heap (pqd_host_alloc).
func (c *Context) NewHostSlab(size int) (*HostSlab, error) {
	var p unsafe.Pointer
	if st := C.pqd_host_alloc(c.handle(), C.uint64_t(size), &p); st != C.PQD_OK {
		return nil, fmt.Errorf("pqd: page-locked host memory: %s: %w", statusName(st), ErrDevice)
	}
	return &HostSlab{ctx: c, ptr: p, size: size}, nil
}

// RegisterHostMemory page-locks memory the caller already holds (pqd_host_register).  The memory
// must not move or be freed before Release: C memory or a pinned slice (runtime.Pinner), never a
// plain Go slice.  Drivers that cannot lock user memory return ErrDevice.
func (c *Context) RegisterHostMemory(mem []byte) (*HostSlab, error) {
	if len(mem) == 0 {
		return nil, fmt.Errorf("pqd: no memory to register: %w", ErrDevice)
	}
	p := unsafe.Pointer(unsafe.SliceData(mem))
	if st := C.pqd_host_register(c.handle(), p, C.uint64_t(len(mem))); st != C.PQD_OK {
		return nil, fmt.Errorf("pqd: page-locked host memory: %s: %w", statusName(st), ErrDevice)
	}
	return &HostSlab{ctx: c, ptr: p, size: len(mem), registered: true}, nil
}

// Bytes returns the slab as a slice, valid until Release.
func (s *HostSlab) Bytes() []byte {
	if s.ptr == nil {
		return nil
	}
	return unsafe.Slice((*byte)(s.ptr), s.size)
}

// Release frees (or unlocks) the memory; no call that uses it may still be running.
func (s *HostSlab) Release() error {
	if s.ptr == nil {
		return nil
	}
	var st C.int
	if s.registered {
		st = C.pqd_host_unregister(s.ctx.handle(), s.ptr)
	} else {
		st = C.pqd_host_free(s.ctx.handle(), s.ptr)
	}
	s.ptr, s.size = nil, 0
	if st != C.PQD_OK {
		return fmt.Errorf("pqd: releasing page-locked host memory: %s: %w", statusName(st), ErrDevice)
	}
	return nil
}
