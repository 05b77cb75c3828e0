// Package gdg is the thin cgo binding of libgdg.so (include/gdg.h): the MI355X-native batch
// implementation of go-dsp-guitar's per-channel effects pipeline.
//
// NOT compiled in the authoring container (no Go toolchain there); it is the stub a maintainer adds
// next to the reference tree, see INTEGRATION.md.  It contains no DSP: every function is one C call.
//
//	CGO_CFLAGS="-I<repo>/include" CGO_LDFLAGS="-L<repo>/go-dsp-guitar_amd/lib -lgdg -Wl,-rpath,<repo>/go-dsp-guitar_amd/lib"
package gdg

/*
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "gdg.h"
*/
import "C"

import (
	"fmt"
	"unsafe"
)

// Context is one shard of channels on one GPU (gdg_ctx).
type Context struct {
	ctx    *C.gdg_ctx
	in     unsafe.Pointer // pinned host slab, row c = channel c (gdg_staging_buffers)
	out    unsafe.Pointer
	stride int
}

func (this *Context) err(rc C.int) error {
	if rc == C.GDG_OK {
		return nil
	}
	return fmt.Errorf("gdg: %s (code %d)", C.GoString(C.gdg_last_error(this.ctx)), int(rc))
}

// CreateContext replaces N x signal.CreateChain + spatializer.Create + tuner.Create for one shard
// (controller/controller.go:3267-3279).
func CreateContext(channels int, maxFrames int, device int) (*Context, error) {
	c := &Context{}
	rc := C.gdg_ctx_create(C.int(channels), C.int(maxFrames), C.int(device), &c.ctx)
	if rc != C.GDG_OK {
		return nil, fmt.Errorf("gdg_ctx_create failed with code %d (no usable HIP device; there is no CPU fallback)", int(rc))
	}
	var in, out *C.double
	var stride C.int
	if err := c.err(C.gdg_staging_buffers(c.ctx, &in, &out, &stride)); err != nil {
		return nil, err
	}
	c.in, c.out, c.stride = unsafe.Pointer(in), unsafe.Pointer(out), int(stride)
	return c, nil
}

func (this *Context) Destroy() { C.gdg_ctx_destroy(this.ctx) }

// UnitCreate: effects.CreateUnit(unitType) on the device side (effects/effects.go:443-516).
func (this *Context) UnitCreate(channel int, unitType int) (int, error) {
	var h C.int
	err := this.err(C.gdg_unit_create(this.ctx, C.int(channel), C.int(unitType), &h))
	return int(h), err
}

func (this *Context) UnitDestroy(handle int) error {
	return this.err(C.gdg_unit_destroy(this.ctx, C.int(handle)))
}

// UnitSetParam passes one RESOLVED parameter: the int32 of a numeric parameter or the index of a
// discrete value.  Name lookup and range checks stay in the reference's effects package.
func (this *Context) UnitSetParam(handle int, index int, value int32) error {
	return this.err(C.gdg_unit_set_param(this.ctx, C.int(handle), C.int(index), C.int32_t(value)))
}

// UnitSetFir hands over the composite taps of a power amp (what poweramp.compile() returns).
func (this *Context) UnitSetFir(handle int, taps []float64) error {
	var p *C.double
	if len(taps) > 0 {
		p = (*C.double)(unsafe.Pointer(&taps[0])) // a []float64 holds no Go pointers: legal for the duration of the call
	}
	return this.err(C.gdg_unit_set_fir(this.ctx, C.int(handle), p, C.int(len(taps))))
}

// UnitCompileFir runs poweramp.compile() on the device (effects/poweramp.go:25-127): slot i = taps of impulse response i at the
// current rate (nil = "- NONE -"), its gain compensation factor and its level in dB.  An array of Go slices would be a
// pointer to Go pointers, which cgo forbids, so every slot is copied into C memory for the duration of the call.
func (this *Context) UnitCompileFir(handle int, taps [][]float64, compensation []float64, levelsDb []int32, targetOrder uint32) error {
	n := len(taps)
	if n == 0 {
		return this.err(C.gdg_unit_compile_fir(this.ctx, C.int(handle), 0, nil, nil, nil, nil, C.uint32_t(targetOrder)))
	}
	ptrSize := C.size_t(unsafe.Sizeof(uintptr(0)))
	ptrs := (*[1 << 20]*C.double)(C.calloc(C.size_t(n), ptrSize))
	defer C.free(unsafe.Pointer(ptrs))
	lens := make([]C.int, n)
	comp := make([]C.double, n)
	lev := make([]C.int32_t, n)
	for i := 0; i < n; i++ {
		comp[i] = C.double(compensation[i])
		lev[i] = C.int32_t(levelsDb[i])
		if len(taps[i]) == 0 {
			continue
		}
		bytes := C.size_t(len(taps[i])) * 8
		mem := C.malloc(bytes)
		defer C.free(mem)
		C.memcpy(mem, unsafe.Pointer(&taps[i][0]), bytes)
		ptrs[i] = (*C.double)(mem)
		lens[i] = C.int(len(taps[i]))
	}
	return this.err(C.gdg_unit_compile_fir(this.ctx, C.int(handle), C.int(n), (**C.double)(unsafe.Pointer(ptrs)), &lens[0], &comp[0], &lev[0], C.uint32_t(targetOrder)))
}

func (this *Context) ChainSet(channel int, handles []int, bypass []bool) error {
	n := len(handles)
	hs := make([]C.int, n+1)
	bs := make([]C.uint8_t, n+1)
	for i := 0; i < n; i++ {
		hs[i] = C.int(handles[i])
		if bypass[i] {
			bs[i] = 1
		}
	}
	return this.err(C.gdg_chain_set(this.ctx, C.int(channel), &hs[0], &bs[0], C.int(n)))
}

// Row returns channel c's rows of the pinned staging slabs as Go slices over C memory.
func (this *Context) Row(channel int, frames int) (in []float64, out []float64) {
	off := uintptr(channel * this.stride * 8)
	in = unsafe.Slice((*float64)(unsafe.Pointer(uintptr(this.in)+off)), frames)
	out = unsafe.Slice((*float64)(unsafe.Pointer(uintptr(this.out)+off)), frames)
	return in, out
}

// ProcessStaged runs the chains of the listed channels on the frames deposited in the staging rows.
func (this *Context) ProcessStaged(channels []int, frames int, sampleRate uint32) error {
	cs := make([]C.int, len(channels))
	for i, c := range channels {
		cs[i] = C.int(c)
	}
	return this.err(C.gdg_process_staged(this.ctx, &cs[0], C.int(len(cs)), C.int(frames), C.uint32_t(sampleRate)))
}
