// Package gdg is the thin cgo binding of libgdg.so (include/gdg.h): the MI355X-native batch
// implementation of go-dsp-guitar's per-channel effects pipeline.
//
// NOT compiled in the authoring container (no Go toolchain there); it is the directory a maintainer copies into the reference
// checkout as <ref>/gdg (package github.com/andrepxx/go-dsp-guitar/gdg, INTEGRATION.md section 3 -- cgo needs the package
// directory on disk, an overlay-only directory will not do).  Written for the language level of the reference's go.mod (go 1.16):
// no unsafe.Slice / unsafe.Add, no generics, no `any`, no typed atomics (tests/test_go_sources.py holds the deny-list).
// It contains no DSP: every function is one C call.
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
	"os"
	"strconv"
	"strings"
	"sync"
	"unsafe"
)

// Context is one shard of channels on one GPU (gdg_ctx).  A context takes ONE call at a time (include/gdg.h): callers that
// share it serialise on Shard.Mutex.
type Context struct {
	ctx      *C.gdg_ctx
	in       unsafe.Pointer // pinned host slab, row c = channel c (gdg_staging_buffers)
	out      unsafe.Pointer
	stride   int
	channels int
	scratch  unsafe.Pointer // C memory for MetersProcess: [ports pointers | ports x frames float64]
	scratchN int
}

// DeviceCount: HIP devices visible to the process (0 without a driver).
func DeviceCount() int { return int(C.gdg_device_count()) }

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
	c.in, c.out, c.stride, c.channels = unsafe.Pointer(in), unsafe.Pointer(out), int(stride), channels
	return c, nil
}

func (this *Context) Destroy() {
	if this.scratch != nil {
		C.free(this.scratch)
		this.scratch = nil
	}
	C.gdg_ctx_destroy(this.ctx)
}

func (this *Context) Channels() int  { return this.channels }
func (this *Context) MaxFrames() int { return this.stride }

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

// Row returns channel c's rows of the pinned staging slabs as Go slices over C memory.  The slab has `channels` rows of
// `stride` (= max_frames) float64: anything outside is an error, never a slice (a longer slice would run into the next
// channel's row and, for the last channel, past the hipHostMalloc slab).
func (this *Context) Row(channel int, frames int) (in []float64, out []float64, err error) {
	if channel < 0 || channel >= this.channels {
		return nil, nil, fmt.Errorf("gdg: channel %d out of range (the context has %d)", channel, this.channels)
	}
	if frames < 0 || frames > this.stride {
		return nil, nil, fmt.Errorf("gdg: %d frames do not fit a staging row of %d", frames, this.stride)
	}
	// slices over C memory the Go 1.16 way (the reference's go.mod:3 says `go 1.16`; unsafe.Slice is 1.17): a pointer to a
	// huge array type, sliced down to the row with its capacity capped
	off := uintptr(channel) * uintptr(this.stride) * 8
	in = (*[1 << 37]float64)(unsafe.Pointer(uintptr(this.in) + off))[:frames:frames]
	out = (*[1 << 37]float64)(unsafe.Pointer(uintptr(this.out) + off))[:frames:frames]
	return in, out, nil
}

// ProcessStaged runs the chains of the listed channels on the frames deposited in the staging rows.
func (this *Context) ProcessStaged(channels []int, frames int, sampleRate uint32) error {
	if len(channels) == 0 {
		return nil // nothing to run; &cs[0] of an empty slice would panic
	}
	cs := make([]C.int, len(channels))
	for i, c := range channels {
		cs[i] = C.int(c)
	}
	return this.err(C.gdg_process_staged(this.ctx, &cs[0], C.int(len(cs)), C.int(frames), C.uint32_t(sampleRate)))
}

// ---- tuner: tuner.Process / tuner.Analyze (tuner/tuner.go:379-587) ------------------------------------------------

type TunerResult struct {
	Frequency float64
	NoteIndex int // index into the 61-note table, -1 = "Unknown"
	Cents     int8
}

// TunerEnqueueStaged: tuner.Process for every channel of the context from the pinned INPUT rows (fill them through Row).
func (this *Context) TunerEnqueueStaged(frames int, sampleRate uint32) error {
	return this.err(C.gdg_tuner_enqueue_staged(this.ctx, C.int(frames), C.uint32_t(sampleRate)))
}

// TunerReplace: the whole ring of one channel at once -- len(samples) must be the ring's length (tuner.NUM_SAMPLES = 96000, oldest
// sample first: what circular.Buffer.Retrieve hands out); anything else is an error from the library, never a partial upload.
func (this *Context) TunerReplace(channel int, samples []float64, sampleRate uint32) error {
	if len(samples) == 0 {
		return fmt.Errorf("gdg: an empty ring")
	}
	return this.err(C.gdg_tuner_replace(this.ctx, C.int(channel), (*C.double)(unsafe.Pointer(&samples[0])), C.int(len(samples)), C.uint32_t(sampleRate)))
}

// TunerAnalyze: tuner.Analyze for every channel of the context.
func (this *Context) TunerAnalyze() ([]TunerResult, error) {
	n := this.channels
	raw := (*[1 << 20]C.gdg_tuner_result)(C.calloc(C.size_t(n), C.size_t(unsafe.Sizeof(C.gdg_tuner_result{}))))
	defer C.free(unsafe.Pointer(raw))
	if err := this.err(C.gdg_tuner_analyze(this.ctx, &raw[0])); err != nil {
		return nil, err
	}
	res := make([]TunerResult, n)
	for i := 0; i < n; i++ {
		res[i] = TunerResult{Frequency: float64(raw[i].frequency), NoteIndex: int(raw[i].note_index), Cents: int8(raw[i].cents)}
	}
	return res, nil
}

// TunerNoteName: name of a note of the reference's table (tuner/tuner.go:79-324), "Unknown" for -1.
func TunerNoteName(index int) string { return C.GoString(C.gdg_tuner_note_name(C.int(index))) }

// ---- spatializer: partial N -> 2 mix of this shard (spatializer/spatializer.go:140-335) -----------------------------

func (this *Context) SpatializerSetPosition(channel int, azimuth float64, distance float64, level float64) error {
	return this.err(C.gdg_spatializer_set_position(this.ctx, C.int(channel), C.double(azimuth), C.double(distance), C.double(level)))
}

func (this *Context) SpatializerSetSampleRate(rate uint32) error {
	return this.err(C.gdg_spatializer_set_sample_rate(this.ctx, C.uint32_t(rate)))
}

// SpatializeStaged mixes this shard's channels into left / right (len = frames).  fromOutputs: the inputs are the chain outputs
// of the last ProcessStaged over ALL channels of the context, still on the device (nothing is uploaded); otherwise the pinned
// INPUT rows.  left / right are []float64 (no Go pointers inside): legal cgo arguments for the duration of the call.
func (this *Context) SpatializeStaged(fromOutputs bool, left []float64, right []float64) error {
	if len(left) == 0 || len(left) != len(right) {
		return fmt.Errorf("gdg: bad output buffers")
	}
	f := C.int(0)
	if fromOutputs {
		f = 1
	}
	return this.err(C.gdg_spatialize_staged(this.ctx, f, (*C.double)(unsafe.Pointer(&left[0])), (*C.double)(unsafe.Pointer(&right[0])), C.int(len(left))))
}

// ---- shards: one context per GPU, channel c on shard c * G / N (contiguous blocks; SURVEY.md 8e, controller.go:3262-3269) ----

type Shard struct {
	Ctx    *Context
	Device int
	First  int // first global channel of the block
	Count  int
	Mutex  sync.Mutex // one call at a time per context
}

var (
	shardMutex sync.Mutex
	shardList  []*Shard
	shardTotal int
)

// devices: GDG_DEVICES="0,1,2,..." (one shard per entry; an entry may repeat), else GDG_DEVICE, else every visible device.
func devices() []int {
	if list := os.Getenv("GDG_DEVICES"); list != "" {
		var devs []int
		for _, item := range strings.Split(list, ",") {
			if d, err := strconv.Atoi(strings.TrimSpace(item)); err == nil {
				devs = append(devs, d)
			}
		}
		if len(devs) > 0 {
			return devs
		}
	}
	if one := os.Getenv("GDG_DEVICE"); one != "" {
		d, _ := strconv.Atoi(one)
		return []int{d}
	}
	n := DeviceCount()
	if n < 1 {
		n = 1
	}
	devs := make([]int, n)
	for i := range devs {
		devs[i] = i
	}
	return devs
}

// Shards creates (once) the contexts for a job of totalChannels channels and returns them.  The same algorithm as
// gdg::Engine::Engine in host/gdg_host.cpp: never more shards than channels, shard g owns [g N / G, (g + 1) N / G).
func Shards(totalChannels int, maxFrames int) ([]*Shard, error) {
	shardMutex.Lock()
	defer shardMutex.Unlock()
	if shardList != nil {
		if totalChannels > shardTotal {
			return nil, fmt.Errorf("gdg: the shards were created for %d channels, %d requested", shardTotal, totalChannels)
		}
		return shardList, nil
	}
	devs := devices()
	G := len(devs)
	if G > totalChannels {
		G = totalChannels
	}
	if G < 1 {
		G = 1
	}
	list := make([]*Shard, 0, G)
	for g := 0; g < G; g++ {
		first := g * totalChannels / G
		count := (g+1)*totalChannels/G - first
		ctx, err := CreateContext(count, maxFrames, devs[g])
		if err != nil {
			for _, sh := range list {
				sh.Ctx.Destroy()
			}
			return nil, err
		}
		list = append(list, &Shard{Ctx: ctx, Device: devs[g], First: first, Count: count})
	}
	shardList, shardTotal = list, totalChannels
	return shardList, nil
}

// ShardOf: the shard of a global channel and the channel's index inside it (nil before Shards() or out of range).
func ShardOf(channel int) (*Shard, int) {
	shardMutex.Lock()
	defer shardMutex.Unlock()
	for _, sh := range shardList {
		if channel >= sh.First && channel < sh.First+sh.Count {
			return sh, channel - sh.First
		}
	}
	return nil, -1
}

// ---- the data formats either side of the chain (optional; include/gdg.h "data formats" section) ----------------

// WaveDecode: bytesToSamples + samplesToChannels (wave/wave.go:237-270, :790-838).  data = the data chunk of a RIFF/WAVE file
// (a []byte holds no Go pointers: legal for the duration of the call); returns planar samples, one slice per channel.
func (this *Context) WaveDecode(format int, data []byte, channels int) ([][]float64, error) {
	width := int(C.gdg_wave_bytes_per_sample(C.int(format)))
	if width == 0 || channels <= 0 {
		return nil, fmt.Errorf("gdg: unknown sample format %d or bad channel count %d", format, channels)
	}
	per := len(data) / (width * channels)
	flat := make([]float64, per*channels)
	if per > 0 {
		rc := C.gdg_wave_decode(this.ctx, C.int(format), unsafe.Pointer(&data[0]), C.size_t(per), C.uint(channels), (*C.double)(unsafe.Pointer(&flat[0])))
		if err := this.err(rc); err != nil {
			return nil, err
		}
	}
	out := make([][]float64, channels)
	for c := range out {
		out[c] = flat[c*per : (c+1)*per]
	}
	return out, nil
}

// WaveEncode: channelsToSamples + samplesToBytes (wave/wave.go:173-232, :737-785) of equally long channels.
func (this *Context) WaveEncode(format int, channels [][]float64) ([]byte, error) {
	width := int(C.gdg_wave_bytes_per_sample(C.int(format)))
	if width == 0 || len(channels) == 0 {
		return nil, fmt.Errorf("gdg: unknown sample format %d or no channels", format)
	}
	per := len(channels[0])
	flat := make([]float64, 0, per*len(channels))
	for _, ch := range channels {
		flat = append(flat, ch[:per]...)
	}
	data := make([]byte, per*len(channels)*width)
	if per == 0 {
		return data, nil
	}
	rc := C.gdg_wave_encode(this.ctx, C.int(format), (*C.double)(unsafe.Pointer(&flat[0])), C.size_t(per), C.uint(len(channels)), unsafe.Pointer(&data[0]))
	return data, this.err(rc)
}

// ResampleTime: resample.Time (resample/resample.go:72-103).
func (this *Context) ResampleTime(samples []float64, sourceRate uint32, targetRate uint32) ([]float64, error) {
	n := C.gdg_resample_time_length(C.int(len(samples)), C.uint32_t(sourceRate), C.uint32_t(targetRate))
	if n <= 0 || len(samples) == 0 {
		return []float64{}, nil
	}
	out := make([]float64, int(n))
	rc := C.gdg_resample_time(this.ctx, (*C.double)(unsafe.Pointer(&samples[0])), C.int(len(samples)), C.uint32_t(sourceRate), C.uint32_t(targetRate),
		(*C.double)(unsafe.Pointer(&out[0])), n)
	return out, this.err(rc)
}

// MetersConfigure / MetersSetEnabled / MetersAnalyze: level.Meter over n ports (level/level.go:100-279).  Buffers reach the meters
// through MetersProcessDevice on rows already resident on the device (the staging slab's device twin) or through
// gdg_meter_process with C-allocated rows; Go slices of slices cannot cross cgo.
func (this *Context) MetersConfigure(ports int) error { return this.err(C.gdg_meter_configure(this.ctx, C.int(ports))) }
func (this *Context) MetersSetEnabled(port int, enabled bool) error {
	e := C.int(0)
	if enabled {
		e = 1
	}
	return this.err(C.gdg_meter_set_enabled(this.ctx, C.int(port), e))
}

// MetersProcess: level.Meter.Process over `buffers` (one per port, level/level.go:302-325).  A [][]float64 is a pointer to Go
// pointers and cannot cross cgo, so the buffers are copied into C memory owned by the context.
func (this *Context) MetersProcess(buffers [][]float64, sampleRate uint32) error {
	ports := len(buffers)
	if ports == 0 {
		return nil
	}
	frames := len(buffers[0])
	ptrBytes := ports * int(unsafe.Sizeof(uintptr(0)))
	need := ptrBytes + ports*frames*8
	if need > this.scratchN {
		if this.scratch != nil {
			C.free(this.scratch)
		}
		this.scratch = C.malloc(C.size_t(need))
		this.scratchN = need
	}
	ptrs := (*[1 << 20]*C.double)(this.scratch)
	data := unsafe.Pointer(uintptr(this.scratch) + uintptr(ptrBytes))
	for p, buf := range buffers {
		if len(buf) != frames {
			return fmt.Errorf("gdg: meter buffers must be of equal length")
		}
		row := unsafe.Pointer(uintptr(data) + uintptr(p*frames*8))
		if frames > 0 {
			C.memcpy(row, unsafe.Pointer(&buf[0]), C.size_t(frames*8))
		}
		ptrs[p] = (*C.double)(row) // a C pointer stored in C memory: allowed
	}
	return this.err(C.gdg_meter_process(this.ctx, (**C.double)(this.scratch), C.int(frames), C.uint32_t(sampleRate)))
}

func (this *Context) MetersAnalyze(ports int) (levels []int32, peaks []int32, err error) {
	levels, peaks = make([]int32, ports), make([]int32, ports)
	if ports == 0 {
		return levels, peaks, nil
	}
	err = this.err(C.gdg_meter_analyze(this.ctx, (*C.int32_t)(unsafe.Pointer(&levels[0])), (*C.int32_t)(unsafe.Pointer(&peaks[0]))))
	return levels, peaks, err
}

// Metronome: metronome.Metronome (metronome/metronome.go:63-131).
func (this *Context) MetronomeSetTick(coefficients []float64) error {
	if coefficients == nil {
		return this.err(C.gdg_metronome_set_tick(this.ctx, nil, 0))
	}
	var dummy C.double
	p := &dummy
	if len(coefficients) > 0 {
		p = (*C.double)(unsafe.Pointer(&coefficients[0]))
	}
	return this.err(C.gdg_metronome_set_tick(this.ctx, p, C.int(len(coefficients))))
}
func (this *Context) MetronomeSetTock(coefficients []float64) error {
	if coefficients == nil {
		return this.err(C.gdg_metronome_set_tock(this.ctx, nil, 0))
	}
	var dummy C.double
	p := &dummy
	if len(coefficients) > 0 {
		p = (*C.double)(unsafe.Pointer(&coefficients[0]))
	}
	return this.err(C.gdg_metronome_set_tock(this.ctx, p, C.int(len(coefficients))))
}
func (this *Context) MetronomeConfigure(beatsPerPeriod uint32, bpmSpeed uint32, sampleRate uint32) error {
	return this.err(C.gdg_metronome_configure(this.ctx, C.uint32_t(beatsPerPeriod), C.uint32_t(bpmSpeed), C.uint32_t(sampleRate)))
}
func (this *Context) MetronomeProcess(out []float64) error {
	if len(out) == 0 {
		return nil
	}
	return this.err(C.gdg_metronome_process(this.ctx, (*C.double)(unsafe.Pointer(&out[0])), C.int(len(out))))
}

// SetWindow: the batch run steps through the files `frames` (1, 2, 4, 8 or 16) blocks at a time; every power amp then reads its IR
// spectra and its delay line once per step instead of once per block (gdg_ctx_set_window).
func (this *Context) SetWindow(frames int) error {
	return this.err(C.gdg_ctx_set_window(this.ctx, C.int(frames)))
}

// SetOverlap: free-running channel groups of the device-resident calls, opt-in (gdg_ctx_set_overlap: 0 or 1 = one group on the
// context's stream, the default; > 1: groups on streams of their own, joined by the next call of any other kind).
func (this *Context) SetOverlap(groups int) error {
	return this.err(C.gdg_ctx_set_overlap(this.ctx, C.int(groups)))
}

// SetOption / Option: the library's launch-shape options (gdg_ctx_set_option in include/gdg.h lists the keys) -- what used to be
// environment variables of the process.  The drop-in needs none of them; they are here for deployments that tune a node
// ("numa", "copy_threads", "seg_wave_max_channels", ...).
func (this *Context) SetOption(key string, value int64) error {
	k := C.CString(key)
	defer C.free(unsafe.Pointer(k))
	return this.err(C.gdg_ctx_set_option(this.ctx, k, C.longlong(value)))
}
func (this *Context) Option(key string) (int64, error) {
	k := C.CString(key)
	defer C.free(unsafe.Pointer(k))
	var v C.longlong
	if e := this.err(C.gdg_ctx_get_option(this.ctx, k, &v)); e != nil {
		return 0, e
	}
	return int64(v), nil
}

// BatchInput: one input file of the batch run -- the data section of a RIFF/WAVE file (wave.go:840-1100 parses the header and
// knows Format / BitDepth / SampleRate / ChannelCount), and the channel of it that feeds the input.  Data == nil leaves the
// channel empty (controller.go:2935).
type BatchInput struct {
	Data       []byte
	Format     int // gdg_wave_format
	SampleRate uint32
	Channels   int
	Channel    int
}

type BatchOptions struct {
	TargetRate        uint32
	OutFormat         int
	MetronomeToMaster bool
	RunMeters         bool
	TunerEnqueue      bool
}

func cbool(b bool) C.int {
	if b {
		return 1
	}
	return 0
}

// BatchRun: controller.processFiles between "the files are read" and "the files are written" (controller/controller.go:2884-3219)
// in one call.  The C structs hold pointers, so the file bytes live in C memory for the duration of the call (C.CBytes: one copy), and so do the N + 3 output data sections.
func (this *Context) BatchRun(inputs []BatchInput, opt BatchOptions) ([][]byte, error) {
	n := len(inputs)
	if n == 0 {
		return nil, fmt.Errorf("gdg: no inputs")
	}
	arr, release, err := batchInputs(inputs)
	if err != nil {
		return nil, err
	}
	defer release()
	var owned []unsafe.Pointer
	defer func() {
		for _, p := range owned {
			C.free(p)
		}
	}()
	o := C.gdg_batch_options{target_rate: C.uint32_t(opt.TargetRate), out_format: C.int(opt.OutFormat),
		metronome_to_master: cbool(opt.MetronomeToMaster), run_meters: cbool(opt.RunMeters), tuner_enqueue: cbool(opt.TunerEnqueue)}
	var samples C.size_t
	if e := this.err(C.gdg_batch_length(this.ctx, &arr[0], C.int(n), o.target_rate, &samples)); e != nil {
		return nil, e
	}
	each := int(samples) * int(C.gdg_wave_bytes_per_sample(o.out_format))
	outs := make([][]byte, n+3)
	if each == 0 {
		for i := range outs {
			outs[i] = []byte{}
		}
		return outs, nil
	}
	ptrs := (*[1 << 20]unsafe.Pointer)(C.calloc(C.size_t(n+3), C.size_t(unsafe.Sizeof(uintptr(0)))))
	if ptrs == nil {
		return nil, fmt.Errorf("gdg: out of memory")
	}
	defer C.free(unsafe.Pointer(ptrs))
	for i := 0; i < n+3; i++ {
		ptrs[i] = C.malloc(C.size_t(each))
		if ptrs[i] == nil {
			return nil, fmt.Errorf("gdg: out of memory (%d bytes per output)", each)
		}
		owned = append(owned, ptrs[i])
	}
	if e := this.err(C.gdg_batch_run(this.ctx, &arr[0], C.int(n), &o, (*unsafe.Pointer)(unsafe.Pointer(ptrs)))); e != nil {
		return nil, e
	}
	for i := range outs {
		outs[i] = goBytes(ptrs[i], each)
	}
	return outs, nil
}

// goBytes copies n bytes of C memory into a fresh slice.  C.GoBytes takes a C.int length and overflows from 2 GiB on
// (46 minutes of float64 at 96 kHz); this goes through a slice header over the C memory instead.
func goBytes(p unsafe.Pointer, n int) []byte {
	out := make([]byte, n)
	if n > 0 {
		copy(out, (*[1 << 40]byte)(p)[:n:n])
	}
	return out
}

func goFloats(p unsafe.Pointer, n int) []float64 {
	out := make([]float64, n)
	if n > 0 {
		copy(out, (*[1 << 37]float64)(p)[:n:n])
	}
	return out
}

// batchInputs builds the C array of gdg_batch_input; the file bytes live in C memory until release() is called.
func batchInputs(inputs []BatchInput) (arr *[1 << 20]C.gdg_batch_input, release func(), err error) {
	n := len(inputs)
	arr = (*[1 << 20]C.gdg_batch_input)(C.calloc(C.size_t(n), C.size_t(unsafe.Sizeof(C.gdg_batch_input{}))))
	if arr == nil {
		return nil, func() {}, fmt.Errorf("gdg: out of memory")
	}
	owned := []unsafe.Pointer{unsafe.Pointer(arr)}
	release = func() {
		for _, p := range owned {
			C.free(p)
		}
	}
	for i, in := range inputs {
		width := int(C.gdg_wave_bytes_per_sample(C.int(in.Format)))
		if len(in.Data) == 0 || width == 0 || in.Channels <= 0 {
			continue
		}
		p := C.CBytes(in.Data)
		if p == nil {
			release()
			return nil, func() {}, fmt.Errorf("gdg: out of memory")
		}
		owned = append(owned, p)
		arr[i].bytes = p
		arr[i].samples_per_channel = C.size_t(len(in.Data) / (width * in.Channels))
		arr[i].format = C.int(in.Format)
		arr[i].sample_rate = C.uint32_t(in.SampleRate)
		arr[i].channels = C.uint(in.Channels)
		arr[i].channel = C.uint(in.Channel)
	}
	return arr, release, nil
}

// BatchLength: samples of every output for these inputs (gdg_batch_length): the longest resampled input, rounded up to 8192.
func (this *Context) BatchLength(inputs []BatchInput, targetRate uint32) (int, error) {
	if len(inputs) == 0 {
		return 0, fmt.Errorf("gdg: no inputs")
	}
	arr, release, err := batchInputs(inputs)
	if err != nil {
		return 0, err
	}
	defer release()
	var samples C.size_t
	if e := this.err(C.gdg_batch_length(this.ctx, &arr[0], C.int(len(inputs)), C.uint32_t(targetRate), &samples)); e != nil {
		return 0, e
	}
	return int(samples), nil
}

// ShardResult: what one shard of a split job hands back (gdg_batch_run_shard).
type ShardResult struct {
	Outputs        [][]byte  // the shard's n encoded chain outputs
	Left, Right    []float64 // its PARTIAL master mix (no aux input, not clipped)
	MetronomeBytes []byte    // only on the shard that runs the metronome
	Metronome      []float64 // its float64 samples (the master's aux input when metrMasterOutput is set)
}

// BatchRunShard: the batch run of ONE shard of a job whose channels are split over several contexts / GPUs (contiguous channel
// blocks, gdg.Shards).  jobSamples = the longest BatchLength over all shards (every output of the job has that length,
// controller.go:3005-3045); exactly one shard passes runMetronome.  The master is finished once with FinishMaster.
func (this *Context) BatchRunShard(inputs []BatchInput, opt BatchOptions, jobSamples int, runMetronome bool) (*ShardResult, error) {
	n := len(inputs)
	if n == 0 {
		return nil, fmt.Errorf("gdg: no inputs")
	}
	arr, release, err := batchInputs(inputs)
	if err != nil {
		return nil, err
	}
	defer release()
	o := C.gdg_batch_options{target_rate: C.uint32_t(opt.TargetRate), out_format: C.int(opt.OutFormat),
		metronome_to_master: 0, run_meters: cbool(opt.RunMeters), tuner_enqueue: cbool(opt.TunerEnqueue)}
	width := int(C.gdg_wave_bytes_per_sample(o.out_format))
	res := &ShardResult{Outputs: make([][]byte, n)}
	if jobSamples == 0 || width == 0 {
		return res, nil
	}
	var owned []unsafe.Pointer
	defer func() {
		for _, p := range owned {
			C.free(p)
		}
	}()
	alloc := func(bytes int) unsafe.Pointer {
		p := C.malloc(C.size_t(bytes))
		if p != nil {
			owned = append(owned, p)
		}
		return p
	}
	ptrs := (*[1 << 20]unsafe.Pointer)(C.calloc(C.size_t(n), C.size_t(unsafe.Sizeof(uintptr(0)))))
	if ptrs == nil {
		return nil, fmt.Errorf("gdg: out of memory")
	}
	owned = append(owned, unsafe.Pointer(ptrs))
	for i := 0; i < n; i++ {
		if ptrs[i] = alloc(jobSamples * width); ptrs[i] == nil {
			return nil, fmt.Errorf("gdg: out of memory")
		}
	}
	var so C.gdg_batch_shard_out
	so.master_left = (*C.double)(alloc(jobSamples * 8))
	so.master_right = (*C.double)(alloc(jobSamples * 8))
	so.job_samples = C.size_t(jobSamples)
	if so.master_left == nil || so.master_right == nil {
		return nil, fmt.Errorf("gdg: out of memory")
	}
	if runMetronome {
		so.metronome_bytes = alloc(jobSamples * width)
		so.metronome = (*C.double)(alloc(jobSamples * 8))
		if so.metronome_bytes == nil || so.metronome == nil {
			return nil, fmt.Errorf("gdg: out of memory")
		}
	}
	if e := this.err(C.gdg_batch_run_shard(this.ctx, &arr[0], C.int(n), &o, (*unsafe.Pointer)(unsafe.Pointer(ptrs)), &so)); e != nil {
		return nil, e
	}
	for i := range res.Outputs {
		res.Outputs[i] = goBytes(ptrs[i], jobSamples*width)
	}
	res.Left = goFloats(unsafe.Pointer(so.master_left), jobSamples)
	res.Right = goFloats(unsafe.Pointer(so.master_right), jobSamples)
	if runMetronome {
		res.MetronomeBytes = goBytes(so.metronome_bytes, jobSamples*width)
		res.Metronome = goFloats(unsafe.Pointer(so.metronome), jobSamples)
	}
	return res, nil
}

// BatchRelease returns the device buffers of the last batch run to the context's arena (gdg_batch_release); the next run re-makes them.
func (this *Context) BatchRelease() error { return this.err(C.gdg_batch_release(this.ctx)) }

// Synchronize waits for everything the context has launched (gdg_ctx_synchronize); the host-buffer calls above already do.
func (this *Context) Synchronize() error { return this.err(C.gdg_ctx_synchronize(this.ctx)) }

// Version of the library behind the binding (gdg_version).
func Version() string { return C.GoString(C.gdg_version()) }

// FinishMaster: master = ((p_0 + p_1) + ... + p_{G-1}) + aux per side, summed and encoded on this context's device
// (gdg_batch_finish_master; spatializer/spatializer.go:300-310, controller/controller.go:3123-3219).  aux == nil: no aux input.
func (this *Context) FinishMaster(outFormat int, shards []*ShardResult, aux []float64, sampleRate uint32, runMeters bool) (left []byte, right []byte, err error) {
	g := len(shards)
	if g == 0 {
		return nil, nil, fmt.Errorf("gdg: no shards")
	}
	samples := len(shards[0].Left)
	width := int(C.gdg_wave_bytes_per_sample(C.int(outFormat)))
	if samples == 0 || width == 0 {
		return []byte{}, []byte{}, nil
	}
	var owned []unsafe.Pointer
	defer func() {
		for _, p := range owned {
			C.free(p)
		}
	}()
	cFloats := func(v []float64) *C.double {
		p := C.malloc(C.size_t(len(v) * 8))
		if p == nil {
			return nil
		}
		owned = append(owned, p)
		copy((*[1 << 37]float64)(p)[:len(v):len(v)], v)
		return (*C.double)(p)
	}
	lp := (*[1 << 20]*C.double)(C.calloc(C.size_t(2*g), C.size_t(unsafe.Sizeof(uintptr(0)))))
	if lp == nil {
		return nil, nil, fmt.Errorf("gdg: out of memory")
	}
	owned = append(owned, unsafe.Pointer(lp))
	for i, s := range shards {
		if len(s.Left) != samples || len(s.Right) != samples {
			return nil, nil, fmt.Errorf("gdg: shard %d has %d samples, shard 0 has %d", i, len(s.Left), samples)
		}
		lp[i], lp[g+i] = cFloats(s.Left), cFloats(s.Right)
		if lp[i] == nil || lp[g+i] == nil {
			return nil, nil, fmt.Errorf("gdg: out of memory")
		}
	}
	var cAux *C.double
	if aux != nil {
		if len(aux) != samples {
			return nil, nil, fmt.Errorf("gdg: the aux input has %d samples, the job %d", len(aux), samples)
		}
		if cAux = cFloats(aux); cAux == nil {
			return nil, nil, fmt.Errorf("gdg: out of memory")
		}
	}
	lb, rb := C.malloc(C.size_t(samples*width)), C.malloc(C.size_t(samples*width))
	if lb == nil || rb == nil {
		C.free(lb)
		C.free(rb)
		return nil, nil, fmt.Errorf("gdg: out of memory")
	}
	owned = append(owned, lb, rb)
	if e := this.err(C.gdg_batch_finish_master(this.ctx, C.int(outFormat), &lp[0], &lp[g], C.int(g), cAux, C.size_t(samples), C.uint32_t(sampleRate),
		cbool(runMeters), lb, rb)); e != nil {
		return nil, nil, e
	}
	return goBytes(lb, samples*width), goBytes(rb, samples*width), nil
}
