// Package spatializer is the drop-in replacement for the reference's spatializer/spatializer.go (overlay, like ../signal):
// same exported Spatializer interface (spatializer/spatializer.go:30-41), constants and Create (:436-469).
//
// Every shard (GPU) mixes ITS block of channels to a partial (left, right) pair -- gains level * min(1, 1 / ear distance), the
// inter-aural delay with the reference's 96000 Hz quirk, linear interpolation, the per-channel history -- and this file adds the
// partials in shard order and then the aux input (spatializer.go:300-310).  No collective, no peer access (SURVEY.md 8e).
//
// Inputs: Process copies inputBuffers into the pinned INPUT rows of the shards (the chains' rows are free again by then:
// controller.process() runs the spatializer after the N Chain.Process calls have returned, controller.go:2703-2761).  With
// GDG_SPATIALIZER_REUSE_OUTPUTS=1 the copy and the upload are skipped and every shard mixes the chain outputs of the block it
// has just computed, which are still on the device -- valid exactly when the caller passes the chain outputs, as
// controller.process() does (spatializerInputs := outputBuffers[0:nIn], controller.go:2751).
//
// NOT compiled in the authoring container (no Go toolchain); the C++ twin gdg::spatializer::Spatializer
// (../../host/gdg_host.cpp) has the same ten methods and the same partial-sum structure and is tested on the GPU against the
// oracle over two shards, both input paths (tests/test_host_mirror.py).
package spatializer

import (
	"fmt"
	"math"
	"os"
	"sync"

	"github.com/andrepxx/go-dsp-guitar/gdg" // the cgo binding: a directory added to the reference checkout (INTEGRATION.md section 3)
)

const (
	MATH_DEGREE_TO_RADIANS  = math.Pi / 180.0
	DEFAULT_SAMPLE_RATE     = 96000
	EFFECTIVE_DISTANCE      = 0.215
	HALF_EFFECTIVE_DISTANCE = 0.5 * EFFECTIVE_DISTANCE
	GROUP_DELAY             = 6.3e-4
	OUTPUT_COUNT            = 2
	blockSize               = 8192 // controller/controller.go:36 BLOCK_SIZE
)

// Spatializer: identical to the reference (spatializer/spatializer.go:30-41).
type Spatializer interface {
	GetAzimuth(inputChannel uint32) (float64, error)
	GetDistance(inputChannel uint32) (float64, error)
	GetLevel(inputChannel uint32) (float64, error)
	GetInputCount() uint32
	GetOutputCount() uint32
	Process(inputBuffers [][]float64, auxInputBuffer []float64, outputBuffers [][]float64)
	SetAzimuth(inputChannel uint32, azimuth float64) error
	SetDistance(inputChannel uint32, distance float64) error
	SetLevel(inputChannel uint32, level float64) error
	SetSampleRate(rate uint32)
}

type position struct {
	azimuth  float64
	distance float64
	level    float64
}

type spatializerStruct struct {
	inputCount uint32
	mutex      sync.RWMutex
	positions  []position
	sampleRate uint32 // rate of the history buffers, pushed to the shards when they exist
	pushed     bool   // positions and rate have reached the shards
	reuse      bool
}

// The reference checks `inputChannel > inputCount` and would index out of range for inputChannel == inputCount
// (spatializer.go:73, :93 ...); here that one value is an error as well (twin: SPAT_CHECK).
func (this *spatializerStruct) check(inputChannel uint32, what string) error {
	if inputChannel >= this.inputCount {
		return fmt.Errorf("Cannot %s for channel %d: Only %d channels exist.", what, inputChannel, this.inputCount)
	}
	return nil
}

func (this *spatializerStruct) GetAzimuth(inputChannel uint32) (float64, error) {
	if err := this.check(inputChannel, "get azimuth"); err != nil {
		return 0.0, err
	}
	this.mutex.RLock()
	defer this.mutex.RUnlock()
	return this.positions[inputChannel].azimuth, nil
}

func (this *spatializerStruct) GetDistance(inputChannel uint32) (float64, error) {
	if err := this.check(inputChannel, "get distance"); err != nil {
		return 0.0, err
	}
	this.mutex.RLock()
	defer this.mutex.RUnlock()
	return this.positions[inputChannel].distance, nil
}

func (this *spatializerStruct) GetLevel(inputChannel uint32) (float64, error) {
	if err := this.check(inputChannel, "get level"); err != nil {
		return 0.0, err
	}
	this.mutex.RLock()
	defer this.mutex.RUnlock()
	return this.positions[inputChannel].level, nil
}

func (this *spatializerStruct) GetInputCount() uint32  { return this.inputCount }
func (this *spatializerStruct) GetOutputCount() uint32 { return OUTPUT_COUNT }

// push one position to its shard (mutex held).  Before the shards exist nothing is sent: pushAll() does it at the first block.
func (this *spatializerStruct) push(channel uint32) {
	sh, local := gdg.ShardOf(int(channel))
	if sh == nil {
		this.pushed = false
		return
	}
	p := this.positions[channel]
	sh.Mutex.Lock()
	sh.Ctx.SpatializerSetPosition(local, p.azimuth, p.distance, p.level)
	sh.Mutex.Unlock()
}

func (this *spatializerStruct) SetAzimuth(inputChannel uint32, azimuth float64) error {
	if err := this.check(inputChannel, "set azimuth"); err != nil {
		return err
	}
	this.mutex.Lock()
	this.positions[inputChannel].azimuth = azimuth
	this.push(inputChannel)
	this.mutex.Unlock()
	return nil
}

func (this *spatializerStruct) SetDistance(inputChannel uint32, distance float64) error {
	if err := this.check(inputChannel, "set distance"); err != nil {
		return err
	}
	if distance < 0.0 || distance > 10.0 {
		return fmt.Errorf("%s", "Failed to set distance: Value must be within [0, 10].")
	}
	this.mutex.Lock()
	this.positions[inputChannel].distance = distance
	this.push(inputChannel)
	this.mutex.Unlock()
	return nil
}

func (this *spatializerStruct) SetLevel(inputChannel uint32, level float64) error {
	if err := this.check(inputChannel, "set distance"); err != nil { // the reference's message says "distance" here too (spatializer.go:395)
		return err
	}
	if level < 0.0 || level > 1.0 {
		return fmt.Errorf("%s", "Failed to set level: Value must be within [0, 1].")
	}
	this.mutex.Lock()
	this.positions[inputChannel].level = level
	this.push(inputChannel)
	this.mutex.Unlock()
	return nil
}

// SetSampleRate: spatializer.go:418-431 -- new (zeroed) history buffers of ceil(rate * 6.3e-4) samples on every shard.
func (this *spatializerStruct) SetSampleRate(rate uint32) {
	this.mutex.Lock()
	this.sampleRate = rate
	list, err := gdg.Shards(int(this.inputCount), blockSize)
	if err == nil {
		for _, sh := range list {
			sh.Mutex.Lock()
			sh.Ctx.SpatializerSetSampleRate(rate)
			sh.Mutex.Unlock()
		}
	}
	this.mutex.Unlock()
}

// everything that was set before the shards existed (mutex held)
func (this *spatializerStruct) pushAll(list []*gdg.Shard) {
	for _, sh := range list {
		sh.Mutex.Lock()
		if this.sampleRate != DEFAULT_SAMPLE_RATE {
			sh.Ctx.SpatializerSetSampleRate(this.sampleRate)
		}
		for local := 0; local < sh.Count; local++ {
			c := sh.First + local
			if c < len(this.positions) {
				p := this.positions[c]
				sh.Ctx.SpatializerSetPosition(local, p.azimuth, p.distance, p.level)
			}
		}
		sh.Mutex.Unlock()
	}
	this.pushed = true
}

// Process: spatializer/spatializer.go:140-335.  Twin: gdg::spatializer::Spatializer::Process.
func (this *spatializerStruct) Process(inputBuffers [][]float64, auxInputBuffer []float64, outputBuffers [][]float64) {
	if len(outputBuffers) < OUTPUT_COUNT || len(inputBuffers) < int(this.inputCount) {
		return
	}
	left, right := outputBuffers[0], outputBuffers[1]
	n := len(left)
	for i := 0; i < n; i++ {
		left[i], right[i] = 0.0, 0.0
	}
	list, err := gdg.Shards(int(this.inputCount), blockSize)
	if err != nil || n == 0 || n > blockSize || len(right) != n {
		return
	}
	this.mutex.Lock()
	if !this.pushed {
		this.pushAll(list)
	}
	this.mutex.Unlock()
	partL := make([][]float64, len(list))
	partR := make([][]float64, len(list))
	var wg sync.WaitGroup
	for g, sh := range list {
		partL[g], partR[g] = make([]float64, n), make([]float64, n)
		wg.Add(1)
		go func(g int, sh *gdg.Shard) {
			defer wg.Done()
			sh.Mutex.Lock()
			defer sh.Mutex.Unlock()
			if !this.reuse {
				for local := 0; local < sh.Count; local++ {
					row, _, err := sh.Ctx.Row(local, n)
					if err != nil || len(inputBuffers[sh.First+local]) != n {
						return
					}
					copy(row, inputBuffers[sh.First+local]) // Go memory -> pinned C slab
				}
			}
			if sh.Ctx.SpatializeStaged(this.reuse, partL[g], partR[g]) != nil {
				for i := 0; i < n; i++ { // a failed shard contributes silence
					partL[g][i], partR[g][i] = 0.0, 0.0
				}
			}
		}(g, sh)
	}
	wg.Wait()
	for g := range list { // host-side sum of the partials in shard order ...
		for i := 0; i < n; i++ {
			left[i] += partL[g][i]
			right[i] += partR[g][i]
		}
	}
	if auxInputBuffer != nil && len(auxInputBuffer) == n { // ... then the aux input (spatializer.go:300-310)
		for i := 0; i < n; i++ {
			left[i] += auxInputBuffer[i]
			right[i] += auxInputBuffer[i]
		}
	}
}

// Create: spatializer/spatializer.go:436-469 (levels are one by default).
func Create(inputChannels uint32) Spatializer {
	positions := make([]position, inputChannels)
	for i := range positions {
		positions[i].level = 1.0
	}
	s := spatializerStruct{
		inputCount: inputChannels,
		positions:  positions,
		sampleRate: DEFAULT_SAMPLE_RATE,
		reuse:      os.Getenv("GDG_SPATIALIZER_REUSE_OUTPUTS") == "1",
	}
	return &s
}
