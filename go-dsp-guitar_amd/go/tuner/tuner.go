// Package tuner is the drop-in replacement for the reference's tuner/tuner.go (overlay, like ../signal): same exported
// Tuner / Result interfaces (tuner/tuner.go:27-65) and Create (:592-610).  Process copies the samples into the pinned row of a
// private one-channel context and enqueues them into the 96000-sample ring on the device; Analyze runs the 262144-point
// autocorrelation, the arg-max over the note range, the parabolic refinement and the note search on the device
// (tuner.go:379-577) and returns frequency, note name and truncated cents.
//
// NOT compiled in the authoring container (no Go toolchain); the C++ twin gdg::tuner::Tuner (../../host/gdg_host.cpp) has the
// same two methods over the same two C calls and is tested on the GPU against the oracle (tests/test_host_mirror.py).
package tuner

import (
	"fmt"
	"os"
	"strconv"
	"sync"

	"github.com/andrepxx/go-dsp-guitar/gdg" // the cgo binding, added to the reference module by the overlay (go/overlay.json)
)

const (
	NUM_SAMPLES = 96000 // tuner/tuner.go:16
	blockSize   = 8192  // controller/controller.go:36 BLOCK_SIZE: the most the controller enqueues per call
)

type resultStruct struct {
	cents     int8
	frequency float64
	note      string
}

// Result: identical to the reference (tuner/tuner.go:36-46).
type Result interface {
	Cents() int8
	Frequency() float64
	Note() string
}

func (this *resultStruct) Cents() int8        { return this.cents }
func (this *resultStruct) Frequency() float64 { return this.frequency }
func (this *resultStruct) Note() string       { return this.note }

// Tuner: identical to the reference (tuner/tuner.go:62-65).
type Tuner interface {
	Analyze() (Result, error)
	Process(samples []float64, sampleRate uint32)
}

type tunerStruct struct {
	mutex sync.Mutex
	ctx   *gdg.Context
}

func (this *tunerStruct) context() (*gdg.Context, error) {
	if this.ctx == nil {
		dev, _ := strconv.Atoi(os.Getenv("GDG_TUNER_DEVICE")) // default: device 0
		ctx, err := gdg.CreateContext(1, blockSize, dev)
		if err != nil {
			return nil, err // no GPU: fail loudly, there is no CPU fallback
		}
		this.ctx = ctx
	}
	return this.ctx, nil
}

// Process: tuner/tuner.go:582-587 (circular.Enqueue of any number of samples; the rate is remembered for Analyze).
func (this *tunerStruct) Process(samples []float64, sampleRate uint32) {
	this.mutex.Lock()
	defer this.mutex.Unlock()
	ctx, err := this.context()
	if err != nil {
		return
	}
	for at := 0; at < len(samples); at += blockSize {
		n := len(samples) - at
		if n > blockSize {
			n = blockSize
		}
		row, _, err := ctx.Row(0, n)
		if err != nil {
			return
		}
		copy(row, samples[at:at+n]) // Go memory -> pinned C slab
		if ctx.TunerEnqueueStaged(n, sampleRate) != nil {
			return
		}
	}
}

// Analyze: tuner/tuner.go:379-577.
func (this *tunerStruct) Analyze() (Result, error) {
	this.mutex.Lock()
	defer this.mutex.Unlock()
	ctx, err := this.context()
	if err != nil {
		return nil, err
	}
	res, err := ctx.TunerAnalyze()
	if err != nil {
		return nil, fmt.Errorf("Failed to analyze: %s", err.Error())
	}
	r := resultStruct{cents: res[0].Cents, frequency: res[0].Frequency, note: gdg.TunerNoteName(res[0].NoteIndex)}
	return &r, nil
}

// Create: tuner/tuner.go:592-610.
func Create() Tuner {
	return &tunerStruct{}
}
