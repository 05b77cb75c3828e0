// Package tuner is the drop-in replacement for the reference's tuner/tuner.go (overlay, like ../signal): same exported
// Tuner / Result interfaces (tuner/tuner.go:27-65) and Create (:592-610), and the reference's TWO locks (tuner.go:48-57):
//
//	mutexBuffer   guards the ring of the last NUM_SAMPLES samples and the rate.  Process -- the audio path, called from
//	              controller.process() for every block (controller.go:2668-2672) -- takes it for one Enqueue and nothing else:
//	              it never waits for the GPU.
//	mutexAnalyze  serialises Analyze (the HTTP poll, tuner.go:380) and owns the device context.  Analyze holds mutexBuffer
//	              (shared) only while it copies the ring out (tuner.go:408-412), then uploads the copy into the device ring and
//	              runs the autocorrelation, the arg-max over the note range, the parabolic refinement and the note search there
//	              (tuner.go:414-577) with the audio path free to enqueue.
//
// The ring itself is the reference's own circular.Buffer: its contents and order are what the device analyses.
//
// NOT compiled in the authoring container (no Go toolchain); the C++ twin gdg::tuner::Tuner (../../host/gdg_host.cpp) has the
// same two methods, the same two locks and the same C calls and is tested on the GPU against the oracle (tests/test_host_mirror.py).
package tuner

import (
	"fmt"
	"os"
	"strconv"
	"sync"

	"github.com/andrepxx/go-dsp-guitar/circular"
	"github.com/andrepxx/go-dsp-guitar/gdg" // the cgo binding: a directory added to the reference checkout (INTEGRATION.md section 3)
)

const (
	NUM_SAMPLES = 96000 // tuner/tuner.go:16
	blockSize   = 8192  // controller/controller.go:36 BLOCK_SIZE: the most a context takes per call
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
	mutexBuffer  sync.RWMutex
	buffer       circular.Buffer
	sampleRate   uint32
	mutexAnalyze sync.Mutex
	ctx          *gdg.Context // owned by Analyze (mutexAnalyze held)
	snapshot     []float64    // the reference's bufCorrelation[0:n]: the ring, oldest sample first
}

// context (mutexAnalyze held): the private one-channel context, made at the first analysis.
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

// Process: tuner/tuner.go:582-587, statement for statement.
func (this *tunerStruct) Process(samples []float64, sampleRate uint32) {
	this.mutexBuffer.Lock()
	this.buffer.Enqueue(samples...)
	this.sampleRate = sampleRate
	this.mutexBuffer.Unlock()
}

// Analyze: tuner/tuner.go:379-577.
func (this *tunerStruct) Analyze() (Result, error) {
	this.mutexAnalyze.Lock()
	defer this.mutexAnalyze.Unlock()
	ctx, err := this.context()
	if err != nil {
		return nil, err
	}
	n := this.buffer.Length()
	if len(this.snapshot) != n {
		this.snapshot = make([]float64, n)
	}
	this.mutexBuffer.RLock()
	sampleRate := this.sampleRate
	err = this.buffer.Retrieve(this.snapshot)
	this.mutexBuffer.RUnlock()
	if err != nil {
		return nil, fmt.Errorf("Failed to retrieve contents of circular buffer: %s", err.Error())
	}
	// the whole ring, oldest first, replaces the device ring in ONE upload (the library refuses any length but NUM_SAMPLES)
	if err := ctx.TunerReplace(0, this.snapshot, sampleRate); err != nil {
		return nil, fmt.Errorf("Failed to analyze: %s", err.Error())
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
	t := tunerStruct{buffer: circular.CreateBuffer(NUM_SAMPLES)}
	return &t
}
