// Package signal is the drop-in replacement for the reference's signal/signal.go: same package path,
// same exported Chain interface (signal/signal.go:21-36) and CreateChain (:419-431), applied with
//
//	go build -overlay overlay.json       (overlay.json maps <ref>/signal/signal.go to this file)
//
// so that controller.process() and everything above it stay byte-identical.  The reference's
// effects.Unit objects are kept as the PARAMETER STORE (tables, name lookup, range checks and error
// strings are theirs); their Process methods are never called: audio runs in libgdg.so on the GPU.
// NOT compiled in the authoring container (no Go toolchain); the C++ twin in ../../host/ is what the
// test-suite drives.  Structure and rendezvous are the same as gdg::Engine there.
package signal

import (
	"fmt"
	"math"
	"os"
	"strconv"
	"sync"
	"time"

	"github.com/andrepxx/go-dsp-guitar/effects"
	"github.com/andrepxx/go-dsp-guitar/filter"

	"gdg" // ../gdg, the cgo binding
)

type slotStruct struct {
	unit          effects.Unit
	bypass        bool
	handle        int      // device unit, -1 until materialised
	pushed        []int32  // resolved parameters last sent
	firSampleRate uint32   // power amp: rate the composite filter was compiled for
	firKey        string   // power amp: parameter signature of the compiled filter
}

// Chain: identical to the reference (signal/signal.go:21-36).
type Chain interface {
	AppendUnit(unitType int) (int, error)
	RemoveUnit(id int) error
	MoveUp(id int) error
	MoveDown(id int) error
	UnitType(id int) (int, error)
	SetBypass(id int, bypass bool) error
	GetBypass(id int) (bool, error)
	SetDiscreteValue(id int, name string, value string) error
	GetDiscreteValue(id int, name string) (string, error)
	SetNumericValue(id int, name string, value int32) error
	GetNumericValue(id int, name string) (int32, error)
	Parameters(id int) ([]effects.Parameter, error)
	Length() int
	Process(in []float64, out []float64, sampleRate uint32)
}

type chainStruct struct {
	channel   int
	responses filter.ImpulseResponses
	mutex     sync.RWMutex
	slots     []*slotStruct
	retired   []int
	dirty     bool
}

// ---- the shard: one context, N chains, N-way rendezvous (controller.go:2682-2705 keeps exactly N calls in flight) ----

type pendingStruct struct {
	chain      *chainStruct
	in, out    []float64
	sampleRate uint32
}

var (
	g_mutex      sync.Mutex
	g_cond       = sync.NewCond(&g_mutex)
	g_ctx        *gdg.Context
	g_chains     []*chainStruct
	g_pending    []pendingStruct
	g_generation uint64
	g_executing  bool
)

func context() *gdg.Context {
	if g_ctx == nil {
		n, _ := strconv.Atoi(os.Getenv("GDG_CHANNELS")) // = the -channels flag of main.go:14
		if n < len(g_chains) {
			n = len(g_chains)
		}
		dev, _ := strconv.Atoi(os.Getenv("GDG_DEVICE"))
		ctx, err := gdg.CreateContext(n, 8192, dev) // BLOCK_SIZE, controller.go:36
		if err != nil {
			panic(err) // no GPU: fail loudly, there is no CPU fallback
		}
		g_ctx = ctx
	}
	return g_ctx
}

// resolved parameters in table order: numeric value or discrete index
func resolve(unit effects.Unit) []int32 {
	params := unit.Parameters()
	res := make([]int32, len(params))
	for i, p := range params {
		if p.Type == effects.PARAMETER_TYPE_NUMERIC {
			res[i] = p.NumericValue
		} else {
			res[i] = int32(p.DiscreteValueIndex)
		}
	}
	return res
}

// compile mirrors effects/poweramp.go:25-127 through the reference's PUBLIC filter API.
func compile(unit effects.Unit, irs filter.ImpulseResponses, sampleRate uint32) ([]float64, string) {
	key := fmt.Sprint(sampleRate, resolve(unit))
	if irs == nil {
		return nil, key
	}
	orderString, _ := unit.GetDiscreteValue("filter_order")
	order64, _ := strconv.ParseUint(orderString, 10, 32)
	composite := filter.Empty(sampleRate)
	for i := 1; i <= effects.NUM_FILTERS; i++ {
		s := strconv.Itoa(i)
		name, _ := unit.GetDiscreteValue("filter_" + s)
		level, _ := unit.GetNumericValue("level_" + s)
		if name == effects.STRING_NONE {
			continue
		}
		flt := irs.CreateFilter(name, sampleRate)
		if flt == nil {
			return nil, key
		}
		if order64 > 0 {
			flt = flt.Reduce(uint32(order64))
		}
		fac := math.Pow(10.0, 0.05*float64(level))
		flt = flt.Normalize().Multiply(fac)
		composite, _ = composite.Add(flt)
	}
	return composite.Coefficients(), key
}

// bring the device side of one chain up to date (called by the batch leader only)
func (this *chainStruct) sync(ctx *gdg.Context, sampleRate uint32) {
	this.mutex.Lock()
	defer this.mutex.Unlock()
	for _, h := range this.retired {
		ctx.UnitDestroy(h)
	}
	this.retired = nil
	for _, slot := range this.slots {
		if slot.handle < 0 {
			slot.handle, _ = ctx.UnitCreate(this.channel, slot.unit.Type())
			slot.pushed = nil
			this.dirty = true
		}
		res := resolve(slot.unit)
		if slot.unit.Type() == effects.UNIT_POWERAMP {
			if !slot.bypass {
				taps, key := compile(slot.unit, this.responses, sampleRate)
				if key != slot.firKey { // any parameter set or a rate change => new filter => fresh state
					ctx.UnitSetFir(slot.handle, taps)
					slot.firKey = key
				}
			}
			continue
		}
		for i, v := range res {
			if i >= 8 {
				break
			}
			if slot.pushed == nil || slot.pushed[i] != v {
				ctx.UnitSetParam(slot.handle, i, v)
			}
		}
		slot.pushed = res
	}
	if this.dirty {
		handles := make([]int, len(this.slots))
		bypass := make([]bool, len(this.slots))
		for i, slot := range this.slots {
			handles[i], bypass[i] = slot.handle, slot.bypass
		}
		ctx.ChainSet(this.channel, handles, bypass)
		this.dirty = false
	}
}

func runBatch(batch []pendingStruct) {
	ctx := context()
	frames, sampleRate := len(batch[0].in), batch[0].sampleRate
	channels := make([]int, 0, len(batch))
	for _, p := range batch {
		p.chain.sync(ctx, sampleRate)
		row, _ := ctx.Row(p.chain.channel, frames)
		copy(row, p.in) // Go memory -> pinned C slab: no Go pointer crosses the boundary
		channels = append(channels, p.chain.channel)
	}
	err := ctx.ProcessStaged(channels, frames, sampleRate)
	for _, p := range batch {
		_, row := ctx.Row(p.chain.channel, frames)
		if err != nil {
			for i := range p.out { // the reference's failure mode inside Process: zeros (poweramp.go:210-214)
				p.out[i] = 0.0
			}
		} else {
			copy(p.out, row)
		}
	}
}

// Process: signal/signal.go:361-414.  Length mismatch is a silent no-op.
func (this *chainStruct) Process(in []float64, out []float64, sampleRate uint32) {
	if len(in) != len(out) {
		return
	}
	g_mutex.Lock()
	for g_executing {
		g_cond.Wait()
	}
	g_pending = append(g_pending, pendingStruct{this, in, out, sampleRate})
	gen := g_generation
	if len(g_pending) < len(g_chains) {
		// not the last arrival: wait for the leader (or become it after a grace period, e.g. when a
		// caller processes a single chain outside controller.process())
		timer := time.AfterFunc(50*time.Millisecond, func() { g_mutex.Lock(); g_cond.Broadcast(); g_mutex.Unlock() })
		deadline := time.Now().Add(50 * time.Millisecond)
		for g_generation == gen && (g_executing || time.Now().Before(deadline)) {
			g_cond.Wait()
		}
		timer.Stop()
		if g_generation != gen {
			g_mutex.Unlock()
			return
		}
	}
	batch := g_pending
	g_pending = nil
	g_executing = true
	g_mutex.Unlock()
	runBatch(batch) // all frames have the same length and rate in controller.process()
	g_mutex.Lock()
	g_executing = false
	g_generation++
	g_cond.Broadcast()
	g_mutex.Unlock()
}

// ---- slot bookkeeping: the reference's code with the device hand-off flags added ----

func (this *chainStruct) AppendUnit(unitType int) (int, error) {
	unit := effects.CreateUnit(unitType)
	if unit == nil {
		return -1, fmt.Errorf("%s", "Failed to create effects unit.")
	}
	if unitType == effects.UNIT_POWERAMP {
		effects.PreparePowerAmp(unit, this.responses)
	}
	this.mutex.Lock()
	this.slots = append(this.slots, &slotStruct{unit: unit, bypass: true, handle: -1})
	this.dirty = true
	n := len(this.slots) - 1
	this.mutex.Unlock()
	return n, nil
}

func (this *chainStruct) RemoveUnit(id int) error {
	this.mutex.Lock()
	defer this.mutex.Unlock()
	if id < 0 || id >= len(this.slots) {
		return fmt.Errorf("Cannot remove unit %d.", id)
	}
	if this.slots[id].handle >= 0 {
		this.retired = append(this.retired, this.slots[id].handle)
	}
	this.slots = append(this.slots[:id], this.slots[id+1:]...)
	this.dirty = true
	return nil
}

func (this *chainStruct) MoveUp(id int) error {
	this.mutex.Lock()
	defer this.mutex.Unlock()
	if id <= 0 || id >= len(this.slots) {
		return fmt.Errorf("Cannot move unit %d up.", id)
	}
	this.slots[id], this.slots[id-1] = this.slots[id-1], this.slots[id]
	this.dirty = true
	return nil
}

func (this *chainStruct) MoveDown(id int) error {
	this.mutex.Lock()
	defer this.mutex.Unlock()
	if id < 0 || id >= len(this.slots)-1 {
		return fmt.Errorf("Cannot move unit %d down.", id)
	}
	this.slots[id], this.slots[id+1] = this.slots[id+1], this.slots[id]
	this.dirty = true
	return nil
}

func (this *chainStruct) slot(id int, what string) (*slotStruct, error) {
	this.mutex.RLock()
	defer this.mutex.RUnlock()
	if id < 0 || id >= len(this.slots) {
		return nil, fmt.Errorf("Cannot %s: No unit %d.", what, id)
	}
	return this.slots[id], nil
}

func (this *chainStruct) UnitType(id int) (int, error) {
	s, err := this.slot(id, "get unit type")
	if err != nil {
		return -1, err
	}
	return s.unit.Type(), nil
}

func (this *chainStruct) SetBypass(id int, bypass bool) error {
	action := "disable"
	if bypass {
		action = "enable"
	}
	s, err := this.slot(id, action+" bypass")
	if err != nil {
		return err
	}
	this.mutex.Lock()
	s.bypass = bypass
	this.dirty = true
	this.mutex.Unlock()
	return nil
}

func (this *chainStruct) GetBypass(id int) (bool, error) {
	s, err := this.slot(id, "get bypass value")
	if err != nil {
		return false, err
	}
	return s.bypass, nil
}

func (this *chainStruct) SetDiscreteValue(id int, name string, value string) error {
	s, err := this.slot(id, "set discrete value")
	if err != nil {
		return err
	}
	return s.unit.SetDiscreteValue(name, value)
}

func (this *chainStruct) GetDiscreteValue(id int, name string) (string, error) {
	s, err := this.slot(id, "get discrete value")
	if err != nil {
		return "", err
	}
	return s.unit.GetDiscreteValue(name)
}

func (this *chainStruct) SetNumericValue(id int, name string, value int32) error {
	s, err := this.slot(id, "set numeric value")
	if err != nil {
		return err
	}
	return s.unit.SetNumericValue(name, value)
}

func (this *chainStruct) GetNumericValue(id int, name string) (int32, error) {
	s, err := this.slot(id, "get numeric value")
	if err != nil {
		return 0, err
	}
	return s.unit.GetNumericValue(name)
}

func (this *chainStruct) Parameters(id int) ([]effects.Parameter, error) {
	s, err := this.slot(id, "get parameters")
	if err != nil {
		return nil, err
	}
	return s.unit.Parameters(), nil
}

func (this *chainStruct) Length() int {
	this.mutex.RLock()
	defer this.mutex.RUnlock()
	return len(this.slots)
}

// CreateChain: signal/signal.go:419-431; called once per input channel by controller.initialize (:3267-3269).
func CreateChain(responses filter.ImpulseResponses) Chain {
	g_mutex.Lock()
	defer g_mutex.Unlock()
	chain := &chainStruct{channel: len(g_chains), responses: responses}
	g_chains = append(g_chains, chain)
	return chain
}
