// Package signal is the drop-in replacement for the reference's signal/signal.go: same package path,
// same exported Chain interface (signal/signal.go:21-36) and CreateChain (:419-431), applied with
//
//	go build -overlay overlay.json       (overlay.json maps <ref>/signal/signal.go to this file)
//
// so that controller.process() and everything above it stay byte-identical.  The reference's
// effects.Unit objects are kept as the PARAMETER STORE (tables, name lookup, range checks and error
// strings are theirs); their Process methods are never called: audio runs in libgdg.so on the GPUs.
//
// NOT compiled in the authoring container (no Go toolchain).  The algorithm below is the one of the
// C++ twin (../../host/gdg_host.cpp), which the test-suite drives on the GPU -- INTEGRATION.md
// section 3 lists it once, and the two files follow it statement for statement:
//
//	gdg::Engine::process   <->  chainStruct.Process     rendezvous: deposit, last arrival (or grace period) runs the batch
//	gdg::Engine::runBatch  <->  runBatch                per shard (GPU): sync, stage, ONE batched call, copy back; shards in parallel
//	gdg::Engine::sync      <->  chainStruct.sync        retire, create, rate change -> recompile, push from ONE snapshot, slot list
//	effects::Unit setters  <->  chainStruct.Set*Value   count every successful set; a power amp recompiles AT the set (poweramp.go:131-181)
package signal

import (
	"fmt"
	"log"
	"math"
	"os"
	"strconv"
	"sync"
	"time"

	"github.com/andrepxx/go-dsp-guitar/effects"
	"github.com/andrepxx/go-dsp-guitar/filter"

	"github.com/andrepxx/go-dsp-guitar/gdg" // the cgo binding: a directory added to the reference checkout (INTEGRATION.md section 3)
)

const blockSize = 8192 // controller/controller.go:36 BLOCK_SIZE

// slotStruct: one unit of a chain.  Everything below `unit` and `bypass` is guarded by the chain's mutex.
type slotStruct struct {
	unit   effects.Unit
	bypass bool
	handle int // device unit inside the chain's shard, -1 until materialised

	paramVersion, pushedParamVersion uint64 // bumped by every successful Set*Value (twin: Unit::paramVersion)
	firVersion, pushedFirVersion     uint64 // power amp: bumped by every successful compile (twin: Unit::firVersion)
	firTaps                          []float64
	sampleRate                       uint32 // power amp: poweramp.sampleRate, changes only when the unit is PROCESSED
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
	channel   int // GLOBAL channel number = index of the CreateChain call (controller.go:3267-3269)
	responses filter.ImpulseResponses
	mutex     sync.RWMutex
	slots     []*slotStruct
	retired   []int // device handles of removed units

	layoutVersion, pushedLayoutVersion uint64
}

// ---- the job: N chains, G shards, ONE N-way rendezvous (controller.go:2682-2705 keeps exactly N calls in flight) ----

type pendingStruct struct {
	chain      *chainStruct
	in, out    []float64
	sampleRate uint32
}

var (
	g_mutex      sync.Mutex
	g_cond       = sync.NewCond(&g_mutex)
	g_chains     []*chainStruct
	g_pending    []pendingStruct
	g_generation uint64
	g_executing  bool
	g_logOnce    sync.Once
)

func logOnce(err error) {
	g_logOnce.Do(func() { log.Printf("gdg: %v (outputs are zeros, like the reference's failure mode inside Process)", err) })
}

// shards: the contexts of the job, created at the first block (GDG_CHANNELS = the -channels flag of main.go:14, if set).
func shards() ([]*gdg.Shard, error) {
	n, _ := strconv.Atoi(os.Getenv("GDG_CHANNELS"))
	g_mutex.Lock()
	if n < len(g_chains) {
		n = len(g_chains)
	}
	g_mutex.Unlock()
	return gdg.Shards(n, blockSize)
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

// compile mirrors effects/poweramp.go:25-127 through the reference's PUBLIC filter API (twin: effects::Unit::compile).
func compile(unit effects.Unit, irs filter.ImpulseResponses, sampleRate uint32) ([]float64, error) {
	if irs == nil {
		return nil, fmt.Errorf("%s", "Could not compile filter: No impulse responses were loaded.")
	}
	orderString, err := unit.GetDiscreteValue("filter_order")
	if err != nil {
		return nil, err
	}
	order64, err := strconv.ParseUint(orderString, 10, 32)
	if err != nil {
		return nil, fmt.Errorf("Could not parse filter target order: '%s'", orderString)
	}
	composite := filter.Empty(sampleRate)
	for i := 1; i <= effects.NUM_FILTERS; i++ {
		s := strconv.Itoa(i)
		name, errName := unit.GetDiscreteValue("filter_" + s)
		level, errLevel := unit.GetNumericValue("level_" + s)
		if errName != nil || errLevel != nil {
			return nil, fmt.Errorf("Error parsing values for filter %d.", i-1)
		}
		if name == effects.STRING_NONE {
			continue
		}
		flt := irs.CreateFilter(name, sampleRate)
		if flt == nil {
			return nil, fmt.Errorf("Failed to load filter '%s' for sample rate '%d'.", name, sampleRate)
		}
		if order64 > 0 {
			flt = flt.Reduce(uint32(order64))
		}
		fac := math.Pow(10.0, 0.05*float64(level))
		flt = flt.Normalize().Multiply(fac)
		sum, errAdd := composite.Add(flt)
		if errAdd != nil {
			return nil, fmt.Errorf("Failed to add filter: %s", errAdd.Error())
		}
		composite = sum
	}
	return composite.Coefficients(), nil
}

// recompile (chain mutex held): on success the power amp gets a NEW filter -- fresh convolution state on the device --
// on failure the previous one stays (poweramp.go:147-151).  Twin: effects::Unit::recompile.
func (this *chainStruct) recompile(slot *slotStruct) {
	taps, err := compile(slot.unit, this.responses, slot.sampleRate)
	if err == nil {
		slot.firTaps = taps
		slot.firVersion++
	}
}

// bring the device side of this chain up to date (batch leader of the chain's shard only; shard mutex held).
// Twin: gdg::Engine::sync.
func (this *chainStruct) sync(ctx *gdg.Context, local int, sampleRate uint32) error {
	this.mutex.Lock()
	defer this.mutex.Unlock()
	// 1. removed units
	for _, h := range this.retired {
		ctx.UnitDestroy(h)
	}
	this.retired = nil
	for _, slot := range this.slots {
		// 2a. materialise
		if slot.handle < 0 {
			h, err := ctx.UnitCreate(local, slot.unit.Type())
			if err != nil {
				return err
			}
			slot.handle = h
			slot.pushedParamVersion, slot.pushedFirVersion = 0, 0
			this.pushedLayoutVersion = 0
		}
		isAmp := slot.unit.Type() == effects.UNIT_POWERAMP
		// 2b. only a PROCESSED power amp notices a new sample rate (poweramp.go:191-203)
		if isAmp && !slot.bypass && sampleRate != slot.sampleRate {
			slot.sampleRate = sampleRate
			this.recompile(slot)
		}
		// 2c. push from ONE consistent snapshot: the chain mutex is held, setters bump the versions under the same mutex
		if slot.pushedParamVersion != slot.paramVersion {
			res := resolve(slot.unit)
			n := len(res)
			if n > 8 {
				n = 8
			}
			if isAmp {
				n = 1 // filter_order; the filter_N / level_N values only matter through the compiled taps
			}
			for i := 0; i < n; i++ {
				if err := ctx.UnitSetParam(slot.handle, i, res[i]); err != nil {
					return err
				}
			}
			slot.pushedParamVersion = slot.paramVersion
		}
		if isAmp && slot.pushedFirVersion != slot.firVersion {
			// every successful Set -- the same value or not -- replaced currentFilter in the reference: fresh tail
			if err := ctx.UnitSetFir(slot.handle, slot.firTaps); err != nil {
				return err
			}
			slot.pushedFirVersion = slot.firVersion
		}
	}
	// 3. slot list
	if this.pushedLayoutVersion != this.layoutVersion {
		handles := make([]int, len(this.slots))
		bypass := make([]bool, len(this.slots))
		for i, slot := range this.slots {
			handles[i], bypass[i] = slot.handle, slot.bypass
		}
		if err := ctx.ChainSet(local, handles, bypass); err != nil {
			return err
		}
		this.pushedLayoutVersion = this.layoutVersion
	}
	return nil
}

func zero(buf []float64) {
	for i := range buf {
		buf[i] = 0.0
	}
}

// one shard's part of a batch: sync, stage, ONE gdg_process_staged over its channels, copy back.  Twin: gdg::Engine::runShard.
func runShard(sh *gdg.Shard, group []pendingStruct, frames int, sampleRate uint32) {
	sh.Mutex.Lock()
	defer sh.Mutex.Unlock()
	fail := func(err error) {
		logOnce(err)
		for _, p := range group { // the reference's failure mode inside Process: zeros (poweramp.go:210-214)
			zero(p.out)
		}
	}
	channels := make([]int, 0, len(group))
	for _, p := range group {
		local := p.chain.channel - sh.First
		if err := p.chain.sync(sh.Ctx, local, sampleRate); err != nil {
			fail(err)
			return
		}
		row, _, err := sh.Ctx.Row(local, frames) // checked: frames <= row stride, channel inside the shard
		if err != nil {
			fail(err)
			return
		}
		copy(row, p.in) // Go memory -> pinned C slab: no Go pointer crosses the boundary
		channels = append(channels, local)
	}
	if err := sh.Ctx.ProcessStaged(channels, frames, sampleRate); err != nil {
		fail(err)
		return
	}
	for _, p := range group {
		_, row, _ := sh.Ctx.Row(p.chain.channel-sh.First, frames)
		copy(p.out, row)
	}
}

// Twin: gdg::Engine::runBatch.  One group per distinct (frames, sample rate) -- in controller.process() exactly one --
// and inside it one goroutine per shard: channels are independent, the GPUs have nothing to exchange.
func runBatch(batch []pendingStruct) {
	list, err := shards()
	if err != nil {
		logOnce(err) // no GPU: fail loudly, there is no CPU fallback
		for _, p := range batch {
			zero(p.out)
		}
		return
	}
	for len(batch) > 0 {
		frames, sampleRate := len(batch[0].in), batch[0].sampleRate
		var rest []pendingStruct
		perShard := make([][]pendingStruct, len(list))
		for _, p := range batch {
			if len(p.in) != frames || p.sampleRate != sampleRate {
				rest = append(rest, p)
				continue
			}
			placed := false
			for g, sh := range list {
				if p.chain.channel >= sh.First && p.chain.channel < sh.First+sh.Count {
					perShard[g] = append(perShard[g], p)
					placed = true
					break
				}
			}
			if !placed {
				zero(p.out)
			}
		}
		var wg sync.WaitGroup
		for g, group := range perShard {
			if len(group) == 0 {
				continue
			}
			wg.Add(1)
			go func(sh *gdg.Shard, group []pendingStruct) {
				defer wg.Done()
				runShard(sh, group, frames, sampleRate)
			}(list[g], group)
		}
		wg.Wait()
		batch = rest
	}
}

// Process: signal/signal.go:361-414.  Length mismatch is a silent no-op.  Twin: gdg::Engine::process.
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
	runBatch(batch)
	g_mutex.Lock()
	g_executing = false
	g_generation++
	g_cond.Broadcast()
	g_mutex.Unlock()
}

// ---- slot bookkeeping: the reference's code with the device hand-off versions added ----

func (this *chainStruct) AppendUnit(unitType int) (int, error) {
	unit := effects.CreateUnit(unitType)
	if unit == nil {
		return -1, fmt.Errorf("%s", "Failed to create effects unit.")
	}
	if unitType == effects.UNIT_POWERAMP {
		effects.PreparePowerAmp(unit, this.responses)
	}
	this.mutex.Lock()
	this.slots = append(this.slots, &slotStruct{unit: unit, bypass: true, handle: -1, paramVersion: 1}) // new units start bypassed (signal.go:74)
	this.layoutVersion++
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
	this.layoutVersion++
	return nil
}

func (this *chainStruct) MoveUp(id int) error {
	this.mutex.Lock()
	defer this.mutex.Unlock()
	if id <= 0 || id >= len(this.slots) {
		return fmt.Errorf("Cannot move unit %d up.", id)
	}
	this.slots[id], this.slots[id-1] = this.slots[id-1], this.slots[id]
	this.layoutVersion++
	return nil
}

func (this *chainStruct) MoveDown(id int) error {
	this.mutex.Lock()
	defer this.mutex.Unlock()
	if id < 0 || id >= len(this.slots)-1 {
		return fmt.Errorf("Cannot move unit %d down.", id)
	}
	this.slots[id], this.slots[id+1] = this.slots[id+1], this.slots[id]
	this.layoutVersion++
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
	s.bypass = bypass // a bypass toggle does not touch the filter: state stays with the unit (signal.go:390-401)
	this.layoutVersion++
	this.mutex.Unlock()
	return nil
}

func (this *chainStruct) GetBypass(id int) (bool, error) {
	s, err := this.slot(id, "get bypass value")
	if err != nil {
		return false, err
	}
	this.mutex.RLock()
	defer this.mutex.RUnlock()
	return s.bypass, nil
}

// after a SUCCESSFUL set: count it, and a power amp compiles a new filter right here, at the rate it was last processed at
// (poweramp.go:131-181) -- the same value or not.  Twin: effects::Unit::SetDiscreteValue / SetNumericValue.
func (this *chainStruct) afterSet(s *slotStruct) {
	this.mutex.Lock()
	s.paramVersion++
	if s.unit.Type() == effects.UNIT_POWERAMP {
		this.recompile(s)
	}
	this.mutex.Unlock()
}

func (this *chainStruct) SetDiscreteValue(id int, name string, value string) error {
	s, err := this.slot(id, "set discrete value")
	if err != nil {
		return err
	}
	err = s.unit.SetDiscreteValue(name, value)
	if err == nil {
		this.afterSet(s)
	}
	return err
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
	err = s.unit.SetNumericValue(name, value)
	if err == nil {
		this.afterSet(s)
	}
	return err
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
	chain := &chainStruct{channel: len(g_chains), responses: responses, layoutVersion: 1}
	g_chains = append(g_chains, chain)
	return chain
}
