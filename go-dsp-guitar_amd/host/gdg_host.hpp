/*
 * gdg_host.hpp -- host-side mirror of the reference's Go interfaces for the hot path, written in
 * C++ because the build image has no Go toolchain (the Go shim a maintainer would compile is in
 * ../go/ and INTEGRATION.md; it has the same structure as this file).
 *
 *   effects::Unit       <->  effects.Unit       (effects/effects.go:83-91)     7 methods
 *   effects::Parameter  <->  effects.Parameter  (effects/effects.go:69-78)
 *   signal::Chain       <->  signal.Chain       (signal/signal.go:21-36)       14 methods
 *   filter::Filter / ImpulseResponses  <->  filter.Filter / filter.ImpulseResponses (filter/filter.go:64-95)
 *
 * Same names, argument meaning and error strings.  Go's `error` is a std::string here (empty =
 * nil).  Parameter tables, name lookup, range checks and the power-amp filter compile
 * (Reduce / Normalize / Multiply / Add, effects/poweramp.go:25-127) live here; only resolved
 * integers, taps and sample buffers cross the C-ABI of include/gdg.h.  All audio is computed by
 * libgdg.so on the GPU: nothing in this file processes samples.
 */
#ifndef GDG_HOST_HPP
#define GDG_HOST_HPP

#include <condition_variable>
#include <cstdint>
#include <memory>
#include <mutex>
#include <shared_mutex>
#include <string>
#include <utility>
#include <vector>

#include "../../include/gdg.h"      /* the C-ABI the twin is written against (gdg_ctx, gdg_batch_input, gdg_batch_options ...) */

#define GDG_HOST_MAX_FRAMES 8192      /* controller/controller.go:36 BLOCK_SIZE */

namespace gdg {

using Error = std::string;          /* "" == nil */

namespace filter {

class Filter {
public:
    Filter(std::vector<double> coeffs, uint32_t sampleRate, double gainCompensation, std::string name);
    std::pair<std::shared_ptr<Filter>, Error> Add(const std::shared_ptr<Filter> &other) const;   /* filter.go:167-253 */
    std::vector<double> Coefficients() const;                                                     /* :258-265 */
    std::shared_ptr<Filter> Multiply(double scalar) const;                                        /* :270-323 */
    std::shared_ptr<Filter> Normalize() const;                                                    /* :328-336 */
    std::shared_ptr<Filter> Reduce(uint32_t order) const;                                         /* :520-604 */
    uint32_t SampleRate() const { return sampleRate_; }                                           /* :609-613 */
    const std::string &Name() const { return name_; }
private:
    std::vector<double> data_;
    uint32_t sampleRate_;
    double gainCompensation_;
    std::string name_;
};

std::shared_ptr<Filter> Empty(uint32_t sampleRate);                                               /* :807-845 */
std::shared_ptr<Filter> FromCoefficients(const std::vector<double> &coeffs, uint32_t sampleRate, const std::string &name);  /* :850-890 */
std::vector<uint32_t> SampleRates();                                                              /* :895-900 */

/* filter.ImpulseResponses (filter.go:64-67).  The reference fills it from ir/index.json + WAV files
 * (filter.Import, host I/O, out of scope); here entries are added from memory. */
class ImpulseResponses {
public:
    void Add(const std::string &name, uint32_t sampleRate, int32_t compensationDecibels, const std::vector<double> &taps);
    std::shared_ptr<Filter> CreateFilter(const std::string &name, uint32_t sampleRate) const;    /* :619-658 */
    std::vector<std::string> Names() const;                                                       /* :663-699 */
private:
    struct Entry { std::string name; uint32_t sampleRate; double gainCompensation; std::vector<double> data; };
    std::vector<Entry> responses_;
};

}  // namespace filter

namespace effects {

enum { PARAMETER_TYPE_INVALID = 0, PARAMETER_TYPE_DISCRETE, PARAMETER_TYPE_NUMERIC };   /* effects.go:11-15 */
enum {                                                                                     /* effects.go:21-43 */
    UNIT_SIGNALGENERATOR = 0, UNIT_NOISEGATE, UNIT_BANDPASS, UNIT_AUTOWAH, UNIT_AUTOYOY, UNIT_COMPRESSOR, UNIT_OCTAVER,
    UNIT_EXCESS, UNIT_FUZZ, UNIT_OVERDRIVE, UNIT_DISTORTION, UNIT_TONESTACK, UNIT_CHORUS, UNIT_FLANGER, UNIT_PHASER,
    UNIT_TREMOLO, UNIT_RINGMODULATOR, UNIT_DELAY, UNIT_REVERB, UNIT_POWERAMP, UNIT_CABINET, UNIT_COUNT
};
constexpr int NUM_FILTERS = 8;                 /* effects.go:62 */
extern const char *const STRING_NONE;          /* "- NONE -" */

struct Parameter {                             /* effects.go:69-78 */
    std::string Name;
    int32_t Type = PARAMETER_TYPE_INVALID;
    std::string PhysicalUnit;
    int32_t Minimum = -1, Maximum = -1, NumericValue = -1;
    int DiscreteValueIndex = -1;
    std::vector<std::string> DiscreteValues;
};

class Unit {                                   /* effects.go:83-91 + unitStruct :96-100 */
public:
    explicit Unit(int unitType);
    ~Unit();
    std::vector<Parameter> Parameters() const;
    /* Stand-alone Process of a unit that is not in a chain: runs it as a one-slot chain on a private
     * one-channel context (device 0).  Units owned by a Chain are processed through Chain::Process. */
    void Process(const double *in, double *out, size_t n, uint32_t sampleRate);
    int Type() const { return unitType_; }
    Error SetDiscreteValue(const std::string &name, const std::string &value);
    std::pair<std::string, Error> GetDiscreteValue(const std::string &name) const;
    Error SetNumericValue(const std::string &name, int32_t value);
    std::pair<int32_t, Error> GetNumericValue(const std::string &name) const;

    /* ---- plumbing used by signal::Chain / Engine (not part of the mirrored interface) ---- */
    struct Backing { gdg_ctx *ctx = nullptr; int handle = -1; bool owns_ctx = false; };
    Backing backing;
    uint64_t paramVersion = 1, pushedParamVersion = 0;     /* resolved-integer parameters */
    uint64_t firVersion = 0, pushedFirVersion = 0;         /* power amp: current composite filter */
    std::vector<double> firTaps;                           /* taps of currentFilter (empty: zeros out) */
    uint32_t sampleRate = 0;                               /* power amp: poweramp.sampleRate */
    const filter::ImpulseResponses *impulseResponses = nullptr;
    void resolved(int32_t out[8]) const;                   /* numeric values / discrete indices of the first <= 8 parameters */
    void onSampleRate(uint32_t sampleRate);                /* poweramp.go:191-203 */
    /* Everything sync() sends to the device, taken under the unit's lock in ONE step: control-plane setters run concurrently with
     * the batch leader (controller.go:3493-3498), so versions, resolved values and taps must belong together. */
    struct Snapshot { uint64_t paramVersion, firVersion; int32_t values[8]; int nValues; std::vector<double> taps; bool withTaps; };
    Snapshot snapshot(uint64_t pushedFir) const;
private:
    friend Error PreparePowerAmp(Unit &unit, const filter::ImpulseResponses *responses);
    Error setDiscrete(const std::string &name, const std::string &value);
    Error setNumeric(const std::string &name, int32_t value);
    std::pair<std::shared_ptr<filter::Filter>, Error> compile(uint32_t sampleRate) const;   /* poweramp.go:25-127 */
    void recompile();
    int unitType_;
    mutable std::mutex mutex_;
    std::vector<Parameter> params_;
};

std::shared_ptr<Unit> CreateUnit(int unitType);                                            /* effects.go:443-516 */
Error PreparePowerAmp(Unit &unit, const filter::ImpulseResponses *responses);               /* poweramp.go:221-289 */
std::vector<std::string> ParameterTypes();                                                  /* effects.go:521-533 */
std::vector<std::string> UnitTypes();                                                       /* effects.go:538-568 */

}  // namespace effects

class Engine;

namespace signal {

class Chain {                                  /* signal.go:21-36 */
public:
    std::pair<int, Error> AppendUnit(int unitType);
    Error RemoveUnit(int id);
    Error MoveUp(int id);
    Error MoveDown(int id);
    std::pair<int, Error> UnitType(int id) const;
    Error SetBypass(int id, bool bypass);
    std::pair<bool, Error> GetBypass(int id) const;
    Error SetDiscreteValue(int id, const std::string &name, const std::string &value);
    std::pair<std::string, Error> GetDiscreteValue(int id, const std::string &name) const;
    Error SetNumericValue(int id, const std::string &name, int32_t value);
    std::pair<int32_t, Error> GetNumericValue(int id, const std::string &name) const;
    std::pair<std::vector<effects::Parameter>, Error> Parameters(int id) const;
    int Length() const;
    /* len(in) != len(out) is a silent no-op (signal.go:366).  Blocks until the batch this call joined has run. */
    void Process(const double *in, size_t nIn, double *out, size_t nOut, uint32_t sampleRate);

    int channel() const { return channel_; }
private:
    friend class gdg::Engine;
    struct Slot { std::shared_ptr<effects::Unit> unit; bool bypass; };
    Chain(Engine *engine, int channel, const filter::ImpulseResponses *responses) : engine_(engine), channel_(channel), responses_(responses) {}
    Engine *engine_;
    int channel_;
    const filter::ImpulseResponses *responses_;
    mutable std::mutex mutex_;
    std::vector<Slot> slots_;
    uint64_t layoutVersion_ = 1, pushedLayoutVersion_ = 0;
    std::vector<std::shared_ptr<effects::Unit>> retired_;   /* removed units whose device state must be released */
};

/* signal.CreateChain(responses) (signal.go:419-431): the next free channel of the default engine. */
std::shared_ptr<Chain> CreateChain(const filter::ImpulseResponses *responses);

}  // namespace signal

/*
 * One shard of channels on one GPU.  controller.process() has exactly N Chain.Process calls in
 * flight at a time (controller.go:2682-2705, N worker goroutines :3339-3341), so Chain::Process is
 * a rendezvous: every call deposits its buffers, the last arrival launches ONE batched
 * gdg_process_subset for all of them, everybody returns together.  If fewer than the expected
 * number of calls arrive within the timeout, the calls that did arrive are processed as a subset.
 */
class Engine {
public:
    /* one shard (context) per entry of `devices`; channel c lives on shard c * G / N (contiguous blocks).  The same
     * device may be listed more than once (independent contexts): that is how the routing is tested on a one-GPU box. */
    Engine(int nChannels, int maxFrames, std::vector<int> devices);
    Engine(int nChannels, int maxFrames, int device) : Engine(nChannels, maxFrames, std::vector<int>{ device }) {}
    ~Engine();
    static Engine &Default();                              /* configured by Configure() or GDG_CHANNELS / GDG_DEVICES */
    static void Configure(int nChannels, int maxFrames, int device);
    std::pair<std::shared_ptr<signal::Chain>, Error> CreateChain(const filter::ImpulseResponses *responses);
    void SetRendezvous(int expected, int timeoutMs) { expected_ = expected; timeoutMs_ = timeoutMs; }
    /* direct batch call for callers that already hold all channels' buffers (bench, tests) */
    Error ProcessAll(const double *const *in, double *const *out, int frames, uint32_t sampleRate);
    /* controller.processFiles between "the files are read" and "the files are written" (controller.go:2884-3219) over ALL shards:
     * the chains are synchronised to the devices, every shard runs its block of channels on its own thread, `window` blocks per step
     * (gdg_batch_run_shard; shard 0 also runs the metronome), and the master is finished once on shard 0's device: the shards'
     * float64 partial mixes added in shard order, then the metronome as aux input (options.metronome_to_master), then the encoder
     * (gdg_batch_finish_master) -- the reference's sum over all channels -> + aux -> encode (spatializer.go:300-310,
     * controller.go:3123-3219).  inputs: one per channel of the engine; outs: N + 3 host buffers as for gdg_batch_run
     * (NULL = "skipping output"); `samples` receives the length of every output. */
    Error BatchRun(const gdg_batch_input *inputs, int nInputs, const gdg_batch_options &options, int window, void *const *outs, size_t *samples);
    std::string LastError() const;
    int channels() const { return nChannels_; }
    int shards() const { return (int)shards_.size(); }
    /* shard of a channel and the channel's index inside it */
    int shardOf(int channel, int *local = nullptr) const;
    void shardRange(int shard, int *first, int *count) const;
    gdg_ctx *context(int shard = 0);                       /* creates the device context on first use */
    std::mutex &shardMutex(int shard);                     /* a context takes one call at a time (include/gdg.h) */
    void setError(const Error &e);                         /* what lastError() returns; also used by the spatializer / tuner twins for per-shard failures */

private:
    friend class signal::Chain;
    struct Pending { signal::Chain *chain; const double *in; double *out; int frames; uint32_t sampleRate; };
    struct Shard { int device = 0, first = 0, count = 0; gdg_ctx *ctx = nullptr; std::mutex mu; };
    void process(signal::Chain *chain, const double *in, double *out, int frames, uint32_t sampleRate);
    void runBatch(std::vector<Pending> batch);
    Error runShard(int shard, std::vector<Pending> &group, int frames, uint32_t sampleRate);
    Error sync(int shard, const std::vector<signal::Chain *> &chains, uint32_t sampleRate);
    int nChannels_, maxFrames_;
    std::vector<std::unique_ptr<Shard>> shards_;
    std::vector<std::shared_ptr<signal::Chain>> chains_;
    std::mutex mu_;
    std::condition_variable cv_;
    std::vector<Pending> pending_;
    bool executing_ = false;
    uint64_t generation_ = 0;
    int expected_ = 0, timeoutMs_ = 50;
    mutable std::mutex errMu_;
    std::string lastError_;
};

/* ---- spatializer.Spatializer (spatializer/spatializer.go:30-41): the ten methods, on top of an engine's shards -------------
 * Every shard mixes its block of channels to a partial (left, right) pair; the host adds the partials in shard order and then the
 * aux input (spatializer.go:300-310).  Error strings are the reference's. */
namespace spatializer {
constexpr uint32_t OUTPUT_COUNT = 2;
class Spatializer {
public:
    Spatializer(Engine *engine, uint32_t inputChannels);
    std::pair<double, Error> GetAzimuth(uint32_t inputChannel) const;
    std::pair<double, Error> GetDistance(uint32_t inputChannel) const;
    std::pair<double, Error> GetLevel(uint32_t inputChannel) const;
    uint32_t GetInputCount() const { return inputCount_; }
    uint32_t GetOutputCount() const { return OUTPUT_COUNT; }
    /* inputBuffers: inputCount rows of n samples; auxInputBuffer may be null; outputBuffers: 2 rows of n samples.
     * reuseChainOutputs: the inputs ARE the outputs of the engine's last block (what controller.process() passes,
     * controller.go:2744-2761) and are still on the devices: nothing is uploaded. */
    void Process(const double *const *inputBuffers, const double *auxInputBuffer, double *const *outputBuffers, size_t n, bool reuseChainOutputs = false);
    Error SetAzimuth(uint32_t inputChannel, double azimuth);
    Error SetDistance(uint32_t inputChannel, double distance);
    Error SetLevel(uint32_t inputChannel, double level);
    void SetSampleRate(uint32_t rate);
private:
    void push(uint32_t channel);
    Engine *engine_;
    uint32_t inputCount_;
    mutable std::mutex mutex_;
    struct Position { double azimuth = 0.0, distance = 0.0, level = 1.0; };
    std::vector<Position> positions_;
};
std::shared_ptr<Spatializer> Create(Engine *engine, uint32_t inputChannels);       /* spatializer.go:436-469 */
}  // namespace spatializer

/* ---- tuner.Tuner (tuner/tuner.go:62-65): Process enqueues, Analyze runs the 262144-point autocorrelation on the device ---- */
namespace tuner {
struct Result { int8_t cents; double frequency; std::string note; };              /* tuner.go:30-46 */
class Tuner {
public:
    explicit Tuner(int device);
    ~Tuner();
    std::pair<Result, Error> Analyze();
    void Process(const double *samples, size_t n, uint32_t sampleRate);
private:
    /* the reference's two locks (tuner.go:48-57): Process -- the audio path -- only ever takes mutexBuffer_ for one enqueue into the host ring;
     * Analyze owns the device context under mutexAnalyze_ and holds mutexBuffer_ (shared) just while it copies the ring out (tuner.go:408-412) */
    std::shared_mutex mutexBuffer_;
    std::vector<double> ring_;             /* circular.Buffer of NUM_SAMPLES (circular.go:33-105) */
    size_t pointer_ = 0;
    uint32_t sampleRate_ = 0;
    std::mutex mutexAnalyze_;
    std::vector<double> snapshot_;         /* the ring, oldest sample first */
    gdg_ctx *ctx_ = nullptr;
    int device_;
};
std::shared_ptr<Tuner> Create(int device = 0);                                     /* tuner.go:592-610 */
}  // namespace tuner

}  // namespace gdg

#endif
