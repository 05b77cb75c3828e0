/*
 * gdg_host.cpp -- implementation of the host-side mirror (see gdg_host.hpp) plus a flat C API
 * (gdgh_*) so that the Python test-suite can drive it through ctypes.
 * No sample is computed here: Process() always ends in libgdg.so (HIP).
 */
#include "gdg_host.hpp"

#include "../../include/gdg.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <algorithm>
#include <chrono>
#include <complex>
#include <map>
#include <sstream>
#include <thread>

namespace gdg {

static std::string format(const char *fmt, ...) __attribute__((format(printf, 1, 2)));
static std::string format(const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    return std::string(buf);
}

static double decibelsToFactor(int32_t decibels) {      /* effects.go:389-394 */
    double e = 0.05 * (double)decibels;
    return pow(10.0, e);
}

/* ================================ filter ================================================== */
namespace filter {

Filter::Filter(std::vector<double> coeffs, uint32_t sampleRate, double gainCompensation, std::string name)
    : data_(std::move(coeffs)), sampleRate_(sampleRate), gainCompensation_(gainCompensation), name_(std::move(name)) {}

std::pair<std::shared_ptr<Filter>, Error> Filter::Add(const std::shared_ptr<Filter> &other) const {
    if (!other) return { std::make_shared<Filter>(data_, sampleRate_, gainCompensation_, name_), "" };
    if (sampleRate_ != other->sampleRate_) return { nullptr, "Cannot add filters: Sample rates do not match." };
    std::vector<double> r(std::max(data_.size(), other->data_.size()), 0.0);
    std::copy(data_.begin(), data_.end(), r.begin());
    for (size_t i = 0; i < other->data_.size(); i++) r[i] += other->data_[i];
    return { std::make_shared<Filter>(r, sampleRate_, 0.0, name_ + " + " + other->name_), "" };
}

std::vector<double> Filter::Coefficients() const { return data_; }

std::shared_ptr<Filter> Filter::Multiply(double scalar) const {
    std::vector<double> r(data_.size());
    for (size_t i = 0; i < data_.size(); i++) r[i] = scalar * data_[i];
    return std::make_shared<Filter>(r, sampleRate_, 0.0, format("%g", scalar) + " * (" + name_ + ")");
}

std::shared_ptr<Filter> Filter::Normalize() const {
    double sum = 0.0;                                   /* estimateGain, filter.go:127-138 */
    for (double c : data_) sum += c * c;
    double gain = sqrt(sum);
    double fac = gainCompensation_ / gain;
    return Multiply(fac);
}

/* plain iterative radix-2 FFT for the set-up time Reduce (not on the audio path) */
static void hostFft(std::vector<std::complex<double>> &a, bool inverse) {
    const size_t n = a.size();
    for (size_t i = 1, j = 0; i < n; i++) {
        size_t bit = n >> 1;
        for (; j & bit; bit >>= 1) j ^= bit;
        j ^= bit;
        if (i < j) std::swap(a[i], a[j]);
    }
    for (size_t len = 2; len <= n; len <<= 1) {
        for (size_t i = 0; i < n; i += len)
            for (size_t k = 0; k < len / 2; k++) {
                double ang = (inverse ? 2.0 : -2.0) * M_PI * (double)k / (double)len;
                std::complex<double> w(cos(ang), sin(ang));
                std::complex<double> u = a[i + k], v = a[i + k + len / 2] * w;
                a[i + k] = u + v;
                a[i + k + len / 2] = u - v;
            }
    }
    if (inverse) for (auto &x : a) x /= (double)n;
}

static uint64_t nextPowerOfTwo(uint64_t v) { uint64_t p = 1; while (p < v) p <<= 1; return p; }

static double lanczosKernel(double x, double a) {       /* resample.go:10-31 */
    if (x == 0) return 1.0;
    if ((-a < x) && (x < a)) {
        double pi_x = M_PI * x;
        double pi_xa = pi_x / a;
        return (a * (sin(pi_x) * sin(pi_xa))) / (pi_x * pi_x);
    }
    return 0.0;
}

static double lanczosInterpolate(const std::vector<double> &s, double x) {   /* resample.go:36-66, a = 3 */
    int idx = (int)floor(x);
    double sum = 0.0;
    for (int i = idx + 1 - 3; i < idx + 1 + 3; i++)
        if (i >= 0 && i < (int)s.size()) sum += s[(size_t)i] * lanczosKernel(x - (double)i, 3.0);
    return sum;
}

std::shared_ptr<Filter> Filter::Reduce(uint32_t order) const {
    const size_t n = data_.size();
    if ((uint64_t)n <= (uint64_t)order) return std::make_shared<Filter>(data_, sampleRate_, gainCompensation_, name_);
    const size_t nSrc = (size_t)nextPowerOfTwo(n), nTgt = (size_t)nextPowerOfTwo(order);
    std::vector<std::complex<double>> fr(nSrc);
    for (size_t i = 0; i < n; i++) fr[i] = data_[i];
    hostFft(fr, false);
    const size_t posSrc = (nSrc >> 1) + 1, tgtHalf = nTgt >> 1, posTgt = tgtHalf + 1;
    std::vector<double> re(posSrc), im(posSrc);
    for (size_t i = 0; i < posSrc; i++) { re[i] = fr[i].real(); im[i] = fr[i].imag(); }
    double dx = (double)posSrc / (double)posTgt;         /* resample.Frequency, resample.go:109-142 */
    std::vector<std::complex<double>> frNew(nTgt);
    std::vector<std::complex<double>> pos(posTgt);
    for (size_t i = 0; i < posTgt; i++) pos[i] = { lanczosInterpolate(re, (double)i * dx), lanczosInterpolate(im, (double)i * dx) };
    for (size_t i = 0; i < std::min(posTgt, nTgt); i++) frNew[i] = pos[i];
    for (size_t i = 1; i < tgtHalf; i++) frNew[nTgt - i] = std::conj(pos[i]);
    /* RealInverseFourier first symmetrises its input (fft.go:899-906) and uses Re(DC), Re(Nyquist) */
    for (size_t i = 1; i < tgtHalf; i++) {
        std::complex<double> avg = 0.5 * (frNew[i] + std::conj(frNew[nTgt - i]));
        frNew[i] = avg;
        frNew[nTgt - i] = std::conj(avg);
    }
    frNew[0] = frNew[0].real();
    if (tgtHalf > 0) frNew[tgtHalf] = frNew[tgtHalf].real();
    hostFft(frNew, true);
    std::vector<double> out(order);
    for (size_t i = 0; i < order; i++) out[i] = frNew[i].real();
    return std::make_shared<Filter>(out, sampleRate_, gainCompensation_, name_);
}

std::shared_ptr<Filter> Empty(uint32_t sampleRate) { return std::make_shared<Filter>(std::vector<double>(), sampleRate, 0.0, "(EMPTY)"); }

std::shared_ptr<Filter> FromCoefficients(const std::vector<double> &coeffs, uint32_t sampleRate, const std::string &name) {
    return std::make_shared<Filter>(coeffs, sampleRate, 0.0, name);
}

std::vector<uint32_t> SampleRates() { return { 22050, 32000, 44100, 48000, 88200, 96000, 192000 }; }   /* filter.go:25-33 */

void ImpulseResponses::Add(const std::string &name, uint32_t sampleRate, int32_t compensationDecibels, const std::vector<double> &taps) {
    double fac = pow(10.0, 0.05 * (double)compensationDecibels);      /* filter.go:731-734 */
    responses_.push_back(Entry{ name, sampleRate, fac, taps });
}

std::shared_ptr<Filter> ImpulseResponses::CreateFilter(const std::string &name, uint32_t sampleRate) const {
    for (const auto &ir : responses_)
        if (ir.name == name && ir.sampleRate == sampleRate) return std::make_shared<Filter>(ir.data, ir.sampleRate, ir.gainCompensation, ir.name);
    return nullptr;
}

std::vector<std::string> ImpulseResponses::Names() const {
    std::vector<std::string> names;
    for (const auto &ir : responses_)
        if (std::find(names.begin(), names.end(), ir.name) == names.end()) names.push_back(ir.name);
    return names;
}

}  // namespace filter

/* ================================ effects ================================================= */
namespace effects {

const char *const STRING_NONE = "- NONE -";

static Parameter N(const char *name, const char *unit, int32_t mn, int32_t mx, int32_t def) {
    Parameter p;
    p.Name = name; p.Type = PARAMETER_TYPE_NUMERIC; p.PhysicalUnit = unit; p.Minimum = mn; p.Maximum = mx; p.NumericValue = def;
    p.DiscreteValueIndex = -1;
    return p;
}

static Parameter D(const char *name, int def, std::vector<std::string> values) {
    Parameter p;
    p.Name = name; p.Type = PARAMETER_TYPE_DISCRETE; p.PhysicalUnit = ""; p.Minimum = -1; p.Maximum = -1; p.NumericValue = -1;
    p.DiscreteValueIndex = def; p.DiscreteValues = std::move(values);
    return p;
}

/* parameter tables of the create*() functions, effects/<unit>.go */
static std::vector<Parameter> defaultParameters(int t) {
    const std::vector<std::string> follow = { "envelope", "level" };
    const std::vector<std::string> os = { "- NONE -", "2", "4" };
    switch (t) {
    case UNIT_SIGNALGENERATOR:
        return { N("input_amplitude", "%", 0, 100, 100), N("input_gain", "dB", -60, 0, 0),
                 D("signal_type", 0, { "sine", "triangle", "square", "sawtooth", "noise" }), N("signal_frequency", "Hz", 1, 20000, 440),
                 N("signal_amplitude", "%", 0, 100, 100), N("signal_gain", "dB", -60, 0, 0) };
    case UNIT_NOISEGATE:
        return { N("threshold_open", "dB", -60, 0, -20), N("threshold_close", "dB", -60, 0, -40), N("hold_time", "ms", 0, 1000, 50) };
    case UNIT_BANDPASS:
        return { D("filter_order", 0, { "2", "4", "6", "8" }), N("frequency_1", "Hz", 1, 20000, 300), N("frequency_2", "Hz", 1, 20000, 3000) };
    case UNIT_AUTOWAH:
        return { D("follow", 1, follow), N("level_1", "dB", -60, 0, -40), N("level_2", "dB", -60, 0, -10),
                 N("frequency_1", "Hz", 1, 20000, 300), N("frequency_2", "Hz", 1, 20000, 6000) };
    case UNIT_AUTOYOY:
        return { D("follow", 1, follow), N("level_1", "dB", -60, 0, -40), N("level_2", "dB", -60, 0, -10), N("depth", "%", 0, 100, 100) };
    case UNIT_COMPRESSOR:
        return { D("follow", 1, follow), N("gain_limit", "dB", 0, 30, 30), N("target_level", "dB", -30, 0, -20) };
    case UNIT_OCTAVER:
        return { D("follow", 1, follow), N("level_octave_up", "dB", -60, 0, -20), N("level_clean", "dB", -60, 0, -20),
                 N("level_dist", "dB", -60, 0, -20), N("level_octave_down_first", "dB", -60, 0, -20),
                 N("level_octave_down_second", "dB", -60, 0, -20), N("level_hysteresis", "dB", -60, 0, -20) };
    case UNIT_EXCESS:
        return { N("gain", "dB", -30, 30, 0), N("level", "dB", -30, 0, 0), D("oversampling", 0, os) };
    case UNIT_FUZZ:
        return { D("follow", 1, follow), N("bias", "%", -100, 100, 50), N("boost", "dB", 0, 30, 0), N("gain", "dB", -30, 30, 0),
                 N("fuzz", "%", 0, 100, 100), N("level", "dB", -30, 0, 0), D("oversampling", 0, os) };
    case UNIT_OVERDRIVE:
        return { N("boost", "dB", 0, 30, 0), N("gain", "dB", -30, 30, 0), N("drive", "%", 0, 100, 100), N("level", "dB", -30, 0, 0),
                 D("valve", 1, { "ECC82 (12AU7)", "ECC83 (12AX7)" }), D("oversampling", 0, os) };
    case UNIT_DISTORTION:
        return { N("boost", "dB", 0, 30, 0), N("gain", "dB", -30, 30, 0), N("level", "dB", -30, 0, 0), D("oversampling", 0, os) };
    case UNIT_TONESTACK:
        return { N("low", "dB", -30, 0, 0), N("middle", "dB", -30, 0, -2), N("presence", "dB", -30, 0, -5), N("high", "dB", -30, 0, -5) };
    case UNIT_CHORUS:
        return { N("depth", "%", 0, 100, 100), N("speed", "%", 1, 100, 30) };
    case UNIT_FLANGER:
        return { N("depth", "%", 0, 100, 100), N("speed", "%", 1, 100, 10) };
    case UNIT_PHASER:
        return { N("depth", "%", 0, 100, 100), N("speed", "%", 1, 100, 10), N("phase", "\xc2\xb0", -90, 90, 45) };
    case UNIT_TREMOLO:
        return { N("frequency", "0.1 Hz", 10, 100, 100), N("phase", "%", 0, 100, 50), N("depth", "dB", -60, 0, -10) };
    case UNIT_RINGMODULATOR:
        return { N("frequency", "Hz", 1, 100, 100) };
    case UNIT_DELAY:
        return { N("delay_time", "ms", 0, 1000, 200), N("feedback", "dB", -60, 0, -5), N("level", "dB", -30, 0, -5) };
    case UNIT_REVERB:
        return { N("mix", "%", 0, 100, 50) };
    case UNIT_POWERAMP:
        return { D("filter_order", 14, { "64", "128", "256", "512", "1024", "2048", "4096", "8192", "16384", "32768", "65536",
                                         "131072", "262144", "524288", "1048576" }) };
    case UNIT_CABINET:
        return { D("type", 0, { "- DEFAULT -" }) };
    default:
        return {};
    }
}

Unit::Unit(int unitType) : unitType_(unitType), params_(defaultParameters(unitType)) {}

Unit::~Unit() {
    if (backing.ctx && backing.owns_ctx) gdg_ctx_destroy(backing.ctx);
}

std::vector<Parameter> Unit::Parameters() const {
    std::lock_guard<std::mutex> lk(mutex_);
    return params_;
}

static int findParam(const std::vector<Parameter> &params, const std::string &name) {
    int idx = -1;
    for (size_t i = 0; i < params.size(); i++) if (params[i].Name == name) idx = (int)i;    /* last match wins, as upstream */
    return idx;
}

/* effects.go:144-207 */
Error Unit::setDiscrete(const std::string &name, const std::string &value) {
    int idx = findParam(params_, name);
    if (idx == -1) return format("Failed to set discrete value: Could not find parameter with name '%s'.", name.c_str());
    Parameter &p = params_[(size_t)idx];
    if (p.Type != PARAMETER_TYPE_DISCRETE) return format("Failed to set discrete value: Parameter '%s' is not discrete.", name.c_str());
    int valIdx = -1;
    for (size_t i = 0; i < p.DiscreteValues.size(); i++) if (p.DiscreteValues[i] == value) valIdx = (int)i;
    if (valIdx == -1) return format("Failed to set discrete value: Value '%s' is not valid for parameter '%s'.", value.c_str(), name.c_str());
    p.DiscreteValueIndex = valIdx;
    paramVersion++;
    return "";
}

/* effects.go:273-321 */
Error Unit::setNumeric(const std::string &name, int32_t value) {
    int idx = findParam(params_, name);
    if (idx == -1) return format("Failed to set numeric value: Could not find parameter with name '%s'.", name.c_str());
    Parameter &p = params_[(size_t)idx];
    if (p.Type != PARAMETER_TYPE_NUMERIC) return format("Failed to set numeric value: Parameter '%s' is not numeric.", name.c_str());
    if (value < p.Minimum || value > p.Maximum)
        return format("Failed to set numeric value: Parameter '%s' must be between '%d' and '%d' - got '%d'.", name.c_str(), p.Minimum, p.Maximum, value);
    p.NumericValue = value;
    paramVersion++;
    return "";
}

Error Unit::SetDiscreteValue(const std::string &name, const std::string &value) {
    std::lock_guard<std::mutex> lk(mutex_);
    Error err = setDiscrete(name, value);
    if (err.empty() && unitType_ == UNIT_POWERAMP) recompile();      /* poweramp.go:132-154 */
    return err;
}

Error Unit::SetNumericValue(const std::string &name, int32_t value) {
    std::lock_guard<std::mutex> lk(mutex_);
    Error err = setNumeric(name, value);
    if (err.empty() && unitType_ == UNIT_POWERAMP) recompile();      /* poweramp.go:159-181 */
    return err;
}

/* effects.go:212-255 */
std::pair<std::string, Error> Unit::GetDiscreteValue(const std::string &name) const {
    std::lock_guard<std::mutex> lk(mutex_);
    int idx = findParam(params_, name);
    if (idx == -1) return { "", format("Failed to get discrete value: Could not find parameter with name '%s'.", name.c_str()) };
    const Parameter &p = params_[(size_t)idx];
    if (p.Type != PARAMETER_TYPE_DISCRETE) return { "", format("Failed to get discrete value: Parameter '%s' is not discrete.", name.c_str()) };
    return { p.DiscreteValues[(size_t)p.DiscreteValueIndex], "" };
}

/* effects.go:337-384 */
std::pair<int32_t, Error> Unit::GetNumericValue(const std::string &name) const {
    std::lock_guard<std::mutex> lk(mutex_);
    int idx = findParam(params_, name);
    if (idx == -1) return { 0, format("Failed to get numeric value: Could not find parameter with name '%s'.", name.c_str()) };
    const Parameter &p = params_[(size_t)idx];
    if (p.Type != PARAMETER_TYPE_NUMERIC) return { 0, format("Failed to get numeric value: Parameter '%s' is not numeric.", name.c_str()) };
    return { p.NumericValue, "" };
}

void Unit::resolved(int32_t out[8]) const {
    std::lock_guard<std::mutex> lk(mutex_);
    for (int i = 0; i < 8; i++) {
        out[i] = 0;
        if ((size_t)i < params_.size()) out[i] = (params_[(size_t)i].Type == PARAMETER_TYPE_NUMERIC) ? params_[(size_t)i].NumericValue : params_[(size_t)i].DiscreteValueIndex;
    }
}

/* poweramp.go:25-127 (called with the unit's mutex held) */
std::pair<std::shared_ptr<filter::Filter>, Error> Unit::compile(uint32_t sr) const {
    if (!impulseResponses) return { nullptr, "Could not compile filter: No impulse responses were loaded." };
    uint32_t targetOrder = 0;
    {
        int idx = findParam(params_, "filter_order");
        if (idx >= 0) {
            const std::string &s = params_[(size_t)idx].DiscreteValues[(size_t)params_[(size_t)idx].DiscreteValueIndex];
            char *end = nullptr;
            unsigned long long v = strtoull(s.c_str(), &end, 10);
            if (!end || *end != '\0' || v > 0xffffffffULL) return { nullptr, format("Could not parse filter target order: '%s'", s.c_str()) };
            targetOrder = (uint32_t)v;
        }
    }
    std::vector<std::shared_ptr<filter::Filter>> filters(NUM_FILTERS);
    for (int i = 0; i < NUM_FILTERS; i++) {
        std::string pf = "filter_" + std::to_string(i + 1), pl = "level_" + std::to_string(i + 1);
        int fi = findParam(params_, pf), li = findParam(params_, pl);
        if (fi < 0 || li < 0 || params_[(size_t)fi].Type != PARAMETER_TYPE_DISCRETE || params_[(size_t)li].Type != PARAMETER_TYPE_NUMERIC)
            return { nullptr, format("Error parsing values for filter %d.", i) };
        const std::string &name = params_[(size_t)fi].DiscreteValues[(size_t)params_[(size_t)fi].DiscreteValueIndex];
        int32_t level = params_[(size_t)li].NumericValue;
        if (name != STRING_NONE) {
            double fac = decibelsToFactor(level);
            auto flt = impulseResponses->CreateFilter(name, sr);
            if (!flt) return { nullptr, format("Failed to load filter '%s' for sample rate '%u'.", name.c_str(), sr) };
            if (targetOrder > 0) flt = flt->Reduce(targetOrder);
            flt = flt->Normalize();
            flt = flt->Multiply(fac);
            filters[(size_t)i] = flt;
        }
    }
    std::shared_ptr<filter::Filter> composite = filter::Empty(sr);
    for (auto &flt : filters) {
        auto r = composite->Add(flt);
        if (!r.second.empty()) return { nullptr, "Failed to add filter: " + r.second };
        composite = r.first;
    }
    return { composite, "" };
}

void Unit::recompile() {
    auto r = compile(sampleRate);
    if (r.second.empty()) {                    /* on error the previous filter stays (poweramp.go:147-151) */
        firTaps = r.first->Coefficients();
        firVersion++;
    }
}

void Unit::onSampleRate(uint32_t sr) {
    if (unitType_ != UNIT_POWERAMP) return;
    std::lock_guard<std::mutex> lk(mutex_);
    if (sr != sampleRate) {                    /* poweramp.go:191-203 */
        sampleRate = sr;
        recompile();
    }
}

std::shared_ptr<Unit> CreateUnit(int unitType) {
    if (unitType < 0 || unitType >= UNIT_COUNT) return nullptr;
    return std::make_shared<Unit>(unitType);
}

Error PreparePowerAmp(Unit &unit, const filter::ImpulseResponses *responses) {
    if (unit.Type() != UNIT_POWERAMP) return "Cannot prepare power amp: Unit is not a power amp.";
    if (!responses) return "Cannot prepare power amp: Impulse responses are nil.";
    std::lock_guard<std::mutex> lk(unit.mutex_);
    std::vector<std::string> names = { STRING_NONE };
    for (auto &n : responses->Names()) names.push_back(n);
    for (int i = 0; i < NUM_FILTERS; i++) {
        std::string s = std::to_string(i + 1);
        unit.params_.push_back(D(("filter_" + s).c_str(), 0, names));
        unit.params_.push_back(N(("level_" + s).c_str(), "dB", -60, 0, 0));
    }
    unit.impulseResponses = responses;
    return "";
}

std::vector<std::string> ParameterTypes() { return { "invalid", "discrete", "numeric" }; }

std::vector<std::string> UnitTypes() {
    return { "signal_generator", "noise_gate", "bandpass", "auto_wah", "auto_yoy", "compressor", "octaver", "excess", "fuzz",
             "overdrive", "distortion", "tone_stack", "chorus", "flanger", "phaser", "tremolo", "ring_modulator", "delay",
             "reverb", "power_amp", "cabinet" };
}

Unit::Snapshot Unit::snapshot(uint64_t pushedFir) const {
    std::lock_guard<std::mutex> lk(mutex_);
    Snapshot snap;
    snap.paramVersion = paramVersion;
    snap.firVersion = firVersion;
    snap.nValues = std::min<int>(8, (int)params_.size());
    if (unitType_ == UNIT_POWERAMP) snap.nValues = 1;
    for (int i = 0; i < 8; i++) {
        snap.values[i] = 0;
        if ((size_t)i < params_.size()) snap.values[i] = (params_[(size_t)i].Type == PARAMETER_TYPE_NUMERIC) ? params_[(size_t)i].NumericValue : params_[(size_t)i].DiscreteValueIndex;
    }
    snap.withTaps = unitType_ == UNIT_POWERAMP && pushedFir != firVersion;
    if (snap.withTaps) snap.taps = firTaps;            /* a COPY: a concurrent setter may reassign firTaps as soon as the lock is gone */
    return snap;
}

/* Push parameters / taps of one unit to the device context it lives in.  Step 1: ONE snapshot under the unit's lock (version
 * counters, resolved values, a copy of the taps).  Step 2: the ABI calls, without the lock.  Step 3: remember the SNAPSHOT's
 * versions -- a setter that lands in between bumps the live counters and is pushed with the next block. */
static Error pushUnit(Unit &u) {
    gdg_ctx *ctx = u.backing.ctx;
    const Unit::Snapshot snap = u.snapshot(u.pushedFirVersion);
    if (u.pushedParamVersion != snap.paramVersion) {
        for (int i = 0; i < snap.nValues; i++)
            if (gdg_unit_set_param(ctx, u.backing.handle, i, snap.values[i]) != GDG_OK) return gdg_last_error(ctx);
        u.pushedParamVersion = snap.paramVersion;
    }
    if (snap.withTaps) {
        /* every successful Set of a power amp parameter -- the same value or not -- replaced currentFilter in the reference
         * (poweramp.go:131-181), i.e. fresh convolution state: gdg_unit_set_fir resets it */
        if (gdg_unit_set_fir(ctx, u.backing.handle, snap.taps.data(), (int)snap.taps.size()) != GDG_OK) return gdg_last_error(ctx);
        u.pushedFirVersion = snap.firVersion;
    }
    return "";
}

void Unit::Process(const double *in, double *out, size_t n, uint32_t sr) {
    if (!backing.ctx) {
        gdg_ctx *ctx = nullptr;
        if (gdg_ctx_create(1, GDG_HOST_MAX_FRAMES, 0, &ctx) != GDG_OK) { for (size_t i = 0; i < n; i++) out[i] = 0.0; return; }
        backing.ctx = ctx;
        backing.owns_ctx = true;
        gdg_unit_create(ctx, 0, unitType_, &backing.handle);
        uint8_t bypass = 0;
        gdg_chain_set(ctx, 0, &backing.handle, &bypass, 1);
    }
    if (!backing.owns_ctx) { for (size_t i = 0; i < n; i++) out[i] = 0.0; return; }   /* owned by a chain: use Chain::Process */
    onSampleRate(sr);
    Error err = pushUnit(*this);
    const double *ins[1] = { in };
    double *outs[1] = { out };
    if (!err.empty() || gdg_process(backing.ctx, ins, outs, (int)n, sr) != GDG_OK)
        for (size_t i = 0; i < n; i++) out[i] = 0.0;        /* the reference's failure mode inside Process: zeros, no error */
}

}  // namespace effects

/* ================================ signal ================================================== */
namespace signal {

std::pair<int, Error> Chain::AppendUnit(int unitType) {            /* signal.go:52-86 */
    auto unit = effects::CreateUnit(unitType);
    if (!unit) return { -1, "Failed to create effects unit." };
    if (unitType == effects::UNIT_POWERAMP) effects::PreparePowerAmp(*unit, responses_);
    std::lock_guard<std::mutex> lk(mutex_);
    slots_.push_back(Slot{ unit, true });                          /* new units start in bypass mode */
    layoutVersion_++;
    return { (int)slots_.size() - 1, "" };
}

Error Chain::RemoveUnit(int id) {                                  /* signal.go:91-113 */
    std::lock_guard<std::mutex> lk(mutex_);
    if (id < 0 || id >= (int)slots_.size()) return format("Cannot remove unit %d.", id);
    retired_.push_back(slots_[(size_t)id].unit);
    slots_.erase(slots_.begin() + id);
    layoutVersion_++;
    return "";
}

Error Chain::MoveUp(int id) {                                      /* signal.go:118-135 */
    std::lock_guard<std::mutex> lk(mutex_);
    if (id <= 0 || id >= (int)slots_.size()) return format("Cannot move unit %d up.", id);
    std::swap(slots_[(size_t)id], slots_[(size_t)id - 1]);
    layoutVersion_++;
    return "";
}

Error Chain::MoveDown(int id) {                                    /* signal.go:140-157 */
    std::lock_guard<std::mutex> lk(mutex_);
    if (id < 0 || id >= (int)slots_.size() - 1) return format("Cannot move unit %d down.", id);
    std::swap(slots_[(size_t)id], slots_[(size_t)id + 1]);
    layoutVersion_++;
    return "";
}

std::pair<int, Error> Chain::UnitType(int id) const {
    std::lock_guard<std::mutex> lk(mutex_);
    if (id < 0 || id >= (int)slots_.size()) return { -1, format("Cannot get unit type: No unit %d.", id) };
    return { slots_[(size_t)id].unit->Type(), "" };
}

Error Chain::SetBypass(int id, bool bypass) {
    std::lock_guard<std::mutex> lk(mutex_);
    if (id < 0 || id >= (int)slots_.size()) return format("Cannot %s bypass: No unit %d.", bypass ? "enable" : "disable", id);
    slots_[(size_t)id].bypass = bypass;
    layoutVersion_++;
    return "";
}

std::pair<bool, Error> Chain::GetBypass(int id) const {
    std::lock_guard<std::mutex> lk(mutex_);
    if (id < 0 || id >= (int)slots_.size()) return { false, format("Cannot get bypass value: No unit %d.", id) };
    return { slots_[(size_t)id].bypass, "" };
}

#define CHAIN_UNIT_OR(id, what, failure)                                                   \
    std::shared_ptr<effects::Unit> unit;                                                   \
    {                                                                                      \
        std::lock_guard<std::mutex> lk(mutex_);                                            \
        if (id < 0 || id >= (int)slots_.size()) return failure(format("Cannot " what ": No unit %d.", id)); \
        unit = slots_[(size_t)id].unit;                                                    \
    }

Error Chain::SetDiscreteValue(int id, const std::string &name, const std::string &value) {
    CHAIN_UNIT_OR(id, "set discrete value", Error)
    return unit->SetDiscreteValue(name, value);
}

std::pair<std::string, Error> Chain::GetDiscreteValue(int id, const std::string &name) const {
    auto failure = [](const std::string &e) { return std::make_pair(std::string(), e); };
    CHAIN_UNIT_OR(id, "get discrete value", failure)
    return unit->GetDiscreteValue(name);
}

Error Chain::SetNumericValue(int id, const std::string &name, int32_t value) {
    CHAIN_UNIT_OR(id, "set numeric value", Error)
    return unit->SetNumericValue(name, value);
}

std::pair<int32_t, Error> Chain::GetNumericValue(int id, const std::string &name) const {
    auto failure = [](const std::string &e) { return std::make_pair((int32_t)0, e); };
    CHAIN_UNIT_OR(id, "get numeric value", failure)
    return unit->GetNumericValue(name);
}

std::pair<std::vector<effects::Parameter>, Error> Chain::Parameters(int id) const {
    auto failure = [](const std::string &e) { return std::make_pair(std::vector<effects::Parameter>(), e); };
    CHAIN_UNIT_OR(id, "get parameters", failure)
    return { unit->Parameters(), "" };
}

int Chain::Length() const {
    std::lock_guard<std::mutex> lk(mutex_);
    return (int)slots_.size();
}

void Chain::Process(const double *in, size_t nIn, double *out, size_t nOut, uint32_t sampleRate) {
    if (nIn != nOut) return;                                        /* signal.go:366: silent no-op */
    engine_->process(this, in, out, (int)nIn, sampleRate);
}

std::shared_ptr<Chain> CreateChain(const filter::ImpulseResponses *responses) {
    auto r = Engine::Default().CreateChain(responses);
    return r.first;
}

}  // namespace signal

/* ================================ Engine ================================================== */

static Engine *g_default = nullptr;
static std::mutex g_default_mu;

static std::vector<int> devicesFromEnv() {
    /* GDG_DEVICES = "0,1,2,..." (one shard per entry); GDG_DEVICE = a single device; default: every visible device */
    std::vector<int> devs;
    if (const char *list = getenv("GDG_DEVICES")) {
        std::stringstream ss(list);
        std::string item;
        while (std::getline(ss, item, ',')) if (!item.empty()) devs.push_back(atoi(item.c_str()));
    } else if (const char *one = getenv("GDG_DEVICE")) {
        devs.push_back(atoi(one));
    } else {
        int n = gdg_device_count();
        for (int d = 0; d < std::max(n, 1); d++) devs.push_back(d);
    }
    return devs;
}

void Engine::Configure(int nChannels, int maxFrames, int device) {
    std::lock_guard<std::mutex> lk(g_default_mu);
    delete g_default;
    g_default = new Engine(nChannels, maxFrames, device);
}

Engine &Engine::Default() {
    std::lock_guard<std::mutex> lk(g_default_mu);
    if (!g_default) {
        const char *ch = getenv("GDG_CHANNELS");
        g_default = new Engine(ch ? atoi(ch) : 1, GDG_HOST_MAX_FRAMES, devicesFromEnv());
    }
    return *g_default;
}

Engine::Engine(int nChannels, int maxFrames, std::vector<int> devices) : nChannels_(nChannels), maxFrames_(maxFrames) {
    if (devices.empty()) devices.push_back(0);
    int G = std::min<int>((int)devices.size(), std::max(nChannels, 1));      /* never more shards than channels */
    for (int g = 0; g < G; g++) {
        std::unique_ptr<Shard> sh(new Shard());
        sh->device = devices[(size_t)g];
        sh->first = (int)(((long long)g * nChannels) / G);                    /* contiguous blocks */
        sh->count = (int)(((long long)(g + 1) * nChannels) / G) - sh->first;
        shards_.push_back(std::move(sh));
    }
}

Engine::~Engine() {
    chains_.clear();
    for (auto &sh : shards_) if (sh->ctx) gdg_ctx_destroy(sh->ctx);
}

int Engine::shardOf(int channel, int *local) const {
    for (size_t g = 0; g < shards_.size(); g++)
        if (channel >= shards_[g]->first && channel < shards_[g]->first + shards_[g]->count) {
            if (local) *local = channel - shards_[g]->first;
            return (int)g;
        }
    return -1;
}

void Engine::shardRange(int shard, int *first, int *count) const {
    *first = shards_[(size_t)shard]->first;
    *count = shards_[(size_t)shard]->count;
}

std::mutex &Engine::shardMutex(int shard) { return shards_[(size_t)shard]->mu; }

void Engine::setError(const Error &e) {
    std::lock_guard<std::mutex> lk(errMu_);
    lastError_ = e;
}

gdg_ctx *Engine::context(int shard) {
    Shard &sh = *shards_[(size_t)shard];
    if (!sh.ctx && sh.count > 0) {
        int rc = gdg_ctx_create(sh.count, maxFrames_, sh.device, &sh.ctx);
        if (rc != GDG_OK) { sh.ctx = nullptr; setError(format("gdg_ctx_create failed with %d on device %d (no usable HIP device; there is no CPU fallback)", rc, sh.device)); }
    }
    return sh.ctx;
}

std::string Engine::LastError() const {
    std::lock_guard<std::mutex> lk(errMu_);
    return lastError_;
}

std::pair<std::shared_ptr<signal::Chain>, Error> Engine::CreateChain(const filter::ImpulseResponses *responses) {
    std::lock_guard<std::mutex> lk(mu_);
    if ((int)chains_.size() >= nChannels_) return { nullptr, format("engine has only %d channels", nChannels_) };
    std::shared_ptr<signal::Chain> c(new signal::Chain(this, (int)chains_.size(), responses));
    chains_.push_back(c);
    if (expected_ < (int)chains_.size()) expected_ = (int)chains_.size();
    return { c, "" };
}

/* Bring the device side of the listed chains (all on `shard`) up to date: units, parameters, taps, slot lists.  THE algorithm
 * (the Go shim's chainStruct.sync is the same, statement for statement -- INTEGRATION.md section 3):
 *   1. destroy the device units of removed slots;
 *   2. per slot: create the device unit if it has none; a NON-bypassed power amp notices a new sample rate and recompiles
 *      (poweramp.go:191-203); push the unit from ONE snapshot taken under its lock (pushUnit);
 *   3. send the slot list (handles + bypass flags) if the layout changed. */
Error Engine::sync(int shard, const std::vector<signal::Chain *> &chains, uint32_t sampleRate) {
    gdg_ctx *ctx = context(shard);
    if (!ctx) return LastError();
    for (signal::Chain *ch : chains) {
        int local = 0;
        shardOf(ch->channel_, &local);
        std::lock_guard<std::mutex> lk(ch->mutex_);
        for (auto &u : ch->retired_)
            if (u->backing.ctx == ctx && u->backing.handle >= 0) { gdg_unit_destroy(ctx, u->backing.handle); u->backing.handle = -1; u->backing.ctx = nullptr; }
        ch->retired_.clear();
        for (auto &s : ch->slots_) {
            effects::Unit &u = *s.unit;
            if (u.backing.handle < 0) {
                if (gdg_unit_create(ctx, local, u.Type(), &u.backing.handle) != GDG_OK) return gdg_last_error(ctx);
                u.backing.ctx = ctx;
                u.pushedParamVersion = 0;
                u.pushedFirVersion = 0;
                ch->pushedLayoutVersion_ = 0;
            }
            if (!s.bypass) u.onSampleRate(sampleRate);               /* only a processed power amp notices the rate */
            Error e = effects::pushUnit(u);
            if (!e.empty()) return e;
        }
        if (ch->pushedLayoutVersion_ != ch->layoutVersion_) {
            std::vector<int> handles;
            std::vector<uint8_t> bypass;
            for (auto &s : ch->slots_) { handles.push_back(s.unit->backing.handle); bypass.push_back(s.bypass ? 1 : 0); }
            if (gdg_chain_set(ctx, local, handles.data(), bypass.data(), (int)handles.size()) != GDG_OK) return gdg_last_error(ctx);
            ch->pushedLayoutVersion_ = ch->layoutVersion_;
        }
    }
    return "";
}

/* one shard's part of a batch: sync, then ONE gdg_process_subset over its channels (local indices) */
Error Engine::runShard(int shard, std::vector<Pending> &group, int frames, uint32_t sr) {
    std::lock_guard<std::mutex> lk(shards_[(size_t)shard]->mu);
    std::vector<signal::Chain *> chains;
    std::vector<int> channels;
    std::vector<const double *> ins;
    std::vector<double *> outs;
    for (auto &p : group) {
        int local = 0;
        shardOf(p.chain->channel(), &local);
        chains.push_back(p.chain); channels.push_back(local); ins.push_back(p.in); outs.push_back(p.out);
    }
    Error e = sync(shard, chains, sr);
    gdg_ctx *ctx = shards_[(size_t)shard]->ctx;
    if (e.empty() && gdg_process_subset(ctx, channels.data(), (int)channels.size(), ins.data(), outs.data(), frames, sr) != GDG_OK)
        e = gdg_last_error(ctx);
    if (!e.empty()) {
        /* the reference's Process has no error return: failures produce zeros (effects/poweramp.go:210-214) */
        setError(e);
        for (auto &p : group) memset(p.out, 0, sizeof(double) * (size_t)p.frames);
    }
    return e;
}

void Engine::runBatch(std::vector<Pending> batch) {
    /* one launch per distinct (frames, sample rate) among the deposited calls -- in practice exactly one -- and per shard;
     * the shards (GPUs) of one group run concurrently: channels are independent, there is nothing to exchange */
    while (!batch.empty()) {
        int frames = batch[0].frames;
        uint32_t sr = batch[0].sampleRate;
        std::vector<Pending> group, rest;
        for (auto &p : batch) ((p.frames == frames && p.sampleRate == sr) ? group : rest).push_back(p);
        std::sort(group.begin(), group.end(), [](const Pending &a, const Pending &b) { return a.chain->channel() < b.chain->channel(); });
        std::vector<std::vector<Pending>> perShard(shards_.size());
        for (auto &p : group) {
            int g = shardOf(p.chain->channel());
            if (g >= 0) perShard[(size_t)g].push_back(p);
        }
        std::vector<std::thread> workers;
        int first = -1;
        for (size_t g = 0; g < perShard.size(); g++) {
            if (perShard[g].empty()) continue;
            if (first < 0) { first = (int)g; continue; }                 /* the caller's thread takes the first shard itself */
            workers.emplace_back([this, g, &perShard, frames, sr]() { runShard((int)g, perShard[g], frames, sr); });
        }
        if (first >= 0) runShard(first, perShard[(size_t)first], frames, sr);
        for (auto &w : workers) w.join();
        batch.swap(rest);
    }
}

/* Chain.Process = rendezvous (controller.go:2682-2705 keeps exactly N calls in flight): deposit, and either wait for the batch
 * this call joined to finish, or -- as the last arrival, or after the grace period -- run it. */
void Engine::process(signal::Chain *chain, const double *in, double *out, int frames, uint32_t sampleRate) {
    std::unique_lock<std::mutex> lk(mu_);
    cv_.wait(lk, [&] { return !executing_; });
    pending_.push_back(Pending{ chain, in, out, frames, sampleRate });
    const uint64_t myGen = generation_;
    bool run = (int)pending_.size() >= std::max(1, expected_);
    if (!run) {
        bool released = cv_.wait_for(lk, std::chrono::milliseconds(timeoutMs_), [&] { return generation_ != myGen; });
        if (released) return;
        if (generation_ != myGen || executing_) { cv_.wait(lk, [&] { return generation_ != myGen; }); return; }
        run = true;                                      /* timed out: process whoever has arrived */
    }
    std::vector<Pending> batch;
    batch.swap(pending_);
    executing_ = true;
    lk.unlock();
    runBatch(batch);
    lk.lock();
    executing_ = false;
    generation_++;
    cv_.notify_all();
}

Error Engine::ProcessAll(const double *const *in, double *const *out, int frames, uint32_t sampleRate) {
    std::vector<Pending> batch;
    {
        std::lock_guard<std::mutex> lk(mu_);
        for (size_t c = 0; c < chains_.size(); c++) batch.push_back(Pending{ chains_[c].get(), in[c], out[c], frames, sampleRate });
    }
    setError("");
    runBatch(batch);
    return LastError();
}

Error Engine::BatchRun(const gdg_batch_input *inputs, int nInputs, const gdg_batch_options &options, int window, void *const *outs, size_t *samples) {
    if (!inputs || !outs) return "BatchRun: no inputs or no outputs";
    if (nInputs != nChannels_) return format("BatchRun: %d inputs for %d channels", nInputs, nChannels_);
    const int G = shards();
    const int N = nChannels_;
    setError("");
    std::vector<std::shared_ptr<signal::Chain>> chains;
    {
        std::lock_guard<std::mutex> lk(mu_);
        chains = chains_;
    }
    /* 1. per shard: the device follows the chains (units, parameters, filters, layout) as before a Process call; the job's length is
     *    the longest shard's (the reference pads every channel to the longest input, controller.go:3005-3045) */
    size_t job = 0;
    for (int g = 0; g < G; g++) {
        int first = 0, count = 0;
        shardRange(g, &first, &count);
        if (count <= 0) continue;
        std::lock_guard<std::mutex> lk(shards_[(size_t)g]->mu);
        gdg_ctx *ctx = context(g);
        if (!ctx) return LastError();
        std::vector<signal::Chain *> mine;
        for (auto &c : chains) if (c->channel() >= first && c->channel() < first + count) mine.push_back(c.get());
        Error e = sync(g, mine, options.target_rate);
        if (!e.empty()) { setError(e); return e; }
        if (gdg_ctx_set_window(ctx, window) != GDG_OK) { setError(gdg_last_error(ctx)); return LastError(); }
        size_t len = 0;
        if (gdg_batch_length(ctx, inputs + first, count, options.target_rate, &len) != GDG_OK) { setError(gdg_last_error(ctx)); return LastError(); }
        job = std::max(job, len);
    }
    if (samples) *samples = job;
    if (job == 0) return "";
    /* 2. the shards, concurrently: encoded chain outputs straight into the caller's buffers, partial master mixes as float64 */
    std::vector<std::vector<double>> left((size_t)G), right((size_t)G);
    std::vector<double> metronome(job, 0.0);
    std::vector<Error> errs((size_t)G);
    auto one = [&](int g) {
        int first = 0, count = 0;
        shardRange(g, &first, &count);
        if (count <= 0) return;
        std::lock_guard<std::mutex> lk(shards_[(size_t)g]->mu);
        gdg_ctx *ctx = shards_[(size_t)g]->ctx;
        left[(size_t)g].assign(job, 0.0);
        right[(size_t)g].assign(job, 0.0);
        gdg_batch_shard_out so;
        memset(&so, 0, sizeof(so));
        so.master_left = left[(size_t)g].data();
        so.master_right = right[(size_t)g].data();
        so.job_samples = job;
        if (g == 0) { so.metronome_bytes = outs[N + 2]; so.metronome = metronome.data(); }      /* shard 0 runs the metronome */
        gdg_batch_options o = options;
        o.metronome_to_master = 0;                       /* the aux input joins the master once, after the shards' sums */
        if (gdg_batch_run_shard(ctx, inputs + first, count, &o, outs + first, &so) != GDG_OK) errs[(size_t)g] = gdg_last_error(ctx);
    };
    {
        std::vector<std::thread> workers;
        for (int g = 1; g < G; g++) workers.emplace_back(one, g);
        one(0);
        for (auto &w : workers) w.join();
    }
    for (int g = 0; g < G; g++) if (!errs[(size_t)g].empty()) { setError(format("shard %d: %s", g, errs[(size_t)g].c_str())); return LastError(); }
    /* 3. the master mix, finished on shard 0's device */
    std::vector<const double *> lp, rp;
    for (int g = 0; g < G; g++) if (!left[(size_t)g].empty()) { lp.push_back(left[(size_t)g].data()); rp.push_back(right[(size_t)g].data()); }
    std::lock_guard<std::mutex> lk(shards_[0]->mu);
    gdg_ctx *ctx0 = shards_[0]->ctx;
    if (gdg_batch_finish_master(ctx0, options.out_format, lp.data(), rp.data(), (int)lp.size(), options.metronome_to_master ? metronome.data() : nullptr, job,
                                options.target_rate, options.run_meters, outs[N], outs[N + 1]) != GDG_OK) {
        setError(gdg_last_error(ctx0));
        return LastError();
    }
    return "";
}

/* ================================ spatializer ============================================= */
namespace spatializer {

Spatializer::Spatializer(Engine *engine, uint32_t inputChannels) : engine_(engine), inputCount_(inputChannels), positions_(inputChannels) {}

/* the reference checks `inputChannel > inputCount` (spatializer.go:73, :93 ...) and would index out of range for
 * inputChannel == inputCount; here that one value is an error as well */
#define SPAT_CHECK(what, failure)                                                                                   \
    if (inputChannel >= inputCount_) return failure(format("Cannot " what " for channel %u: Only %u channels exist.", inputChannel, inputCount_));

std::pair<double, Error> Spatializer::GetAzimuth(uint32_t inputChannel) const {
    auto failure = [](const std::string &e) { return std::make_pair(0.0, e); };
    SPAT_CHECK("get azimuth", failure)
    std::lock_guard<std::mutex> lk(mutex_);
    return { positions_[inputChannel].azimuth, "" };
}
std::pair<double, Error> Spatializer::GetDistance(uint32_t inputChannel) const {
    auto failure = [](const std::string &e) { return std::make_pair(0.0, e); };
    SPAT_CHECK("get distance", failure)
    std::lock_guard<std::mutex> lk(mutex_);
    return { positions_[inputChannel].distance, "" };
}
std::pair<double, Error> Spatializer::GetLevel(uint32_t inputChannel) const {
    auto failure = [](const std::string &e) { return std::make_pair(0.0, e); };
    SPAT_CHECK("get level", failure)
    std::lock_guard<std::mutex> lk(mutex_);
    return { positions_[inputChannel].level, "" };
}

void Spatializer::push(uint32_t channel) {                 /* called with mutex_ held */
    int local = 0;
    int g = engine_->shardOf((int)channel, &local);
    if (g < 0) return;
    std::lock_guard<std::mutex> lk(engine_->shardMutex(g));
    gdg_ctx *ctx = engine_->context(g);
    const Position &p = positions_[channel];
    if (ctx) gdg_spatializer_set_position(ctx, local, p.azimuth, p.distance, p.level);
}

Error Spatializer::SetAzimuth(uint32_t inputChannel, double azimuth) {
    SPAT_CHECK("set azimuth", Error)
    std::lock_guard<std::mutex> lk(mutex_);
    positions_[inputChannel].azimuth = azimuth;
    push(inputChannel);
    return "";
}
Error Spatializer::SetDistance(uint32_t inputChannel, double distance) {
    SPAT_CHECK("set distance", Error)
    if (distance < 0.0 || distance > 10.0) return "Failed to set distance: Value must be within [0, 10].";
    std::lock_guard<std::mutex> lk(mutex_);
    positions_[inputChannel].distance = distance;
    push(inputChannel);
    return "";
}
Error Spatializer::SetLevel(uint32_t inputChannel, double level) {
    SPAT_CHECK("set distance", Error)                      /* the reference's message says "distance" here too (spatializer.go:395) */
    if (level < 0.0 || level > 1.0) return "Failed to set level: Value must be within [0, 1].";
    std::lock_guard<std::mutex> lk(mutex_);
    positions_[inputChannel].level = level;
    push(inputChannel);
    return "";
}

void Spatializer::SetSampleRate(uint32_t rate) {           /* spatializer.go:418-431: new (zeroed) history buffers */
    for (int g = 0; g < engine_->shards(); g++) {
        std::lock_guard<std::mutex> lk(engine_->shardMutex(g));
        gdg_ctx *ctx = engine_->context(g);
        if (ctx) gdg_spatializer_set_sample_rate(ctx, rate);
    }
}

void Spatializer::Process(const double *const *inputBuffers, const double *auxInputBuffer, double *const *outputBuffers, size_t n, bool reuseChainOutputs) {
    double *left = outputBuffers[0], *right = outputBuffers[1];
    for (size_t i = 0; i < n; i++) { left[i] = 0.0; right[i] = 0.0; }
    const int G = engine_->shards();
    std::vector<std::vector<double>> part((size_t)G * 2, std::vector<double>(n, 0.0));
    std::vector<std::thread> workers;
    auto one = [&](int g) {
        int first = 0, count = 0;
        engine_->shardRange(g, &first, &count);
        if (count <= 0) return;
        /* gdg_spatialize reads `count` row pointers: a spatializer with fewer inputs than the engine has channels cannot feed this
         * shard (the reference creates it with the engine's channel count, controller.go:3281-3287) */
        if ((uint64_t)first + (uint64_t)count > (uint64_t)inputCount_) {
            engine_->setError(format("spatializer with %u inputs cannot feed shard %d (channels %d to %d)", inputCount_, g, first, first + count - 1));
            return;
        }
        std::lock_guard<std::mutex> lk(engine_->shardMutex(g));
        gdg_ctx *ctx = engine_->context(g);
        if (!ctx) return;
        double *pl = part[(size_t)g * 2].data(), *pr = part[(size_t)g * 2 + 1].data();
        int rc;
        if (reuseChainOutputs) rc = gdg_spatialize_staged(ctx, 1, pl, pr, (int)n);
        else rc = gdg_spatialize(ctx, inputBuffers + first, pl, pr, (int)n);
        if (rc != GDG_OK) {
            engine_->setError(format("spatializer, shard %d: %s", g, gdg_last_error(ctx)));
            std::fill(pl, pl + n, 0.0);
            std::fill(pr, pr + n, 0.0);
        }
    };
    for (int g = 1; g < G; g++) workers.emplace_back(one, g);
    one(0);
    for (auto &w : workers) w.join();
    /* host-side sum of the partials in shard order, then the aux input (spatializer.go:300-310) */
    for (int g = 0; g < G; g++)
        for (size_t i = 0; i < n; i++) { left[i] += part[(size_t)g * 2][i]; right[i] += part[(size_t)g * 2 + 1][i]; }
    if (auxInputBuffer)
        for (size_t i = 0; i < n; i++) { left[i] += auxInputBuffer[i]; right[i] += auxInputBuffer[i]; }
}

std::shared_ptr<Spatializer> Create(Engine *engine, uint32_t inputChannels) { return std::make_shared<Spatializer>(engine, inputChannels); }

}  // namespace spatializer

/* ================================ tuner =================================================== */
namespace tuner {

static const size_t NUM_SAMPLES = 96000;                                           /* tuner.go:16 */

Tuner::Tuner(int device) : ring_(NUM_SAMPLES, 0.0), device_(device) {}
Tuner::~Tuner() { if (ctx_) gdg_ctx_destroy(ctx_); }

/* tuner.go:582-587: one enqueue into the host ring under mutexBuffer -- the audio path never waits for the GPU.
 * The enqueue is circular.Enqueue (circular.go:33-71): more samples than the ring holds keep the tail and reset the pointer. */
void Tuner::Process(const double *samples, size_t n, uint32_t sampleRate) {
    std::unique_lock<std::shared_mutex> lk(mutexBuffer_);
    const size_t N = ring_.size();
    if (n >= N) {
        memcpy(ring_.data(), samples + (n - N), N * sizeof(double));
        pointer_ = 0;
    } else {
        size_t first = std::min(n, N - pointer_);
        memcpy(ring_.data() + pointer_, samples, first * sizeof(double));
        memcpy(ring_.data(), samples + first, (n - first) * sizeof(double));
        pointer_ = (pointer_ + n) % N;
    }
    sampleRate_ = sampleRate;
}

/* tuner.go:379-577: mutexAnalyze for the whole analysis, mutexBuffer (shared) for the copy of the ring only; the copy then replaces the
 * device ring (NUM_SAMPLES enqueued samples) and the analysis runs there. */
std::pair<Result, Error> Tuner::Analyze() {
    std::lock_guard<std::mutex> lk(mutexAnalyze_);
    Result r{ 0, 0.0, "Unknown" };
    if (!ctx_ && gdg_ctx_create(1, GDG_HOST_MAX_FRAMES, device_, &ctx_) != GDG_OK) { ctx_ = nullptr; return { r, "no usable HIP device; there is no CPU fallback" }; }
    const size_t N = ring_.size();
    snapshot_.resize(N);
    uint32_t sampleRate;
    {
        std::shared_lock<std::shared_mutex> rd(mutexBuffer_);
        sampleRate = sampleRate_;
        memcpy(snapshot_.data(), ring_.data() + pointer_, (N - pointer_) * sizeof(double));      /* circular.Retrieve: oldest first */
        memcpy(snapshot_.data() + (N - pointer_), ring_.data(), pointer_ * sizeof(double));
    }
    for (size_t at = 0; at < N; at += GDG_HOST_MAX_FRAMES) {
        size_t m = std::min<size_t>(GDG_HOST_MAX_FRAMES, N - at);
        const double *rows[1] = { snapshot_.data() + at };
        if (gdg_tuner_enqueue(ctx_, rows, (int)m, sampleRate) != GDG_OK) return { r, std::string("Failed to analyze: ") + gdg_last_error(ctx_) };
    }
    gdg_tuner_result res;
    if (gdg_tuner_analyze(ctx_, &res) != GDG_OK) return { r, std::string("Failed to analyze: ") + gdg_last_error(ctx_) };
    r.cents = res.cents;
    r.frequency = res.frequency;
    r.note = gdg_tuner_note_name(res.note_index);
    return { r, "" };
}

std::shared_ptr<Tuner> Create(int device) { return std::make_shared<Tuner>(device); }

}  // namespace tuner

}  // namespace gdg

/* ================================ flat C API for the test-suite ============================= */

using namespace gdg;

static thread_local std::string t_err;
static const char *ret(const Error &e) {
    if (e.empty()) return nullptr;
    t_err = e;
    return t_err.c_str();
}

struct ChainHandle { std::shared_ptr<signal::Chain> chain; };

extern "C" {

void *gdgh_engine_create(int n_channels, int max_frames, int device) { return new Engine(n_channels, max_frames, device); }
void gdgh_engine_destroy(void *e) { delete (Engine *)e; }
void gdgh_engine_set_rendezvous(void *e, int expected, int timeout_ms) { ((Engine *)e)->SetRendezvous(expected, timeout_ms); }
const char *gdgh_engine_last_error(void *e) { t_err = ((Engine *)e)->LastError(); return t_err.c_str(); }
const char *gdgh_engine_process_all(void *e, const double *const *in, double *const *out, int frames, uint32_t sr) {
    return ret(((Engine *)e)->ProcessAll(in, out, frames, sr));
}

const char *gdgh_engine_batch_run(void *e, const gdg_batch_input *inputs, int n, const gdg_batch_options *opt, int window, void *const *outs, size_t *samples) {
    return ret(((Engine *)e)->BatchRun(inputs, n, *opt, window, outs, samples));
}
void *gdgh_engine_context(void *e, int shard) { return ((Engine *)e)->context(shard); }
void gdgh_engine_shard_range(void *e, int shard, int *first, int *count) { ((Engine *)e)->shardRange(shard, first, count); }
void *gdgh_engine_create_sharded(int n_channels, int max_frames, const int *devices, int n_devices) {
    return new Engine(n_channels, max_frames, std::vector<int>(devices, devices + n_devices));
}
int gdgh_engine_shards(void *e) { return ((Engine *)e)->shards(); }
int gdgh_engine_shard_of(void *e, int channel) { return ((Engine *)e)->shardOf(channel); }

/* spatializer.Spatializer */
void *gdgh_spatializer_create(void *engine, uint32_t input_channels) { return new std::shared_ptr<spatializer::Spatializer>(spatializer::Create((Engine *)engine, input_channels)); }
void gdgh_spatializer_destroy(void *sp) { delete (std::shared_ptr<spatializer::Spatializer> *)sp; }
#define SP(sp) (*(std::shared_ptr<spatializer::Spatializer> *)(sp))
const char *gdgh_spatializer_set(void *sp, int what, uint32_t channel, double value) {
    return ret(what == 0 ? SP(sp)->SetAzimuth(channel, value) : what == 1 ? SP(sp)->SetDistance(channel, value) : SP(sp)->SetLevel(channel, value));
}
const char *gdgh_spatializer_get(void *sp, int what, uint32_t channel, double *value) {
    auto r = what == 0 ? SP(sp)->GetAzimuth(channel) : what == 1 ? SP(sp)->GetDistance(channel) : SP(sp)->GetLevel(channel);
    *value = r.first;
    return ret(r.second);
}
uint32_t gdgh_spatializer_input_count(void *sp) { return SP(sp)->GetInputCount(); }
uint32_t gdgh_spatializer_output_count(void *sp) { return SP(sp)->GetOutputCount(); }
void gdgh_spatializer_set_sample_rate(void *sp, uint32_t rate) { SP(sp)->SetSampleRate(rate); }
void gdgh_spatializer_process(void *sp, const double *const *in, const double *aux, double *left, double *right, int n, int reuse) {
    double *outs[2] = { left, right };
    SP(sp)->Process(in, aux, outs, (size_t)n, reuse != 0);
}

/* tuner.Tuner */
void *gdgh_tuner_create(int device) { return new std::shared_ptr<tuner::Tuner>(tuner::Create(device)); }
void gdgh_tuner_destroy(void *t) { delete (std::shared_ptr<tuner::Tuner> *)t; }
void gdgh_tuner_process(void *t, const double *samples, int n, uint32_t sr) { (*(std::shared_ptr<tuner::Tuner> *)t)->Process(samples, (size_t)n, sr); }
const char *gdgh_tuner_analyze(void *t, int *cents, double *frequency, char *note, int cap) {
    auto r = (*(std::shared_ptr<tuner::Tuner> *)t)->Analyze();
    *cents = r.first.cents;
    *frequency = r.first.frequency;
    snprintf(note, (size_t)cap, "%s", r.first.note.c_str());
    return ret(r.second);
}

void *gdgh_irs_create(void) { return new filter::ImpulseResponses(); }
void gdgh_irs_destroy(void *irs) { delete (filter::ImpulseResponses *)irs; }
void gdgh_irs_add(void *irs, const char *name, uint32_t sample_rate, int32_t compensation_db, const double *taps, int n) {
    ((filter::ImpulseResponses *)irs)->Add(name, sample_rate, compensation_db, std::vector<double>(taps, taps + n));
}

void *gdgh_chain_create(void *engine, void *irs) {
    auto r = ((Engine *)engine)->CreateChain((filter::ImpulseResponses *)irs);
    if (!r.first) { t_err = r.second; return nullptr; }
    return new ChainHandle{ r.first };
}
void gdgh_chain_destroy(void *c) { delete (ChainHandle *)c; }
#define CH(c) (((ChainHandle *)(c))->chain)
const char *gdgh_chain_append_unit(void *c, int unit_type, int *id) { auto r = CH(c)->AppendUnit(unit_type); *id = r.first; return ret(r.second); }
const char *gdgh_chain_remove_unit(void *c, int id) { return ret(CH(c)->RemoveUnit(id)); }
const char *gdgh_chain_move_up(void *c, int id) { return ret(CH(c)->MoveUp(id)); }
const char *gdgh_chain_move_down(void *c, int id) { return ret(CH(c)->MoveDown(id)); }
const char *gdgh_chain_unit_type(void *c, int id, int *t) { auto r = CH(c)->UnitType(id); *t = r.first; return ret(r.second); }
const char *gdgh_chain_set_bypass(void *c, int id, int bypass) { return ret(CH(c)->SetBypass(id, bypass != 0)); }
const char *gdgh_chain_get_bypass(void *c, int id, int *bypass) { auto r = CH(c)->GetBypass(id); *bypass = r.first ? 1 : 0; return ret(r.second); }
const char *gdgh_chain_set_discrete(void *c, int id, const char *name, const char *value) { return ret(CH(c)->SetDiscreteValue(id, name, value)); }
const char *gdgh_chain_get_discrete(void *c, int id, const char *name, char *buf, int cap) {
    auto r = CH(c)->GetDiscreteValue(id, name);
    snprintf(buf, (size_t)cap, "%s", r.first.c_str());
    return ret(r.second);
}
const char *gdgh_chain_set_numeric(void *c, int id, const char *name, int32_t value) { return ret(CH(c)->SetNumericValue(id, name, value)); }
const char *gdgh_chain_get_numeric(void *c, int id, const char *name, int32_t *value) { auto r = CH(c)->GetNumericValue(id, name); *value = r.first; return ret(r.second); }
int gdgh_chain_length(void *c) { return CH(c)->Length(); }
/* parameters as lines "name|type|unit|min|max|numeric|index|v0;v1;..." */
const char *gdgh_chain_parameters(void *c, int id, char *buf, int cap) {
    auto r = CH(c)->Parameters(id);
    std::ostringstream os;
    for (auto &p : r.first) {
        os << p.Name << "|" << p.Type << "|" << p.PhysicalUnit << "|" << p.Minimum << "|" << p.Maximum << "|" << p.NumericValue << "|" << p.DiscreteValueIndex << "|";
        for (size_t i = 0; i < p.DiscreteValues.size(); i++) os << (i ? ";" : "") << p.DiscreteValues[i];
        os << "\n";
    }
    snprintf(buf, (size_t)cap, "%s", os.str().c_str());
    return ret(r.second);
}
void gdgh_chain_process(void *c, const double *in, int n_in, double *out, int n_out, uint32_t sr) { CH(c)->Process(in, (size_t)n_in, out, (size_t)n_out, sr); }

/* stand-alone effects.Unit */
void *gdgh_unit_create(int unit_type) { auto u = effects::CreateUnit(unit_type); return u ? new std::shared_ptr<effects::Unit>(u) : nullptr; }
void gdgh_unit_destroy(void *u) { delete (std::shared_ptr<effects::Unit> *)u; }
const char *gdgh_unit_set_numeric(void *u, const char *name, int32_t v) { return ret((*(std::shared_ptr<effects::Unit> *)u)->SetNumericValue(name, v)); }
const char *gdgh_unit_set_discrete(void *u, const char *name, const char *v) { return ret((*(std::shared_ptr<effects::Unit> *)u)->SetDiscreteValue(name, v)); }
void gdgh_unit_process(void *u, const double *in, double *out, int n, uint32_t sr) { (*(std::shared_ptr<effects::Unit> *)u)->Process(in, out, (size_t)n, sr); }

/* filter algebra (CPU-side set-up code, compared with the oracle in tests/test_host_mirror.py) */
int gdgh_filter_compile(const double *taps, int n, uint32_t sr, int32_t compensation_db, uint32_t order, int32_t level_db, double *out, int cap) {
    filter::ImpulseResponses irs;
    irs.Add("x", sr, compensation_db, std::vector<double>(taps, taps + n));
    auto f = irs.CreateFilter("x", sr);
    if (order > 0) f = f->Reduce(order);
    f = f->Normalize();
    f = f->Multiply(pow(10.0, 0.05 * (double)level_db));
    auto c = f->Coefficients();
    int m = (int)std::min<size_t>((size_t)cap, c.size());
    for (int i = 0; i < m; i++) out[i] = c[(size_t)i];
    return (int)c.size();
}

}  /* extern "C" */
