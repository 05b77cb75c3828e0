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

/* push parameters / taps of one unit to the device context it lives in */
static Error pushUnit(Unit &u) {
    gdg_ctx *ctx = u.backing.ctx;
    if (u.pushedParamVersion != u.paramVersion) {
        int32_t v[8];
        u.resolved(v);
        int n = std::min<int>(8, (int)u.Parameters().size());
        if (u.Type() == UNIT_POWERAMP) n = 1;
        for (int i = 0; i < n; i++)
            if (gdg_unit_set_param(ctx, u.backing.handle, i, v[i]) != GDG_OK) return gdg_last_error(ctx);
        u.pushedParamVersion = u.paramVersion;
    }
    if (u.Type() == UNIT_POWERAMP && u.pushedFirVersion != u.firVersion) {
        if (gdg_unit_set_fir(ctx, u.backing.handle, u.firTaps.data(), (int)u.firTaps.size()) != GDG_OK) return gdg_last_error(ctx);
        u.pushedFirVersion = u.firVersion;
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

void Engine::Configure(int nChannels, int maxFrames, int device) {
    std::lock_guard<std::mutex> lk(g_default_mu);
    delete g_default;
    g_default = new Engine(nChannels, maxFrames, device);
}

Engine &Engine::Default() {
    std::lock_guard<std::mutex> lk(g_default_mu);
    if (!g_default) {
        const char *ch = getenv("GDG_CHANNELS"), *dev = getenv("GDG_DEVICE");
        g_default = new Engine(ch ? atoi(ch) : 1, GDG_HOST_MAX_FRAMES, dev ? atoi(dev) : 0);
    }
    return *g_default;
}

Engine::Engine(int nChannels, int maxFrames, int device) : nChannels_(nChannels), maxFrames_(maxFrames), device_(device) {}

Engine::~Engine() {
    chains_.clear();
    if (ctx_) gdg_ctx_destroy(ctx_);
}

gdg_ctx *Engine::context() {
    if (!ctx_) {
        int rc = gdg_ctx_create(nChannels_, maxFrames_, device_, &ctx_);
        if (rc != GDG_OK) { ctx_ = nullptr; lastError_ = format("gdg_ctx_create failed with %d (no usable HIP device; there is no CPU fallback)", rc); }
    }
    return ctx_;
}

std::string Engine::LastError() const { return lastError_; }

std::pair<std::shared_ptr<signal::Chain>, Error> Engine::CreateChain(const filter::ImpulseResponses *responses) {
    std::lock_guard<std::mutex> lk(mu_);
    if ((int)chains_.size() >= nChannels_) return { nullptr, format("engine has only %d channels", nChannels_) };
    std::shared_ptr<signal::Chain> c(new signal::Chain(this, (int)chains_.size(), responses));
    chains_.push_back(c);
    if (expected_ < (int)chains_.size()) expected_ = (int)chains_.size();
    return { c, "" };
}

/* bring the device side of the listed chains up to date: units, parameters, taps, slot lists */
Error Engine::sync(const std::vector<signal::Chain *> &chains, uint32_t sampleRate) {
    gdg_ctx *ctx = context();
    if (!ctx) return lastError_;
    for (signal::Chain *ch : chains) {
        std::lock_guard<std::mutex> lk(ch->mutex_);
        for (auto &u : ch->retired_)
            if (u->backing.ctx == ctx && u->backing.handle >= 0) { gdg_unit_destroy(ctx, u->backing.handle); u->backing.handle = -1; u->backing.ctx = nullptr; }
        ch->retired_.clear();
        for (auto &s : ch->slots_) {
            effects::Unit &u = *s.unit;
            if (u.backing.handle < 0) {
                if (gdg_unit_create(ctx, ch->channel_, u.Type(), &u.backing.handle) != GDG_OK) return gdg_last_error(ctx);
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
            if (gdg_chain_set(ctx, ch->channel_, handles.data(), bypass.data(), (int)handles.size()) != GDG_OK) return gdg_last_error(ctx);
            ch->pushedLayoutVersion_ = ch->layoutVersion_;
        }
    }
    return "";
}

void Engine::runBatch(std::vector<Pending> batch) {
    /* one launch per distinct (frames, sample rate) among the deposited calls -- in practice exactly one */
    while (!batch.empty()) {
        int frames = batch[0].frames;
        uint32_t sr = batch[0].sampleRate;
        std::vector<Pending> group, rest;
        for (auto &p : batch) ((p.frames == frames && p.sampleRate == sr) ? group : rest).push_back(p);
        std::sort(group.begin(), group.end(), [](const Pending &a, const Pending &b) { return a.chain->channel() < b.chain->channel(); });
        std::vector<signal::Chain *> chains;
        std::vector<int> channels;
        std::vector<const double *> ins;
        std::vector<double *> outs;
        for (auto &p : group) { chains.push_back(p.chain); channels.push_back(p.chain->channel()); ins.push_back(p.in); outs.push_back(p.out); }
        Error e = sync(chains, sr);
        if (e.empty() && gdg_process_subset(ctx_, channels.data(), (int)channels.size(), ins.data(), outs.data(), frames, sr) != GDG_OK)
            e = gdg_last_error(ctx_);
        if (!e.empty()) {
            /* the reference's Process has no error return: failures produce zeros (effects/poweramp.go:210-214) */
            lastError_ = e;
            for (auto &p : group) memset(p.out, 0, sizeof(double) * (size_t)p.frames);
        }
        batch.swap(rest);
    }
}

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
    lastError_.clear();
    runBatch(batch);
    return lastError_;
}

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
