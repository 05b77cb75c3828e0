/*
 * effects.c -- oracle restatement of the 21 effects.Unit implementations
 * (effects/<unit>.go Process functions).  TEST INFRASTRUCTURE ONLY (see gdg_oracle.h).
 *
 * PARITY UNPINNED by the reference: effects/ has no tests upstream and no Go toolchain
 * exists in the build container.  Each Process below follows the Go statement order line by
 * line (floating-point evaluation order included) and is cross-checked against a second,
 * independently written formulation in tests/test_oracle_independent.py.
 *
 * Parameter slots follow the declaration order of each unit's create*() table; discrete
 * parameters are stored as the index into the reference's DiscreteValues list.
 */
#include "gdg_oracle.h"
#include "go_consts.h"
#include <math.h>
#include "libm_jitter.h"
#include <stdlib.h>
#include <string.h>

#define NUM_FILTERS 8                       /* effects/effects.go:62 */

struct gdgo_unit {
    int type;
    int n_params;
    int32_t params[GDGO_MAX_PARAMS];
    /* generic state shared by several units */
    double envelope;                        /* compressor, fuzz, octaver, autowah, autoyoy */
    double cap;                             /* fuzz / octaver coupling capacitor */
    double phase;                           /* chorus/flanger/phaser previousPhase, ringmod, signal generator */
    double hcv[NUM_FILTERS], lcv[NUM_FILTERS];
    int n_hcv;                              /* bandpass: current halfOrder */
    int caps_ready;                         /* tonestack / cabinet slices allocated */
    double *buffer; int buffer_n;           /* delay-type history buffer */
    /* oversampling units */
    gdgo_osd *os2, *os4;
    double *os_in, *os_out; int os_n;
    /* tremolo / noise gate */
    int flag;                               /* attenuated / gateOpen */
    uint32_t counter;                       /* inStateSince / onHoldSince */
    /* octaver */
    double prev_polarity; uint32_t octave_register;
    /* signal generator */
    gdgo_prng prng; int prng_ready;
    /* reverb */
    double *ap_buf[3]; int ap_n[3]; int ap_ptr[3];
    uint32_t dl_idx[4]; double *dl_buf; int dl_n;
    double *rv_front, *rv_back, *rv_dl; int rv_n;
    uint32_t sample_rate;                   /* reverb / poweramp */
    /* poweramp */
    gdgo_filter *fir;
    /* cabinet */
    double *cab_buf; int cab_n;
};

static const int g_param_count[GDGO_UNIT_COUNT] = {
    6, 3, 3, 5, 4, 3, 7, 3, 7, 6, 4, 4, 2, 2, 3, 3, 1, 3, 1, 1, 1
};

/* defaults, effects/<unit>.go create*() tables */
static const int32_t g_param_default[GDGO_UNIT_COUNT][GDGO_MAX_PARAMS] = {
    /* signal_generator */ { 100, 0, 0, 440, 100, 0 },
    /* noise_gate       */ { -20, -40, 50 },
    /* bandpass         */ { 0, 300, 3000 },
    /* auto_wah         */ { 1, -40, -10, 300, 6000 },
    /* auto_yoy         */ { 1, -40, -10, 100 },
    /* compressor       */ { 1, 30, -20 },
    /* octaver          */ { 1, -20, -20, -20, -20, -20, -20 },
    /* excess           */ { 0, 0, 0 },
    /* fuzz             */ { 1, 50, 0, 0, 100, 0, 0 },
    /* overdrive        */ { 0, 0, 100, 0, 1, 0 },
    /* distortion       */ { 0, 0, 0, 0 },
    /* tone_stack       */ { 0, -2, -5, -5 },
    /* chorus           */ { 100, 30 },
    /* flanger          */ { 100, 10 },
    /* phaser           */ { 100, 10, 45 },
    /* tremolo          */ { 100, 50, -10 },
    /* ring_modulator   */ { 100 },
    /* delay            */ { 200, -5, -5 },
    /* reverb           */ { 50 },
    /* power_amp        */ { 14 },
    /* cabinet          */ { 0 },
};

/* effects/effects.go:389-394 */
static double decibels_to_factor(int32_t decibels) {
    double exp_ = 0.05 * (double)decibels;
    return pow(10.0, exp_);
}

/* effects/effects.go:399-402 */
static double factor_to_decibels(double factor) { return 20.0 * log10(factor); }

/* effects/effects.go:424-438 */
static double sign_float(double v) { return v < 0.0 ? -1.0 : (v > 0.0 ? 1.0 : 0.0); }

gdgo_unit *gdgo_unit_create(int unit_type) {
    if (unit_type < 0 || unit_type >= GDGO_UNIT_COUNT) return NULL;
    gdgo_unit *u = (gdgo_unit *)calloc(1, sizeof(gdgo_unit));
    u->type = unit_type;
    u->n_params = g_param_count[unit_type];
    memcpy(u->params, g_param_default[unit_type], sizeof(u->params));
    if (unit_type == GDGO_UNIT_EXCESS || unit_type == GDGO_UNIT_FUZZ ||
        unit_type == GDGO_UNIT_OVERDRIVE || unit_type == GDGO_UNIT_DISTORTION) {
        u->os2 = gdgo_osd_create(2);
        u->os4 = gdgo_osd_create(4);
    }
    return u;
}

void gdgo_unit_destroy(gdgo_unit *u) {
    if (!u) return;
    free(u->buffer); free(u->os_in); free(u->os_out);
    gdgo_osd_destroy(u->os2); gdgo_osd_destroy(u->os4);
    for (int i = 0; i < 3; i++) free(u->ap_buf[i]);
    free(u->dl_buf); free(u->rv_front); free(u->rv_back); free(u->rv_dl);
    gdgo_filter_destroy(u->fir);
    free(u->cab_buf);
    free(u);
}

int gdgo_unit_type(const gdgo_unit *u) { return u->type; }
int gdgo_unit_param_count(const gdgo_unit *u) { return u->n_params; }

int gdgo_unit_set_param(gdgo_unit *u, int idx, int32_t value) {
    if (idx < 0 || idx >= u->n_params) return -1;
    u->params[idx] = value;
    return 0;
}

int32_t gdgo_unit_get_param(const gdgo_unit *u, int idx) {
    if (idx < 0 || idx >= u->n_params) return 0;
    return u->params[idx];
}

int gdgo_unit_set_fir(gdgo_unit *u, const double *taps, int n) {
    if (u->type != GDGO_UNIT_POWERAMP) return -1;
    gdgo_filter_destroy(u->fir);
    u->fir = gdgo_filter_from_coefficients(taps, n, u->sample_rate, 0.0);
    return 0;
}

/* history buffer update shared by chorus/flanger/phaser/autoyoy/delay (e.g. effects/chorus.go:119-131) */
static void history_update(double *buffer, int buffer_size, const double *in, int num_samples) {
    int boundary = buffer_size - num_samples;
    if (boundary >= 0) {
        memmove(buffer, buffer + num_samples, sizeof(double) * (size_t)boundary);
        memcpy(buffer + boundary, in, sizeof(double) * (size_t)num_samples);
    } else {
        memcpy(buffer, in + (-boundary), sizeof(double) * (size_t)buffer_size);
    }
}

static void ensure_history(gdgo_unit *u, int size) {
    if (u->buffer_n != size || u->buffer == NULL) {
        free(u->buffer);
        u->buffer = (double *)calloc((size_t)(size > 0 ? size : 1), sizeof(double));
        u->buffer_n = size;
    }
}

/* envelope follower shared by five units (e.g. effects/compressor.go:37-58); follow: 0 = "envelope", 1 = "level" */
static inline double follow_envelope(int follow, double envelope, double sample_abs, double discharge_inv, double discharge) {
    switch (follow) {
    case 0:
        envelope *= discharge_inv;
        if (sample_abs > envelope) envelope = sample_abs;
        break;
    case 1: {
        double diff = sample_abs - envelope;
        envelope += diff * discharge;
        break;
    }
    default:
        envelope = 1.0;
    }
    return envelope;
}

/* fractional delay read shared by the modulation units (e.g. effects/flanger.go:63-90) */
static inline double frac_delay(const double *in, const double *buffer, int buffer_size, int i, double delay_samples) {
    double early = floor(delay_samples), late = ceil(delay_samples);
    int idx_early = i - (int)early, idx_late = i - (int)late;
    double s_early = (idx_early >= 0) ? in[idx_early] : buffer[buffer_size + idx_early];
    double s_late = (idx_late >= 0) ? in[idx_late] : buffer[buffer_size + idx_late];
    double w_early = 1.0 - (delay_samples - early);
    double w_late = 1.0 - (late - delay_samples);
    return (w_early * s_early) + (w_late * s_late);
}

/* ---- signal generator: effects/signalgenerator.go:20-153 ------------------------------ */
static void process_signal_generator(gdgo_unit *u, const double *in, double *out, int n, uint32_t sample_rate) {
    int32_t input_amplitude = u->params[0], input_gain = u->params[1], signal_type = u->params[2];
    int32_t signal_frequency = u->params[3], signal_amplitude = u->params[4], signal_gain = u->params[5];
    double fac_input = (0.01 * (double)input_amplitude) * decibels_to_factor(input_gain);
    double fac_signal_gain = decibels_to_factor(signal_gain);
    double fac_signal = (0.01 * (double)signal_amplitude) * fac_signal_gain;
    double phase = u->phase;
    double phase_increment = GO_MATH_TWO_PI * ((double)signal_frequency / (double)sample_rate);
    double two_over_pi = GO_MATH_TWO_OVER_PI;
    double n_float = (double)n;
    if (signal_type == 4) {                     /* "noise" */
        if (!u->prng_ready) { gdgo_prng_init(&u->prng, 1337); u->prng_ready = 1; }
        for (int i = 0; i < n; i++) {
            double r = gdgo_prng_next_float(&u->prng);
            double uniform = (1.0 - (2.0 * r));
            out[i] = (fac_input * in[i]) + (fac_signal * uniform);
        }
    } else if (signal_type >= 0 && signal_type <= 3) {
        for (int i = 0; i < n; i++) {
            double updated = phase + ((double)i * phase_increment);
            double cur = fmod(updated, GO_MATH_TWO_PI);
            double signal = 0.0;
            switch (signal_type) {
            case 0: signal = sin(cur); break;
            case 1: signal = (cur < M_PI) ? (two_over_pi * cur) - 1.0 : 3.0 - (two_over_pi * cur); break;
            case 2: signal = sign_float(M_PI - cur); break;
            case 3: signal = cur / M_PI; if (cur > M_PI) signal -= 2.0; break;
            }
            out[i] = (fac_input * in[i]) + (fac_signal * signal);
        }
        phase += n_float * phase_increment;
        phase = fmod(phase, GO_MATH_TWO_PI);
    }
    u->phase = phase;
}

/* ---- noise gate: effects/noisegate.go:19-96 ------------------------------------------- */
static void process_noise_gate(gdgo_unit *u, const double *in, double *out, int n, uint32_t sample_rate) {
    int32_t level_open = u->params[0], level_close = u->params[1], hold_time = u->params[2];
    double fac_open = decibels_to_factor(level_open), fac_close = decibels_to_factor(level_close);
    if (level_open < level_close) {
        memcpy(out, in, sizeof(double) * (size_t)n);
        u->flag = 1;
        u->counter = 0;
        return;
    }
    double hold_seconds = 0.001 * (double)hold_time;
    double hold_samples_f = floor((hold_seconds * (double)sample_rate) + 0.5);
    uint32_t hold_samples = (uint32_t)hold_samples_f;
    int gate_open = u->flag;
    uint32_t on_hold_since = u->counter;
    for (int i = 0; i < n; i++) {
        double sample = in[i];
        double amplitude = fabs(sample);
        if (amplitude > fac_open) gate_open = 1;
        if (amplitude > fac_close) on_hold_since = 0;
        if (on_hold_since >= hold_samples) gate_open = 0;
        double fac = gate_open ? 1.0 : 0.0;
        out[i] = fac * sample;
        if (on_hold_since < UINT32_MAX) on_hold_since++;
    }
    u->flag = gate_open;
    u->counter = on_hold_since;
}

/* ---- bandpass: effects/bandpass.go:20-98 ---------------------------------------------- */
static void process_bandpass(gdgo_unit *u, const double *in, double *out, int n, uint32_t sample_rate) {
    static const int orders[4] = { 2, 4, 6, 8 };
    int32_t oi = u->params[0];
    int half_order = (oi >= 0 && oi < 4) ? orders[oi] >> 1 : 0;
    int32_t freq_a = u->params[1], freq_b = u->params[2];
    if (freq_a > freq_b) { int32_t t = freq_a; freq_a = freq_b; freq_b = t; }
    if (u->n_hcv != half_order) {
        memset(u->hcv, 0, sizeof(u->hcv));
        memset(u->lcv, 0, sizeof(u->lcv));
        u->n_hcv = half_order;
    }
    double m2pi_sr = -GO_MATH_TWO_PI / (double)sample_rate;
    double d_hp_inv = 1.0 - exp(m2pi_sr * (double)freq_a);
    double d_lp_inv = 1.0 - exp(m2pi_sr * (double)freq_b);
    for (int i = 0; i < n; i++) {
        double pre = in[i];
        for (int j = 0; j < half_order; j++) {
            double hcv = u->hcv[j];
            double diff = pre - hcv;
            hcv += diff * d_hp_inv;
            u->hcv[j] = hcv;
            double lcv = u->lcv[j];
            diff -= lcv;
            double iv = lcv;
            lcv += diff * d_lp_inv;
            u->lcv[j] = lcv;
            if (iv < -1.0) pre = -1.0; else if (iv > 1.0) pre = 1.0; else pre = iv;
        }
        out[i] = pre;
    }
}

/* ---- auto-wah: effects/autowah.go:20-130 ---------------------------------------------- */
static void process_autowah(gdgo_unit *u, const double *in, double *out, int n, uint32_t sample_rate) {
    int32_t follow = u->params[0], level_a = u->params[1], level_b = u->params[2];
    int32_t freq_a = u->params[3], freq_b = u->params[4];
    if (level_a > level_b) {
        int32_t t = level_a; level_a = level_b; level_b = t;
        t = freq_a; freq_a = freq_b; freq_b = t;
    }
    double la = (double)level_a, lb = (double)level_b, fa = (double)freq_a, fb = (double)freq_b;
    double slope = (fb - fa) / (lb - la);
    double sr = (double)sample_rate;
    double d_env_inv = exp(-20.0 / sr);
    double d_env = 1.0 - d_env_inv;
    double envelope = u->envelope;
    double gain_comp = pow(2.0, NUM_FILTERS);
    for (int i = 0; i < n; i++) {
        double sample = in[i];
        envelope = follow_envelope(follow, envelope, fabs(sample), d_env_inv, d_env);
        double level = factor_to_decibels(envelope);
        double frequency;
        if (level <= la) frequency = fa;
        else if (level >= lb) frequency = fb;
        else frequency = fa + (slope * (level - la));
        double arg = -frequency / sr;
        double d_inv = 1.0 - exp(arg);
        double lcv = sample;
        for (int j = 0; j < NUM_FILTERS; j++) {
            double hcv = u->hcv[j];
            double diff = lcv - hcv;
            hcv += diff * d_inv;
            u->hcv[j] = hcv;
            lcv = u->lcv[j];
            diff -= lcv;
            lcv += diff * d_inv;
            u->lcv[j] = lcv;
        }
        double pre = gain_comp * lcv;
        if (pre < -1.0) pre = -1.0; else if (pre > 1.0) pre = 1.0;
        out[i] = pre;
    }
    u->envelope = envelope;
}

/* ---- auto-yoy: effects/autoyoy.go:19-157 ---------------------------------------------- */
static void process_autoyoy(gdgo_unit *u, const double *in, double *out, int n, uint32_t sample_rate) {
    int32_t follow = u->params[0], level_a = u->params[1], level_b = u->params[2], depth = u->params[3];
    double depth_a = 0.0, depth_b = 0.01 * (double)depth;
    if (level_a > level_b) {
        int32_t t = level_a; level_a = level_b; level_b = t;
        double d = depth_a; depth_a = depth_b; depth_b = d;
    }
    double la = (double)level_a, lb = (double)level_b;
    double slope = (depth_b - depth_a) / (lb - la);
    double sr = (double)sample_rate;
    double sr_inv = 1.0 / sr;
    double d_env_inv = exp(-20.0 * sr_inv);
    double d_env = 1.0 - d_env_inv;
    int max_delay = (int)floor((0.01 * sr) + 0.5);
    ensure_history(u, max_delay);
    double envelope = u->envelope;
    for (int i = 0; i < n; i++) {
        double sample = in[i];
        envelope = follow_envelope(follow, envelope, fabs(sample), d_env_inv, d_env);
        double level = factor_to_decibels(envelope);
        double delay_fac;
        if (level <= la) delay_fac = depth_a;
        else if (level >= lb) delay_fac = depth_b;
        else delay_fac = depth_a + (slope * (level - la));
        double delay_time = 0.01 * delay_fac;
        double delay_samples = delay_time * sr;
        double delayed = frac_delay(in, u->buffer, max_delay, i, delay_samples);
        out[i] = (0.5 * sample) + (0.5 * delayed);
    }
    u->envelope = envelope;
    history_update(u->buffer, max_delay, in, n);
}

/* ---- compressor: effects/compressor.go:18-84 ------------------------------------------ */
static void process_compressor(gdgo_unit *u, const double *in, double *out, int n, uint32_t sample_rate) {
    int32_t follow = u->params[0];
    double gain_limit = decibels_to_factor(u->params[1]);
    double target = decibels_to_factor(u->params[2]);
    double d_inv = exp(-20.0 / (double)sample_rate);
    double d = 1.0 - d_inv;
    double envelope = u->envelope;
    for (int i = 0; i < n; i++) {
        double sample = in[i];
        envelope = follow_envelope(follow, envelope, fabs(sample), d_inv, d);
        double gain = target / envelope;
        if (gain > gain_limit) gain = gain_limit;
        double pre = gain * sample;
        if (pre < -1.0) pre = -1.0; else if (pre > 1.0) pre = 1.0;
        out[i] = pre;
    }
    u->envelope = envelope;
}

/* ---- octaver: effects/octaver.go:21-139 ----------------------------------------------- */
static void process_octaver(gdgo_unit *u, const double *in, double *out, int n, uint32_t sample_rate) {
    int32_t follow = u->params[0];
    double fac_up = decibels_to_factor(u->params[1]);
    double fac_clean = decibels_to_factor(u->params[2]);
    double fac_dist = decibels_to_factor(u->params[3]);
    double fac_d1 = decibels_to_factor(u->params[4]);
    double fac_d2 = decibels_to_factor(u->params[5]);
    double fac_hyst = decibels_to_factor(u->params[6]);
    double prev_polarity = u->prev_polarity;
    uint32_t reg = u->octave_register;
    double envelope = u->envelope, cap = u->cap;
    double d_inv = exp(-20.0 / (double)sample_rate);
    double d = 1.0 - d_inv;
    for (int i = 0; i < n; i++) {
        double sample = in[i];
        double sample_abs = fabs(sample);
        envelope = follow_envelope(follow, envelope, sample_abs, d_inv, d);
        double square = sample * sample;
        double sign = sign_float(sample);
        double hysteresis = envelope * fac_hyst;
        if ((sign != 0.0) && (sign != prev_polarity) && (sample_abs > hysteresis)) {
            reg = (reg + 1) & 0x7;
            prev_polarity = sign;
        }
        double first_down = (reg & 0x2) ? -1.0 : 1.0;
        double second_down = (reg & 0x4) ? -1.0 : 1.0;
        double pre = fac_clean * sample;
        if (envelope > 0.0001) pre += fac_up * (square / envelope);
        pre += fac_dist * (sign * envelope);
        pre += fac_d1 * (first_down * envelope);
        pre += fac_d2 * (second_down * envelope);
        cap += (pre - cap) * d;
        pre -= cap;
        if (pre < -1.0) out[i] = -1.0; else if (pre > 1.0) out[i] = 1.0; else out[i] = pre;
    }
    u->prev_polarity = prev_polarity;
    u->octave_register = reg;
    u->envelope = envelope;
    u->cap = cap;
}

/* ---- waveshapers (inner loops at the possibly oversampled rate) ------------------------ */

/* effects/excess.go:22-66 */
static void excess_inner(gdgo_unit *u, const double *in, double *out, int n, uint32_t sample_rate) {
    (void)sample_rate;
    double gain = decibels_to_factor(u->params[0]), level = decibels_to_factor(u->params[1]);
    for (int i = 0; i < n; i++) {
        double pre = gain * in[i];
        double abs_pre = fabs(pre);
        int exceeded = abs_pre > 1.0;
        int negative = pre < 0.0;
        double abs_pre_biased = abs_pre + 1.0;
        double floor_ = floor(abs_pre_biased);
        int32_t section = (int32_t)(0.5 * floor_);
        int section_odd = (section % 2) != 0;
        int inverted = section_odd != (exceeded && negative);
        double excess = fmod(abs_pre + 1.0, 2.0);
        if (exceeded) pre = inverted ? 1.0 - excess : excess - 1.0;
        out[i] = level * pre;
    }
}

/* effects/fuzz.go:24-108 */
static void fuzz_inner(gdgo_unit *u, const double *in, double *out, int n, uint32_t sample_rate) {
    int32_t follow = u->params[0];
    double bias = 0.01 * (double)u->params[1];
    double gain = decibels_to_factor(u->params[2] + u->params[3]);
    double fuzz = 0.01 * (double)u->params[4];
    double fuzz_inv = 1.0 - fuzz;
    double level = decibels_to_factor(u->params[5]);
    double envelope = u->envelope, cap = u->cap;
    double d_inv = exp(-20.0 / (double)sample_rate);
    double d = 1.0 - d_inv;
    for (int i = 0; i < n; i++) {
        double sample = in[i];
        envelope = follow_envelope(follow, envelope, fabs(sample), d_inv, d);
        double bias_voltage = bias * envelope;
        double pre = gain * (sample - bias_voltage);
        if (pre < -1.0) pre = -1.0; else if (pre > 1.0) pre = 1.0;
        double fuzz_fraction = fuzz * pre;
        double clean_fraction = fuzz_inv * sample;
        pre = fuzz_fraction + clean_fraction;
        double diff = pre - cap;
        cap += diff * d;
        pre -= cap;
        if (pre < -1.0) pre = -1.0; else if (pre > 1.0) pre = 1.0;
        out[i] = level * pre;
    }
    u->envelope = envelope;
    u->cap = cap;
}

/* effects/overdrive.go:28-78 */
static void overdrive_inner(gdgo_unit *u, const double *in, double *out, int n, uint32_t sample_rate) {
    (void)sample_rate;
    double gain = decibels_to_factor(u->params[0] + u->params[1]);
    double drive = 0.01 * (double)u->params[2];
    double clean = 1.0 - drive;
    double level = decibels_to_factor(u->params[3]);
    int32_t valve = u->params[4];               /* 0 = ECC82, 1 = ECC83, anything else = invalid */
    for (int i = 0; i < n; i++) {
        double sample = in[i];
        double arg = gain * sample;
        double dist = 0.0;
        if (valve == 0) {
            double aarg = GO_MATH_QUARTER_PI * arg;
            double x = atan(aarg);
            dist = GO_MATH_TWO_OVER_PI * x;
        } else if (valve == 1) {
            double x = exp(-arg);
            dist = (2.0 / (1.0 + x)) - 1.0;
        }
        double mix = (drive * dist) + (clean * sample);
        out[i] = level * mix;
    }
}

/* effects/distortion.go:21-49 */
static void distortion_inner(gdgo_unit *u, const double *in, double *out, int n, uint32_t sample_rate) {
    (void)sample_rate;
    double gain = decibels_to_factor(u->params[0] + u->params[1]);
    double level = decibels_to_factor(u->params[2]);
    for (int i = 0; i < n; i++) {
        double pre = gain * in[i];
        if (pre < -1.0) pre = -1.0; else if (pre > 1.0) pre = 1.0;
        out[i] = level * pre;
    }
}

typedef void (*inner_fn)(gdgo_unit *, const double *, double *, int, uint32_t);

/* the oversampling wrapper common to excess/fuzz/overdrive/distortion (e.g. effects/overdrive.go:83-144) */
static void process_oversampled(gdgo_unit *u, inner_fn inner, int os_param, const double *in, double *out, int n, uint32_t sample_rate) {
    int32_t os = u->params[os_param];           /* 0 = "- NONE -", 1 = "2", 2 = "4" */
    int factor = (os == 1) ? 2 : (os == 2) ? 4 : 1;
    if (factor > 1) {
        int num = factor * n;
        if (u->os_n != num) {
            free(u->os_in); free(u->os_out);
            u->os_in = (double *)calloc((size_t)(num > 0 ? num : 1), sizeof(double));
            u->os_out = (double *)calloc((size_t)(num > 0 ? num : 1), sizeof(double));
            u->os_n = num;
        }
        gdgo_osd *o = (factor == 4) ? u->os4 : u->os2;
        gdgo_osd_oversample(o, in, n, u->os_in, num);
        inner(u, u->os_in, u->os_out, num, (uint32_t)factor * sample_rate);
        gdgo_osd_decimate(o, u->os_out, num, out, n);
    } else {
        inner(u, in, out, n, sample_rate);
    }
}

/* ---- tone stack: effects/tonestack.go:19-100 ------------------------------------------ */
static void process_tonestack(gdgo_unit *u, const double *in, double *out, int n, uint32_t sample_rate) {
    static const double frequencies[5] = { 20.0, 300.0, 3000.0, 6000.0, 20000.0 };
    double facs[4];
    for (int i = 0; i < 4; i++) facs[i] = decibels_to_factor(u->params[i]);
    double m2pi_sr = -GO_MATH_TWO_PI / (double)sample_rate;
    for (int i = 0; i < n; i++) {
        double sample = in[i];
        double sum = 0.0;
        for (int j = 0; j < 4; j++) {
            double hcv = u->hcv[j], lcv = u->lcv[j];
            double d_hp_inv = 1.0 - exp(m2pi_sr * frequencies[j]);     /* 8 exp() per sample, as upstream */
            double d_lp_inv = 1.0 - exp(m2pi_sr * frequencies[j + 1]);
            double diff = sample - hcv;
            hcv += diff * d_hp_inv;
            diff -= lcv;
            double pre = lcv;
            lcv += diff * d_lp_inv;
            u->hcv[j] = hcv;
            u->lcv[j] = lcv;
            sum += facs[j] * pre;
        }
        if (sum < -1.0) out[i] = -1.0; else if (sum > 1.0) out[i] = 1.0; else out[i] = sum;
    }
}

/* ---- chorus: effects/chorus.go:19-131 -------------------------------------------------- */
static void process_chorus(gdgo_unit *u, const double *in, double *out, int n, uint32_t sample_rate) {
    double depth = 0.1 * (double)u->params[0];
    if (depth < 0.0) depth = 0.0; else if (depth > 10.0) depth = 10.0;
    double angular_speed = GO_MATH_PI_THOUSANDTH * (double)u->params[1];
    double sr = (double)sample_rate;
    int max_delay = (int)floor((0.05 * sr) + 0.5);
    double previous_phase = u->phase;
    ensure_history(u, max_delay);
    int buffer_size = max_delay;
    for (int i = 0; i < n; i++) {
        double time = (double)i / sr;
        double phase_changed = previous_phase + (angular_speed * time);
        double zero_phase = fmod(phase_changed, GO_MATH_TWO_PI);
        double effected = 0.0;
        for (int j = 0; j < 5; j++) {
            double phase_offset = GO_MATH_TWO_PI_FIFTH * (double)j;
            double phase = fmod(zero_phase + phase_offset, GO_MATH_TWO_PI);
            double offset = depth * sin(phase);
            double delay_time = 0.001 * (40.0 + offset);
            double delay_samples = delay_time * sr;
            effected += 0.2 * frac_delay(in, u->buffer, buffer_size, i, delay_samples);
        }
        out[i] = (0.5 * in[i]) + (0.5 * effected);
    }
    double buffer_time = (double)buffer_size / sr;        /* quirk: advances by the buffer length, not by n */
    u->phase = fmod(previous_phase + (angular_speed * buffer_time), GO_MATH_TWO_PI);
    history_update(u->buffer, buffer_size, in, n);
}

/* ---- flanger / phaser: effects/flanger.go:19-119, effects/phaser.go:19-125 ------------- */
static void process_flanger_phaser(gdgo_unit *u, int is_phaser, const double *in, double *out, int n, uint32_t sample_rate) {
    double depth = 0.01 * (double)u->params[0];
    if (depth < 0.0) depth = 0.0; else if (depth > 1.0) depth = 1.0;
    double angular_speed = GO_MATH_TWO_PI_HUNDREDTH * (double)u->params[1];
    double mix_dry = 0.5, mix_wet = 0.5;
    if (is_phaser) {
        double radians = GO_MATH_DEGREE_TO_RADIANS * (double)u->params[2];
        mix_wet = 0.5 * sin(radians);
        mix_dry = 1.0 - fabs(mix_wet);
    }
    double sr = (double)sample_rate;
    double sr_inv = 1.0 / sr;
    int max_delay = (int)floor((0.002 * sr) + 0.5);
    double previous_phase = u->phase;
    ensure_history(u, max_delay);
    int buffer_size = max_delay;
    for (int i = 0; i < n; i++) {
        double time = (double)i * sr_inv;
        double phase = fmod(previous_phase + (angular_speed * time), GO_MATH_TWO_PI);
        double offset = depth * sin(phase);
        double delay_time = 0.001 * (depth + offset);
        double delay_samples = delay_time * sr;
        double delayed = frac_delay(in, u->buffer, buffer_size, i, delay_samples);
        out[i] = (mix_dry * in[i]) + (mix_wet * delayed);
    }
    double duration = (double)buffer_size * sr_inv;
    u->phase = fmod(previous_phase + (angular_speed * duration), GO_MATH_TWO_PI);
    history_update(u->buffer, buffer_size, in, n);
}

/* ---- tremolo: effects/tremolo.go:15-65 ------------------------------------------------- */
static void process_tremolo(gdgo_unit *u, const double *in, double *out, int n, uint32_t sample_rate) {
    double frequency = 0.1 * (double)u->params[0];
    double period_f = (double)sample_rate / frequency;
    uint32_t period = (uint32_t)period_f;
    double phase = 0.01 * (double)u->params[1];
    uint32_t unatt = (uint32_t)(period_f * phase);
    uint32_t att = period - unatt;
    double fac = decibels_to_factor(u->params[2]);
    int attenuated = u->flag;
    uint32_t since = u->counter;
    for (int i = 0; i < n; i++) {
        double result = in[i];
        if (attenuated && (since >= att)) { attenuated = 0; since = 0; }
        else if (!attenuated && (since >= unatt)) { attenuated = 1; since = 0; }
        if (attenuated) result *= fac;
        out[i] = result;
        since++;
    }
    u->flag = attenuated;
    u->counter = since;
}

/* ---- ring modulator: effects/ringmodulator.go:18-45 ------------------------------------ */
static void process_ringmod(gdgo_unit *u, const double *in, double *out, int n, uint32_t sample_rate) {
    double phase = u->phase;
    double angular = GO_MATH_TWO_PI * (double)u->params[0];
    double fraction = angular / (double)sample_rate;
    for (int i = 0; i < n; i++) {
        double cur = fmod(phase + ((double)i * fraction), GO_MATH_TWO_PI);
        out[i] = sin(cur) * in[i];
    }
    u->phase = fmod(phase + ((double)n * fraction), GO_MATH_TWO_PI);
}

/* ---- delay: effects/delay.go:18-89 ----------------------------------------------------- */
static void process_delay(gdgo_unit *u, const double *in, double *out, int n, uint32_t sample_rate) {
    double seconds = 0.001 * (double)u->params[0];
    int delay_samples = (int)floor((seconds * (double)sample_rate) + 0.5);
    double feedback = decibels_to_factor(u->params[1]);
    double level = decibels_to_factor(u->params[2]);
    ensure_history(u, delay_samples);
    int buffer_size = delay_samples;
    for (int i = 0; i < n; i++) {
        int idx = i - delay_samples;
        double delayed = (idx >= 0) ? in[idx] : u->buffer[buffer_size + idx];
        double pre = level * (in[i] + (feedback * delayed));
        if (pre < -1.0) out[i] = -1.0; else if (pre > 1.0) out[i] = 1.0; else out[i] = pre;
    }
    history_update(u->buffer, buffer_size, in, n);
}

/* ---- reverb: effects/reverb.go:41-116, :179-338 ----------------------------------------- */
static void reverb_allpass(gdgo_unit *u, int k, const double *in, double *out, int n) {
    double *buf = u->ap_buf[k];
    int size = u->ap_n[k], ptr_write = u->ap_ptr[k];
    const double feedback = 0.7;
    for (int i = 0; i < n; i++) {
        int ptr_read = (ptr_write + 1) % size;
        double delayed = buf[ptr_read];
        double pre = in[i] - (feedback * delayed);
        buf[ptr_write] = pre;
        out[i] = (feedback * pre) + delayed;
        ptr_write = ptr_read;
    }
    u->ap_ptr[k] = ptr_write;
}

static void process_reverb(gdgo_unit *u, const double *in, double *out, int n, uint32_t sample_rate) {
    static const double ap_delays[3] = { 0.04204, 0.01348, 0.00452 };
    static const double tap_times[4] = { 0.19196, 0.19996, 0.21596, 0.23204 };
    static const double tap_coeffs[4] = { 0.1855, 0.18325, 0.17875, 0.17425 };
    double wet = 0.01 * (double)u->params[0];
    double dry = 1.0 - wet;
    double sr = (double)sample_rate;
    if (u->sample_rate != sample_rate) {
        for (int i = 0; i < 3; i++) {
            int ds = (int)round(ap_delays[i] * sr);
            free(u->ap_buf[i]);
            u->ap_buf[i] = (double *)calloc((size_t)(ds > 0 ? ds : 1), sizeof(double));
            u->ap_n[i] = ds;
            u->ap_ptr[i] = 0;
        }
        uint32_t max_index = 0;
        for (int i = 0; i < 4; i++) {
            u->dl_idx[i] = (uint32_t)round(tap_times[i] * sr);
            if (u->dl_idx[i] > max_index) max_index = u->dl_idx[i];
        }
        free(u->dl_buf);
        u->dl_buf = (double *)calloc((size_t)(max_index > 0 ? max_index : 1), sizeof(double));
        u->dl_n = (int)max_index;
        u->sample_rate = sample_rate;
    }
    if (u->rv_n != n) {
        free(u->rv_front); free(u->rv_back); free(u->rv_dl);
        u->rv_front = (double *)calloc((size_t)(n > 0 ? n : 1), sizeof(double));
        u->rv_back = (double *)calloc((size_t)(n > 0 ? n : 1), sizeof(double));
        u->rv_dl = (double *)calloc((size_t)(n > 0 ? n : 1), sizeof(double));
        u->rv_n = n;
    }
    /* tapped delay line, reverb.go:65-116 */
    int buffer_size = u->dl_n;
    for (int i = 0; i < n; i++) {
        double pre = 0.0;
        for (int j = 0; j < 4; j++) {
            int idx = i - (int)u->dl_idx[j];
            double cur = 0.0;
            if (idx >= 0) cur = in[idx];
            else if (idx >= -buffer_size) cur = u->dl_buf[buffer_size + idx];
            pre += tap_coeffs[j] * cur;
        }
        u->rv_dl[i] = pre;
    }
    history_update(u->dl_buf, buffer_size, in, n);
    memcpy(u->rv_front, u->rv_dl, sizeof(double) * (size_t)n);
    double *front = u->rv_front, *back = u->rv_back;
    for (int k = 0; k < 3; k++) {
        reverb_allpass(u, k, front, back, n);
        double *t = back; back = front; front = t;
    }
    double half_wet = 0.5 * wet;
    for (int i = 0; i < n; i++) {
        double sum = u->rv_dl[i] + front[i];
        double pre = (dry * in[i]) + (half_wet * sum);
        if (pre < -1.0) out[i] = -1.0; else if (pre > 1.0) out[i] = 1.0; else out[i] = pre;
    }
    u->rv_front = front;
    u->rv_back = back;
}

/* ---- power amp: effects/poweramp.go:186-216 (compile stays with the caller) -------------- */
static void process_poweramp(gdgo_unit *u, const double *in, double *out, int n, uint32_t sample_rate) {
    if (sample_rate != u->sample_rate) {
        /* upstream recompiles here, which replaces the filter and thereby resets its state */
        u->sample_rate = sample_rate;
        if (u->fir) {
            gdgo_filter *fresh = gdgo_filter_from_coefficients(gdgo_filter_coefficients(u->fir), gdgo_filter_length(u->fir), sample_rate, 0.0);
            gdgo_filter_destroy(u->fir);
            u->fir = fresh;
        }
    }
    if (u->fir) gdgo_filter_process(u->fir, in, out, n);
    else for (int i = 0; i < n; i++) out[i] = 0.0;
}

/* ---- cabinet (IIR): effects/cabinet.go:27-162 ------------------------------------------- */
static void process_cabinet(gdgo_unit *u, const double *in, double *out, int n, uint32_t sample_rate) {
    static const double hp_freqs[3] = { 300.0, 120.0, 80.0 };
    static const double lp_freqs[4] = { 3000.0, 4000.0, 5000.0, 6000.0 };
    if (u->cab_n != n) {
        free(u->cab_buf);
        u->cab_buf = (double *)calloc((size_t)(n > 0 ? n : 1), sizeof(double));
        u->cab_n = n;
    }
    double *buffer = u->cab_buf;
    memcpy(buffer, in, sizeof(double) * (size_t)n);
    double m2pi_sr = -GO_MATH_TWO_PI / (double)sample_rate;
    for (int i = 0; i < 3; i++) {
        double hcv = u->hcv[i];
        double d_inv = 1.0 - exp(m2pi_sr * hp_freqs[i]);
        for (int j = 0; j < n; j++) {
            double diff = buffer[j] - hcv;
            buffer[j] = diff;
            hcv += diff * d_inv;
        }
        u->hcv[i] = hcv;
    }
    for (int i = 0; i < 4; i++) {
        double lcv = u->lcv[i];
        double d_inv = 1.0 - exp(m2pi_sr * lp_freqs[i]);
        for (int j = 0; j < n; j++) {
            double diff = buffer[j] - lcv;
            buffer[j] = lcv;
            lcv += diff * d_inv;
        }
        u->lcv[i] = lcv;
    }
    for (int i = 0; i < n; i++) {
        double pre = buffer[i];
        if (pre < -1.0) pre = -1.0; else if (pre > 1.0) pre = 1.0;
        out[i] = pre;
    }
}

void gdgo_unit_process(gdgo_unit *u, const double *in, double *out, int n, uint32_t sample_rate) {
    switch (u->type) {
    case GDGO_UNIT_SIGNALGENERATOR: process_signal_generator(u, in, out, n, sample_rate); break;
    case GDGO_UNIT_NOISEGATE: process_noise_gate(u, in, out, n, sample_rate); break;
    case GDGO_UNIT_BANDPASS: process_bandpass(u, in, out, n, sample_rate); break;
    case GDGO_UNIT_AUTOWAH: process_autowah(u, in, out, n, sample_rate); break;
    case GDGO_UNIT_AUTOYOY: process_autoyoy(u, in, out, n, sample_rate); break;
    case GDGO_UNIT_COMPRESSOR: process_compressor(u, in, out, n, sample_rate); break;
    case GDGO_UNIT_OCTAVER: process_octaver(u, in, out, n, sample_rate); break;
    case GDGO_UNIT_EXCESS: process_oversampled(u, excess_inner, 2, in, out, n, sample_rate); break;
    case GDGO_UNIT_FUZZ: process_oversampled(u, fuzz_inner, 6, in, out, n, sample_rate); break;
    case GDGO_UNIT_OVERDRIVE: process_oversampled(u, overdrive_inner, 5, in, out, n, sample_rate); break;
    case GDGO_UNIT_DISTORTION: process_oversampled(u, distortion_inner, 3, in, out, n, sample_rate); break;
    case GDGO_UNIT_TONESTACK: process_tonestack(u, in, out, n, sample_rate); break;
    case GDGO_UNIT_CHORUS: process_chorus(u, in, out, n, sample_rate); break;
    case GDGO_UNIT_FLANGER: process_flanger_phaser(u, 0, in, out, n, sample_rate); break;
    case GDGO_UNIT_PHASER: process_flanger_phaser(u, 1, in, out, n, sample_rate); break;
    case GDGO_UNIT_TREMOLO: process_tremolo(u, in, out, n, sample_rate); break;
    case GDGO_UNIT_RINGMODULATOR: process_ringmod(u, in, out, n, sample_rate); break;
    case GDGO_UNIT_DELAY: process_delay(u, in, out, n, sample_rate); break;
    case GDGO_UNIT_REVERB: process_reverb(u, in, out, n, sample_rate); break;
    case GDGO_UNIT_POWERAMP: process_poweramp(u, in, out, n, sample_rate); break;
    case GDGO_UNIT_CABINET: process_cabinet(u, in, out, n, sample_rate); break;
    default: break;
    }
}
