/*
 * io.c -- oracle restatements of the data formats either side of the hot path:
 *   - wave/wave.go:275-735 sample codecs (LPCM 8/16/24/32, IEEE 32/64), mono data section only;
 *   - level/level.go:147-210 peak-programme meter (current value + held peak).
 * TEST INFRASTRUCTURE ONLY (see gdg_oracle.h).  Pinned by wave/wave_test.go (byte-exact data sections of the
 * twelve import/export tests, tests/golden/wave.json) and level/level_test.go (known dB readings).
 */
#include "gdg_oracle.h"
#include <math.h>
#include "libm_jitter.h"
#include <stdlib.h>
#include <string.h>

#define MAX_INT24 0x007fffff
#define MIN_INT24 (-(MAX_INT24 + 1))
#define SIGN_BIT_INT24 0x00800000

int gdgo_wave_bytes_per_sample(int fmt) {
    switch (fmt) {
    case GDGO_FMT_LPCM8: return 1;
    case GDGO_FMT_LPCM16: return 2;
    case GDGO_FMT_LPCM24: return 3;
    case GDGO_FMT_LPCM32: return 4;
    case GDGO_FMT_IEEE32: return 4;
    case GDGO_FMT_IEEE64: return 8;
    default: return 0;
    }
}

static double clamp1(double s) { return s < -1.0 ? -1.0 : (s > 1.0 ? 1.0 : s); }

/* wave.go:275-735 samplesToBytes*: little endian */
int gdgo_wave_encode(int fmt, const double *samples, size_t n, uint8_t *data) {
    for (size_t i = 0; i < n; i++) {
        double sample = samples[i];
        switch (fmt) {
        case GDGO_FMT_LPCM8: {                                  /* :275-311 */
            sample = clamp1(sample);
            int16_t temp = (int16_t)(127.0 * sample);
            int res = temp - (-128);
            data[i] = (uint8_t)(res < 0 ? 0 : (res > 255 ? 255 : res));
            break;
        }
        case GDGO_FMT_LPCM16: {                                 /* :347-395 */
            sample = clamp1(sample);
            double scale = 0.5 * 65535.0;
            int32_t tmp = (int32_t)(scale * sample);
            if (tmp > 32767) tmp = 32767; else if (tmp < -32768) tmp = -32768;
            uint16_t u = (uint16_t)(int16_t)tmp;
            data[2 * i] = (uint8_t)(u & 0xff); data[2 * i + 1] = (uint8_t)(u >> 8);
            break;
        }
        case GDGO_FMT_LPCM24: {                                 /* :433-470 */
            sample = clamp1(sample);
            double scale = 0.5 * 16777215.0;
            int32_t tmp = (int32_t)(scale * sample);
            if (tmp > MAX_INT24) tmp = MAX_INT24; else if (tmp < MIN_INT24) tmp = MIN_INT24;
            uint32_t u = (uint32_t)tmp;
            for (int j = 0; j < 3; j++) data[3 * i + j] = (uint8_t)((u >> (8 * j)) & 0xff);
            break;
        }
        case GDGO_FMT_LPCM32: {                                 /* :519-562 */
            sample = clamp1(sample);
            double scale = 0.5 * 4294967295.0;
            int64_t tmp = (int64_t)(scale * sample);
            if (tmp > 2147483647LL) tmp = 2147483647LL; else if (tmp < -2147483648LL) tmp = -2147483648LL;
            uint32_t u = (uint32_t)(int32_t)tmp;
            for (int j = 0; j < 4; j++) data[4 * i + j] = (uint8_t)((u >> (8 * j)) & 0xff);
            break;
        }
        case GDGO_FMT_IEEE32: {                                 /* :599-635 */
            float f = (float)clamp1(sample);
            memcpy(data + 4 * i, &f, 4);
            break;
        }
        case GDGO_FMT_IEEE64:                                   /* :674-690: no clipping */
            memcpy(data + 8 * i, &sample, 8);
            break;
        default:
            return -1;
        }
    }
    return 0;
}

/* wave.go:316-735 bytesToSamples* */
int gdgo_wave_decode(int fmt, const uint8_t *data, size_t n, double *samples) {
    for (size_t i = 0; i < n; i++) {
        switch (fmt) {
        case GDGO_FMT_LPCM8: {                                  /* :316-342 */
            int16_t temp = (int16_t)((int16_t)data[i] + (-128));
            double res = (1.0 / 127.0) * (double)temp;
            samples[i] = res < -1.0 ? -1.0 : (res > 1.0 ? 1.0 : res);
            break;
        }
        case GDGO_FMT_LPCM16: {                                 /* :400-427 */
            int16_t s = (int16_t)(uint16_t)(data[2 * i] | (data[2 * i + 1] << 8));
            samples[i] = (2.0 / 65535.0) * (double)s;
            break;
        }
        case GDGO_FMT_LPCM24: {                                 /* :475-514 */
            uint32_t w = (uint32_t)data[3 * i] | ((uint32_t)data[3 * i + 1] << 8) | ((uint32_t)data[3 * i + 2] << 16);
            int32_t v = (int32_t)w;
            if (w & SIGN_BIT_INT24) v = MIN_INT24 + (v & MAX_INT24);
            samples[i] = (2.0 / 16777215.0) * (double)v;
            break;
        }
        case GDGO_FMT_LPCM32: {                                 /* :567-594 */
            uint32_t w = (uint32_t)data[4 * i] | ((uint32_t)data[4 * i + 1] << 8) | ((uint32_t)data[4 * i + 2] << 16) | ((uint32_t)data[4 * i + 3] << 24);
            samples[i] = (2.0 / 4294967295.0) * (double)(int32_t)w;
            break;
        }
        case GDGO_FMT_IEEE32: {                                 /* :640-669 */
            float f;
            memcpy(&f, data + 4 * i, 4);
            samples[i] = (double)f;
            break;
        }
        case GDGO_FMT_IEEE64:                                   /* :695-714 */
            memcpy(&samples[i], data + 8 * i, 8);
            break;
        default:
            return -1;
        }
    }
    return 0;
}

/* ---- level/level.go: one channel meter ----------------------------------------------------------------- */
#define PEAK_HOLD_TIME_SECONDS 2
#define TIME_CONSTANT 1.7
#define MIN_LEVEL (-200.0)

void gdgo_meter_init(gdgo_meter *m) { memset(m, 0, sizeof(*m)); }

/* level.go:260-279 setEnabled: disabling clears the readings */
void gdgo_meter_set_enabled(gdgo_meter *m, int enabled) {
    if ((enabled != 0) != (m->enabled != 0)) {
        if (!enabled) { m->current_value = 0.0; m->peak_value = 0.0; m->sample_counter = 0; }
        m->enabled = enabled != 0;
    }
}

/* level.go:147-210 */
void gdgo_meter_process(gdgo_meter *m, const double *buffer, size_t n, uint32_t sample_rate) {
    if (!m->enabled) return;
    double current = m->current_value, peak = m->peak_value;
    uint64_t counter = m->sample_counter;
    double sr = (double)sample_rate;
    uint64_t hold = (uint64_t)(PEAK_HOLD_TIME_SECONDS * sr);
    double decay_exp = -1.0 / (TIME_CONSTANT * sr);
    double decay = pow(10.0, decay_exp);
    for (size_t i = 0; i < n; i++) {
        current *= decay;
        if (counter > hold) peak *= decay; else counter++;
        double a = fabs(buffer[i]);
        if (a > current) current = a;
        if (a >= peak) { peak = a; counter = 0; }
    }
    m->current_value = current;
    m->peak_value = peak;
    m->sample_counter = counter;
}

static int32_t to_decibels_int(double value) {           /* level.go:100-118 */
    double level = 20.0 * log10(value);
    if (isnan(level) || level < MIN_LEVEL) level = MIN_LEVEL;
    return (int32_t)round(level);
}

void gdgo_meter_analyze(const gdgo_meter *m, int32_t *level, int32_t *peak) {
    *level = to_decibels_int(m->current_value);
    *peak = to_decibels_int(m->peak_value);
}

/* ---- metronome/metronome.go:63-131: tick/tock generator.  The reference has no test for it: unpinned,
 * cross-checked against a closed form in tests/test_oracle_independent.py ------------------------------------------------ */
void gdgo_metronome_init(gdgo_metronome *m) {
    memset(m, 0, sizeof(*m));
    m->beats_per_period = 4; m->bpm_speed = 120; m->sample_rate = 96000;      /* metronome.go:12-14, Create() */
}

void gdgo_metronome_process(gdgo_metronome *m, const double *tick, int n_tick, const double *tock, int n_tock, double *out, int n) {
    uint32_t sample_counter = m->sample_counter, tick_counter = m->tick_counter;
    uint32_t beats = m->beats_per_period;
    uint32_t samples_per_beat = (60u * m->sample_rate) / m->bpm_speed;
    if (beats == 0) beats = 1;
    for (int i = 0; i < n; i++) {
        double sample = 0.0;
        if (tick_counter == 0) {
            if (tick != NULL && sample_counter < (uint32_t)n_tick) sample = tick[sample_counter];
        } else {
            if (tock != NULL && sample_counter < (uint32_t)n_tock) sample = tock[sample_counter];
        }
        out[i] = sample;
        sample_counter++;
        if (sample_counter >= samples_per_beat) { sample_counter = 0; tick_counter = (tick_counter + 1) % beats; }
    }
    m->sample_counter = sample_counter;
    m->tick_counter = tick_counter;
}
