/*
 * tuner.c -- oracle restatement of tuner/tuner.go (ring capture + FFT autocorrelation pitch
 * detector).  TEST INFRASTRUCTURE ONLY (see gdg_oracle.h).
 * The reference's tuner fixtures (the WAVs under tuner/samples/) are missing from the snapshot
 * (.MISSING_LARGE_BLOBS), so parity is UNPINNED by the reference; tests use synthetic
 * harmonic tones at the same six notes with the reference's pass criterion
 * (tuner/tuner_test.go:95-106: same note, |cents| <= 5).
 */
#include "gdg_oracle.h"
#include "notes.inc"
#include <math.h>
#include "libm_jitter.h"
#include <stdlib.h>
#include <string.h>

#define NUM_SAMPLES 96000                     /* tuner/tuner.go:16 */

struct gdgo_tuner {
    gdgo_ring *buffer;
    uint32_t sample_rate;
    gdgo_fft *ft;
    double *corr; gdgo_cplx *fftbuf; uint64_t fft_size;
};

/* tuner/tuner.go:592-607 */
gdgo_tuner *gdgo_tuner_create(void) {
    gdgo_tuner *t = (gdgo_tuner *)calloc(1, sizeof(gdgo_tuner));
    t->buffer = gdgo_ring_create(NUM_SAMPLES);
    t->ft = gdgo_fft_create();
    return t;
}

void gdgo_tuner_destroy(gdgo_tuner *t) {
    if (!t) return;
    gdgo_ring_destroy(t->buffer);
    gdgo_fft_destroy(t->ft);
    free(t->corr); free(t->fftbuf);
    free(t);
}

int gdgo_tuner_note_count(void) { return GDG_NOTE_COUNT; }
const char *gdgo_tuner_note_name(int idx) { return (idx >= 0 && idx < GDG_NOTE_COUNT) ? GDG_NOTE_NAMES[idx] : "Unknown"; }
double gdgo_tuner_note_frequency(int idx) { return (idx >= 0 && idx < GDG_NOTE_COUNT) ? GDG_NOTE_FREQS[idx] : 0.0; }
const double *gdgo_tuner_correlation(const gdgo_tuner *t, int *n) { *n = (int)t->fft_size; return t->corr; }

/* tuner/tuner.go:582-587 */
void gdgo_tuner_process(gdgo_tuner *t, const double *samples, int n, uint32_t sample_rate) {
    gdgo_ring_enqueue(t->buffer, samples, n);
    t->sample_rate = sample_rate;
}

/* tuner/tuner.go:332-353 (first maximum wins) */
static double find_maximum(const double *buf, int n, int *max_idx) {
    double max_val = -INFINITY;
    int idx = -1;
    for (int i = 0; i < n; i++) if (buf[i] > max_val) { max_val = buf[i]; idx = i; }
    *max_idx = idx;
    return max_val;
}

/* tuner/tuner.go:379-577 */
int gdgo_tuner_analyze(gdgo_tuner *t, gdgo_tuner_result *res) {
    int n = t->buffer->n;
    uint64_t two_n = (uint64_t)(2 * n);
    uint64_t fft_size = gdgo_next_power_of_two(two_n, NULL);
    if (t->fft_size != fft_size) {
        free(t->corr); free(t->fftbuf);
        t->corr = (double *)calloc((size_t)fft_size, sizeof(double));
        t->fftbuf = (gdgo_cplx *)calloc((size_t)fft_size, sizeof(gdgo_cplx));
        t->fft_size = fft_size;
    }
    double *corr = t->corr;
    gdgo_cplx *buf = t->fftbuf;
    uint32_t sample_rate = t->sample_rate;
    gdgo_ring_retrieve(t->buffer, corr, n);
    memset(corr + n, 0, sizeof(double) * (size_t)(fft_size - (uint64_t)n));
    if (gdgo_fft_real_fourier(t->ft, corr, (int)fft_size, buf, (int)fft_size, GDGO_SCALING_DEFAULT) != 0) return -1;
    for (uint64_t i = 0; i < fft_size; i++) {
        /* elem * conj(elem) with Go's complex multiply: (a*a - b*(-b)) + (a*(-b) + b*a)i */
        double a = buf[i].re, b = buf[i].im, nb = -b;
        buf[i].re = a * a - b * nb;
        buf[i].im = a * nb + b * a;
    }
    if (gdgo_fft_real_inverse_fourier(t->ft, buf, (int)fft_size, corr, (int)fft_size, GDGO_SCALING_DEFAULT) != 0) return -2;
    int last = GDG_NOTE_COUNT - 1;
    double low_freq = GDG_NOTE_FREQS[0], high_freq = GDG_NOTE_FREQS[last];
    double sr = (double)sample_rate;
    double lo_f = (sr / high_freq) + 0.5, hi_f = (sr / low_freq) + 0.5;
    /* Go's int(float) of a non-finite/huge value is implementation-specific; mirror the range guard */
    int low_idx = (isfinite(lo_f) && fabs(lo_f) < 9e18) ? (int)lo_f : -1;
    if (low_idx < 0 || (uint64_t)low_idx >= two_n) low_idx = 0;
    int high_idx = (isfinite(hi_f) && fabs(hi_f) < 9e18) ? (int)hi_f : -1;
    if (high_idx < 0 || (uint64_t)high_idx >= two_n) high_idx = (int)(two_n - 1);
    int max_rel;
    double max_val = find_maximum(corr + low_idx, high_idx - low_idx, &max_rel);
    int idx = low_idx + max_rel;
    int idx_up = idx + 1;
    if (idx_up > n) idx_up = n;
    int idx_down = idx - 1;
    if (idx_down < 0) idx_down = 0;
    double value_left = corr[idx_down], value_right = corr[idx_up];
    double idx_float = (double)idx;
    double value_diff = value_right - value_left;
    double value_sum = value_right + value_left;
    double half_diff = 0.5 * value_diff;
    double double_max = 2.0 * max_val;
    double denominator = double_max - value_sum;
    double shift = half_diff / denominator;
    if (shift < -0.5) shift = -0.5; else if (shift > 0.5) shift = 0.5;
    idx_float += shift;
    double actual_frequency = sr / idx_float;
    int actual_note = -1;
    double actual_cents = INFINITY, actual_cents_abs = INFINITY;
    for (int k = 0; k < GDG_NOTE_COUNT; k++) {
        double ratio = actual_frequency / GDG_NOTE_FREQS[k];
        double diff_cents = 1200.0 * log2(ratio);
        double diff_abs = fabs(diff_cents);
        if (diff_abs < actual_cents_abs) { actual_note = k; actual_cents = diff_cents; actual_cents_abs = diff_abs; }
    }
    int8_t cents_int = 0;
    if (!(isinf(actual_cents) || isnan(actual_cents))) cents_int = (int8_t)actual_cents;
    res->frequency = actual_frequency;
    res->note_index = actual_note;
    res->cents = cents_int;
    return 0;
}
