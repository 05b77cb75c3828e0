/*
 * spatializer.c -- oracle restatement of spatializer/spatializer.go (N -> 2 stereo mixdown
 * with 1/r gains and a linearly interpolated inter-aural delay).
 * TEST INFRASTRUCTURE ONLY (see gdg_oracle.h).  PARITY UNPINNED by the reference
 * (spatializer/ has no tests upstream).
 *
 * Quirk kept on purpose: SetSampleRate rebuilds the history buffers but never updates the
 * sampleRate field used for the delay computation, which stays 96000
 * (spatializer/spatializer.go:418-431 vs :149).
 */
#include "gdg_oracle.h"
#include "go_consts.h"
#include <math.h>
#include "libm_jitter.h"
#include <stdlib.h>
#include <string.h>

#define DEFAULT_SAMPLE_RATE 96000            /* spatializer.go:20 */
#define GROUP_DELAY 6.3e-4                    /* spatializer.go:23 */

typedef struct { double azimuth, distance, level; } position_t;

struct gdgo_spatializer {
    uint32_t input_count;
    uint32_t sample_rate;
    position_t *positions;
    double **buffers; int buffer_size;
};

/* spatializer.go:436-469 */
gdgo_spatializer *gdgo_spatializer_create(uint32_t input_channels) {
    gdgo_spatializer *s = (gdgo_spatializer *)calloc(1, sizeof(gdgo_spatializer));
    s->input_count = input_channels;
    s->sample_rate = DEFAULT_SAMPLE_RATE;
    s->positions = (position_t *)calloc(input_channels ? input_channels : 1, sizeof(position_t));
    for (uint32_t i = 0; i < input_channels; i++) s->positions[i].level = 1.0;
    s->buffers = (double **)calloc(input_channels ? input_channels : 1, sizeof(double *));
    s->buffer_size = (int)ceil((double)DEFAULT_SAMPLE_RATE * GROUP_DELAY);
    for (uint32_t i = 0; i < input_channels; i++) s->buffers[i] = (double *)calloc((size_t)s->buffer_size, sizeof(double));
    return s;
}

void gdgo_spatializer_destroy(gdgo_spatializer *s) {
    if (!s) return;
    for (uint32_t i = 0; i < s->input_count; i++) free(s->buffers[i]);
    free(s->buffers); free(s->positions);
    free(s);
}

/* spatializer.go:340-358 (the channel check is `>` upstream, i.e. off by one; indexes are validated here) */
int gdgo_spatializer_set_azimuth(gdgo_spatializer *s, uint32_t ch, double azimuth) {
    if (ch >= s->input_count) return -1;
    s->positions[ch].azimuth = azimuth;
    return 0;
}

/* spatializer.go:363-387 */
int gdgo_spatializer_set_distance(gdgo_spatializer *s, uint32_t ch, double distance) {
    if (ch >= s->input_count) return -1;
    if (distance < 0.0 || distance > 10.0) return -2;
    s->positions[ch].distance = distance;
    return 0;
}

/* spatializer.go:392-416 */
int gdgo_spatializer_set_level(gdgo_spatializer *s, uint32_t ch, double level) {
    if (ch >= s->input_count) return -1;
    if (level < 0.0 || level > 1.0) return -2;
    s->positions[ch].level = level;
    return 0;
}

/* spatializer.go:418-431 */
void gdgo_spatializer_set_sample_rate(gdgo_spatializer *s, uint32_t rate) {
    int size = (int)ceil((double)rate * GROUP_DELAY);
    for (uint32_t i = 0; i < s->input_count; i++) {
        free(s->buffers[i]);
        s->buffers[i] = (double *)calloc((size_t)(size > 0 ? size : 1), sizeof(double));
    }
    s->buffer_size = size;
}

/* spatializer.go:140-335 */
void gdgo_spatializer_process(gdgo_spatializer *s, const double *const *inputs, int n_in, int n,
                              const double *aux, double *out_left, double *out_right) {
    if ((uint32_t)n_in != s->input_count) return;
    double sr = (double)s->sample_rate;
    for (int j = 0; j < n; j++) { out_left[j] = 0.0; out_right[j] = 0.0; }
    for (int i = 0; i < n_in; i++) {
        const double *in = inputs[i];
        position_t p = s->positions[i];
        double azimuth = GO_MATH_DEGREE_TO_RADIANS * p.azimuth;
        double distance = p.distance, level = p.level;
        const double *cur = s->buffers[i];
        int buffer_size = s->buffer_size;
        double sin_az = sin(azimuth), cos_az = cos(azimuth);
        double x_pos = distance * sin_az, y_pos = distance * cos_az;
        double x_left = fabs(x_pos + (GO_HALF_EFFECTIVE_DISTANCE));
        double x_right = fabs(x_pos - (GO_HALF_EFFECTIVE_DISTANCE));
        double y_dist = fabs(y_pos);
        double y_sq = y_dist * y_dist;
        double xl_sq = x_left * x_left;
        double dist_left = sqrt(xl_sq + y_sq);
        double pre_left = 1.0 / dist_left;
        if (pre_left > 1.0) pre_left = 1.0;
        double fac_left = level * pre_left;
        double xr_sq = x_right * x_right;
        double dist_right = sqrt(xr_sq + y_sq);
        double pre_right = 1.0 / dist_right;
        if (pre_right > 1.0) pre_right = 1.0;
        double fac_right = level * pre_right;
        double dist_diff = dist_left - dist_right;
        double delay_time = GO_GROUP_DELAY_OVER_EFFECTIVE_DISTANCE * dist_diff;
        double delay_samples = fabs(delay_time) * sr;
        double early = floor(delay_samples), late = ceil(delay_samples);
        int early_i = (int)early, late_i = (int)late;
        if (early_i >= buffer_size) early_i = buffer_size - 1;
        if (late_i >= buffer_size) late_i = buffer_size - 1;
        for (int j = 0; j < n; j++) {
            double sample = in[j];
            if (delay_time == 0.0) {
                out_left[j] += fac_left * sample;
                out_right[j] += fac_right * sample;
            } else {
                int ie = j - early_i, il = j - late_i;
                double se = (ie >= 0) ? in[ie] : cur[buffer_size + ie];
                double sl = (il >= 0) ? in[il] : cur[buffer_size + il];
                double we = 1.0 - (delay_samples - early);
                double wl = 1.0 - (late - delay_samples);
                double delayed = (we * se) + (wl * sl);
                if (delay_time > 0.0) {
                    out_left[j] += fac_left * delayed;
                    out_right[j] += fac_right * in[j];
                } else {
                    out_left[j] += fac_left * in[j];
                    out_right[j] += fac_right * delayed;
                }
            }
        }
    }
    if (aux != NULL) for (int j = 0; j < n; j++) { out_left[j] += aux[j]; out_right[j] += aux[j]; }
    for (int i = 0; i < n_in; i++) {
        double *cur = s->buffers[i];
        int buffer_size = s->buffer_size;
        int boundary = buffer_size - n;
        if (boundary >= 0) {
            memmove(cur, cur + n, sizeof(double) * (size_t)boundary);
            memcpy(cur + boundary, inputs[i], sizeof(double) * (size_t)n);
        } else {
            memcpy(cur, inputs[i] + (-boundary), sizeof(double) * (size_t)buffer_size);
        }
    }
}
