/*
 * oversampling.c -- oracle restatement of oversampling/oversampling.go
 * (stateful 2x/4x oversampler + decimator).  TEST INFRASTRUCTURE ONLY (see gdg_oracle.h).
 * Pinned by oversampling/oversampling_test.go:53-81, :141-169 (tests/golden/oversampling.json).
 */
#include "gdg_oracle.h"
#include "aa_taps.inc"
#include <stdlib.h>
#include <string.h>

#define ATTENUATION_HALF_DECIBEL 0.9440608762859234   /* oversampling.go:13 */
#define LOOKAHEAD_ONE_SIDE 4                           /* oversampling.go:14 */
#define LOOKAHEAD_BOTH_SIDES 8                         /* oversampling.go:15 */

struct gdgo_osd {
    uint32_t factor;
    gdgo_filter *aa;                 /* antiAliasingFilter */
    double *pre;  int pre_n;         /* bufferPreUpsampling */
    double *post; int post_n;        /* bufferPostUpsampling */
    double *dec;  int dec_n;         /* bufferPreDecimation */
};

static double g_taps2[77], g_taps4[155];
static int g_taps_ready = 0;

static void taps_init(void) {
    if (g_taps_ready) return;
    for (int k = 0; k < 39; k++) { g_taps2[k] = GDG_AA2_HALF[k]; g_taps2[76 - k] = GDG_AA2_HALF[k]; }
    for (int k = 0; k < 78; k++) { g_taps4[k] = GDG_AA4_HALF[k]; g_taps4[154 - k] = GDG_AA4_HALF[k]; }
    g_taps_ready = 1;
}

const double *gdgo_osd_taps(uint32_t factor, int *n_taps) {
    taps_init();
    if (factor == 2) { *n_taps = 77; return g_taps2; }
    if (factor == 4) { *n_taps = 155; return g_taps4; }
    *n_taps = 0;
    return NULL;
}

/* oversampling.go:194-532 */
gdgo_osd *gdgo_osd_create(uint32_t factor) {
    if (factor != 1 && factor != 2 && factor != 4) return NULL;
    gdgo_osd *o = (gdgo_osd *)calloc(1, sizeof(gdgo_osd));
    o->factor = factor;
    if (factor > 1) {
        int n; const double *t = gdgo_osd_taps(factor, &n);
        o->aa = gdgo_filter_from_coefficients(t, n, 0, 0.0);
    }
    return o;
}

void gdgo_osd_destroy(gdgo_osd *o) {
    if (!o) return;
    gdgo_filter_destroy(o->aa);
    free(o->pre); free(o->post); free(o->dec);
    free(o);
}

/* oversampling.go:54-115 */
int gdgo_osd_oversample(gdgo_osd *o, const double *in, int n_in, double *out, int n_out) {
    int factor = (int)o->factor;
    if (factor <= 1) { memcpy(out, in, sizeof(double) * (size_t)(n_in < n_out ? n_in : n_out)); return 0; }
    if (n_out != n_in * factor) return -1;
    int pre_size = n_in + LOOKAHEAD_BOTH_SIDES;
    if (o->pre_n != pre_size) {                    /* re-allocation zeroes the history */
        free(o->pre);
        o->pre = (double *)calloc((size_t)pre_size, sizeof(double));
        o->pre_n = pre_size;
    }
    int tail_start = pre_size - LOOKAHEAD_BOTH_SIDES;
    memmove(o->pre, o->pre + tail_start, sizeof(double) * LOOKAHEAD_BOTH_SIDES);
    memcpy(o->pre + LOOKAHEAD_BOTH_SIDES, in, sizeof(double) * (size_t)n_in);
    int post_size = ((pre_size - 1) * factor) + 1;
    if (o->post_n != post_size) {
        free(o->post);
        o->post = (double *)calloc((size_t)post_size, sizeof(double));
        o->post_n = post_size;
    }
    gdgo_resample_oversample(o->pre, pre_size, o->post, post_size, o->factor);
    memcpy(out, o->post + LOOKAHEAD_ONE_SIDE * factor, sizeof(double) * (size_t)n_out);
    return 0;
}

/* oversampling.go:126-184 */
int gdgo_osd_decimate(gdgo_osd *o, const double *in, int n_in, double *out, int n_out) {
    int factor = (int)o->factor;
    if (factor <= 1) { memcpy(out, in, sizeof(double) * (size_t)(n_in < n_out ? n_in : n_out)); return 0; }
    if (o->dec_n != n_in) {
        free(o->dec);
        o->dec = (double *)calloc((size_t)(n_in > 0 ? n_in : 1), sizeof(double));
        o->dec_n = n_in;
    }
    int rc = gdgo_filter_process(o->aa, in, o->dec, n_in);
    if (rc != 0) return rc;
    for (int i = 0; i < n_out; i++) {
        int idx = factor * i;
        out[i] = (idx < n_in) ? ATTENUATION_HALF_DECIBEL * o->dec[idx] : 0.0;
    }
    return 0;
}
