/*
 * util.c -- oracle restatements of random/random.go, circular/circular.go and
 * resample/resample.go.  TEST INFRASTRUCTURE ONLY (see gdg_oracle.h).
 * Pinned by random_test.go, circular_test.go and resample_test.go golden vectors.
 */
#include "gdg_oracle.h"
#include <math.h>
#include "libm_jitter.h"
#include <stdlib.h>
#include <string.h>

/* ---- random/random.go:23-55 ---------------------------------------------------------- */
void gdgo_prng_init(gdgo_prng *g, uint64_t seed) {
    uint64_t n = ((uint64_t)1 << 31) - 1;
    g->a = 16807; g->b = 0; g->n = n;
    g->x = ((64979 * seed) + 83) % n;             /* uint64 wrap-around like Go */
}

double gdgo_prng_next_float(gdgo_prng *g) {
    uint64_t x = ((g->a * g->x) + g->b) % g->n;
    g->x = x;
    return (double)x / (double)(g->n - 1);
}

/* ---- circular/circular.go:33-124 ----------------------------------------------------- */
gdgo_ring *gdgo_ring_create(int size) {
    gdgo_ring *r = (gdgo_ring *)calloc(1, sizeof(gdgo_ring));
    r->values = (double *)calloc((size_t)(size > 0 ? size : 1), sizeof(double));
    r->n = size;
    r->pointer = 0;
    return r;
}

void gdgo_ring_destroy(gdgo_ring *r) { if (r) { free(r->values); free(r); } }

void gdgo_ring_enqueue(gdgo_ring *r, const double *elems, int num) {
    int n = r->n;
    if (num >= n) {
        memcpy(r->values, elems + (num - n), sizeof(double) * (size_t)n);
        r->pointer = 0;
    } else {
        int ptr = r->pointer, ptr_inc = ptr + num;
        if (ptr_inc < n) {
            memcpy(r->values + ptr, elems, sizeof(double) * (size_t)num);
            r->pointer = ptr_inc;
        } else {
            int head = ptr_inc - n, tail = n - ptr;
            memcpy(r->values + ptr, elems, sizeof(double) * (size_t)tail);
            memcpy(r->values, elems + tail, sizeof(double) * (size_t)head);
            r->pointer = head;
        }
    }
}

int gdgo_ring_retrieve(const gdgo_ring *r, double *buf, int m) {
    int n = r->n;
    if (n != m) return -1;
    int ptr = r->pointer, tail = n - ptr;
    memcpy(buf, r->values + ptr, sizeof(double) * (size_t)tail);
    memcpy(buf + tail, r->values, sizeof(double) * (size_t)ptr);
    return 0;
}

/* ---- resample/resample.go:10-31 ------------------------------------------------------ */
double gdgo_lanczos_kernel(double x, double a) {
    if (x == 0) return 1.0;
    if ((-a < x) && (x < a)) {
        double pi_x = M_PI * x;
        double pi_xa = pi_x / a;
        double pi_x_squared = pi_x * pi_x;
        double x_sin = sin(pi_x);
        double xa_sin = sin(pi_xa);
        double prod = x_sin * xa_sin;
        double arg = a * prod;
        return arg / pi_x_squared;
    }
    return 0.0;
}

/* resample/resample.go:36-66 */
double gdgo_lanczos_interpolate(const double *s, int n, double x, uint16_t a) {
    double floor_x = floor(x);
    int idx = (int)floor_x;
    int idx_inc = idx + 1;
    int l_bound = idx_inc - (int)a, u_bound = idx_inc + (int)a;
    double a_float = (double)a, sum = 0.0;
    for (int i = l_bound; i < u_bound; i++) {
        if (i >= 0 && i < n) {
            double diff = x - (double)i;
            sum += s[i] * gdgo_lanczos_kernel(diff, a_float);
        }
    }
    return sum;
}

/* resample/resample.go:72-87 (output length rule) */
int gdgo_resample_time_length(int input_length, uint32_t source_rate, uint32_t target_rate) {
    double expansion = (double)target_rate / (double)source_rate;
    double out_len_f = (double)input_length * expansion;
    double out_len_floor = floor(out_len_f);
    int out_len = (int)out_len_floor;
    if (out_len_floor == out_len_f) out_len--;
    return out_len;
}

/* resample/resample.go:72-103 */
void gdgo_resample_time(const double *samples, int n, uint32_t source_rate, uint32_t target_rate, double *out, int n_out) {
    double dx = (double)source_rate / (double)target_rate;
    for (int i = 0; i < n_out; i++) out[i] = gdgo_lanczos_interpolate(samples, n, (double)i * dx, 3);
}

/* resample/resample.go:109-142 */
void gdgo_resample_frequency(const gdgo_cplx *bins, int n_src, gdgo_cplx *out, uint32_t n_target) {
    double *re = (double *)malloc(sizeof(double) * (size_t)(n_src > 0 ? n_src : 1) * 2);
    double *im = re + n_src;
    for (int i = 0; i < n_src; i++) { re[i] = bins[i].re; im[i] = bins[i].im; }
    double dx = (double)n_src / (double)n_target;
    for (uint32_t i = 0; i < n_target; i++) {
        double x = (double)i * dx;
        out[i].re = gdgo_lanczos_interpolate(re, n_src, x, 3);
        out[i].im = gdgo_lanczos_interpolate(im, n_src, x, 3);
    }
    free(re);
}

/* resample/resample.go:148-176 */
void gdgo_resample_oversample(const double *source, int n_src, double *target, int n_tgt, uint32_t factor) {
    double dx = 1.0 / (double)factor;
    int f = (int)factor;
    for (int i = 0; i < n_tgt; i++) {
        if (i % f == 0) target[i] = source[i / f];
        else target[i] = gdgo_lanczos_interpolate(source, n_src, (double)i * dx, 3);
    }
}
