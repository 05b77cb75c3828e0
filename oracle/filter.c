/*
 * filter.c -- oracle restatement of filter/filter.go (FIR filter object with the
 * reference's UNPARTITIONED FFT overlap-add Process, and the filter algebra used by the
 * power-amp compile step).  TEST INFRASTRUCTURE ONLY (see gdg_oracle.h).
 *
 * filter.Process is pinned by the reference only through oversampling_test.go (77/155-tap
 * decimator, N = 16/32).  For L >= frame the parity is unpinned by the reference; it is
 * cross-checked against a direct-form convolution in tests/test_oracle_independent.py.
 */
#include "gdg_oracle.h"
#include <math.h>
#include "libm_jitter.h"
#include <stdlib.h>
#include <string.h>

struct gdgo_filter {
    /* impulseResponseStruct, filter/filter.go:47-52 */
    uint32_t sample_rate;
    double gain_compensation;
    double *data;
    int n;
    /* filterStruct work buffers, filter/filter.go:72-82 */
    gdgo_fft *ft;
    gdgo_cplx *filter_complex;      /* H, full fft_size-point spectrum */
    gdgo_cplx *filtered_complex;
    double *input_buffer;
    double *output_buffer;
    double *tail_buffer;
    int fft_size;                   /* current size of the five buffers above (0 = unallocated) */
};

static gdgo_filter *filter_new(const double *coeffs, int n, uint32_t sample_rate, double comp) {
    gdgo_filter *f = (gdgo_filter *)calloc(1, sizeof(gdgo_filter));
    f->sample_rate = sample_rate;
    f->gain_compensation = comp;
    f->n = n;
    f->data = (double *)malloc(sizeof(double) * (size_t)(n > 0 ? n : 1));
    if (n > 0 && coeffs) memcpy(f->data, coeffs, sizeof(double) * (size_t)n);
    f->ft = gdgo_fft_create();
    return f;
}

/* filter/filter.go:850-890 */
gdgo_filter *gdgo_filter_from_coefficients(const double *coeffs, int n, uint32_t sample_rate, double gain_compensation) {
    return filter_new(coeffs, n, sample_rate, gain_compensation);
}

/* filter/filter.go:807-845 */
gdgo_filter *gdgo_filter_empty(uint32_t sample_rate) { return filter_new(NULL, 0, sample_rate, 0.0); }

void gdgo_filter_destroy(gdgo_filter *f) {
    if (!f) return;
    free(f->data); free(f->filter_complex); free(f->filtered_complex);
    free(f->input_buffer); free(f->output_buffer); free(f->tail_buffer);
    gdgo_fft_destroy(f->ft);
    free(f);
}

int gdgo_filter_length(const gdgo_filter *f) { return f->n; }
const double *gdgo_filter_coefficients(const gdgo_filter *f) { return f->data; }

/* filter/filter.go:167-253; "other == nil" returns this (here: a copy, caller owns it) */
gdgo_filter *gdgo_filter_add(const gdgo_filter *a, const gdgo_filter *b) {
    if (b == NULL) return filter_new(a->data, a->n, a->sample_rate, a->gain_compensation);
    if (a->sample_rate != b->sample_rate) return NULL;
    int n = a->n > b->n ? a->n : b->n;
    gdgo_filter *r = filter_new(NULL, n, a->sample_rate, 0.0);
    for (int i = 0; i < n; i++) r->data[i] = 0.0;
    memcpy(r->data, a->data, sizeof(double) * (size_t)a->n);
    for (int i = 0; i < b->n; i++) r->data[i] += b->data[i];
    return r;
}

/* filter/filter.go:270-323 */
gdgo_filter *gdgo_filter_multiply(const gdgo_filter *f, double scalar) {
    gdgo_filter *r = filter_new(NULL, f->n, f->sample_rate, 0.0);
    for (int i = 0; i < f->n; i++) r->data[i] = scalar * f->data[i];
    return r;
}

/* filter/filter.go:127-138 and :328-336 */
gdgo_filter *gdgo_filter_normalize(const gdgo_filter *f) {
    double sum = 0.0;
    for (int i = 0; i < f->n; i++) sum += f->data[i] * f->data[i];
    double gain = sqrt(sum);
    double fac = f->gain_compensation / gain;
    return gdgo_filter_multiply(f, fac);
}

/* filter/filter.go:520-604 */
gdgo_filter *gdgo_filter_reduce(const gdgo_filter *f, uint32_t order) {
    int n = f->n;
    if ((uint64_t)n <= (uint64_t)order) return filter_new(f->data, f->n, f->sample_rate, f->gain_compensation);
    uint64_t n_src = gdgo_next_power_of_two((uint64_t)n, NULL);
    uint64_t n_tgt = gdgo_next_power_of_two((uint64_t)order, NULL);
    double *padded = (double *)calloc((size_t)n_src, sizeof(double));
    memcpy(padded, f->data, sizeof(double) * (size_t)n);
    gdgo_cplx *fr = (gdgo_cplx *)calloc((size_t)n_src, sizeof(gdgo_cplx));
    gdgo_fft_real_fourier(f->ft, padded, (int)n_src, fr, (int)n_src, GDGO_SCALING_DEFAULT);
    uint32_t num_pos_src = ((uint32_t)n_src >> 1) + 1;
    uint32_t tgt_half = (uint32_t)n_tgt >> 1;
    uint32_t num_pos_tgt = tgt_half + 1;
    gdgo_cplx *fr_pos_new = (gdgo_cplx *)calloc((size_t)num_pos_tgt, sizeof(gdgo_cplx));
    gdgo_resample_frequency(fr, (int)num_pos_src, fr_pos_new, num_pos_tgt);
    gdgo_cplx *fr_new = (gdgo_cplx *)calloc((size_t)n_tgt, sizeof(gdgo_cplx));
    /* copy(frNew, frPosNew): copies min(len) elements */
    uint64_t ncopy = num_pos_tgt < n_tgt ? num_pos_tgt : n_tgt;
    memcpy(fr_new, fr_pos_new, sizeof(gdgo_cplx) * (size_t)ncopy);
    for (uint32_t i = 1; i < tgt_half; i++) {
        fr_new[(uint32_t)n_tgt - i].re = fr_pos_new[i].re;
        fr_new[(uint32_t)n_tgt - i].im = -fr_pos_new[i].im;
    }
    double *target = (double *)calloc((size_t)n_tgt, sizeof(double));
    gdgo_fft_real_inverse_fourier(f->ft, fr_new, (int)n_tgt, target, (int)n_tgt, GDGO_SCALING_DEFAULT);
    gdgo_filter *r = filter_new(target, (int)order, f->sample_rate, f->gain_compensation);
    free(padded); free(fr); free(fr_pos_new); free(fr_new); free(target);
    return r;
}

static void ensure_buffers(gdgo_filter *f, int fft_size) {
    if (f->fft_size == fft_size) return;
    free(f->filter_complex); free(f->filtered_complex);
    free(f->input_buffer); free(f->output_buffer); free(f->tail_buffer);
    f->filter_complex = (gdgo_cplx *)calloc((size_t)fft_size, sizeof(gdgo_cplx));
    f->filtered_complex = (gdgo_cplx *)calloc((size_t)fft_size, sizeof(gdgo_cplx));
    f->input_buffer = (double *)calloc((size_t)fft_size, sizeof(double));
    f->output_buffer = (double *)calloc((size_t)fft_size, sizeof(double));
    f->tail_buffer = (double *)calloc((size_t)fft_size, sizeof(double));
    /* filter/filter.go:395-401: pre-calculate the FFT of the zero-padded taps */
    double *padded = (double *)calloc((size_t)fft_size, sizeof(double));
    memcpy(padded, f->data, sizeof(double) * (size_t)f->n);
    gdgo_fft_real_fourier(f->ft, padded, fft_size, f->filter_complex, fft_size, GDGO_SCALING_DEFAULT);
    free(padded);
    f->fft_size = fft_size;
}

/*
 * filter/filter.go:342-515.  Returns 0 on success, -1 on length mismatch (the caller passes
 * one n for both buffers, so that cannot happen here), -3 when the reference would panic
 * on a slice bound (N not a power of two and a block starts beyond N, :443-453).
 */
int gdgo_filter_process(gdgo_filter *f, const double *in, double *out, int n) {
    int L = f->n;
    if (L == 0) { for (int i = 0; i < n; i++) out[i] = 0.0; return 0; }
    if (n == 0) return 0;
    uint64_t N64 = (uint64_t)n;
    uint64_t n_power = gdgo_next_power_of_two(N64, NULL);
    uint64_t block_size = gdgo_next_power_of_two((uint64_t)L, NULL);
    uint64_t num_blocks = n_power / block_size;
    if (n_power % block_size != 0) num_blocks++;
    for (uint64_t i = 0; i < num_blocks; i++) {
        uint64_t fft_size64 = block_size << 1;
        int fft_size = (int)fft_size64;
        ensure_buffers(f, fft_size);
        uint64_t l_bound = i * block_size;
        uint64_t u_bound = l_bound + block_size;
        if (u_bound > N64) u_bound = N64;
        if (l_bound > u_bound) return -3;
        const double *cur_in = in + l_bound;
        double *cur_out = out + l_bound;
        uint64_t num_samples = u_bound - l_bound;
        memcpy(f->input_buffer, cur_in, sizeof(double) * (size_t)num_samples);
        memset(f->input_buffer + num_samples, 0, sizeof(double) * (size_t)(fft_size64 - num_samples));
        gdgo_fft_real_fourier(f->ft, f->input_buffer, fft_size, f->filtered_complex, fft_size, GDGO_SCALING_DEFAULT);
        /* hadamardComplex, filter/filter.go:100-122 */
        for (int k = 0; k < fft_size; k++) {
            gdgo_cplx a = f->filtered_complex[k], b = f->filter_complex[k];
            f->filtered_complex[k].re = a.re * b.re - a.im * b.im;
            f->filtered_complex[k].im = a.re * b.im + a.im * b.re;
        }
        gdgo_fft_real_inverse_fourier(f->ft, f->filtered_complex, fft_size, f->output_buffer, fft_size, GDGO_SCALING_DEFAULT);
        /* filter/filter.go:473-503: overlap with the tail, clip only what is emitted */
        for (int j = 0; j < fft_size; j++) {
            double pre = f->output_buffer[j] + f->tail_buffer[j];
            if ((uint64_t)j < num_samples) {
                if (pre > 1.0) cur_out[j] = 1.0;
                else if (pre < -1.0) cur_out[j] = -1.0;
                else cur_out[j] = pre;
            } else {
                f->tail_buffer[(uint64_t)j - num_samples] = pre;
            }
        }
        uint64_t tail_size = fft_size64 - num_samples;
        memset(f->tail_buffer + tail_size, 0, sizeof(double) * (size_t)(fft_size64 - tail_size));
    }
    return 0;
}
