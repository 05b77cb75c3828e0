/*
 * fft.c -- oracle restatement of fft/fft.go (radix-2 FFT, packed real transforms).
 * TEST INFRASTRUCTURE ONLY (see gdg_oracle.h).  Pinned by fft/fft_test.go golden vectors
 * (tests/golden/fft.json).
 */
#include "gdg_oracle.h"
#include "go_consts.h"
#include <math.h>
#include "libm_jitter.h"
#include <stdlib.h>
#include <string.h>
#include <pthread.h>

#define MATH_MINUS_TWO_PI (-2.0 * M_PI)         /* fft/fft.go:26 */
#define MATH_INV_SQRT_2 GO_MATH_INV_SQRT_2        /* fft/fft.go:25 (exact constant expression) */

static inline gdgo_cplx c_add(gdgo_cplx a, gdgo_cplx b) { gdgo_cplx r = { a.re + b.re, a.im + b.im }; return r; }
static inline gdgo_cplx c_sub(gdgo_cplx a, gdgo_cplx b) { gdgo_cplx r = { a.re - b.re, a.im - b.im }; return r; }
/* Go complex128 multiply: (ac - bd) + (ad + bc)i, no fused operations. */
static inline gdgo_cplx c_mul(gdgo_cplx a, gdgo_cplx b) {
    gdgo_cplx r = { a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re };
    return r;
}
static inline gdgo_cplx c_conj(gdgo_cplx a) { gdgo_cplx r = { a.re, -a.im }; return r; }
static inline gdgo_cplx c_scale(double s, gdgo_cplx a) {
    /* Go: complex(s, 0) * a -> (s*re - 0*im) + (s*im + 0*re)i */
    gdgo_cplx r = { s * a.re - 0.0 * a.im, s * a.im + 0.0 * a.re };
    return r;
}

struct gdgo_fft { gdgo_cplx *scrap; int scrap_n; };

/*
 * Table cache: fft/fft.go:68-135 (small tables, n <= 8192) and :140-240 (large tables kept
 * in maps).  Here one cache indexed by log2(n) serves both; the values are computed with the
 * same expression  exp(i * (-2 pi * j) / n)  = (cos(arg), sin(arg)).
 */
#define MAX_LOG 32
static gdgo_cplx *g_coeffs[MAX_LOG];
static int *g_perm[MAX_LOG];
static pthread_mutex_t g_lock = PTHREAD_MUTEX_INITIALIZER;

static int ilog2(uint64_t n) { int e = 0; while (((uint64_t)1 << e) < n) e++; return e; }

/* fft/fft.go:140-181 */
static const gdgo_cplx *fourier_coefficients(int n) {
    int e = ilog2((uint64_t)n);
    pthread_mutex_lock(&g_lock);
    gdgo_cplx *c = g_coeffs[e];
    if (c == NULL) {
        c = (gdgo_cplx *)malloc(sizeof(gdgo_cplx) * (size_t)n);
        double nf = (double)n;
        for (int j = 0; j < n; j++) {
            double arg = (MATH_MINUS_TWO_PI * (double)j) / nf;
            c[j].re = cos(arg);
            c[j].im = sin(arg);
        }
        g_coeffs[e] = c;
    }
    pthread_mutex_unlock(&g_lock);
    return c;
}

/* fft/fft.go:187-240 (bit-reversal table built by doubling) */
static const int *permutation_coefficients(int n) {
    int e = ilog2((uint64_t)n);
    pthread_mutex_lock(&g_lock);
    int *c = g_perm[e];
    if (c == NULL) {
        c = (int *)calloc((size_t)n, sizeof(int));
        c[0] = 0;
        for (int i = 0; i < e; i++) {
            int m = 1 << i;
            for (int j = 0; j < m; j++) {
                int value = c[j] << 1;
                c[j] = value;
                c[j + m] = value + 1;
            }
        }
        g_perm[e] = c;
    }
    pthread_mutex_unlock(&g_lock);
    return c;
}

gdgo_fft *gdgo_fft_create(void) { return (gdgo_fft *)calloc(1, sizeof(gdgo_fft)); }
void gdgo_fft_destroy(gdgo_fft *ft) { if (ft) { free(ft->scrap); free(ft); } }

/* fft/fft.go:393-409 */
uint64_t gdgo_next_power_of_two(uint64_t value, uint32_t *exponent) {
    uint32_t digit = 0;
    uint64_t v = value;
    while (v) { digit++; v >>= 1; }               /* bits.Len64 */
    uint32_t exp = digit - 1;                      /* wraps for value == 0 exactly like Go's uint32 */
    uint64_t pw = (exp < 64) ? ((uint64_t)1 << exp) : 0;
    if (pw < value) { exp++; pw <<= 1; }
    if (exponent) *exponent = exp;
    return pw;
}

/* fft/fft.go:245-287 (recursive, MODE_STANDARD) */
static void cooley_tukey(const gdgo_cplx *vec, int n, gdgo_cplx *result) {
    if (n <= 1) { if (n == 1) result[0] = vec[0]; return; }
    int half = n / 2;
    gdgo_cplx *even = (gdgo_cplx *)malloc(sizeof(gdgo_cplx) * (size_t)half * 4);
    gdgo_cplx *odd = even + half, *lower = odd + half, *upper = lower + half;
    for (int i = 0; i < half; i++) { even[i] = vec[2 * i]; odd[i] = vec[2 * i + 1]; }
    cooley_tukey(even, half, lower);
    cooley_tukey(odd, half, upper);
    const gdgo_cplx *co = fourier_coefficients(n);
    for (int i = 0; i < half; i++) {
        gdgo_cplx elem = lower[i];
        gdgo_cplx product = c_mul(co[i], upper[i]);
        result[i] = c_add(elem, product);
        result[half + i] = c_sub(elem, product);
    }
    free(even);
}

/* fft/fft.go:528-551 */
static void permute(gdgo_fft *ft, gdgo_cplx *vec, int n) {
    const int *coeff = permutation_coefficients(n);
    if (ft->scrap == NULL || ft->scrap_n < n) {
        free(ft->scrap);
        ft->scrap = (gdgo_cplx *)malloc(sizeof(gdgo_cplx) * (size_t)n);
        ft->scrap_n = n;
    }
    memcpy(ft->scrap, vec, sizeof(gdgo_cplx) * (size_t)n);
    for (int i = 0; i < n; i++) vec[i] = ft->scrap[coeff[i]];
}

/* fft/fft.go:556-607 */
static void inplace_transform(gdgo_fft *ft, gdgo_cplx *vec, int n) {
    permute(ft, vec, n);
    const gdgo_cplx *coeffs = fourier_coefficients(n);
    int size = 1, stride = n;
    uint32_t p;
    gdgo_next_power_of_two((uint64_t)n + 1, &p);
    int pmm = (int)(p - 1);
    for (int r = 1; r <= pmm; r++) {
        size <<= 1;
        stride >>= 1;
        int blocks = n / size;
        for (int j = 0; j < blocks; j++) {
            int half_blocks = blocks << 1;
            int half = n / half_blocks;
            int offset = (j << 1) * half;
            for (int k = 0; k < half; k++) {
                int i = k + offset;
                int jj = i + half;
                gdgo_cplx vi = vec[i], vj = vec[jj];
                int l = k * stride;
                int m = half * stride;
                gdgo_cplx cl = coeffs[l];
                gdgo_cplx cn = coeffs[l + m];      /* table entry l + n/2, not a negation (:592-598) */
                vec[i] = c_add(vi, c_mul(cl, vj));
                vec[jj] = c_add(vi, c_mul(cn, vj));
            }
        }
    }
}

static void swap_complex_inplace(gdgo_cplx *vec, int n) {   /* fft/fft.go:374-386 */
    for (int i = 0; i < n; i++) { double t = vec[i].re; vec[i].re = vec[i].im; vec[i].im = t; }
}

/* fft/fft.go:612-667.  Result is written back into vec for both modes. */
int gdgo_fft_fourier(gdgo_fft *ft, gdgo_cplx *vec, int n, int scaling, int mode) {
    if (mode == GDGO_MODE_STANDARD) {
        gdgo_cplx *res = (gdgo_cplx *)malloc(sizeof(gdgo_cplx) * (size_t)(n > 0 ? n : 1));
        cooley_tukey(vec, n, res);
        memcpy(vec, res, sizeof(gdgo_cplx) * (size_t)n);
        free(res);
    } else if (mode == GDGO_MODE_INPLACE) {
        inplace_transform(ft, vec, n);
    } else {
        return -1;
    }
    if (scaling == GDGO_SCALING_ORTHONORMAL) {
        double r = 1.0 / sqrt((double)n);
        for (int i = 0; i < n; i++) vec[i] = c_mul(vec[i], (gdgo_cplx){ r, 0.0 });
    }
    return 0;
}

/* fft/fft.go:672-739 */
int gdgo_fft_inverse_fourier(gdgo_fft *ft, gdgo_cplx *vec, int n, int scaling, int mode) {
    double nf = (double)n, r = 0.0;
    if (scaling == GDGO_SCALING_DEFAULT) r = 1.0 / nf;
    else if (scaling == GDGO_SCALING_ORTHONORMAL) r = 1.0 / sqrt(nf);
    if (mode == GDGO_MODE_STANDARD) {
        gdgo_cplx *res = (gdgo_cplx *)malloc(sizeof(gdgo_cplx) * (size_t)(n > 0 ? n : 1));
        swap_complex_inplace(vec, n);
        cooley_tukey(vec, n, res);
        swap_complex_inplace(res, n);
        for (int i = 0; i < n; i++) vec[i] = c_scale(r, res[i]);
        free(res);
        return 0;
    } else if (mode == GDGO_MODE_INPLACE) {
        swap_complex_inplace(vec, n);
        inplace_transform(ft, vec, n);
        swap_complex_inplace(vec, n);
        for (int i = 0; i < n; i++) vec[i] = c_scale(r, vec[i]);
        return 0;
    }
    return -1;
}

/* fft/fft.go:744-856 */
int gdgo_fft_real_fourier(gdgo_fft *ft, const double *in, int n_in, gdgo_cplx *out, int n_out, int scaling) {
    if (n_in != n_out) return -1;
    if (n_in % 2 != 0) {
        if (n_in == 1) { out[0].re = in[0]; out[0].im = 0.0; return 0; }
        return -2;
    }
    int half = n_in / 2;
    for (int i = 0; i < half; i++) { out[i].re = in[2 * i]; out[i].im = in[2 * i + 1]; }
    gdgo_fft_fourier(ft, out, half, scaling, GDGO_MODE_INPLACE);
    memcpy(out + half, out, sizeof(gdgo_cplx) * (size_t)half);
    const gdgo_cplx j = { 0.0, 1.0 };
    const gdgo_cplx *coeffs = fourier_coefficients(n_in);
    for (int i = 0; i < half; i++) {
        int idx_low = half + i;
        int idx_high = (i == 0) ? half : n_out - i;
        gdgo_cplx low = out[idx_low];
        gdgo_cplx high_conj = c_conj(out[idx_high]);
        gdgo_cplx coeff = c_mul(j, coeffs[i]);
        gdgo_cplx t = c_sub(c_add(low, high_conj), c_mul(coeff, c_sub(low, high_conj)));
        out[i] = c_scale(0.5, t);
    }
    for (int i = 1; i < half; i++) out[n_out - i] = c_conj(out[i]);
    gdgo_cplx ce = out[half], cc = c_conj(out[half]);
    out[half] = c_scale(0.5, c_add(c_add(ce, cc), c_mul(j, c_sub(ce, cc))));
    if (scaling == GDGO_SCALING_ORTHONORMAL)
        for (int i = 0; i < n_out; i++) out[i] = c_scale(MATH_INV_SQRT_2, out[i]);
    return 0;
}

/* fft/fft.go:863-990 (destroys its input) */
int gdgo_fft_real_inverse_fourier(gdgo_fft *ft, gdgo_cplx *in, int n_in, double *out, int n_out, int scaling) {
    if (n_in != n_out) return -1;
    if (n_in % 2 != 0) {
        if (n_in == 1) { out[0] = in[0].re; return 0; }
        return -2;
    }
    int half = n_in / 2;
    for (int i = 1; i < half; i++)
        in[i] = c_scale(0.5, c_add(in[i], c_conj(in[n_in - i])));
    double dc_real = in[0].re, nyquist_real = in[half].re;
    memcpy(in + half, in, sizeof(gdgo_cplx) * (size_t)half);
    const gdgo_cplx *coeffs = fourier_coefficients(n_in);
    const gdgo_cplx j = { 0.0, 1.0 };
    for (int i = 0; i < half; i++) {
        int idx_low = half + i;
        int idx_high = (i == 0) ? half : n_out - i;
        gdgo_cplx low = in[idx_low];
        gdgo_cplx high_conj = c_conj(in[idx_high]);
        gdgo_cplx even = c_add(low, high_conj);
        gdgo_cplx odd = c_mul(c_sub(low, high_conj), c_conj(coeffs[i]));
        in[i] = c_scale(0.5, c_add(even, c_mul(j, odd)));
    }
    in[0].re = 0.5 * (dc_real + nyquist_real);
    in[0].im = 0.5 * (dc_real - nyquist_real);
    memset(in + half, 0, sizeof(gdgo_cplx) * (size_t)half);
    gdgo_fft_inverse_fourier(ft, in, half, scaling, GDGO_MODE_INPLACE);
    for (int i = 0; i < half; i++) { out[2 * i] = in[i].re; out[2 * i + 1] = in[i].im; }
    if (scaling == GDGO_SCALING_ORTHONORMAL)
        for (int i = 0; i < n_out; i++) out[i] = M_SQRT2 * out[i];
    return 0;
}

/* fft/fft.go:443-523 */
void gdgo_fft_shift(gdgo_cplx *vec, int n, int inverse) {
    int n_neg = n >> 1, n_pos = n_neg;
    int odd = (n & 1) != 0;
    if (odd) n_pos++;
    int a = 0, b = inverse ? n_neg : n_pos;
    while (b < n) { gdgo_cplx t = vec[a]; vec[a] = vec[b]; vec[b] = t; a++; b++; }
    if (odd) {
        if (inverse) {
            b = n - 1; a = b - 1;
            while (a >= n_pos) { gdgo_cplx t = vec[a]; vec[a] = vec[b]; vec[b] = t; a--; b--; }
        } else {
            b = a + 1;
            while (b < n) { gdgo_cplx t = vec[a]; vec[a] = vec[b]; vec[b] = t; a++; b++; }
        }
    }
}
