/*
 * libm_jitter.c -- TEST INFRASTRUCTURE ONLY (see libm_jitter.h): the wrappers of the perturbed oracle build.
 * One generator for the whole library (the tests that use this build are single threaded); gdgo_jitter_seed restarts it, so a test is
 * reproducible; gdgo_jitter_calls says how many results were moved (a test that perturbed nothing proves nothing).
 */
#include <math.h>
#include <stdint.h>

static uint64_t g_state = 0x9E3779B97F4A7C15ull;
static uint64_t g_calls = 0;

void gdgo_jitter_seed(uint64_t seed) { g_state = seed * 0x9E3779B97F4A7C15ull + 0xD1B54A32D192ED03ull; g_calls = 0; }
uint64_t gdgo_jitter_calls(void) { return g_calls; }

/* v moved by k ulp, k uniform in -2 .. +2; exact results that EVERY libm returns exactly stay: 0, +-1 and non-finite values
 * (exp(0) = 1, sin(0) = 0, cos(0) = 1, pow(x, 0) = 1, log10(1) = 0 ... are required or universally honoured special cases) */
static double jitter(double v) {
    if (v == 0.0 || v == 1.0 || v == -1.0 || !isfinite(v)) return v;
    g_state ^= g_state << 13; g_state ^= g_state >> 7; g_state ^= g_state << 17;          /* xorshift64 */
    int k = (int)((g_state >> 33) % 5) - 2;
    g_calls++;
    for (; k > 0; k--) v = nextafter(v, INFINITY);
    for (; k < 0; k++) v = nextafter(v, -INFINITY);
    return v;
}

double gdgo_jit_exp(double x) { return jitter(exp(x)); }
double gdgo_jit_sin(double x) { return jitter(sin(x)); }
double gdgo_jit_cos(double x) { return jitter(cos(x)); }
double gdgo_jit_atan(double x) { return jitter(atan(x)); }
double gdgo_jit_pow(double x, double y) { return jitter(pow(x, y)); }
double gdgo_jit_log10(double x) { return jitter(log10(x)); }
double gdgo_jit_log2(double x) { return jitter(log2(x)); }
double gdgo_jit_log(double x) { return jitter(log(x)); }
