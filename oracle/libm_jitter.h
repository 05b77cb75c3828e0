/*
 * libm_jitter.h -- TEST INFRASTRUCTURE ONLY.  With -DGDGO_LIBM_JITTER (oracle/Makefile: libgdg_oracle_jitter.so) every call of a
 * transcendental function in the oracle goes through a wrapper that moves the result by a seeded pseudo-random -2 .. +2 ulp.
 *
 * Why: the oracle calls glibc, the reference calls Go's math package (pure Go / assembly, within 1-2 ulp of glibc: math.Exp, Sin, Cos, Atan,
 * Pow, Log10, Log2 are not correctly rounded, effects/overdrive.go:28-78, effects/chorus.go:19-131, effects/autowah.go:20-130), the HIP path
 * calls ocml.  "Within 1e-9 RMS of the Go binary" can only be claimed if the output does not care which of them answered -- so the tests
 * run both oracles against each other and the HIP path against the perturbed one (tests/test_libm_sensitivity.py).
 *
 * NOT perturbed, because every conforming implementation returns the same bits: floor, ceil, round, trunc, fabs (exact by definition), fmod
 * (exact by definition: Go's math.Mod and glibc's fmod both return x - n y without rounding), sqrt (correctly rounded by IEEE 754; Go uses the
 * SQRTSD instruction).
 */
#ifndef GDGO_LIBM_JITTER_H
#define GDGO_LIBM_JITTER_H
#ifdef GDGO_LIBM_JITTER
double gdgo_jit_exp(double), gdgo_jit_sin(double), gdgo_jit_cos(double), gdgo_jit_atan(double), gdgo_jit_pow(double, double);
double gdgo_jit_log10(double), gdgo_jit_log2(double), gdgo_jit_log(double);
#define exp(x) gdgo_jit_exp(x)
#define sin(x) gdgo_jit_sin(x)
#define cos(x) gdgo_jit_cos(x)
#define atan(x) gdgo_jit_atan(x)
#define pow(x, y) gdgo_jit_pow(x, y)
#define log10(x) gdgo_jit_log10(x)
#define log2(x) gdgo_jit_log2(x)
#define log(x) gdgo_jit_log(x)
#endif
#endif
