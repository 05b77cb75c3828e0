/*
 * gdg_oracle.h -- CPU restatement (float64, plain C99) of go-dsp-guitar's batch-mode
 * per-channel effects pipeline.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke()
 * and bench.py's cpu_baseline leg may load it.  The product path (libgdg.so, HIP) never
 * links, loads or calls anything in oracle/.
 *
 * Every function cites the reference file:line it follows (paths relative to the
 * go-dsp-guitar v1.8.0 tree).  The structure of the reference is kept on purpose
 * (unpartitioned FFT overlap-add, table-driven radix-2 FFT, per-sample sin() Lanczos,
 * per-sample exp() tone stack) because the same code is the timed CPU baseline ("port").
 *
 * Pinning: fft, resample, oversampling (incl. filter.Process through Decimate), random and
 * circular are pinned by the reference's own golden vectors (tests/golden/ JSON files,
 * tests/test_oracle_golden.py).  The 21 effects units, filter.Process for long IRs,
 * signal.Chain, spatializer and tuner have NO reference tests and no Go toolchain exists
 * here: for those rows parity is UNPINNED by the reference; they are cross-checked against a
 * second independent formulation in tests/test_oracle_independent.py.
 *
 * Build: -O2 -ffp-contract=off (Go on amd64 never fuses multiply-add).
 */
#ifndef GDG_ORACLE_H
#define GDG_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct { double re, im; } gdgo_cplx;

/* ---- fft/fft.go ---------------------------------------------------------------------- */
enum { GDGO_SCALING_DEFAULT = 0, GDGO_SCALING_ORTHONORMAL = 1, GDGO_MODE_STANDARD = 2, GDGO_MODE_INPLACE = 3 };

typedef struct gdgo_fft gdgo_fft;                 /* fourierTransformStruct (scrapspace owner) */
gdgo_fft *gdgo_fft_create(void);
void gdgo_fft_destroy(gdgo_fft *ft);
uint64_t gdgo_next_power_of_two(uint64_t value, uint32_t *exponent);
int gdgo_fft_fourier(gdgo_fft *ft, gdgo_cplx *vec, int n, int scaling, int mode);
int gdgo_fft_inverse_fourier(gdgo_fft *ft, gdgo_cplx *vec, int n, int scaling, int mode);
int gdgo_fft_real_fourier(gdgo_fft *ft, const double *in, int n_in, gdgo_cplx *out, int n_out, int scaling);
int gdgo_fft_real_inverse_fourier(gdgo_fft *ft, gdgo_cplx *in, int n_in, double *out, int n_out, int scaling);
void gdgo_fft_shift(gdgo_cplx *vec, int n, int inverse);

/* ---- random/random.go ---------------------------------------------------------------- */
typedef struct { uint64_t a, b, n, x; } gdgo_prng;
void gdgo_prng_init(gdgo_prng *g, uint64_t seed);
double gdgo_prng_next_float(gdgo_prng *g);

/* ---- circular/circular.go ------------------------------------------------------------ */
typedef struct { double *values; int n; int pointer; } gdgo_ring;
gdgo_ring *gdgo_ring_create(int size);
void gdgo_ring_destroy(gdgo_ring *r);
void gdgo_ring_enqueue(gdgo_ring *r, const double *elems, int num);
int gdgo_ring_retrieve(const gdgo_ring *r, double *buf, int m);

/* ---- resample/resample.go ------------------------------------------------------------ */
double gdgo_lanczos_kernel(double x, double a);
double gdgo_lanczos_interpolate(const double *s, int n, double x, uint16_t a);
int gdgo_resample_time_length(int input_length, uint32_t source_rate, uint32_t target_rate);
void gdgo_resample_time(const double *samples, int n, uint32_t source_rate, uint32_t target_rate, double *out, int n_out);
void gdgo_resample_frequency(const gdgo_cplx *bins, int n_src, gdgo_cplx *out, uint32_t n_target);
void gdgo_resample_oversample(const double *source, int n_src, double *target, int n_tgt, uint32_t factor);

/* ---- filter/filter.go ---------------------------------------------------------------- */
typedef struct gdgo_filter gdgo_filter;
gdgo_filter *gdgo_filter_from_coefficients(const double *coeffs, int n, uint32_t sample_rate, double gain_compensation);
gdgo_filter *gdgo_filter_empty(uint32_t sample_rate);
void gdgo_filter_destroy(gdgo_filter *f);
int gdgo_filter_length(const gdgo_filter *f);
const double *gdgo_filter_coefficients(const gdgo_filter *f);
gdgo_filter *gdgo_filter_add(const gdgo_filter *a, const gdgo_filter *b);      /* NULL on rate mismatch */
gdgo_filter *gdgo_filter_multiply(const gdgo_filter *f, double scalar);
gdgo_filter *gdgo_filter_normalize(const gdgo_filter *f);
gdgo_filter *gdgo_filter_reduce(const gdgo_filter *f, uint32_t order);           /* returns a copy when n <= order */
int gdgo_filter_process(gdgo_filter *f, const double *in, double *out, int n);

/* ---- oversampling/oversampling.go ---------------------------------------------------- */
typedef struct gdgo_osd gdgo_osd;
gdgo_osd *gdgo_osd_create(uint32_t factor);       /* 1, 2 or 4; NULL otherwise */
void gdgo_osd_destroy(gdgo_osd *o);
int gdgo_osd_oversample(gdgo_osd *o, const double *in, int n_in, double *out, int n_out);
int gdgo_osd_decimate(gdgo_osd *o, const double *in, int n_in, double *out, int n_out);
const double *gdgo_osd_taps(uint32_t factor, int *n_taps);

/* ---- effects/ ------------------------------------------------------------------------ */
enum {
    GDGO_UNIT_SIGNALGENERATOR = 0, GDGO_UNIT_NOISEGATE, GDGO_UNIT_BANDPASS, GDGO_UNIT_AUTOWAH,
    GDGO_UNIT_AUTOYOY, GDGO_UNIT_COMPRESSOR, GDGO_UNIT_OCTAVER, GDGO_UNIT_EXCESS, GDGO_UNIT_FUZZ,
    GDGO_UNIT_OVERDRIVE, GDGO_UNIT_DISTORTION, GDGO_UNIT_TONESTACK, GDGO_UNIT_CHORUS,
    GDGO_UNIT_FLANGER, GDGO_UNIT_PHASER, GDGO_UNIT_TREMOLO, GDGO_UNIT_RINGMODULATOR,
    GDGO_UNIT_DELAY, GDGO_UNIT_REVERB, GDGO_UNIT_POWERAMP, GDGO_UNIT_CABINET, GDGO_UNIT_COUNT
};

#define GDGO_MAX_PARAMS 8

/*
 * A unit carries its parameters as resolved integers: numeric parameters hold the int32
 * value, discrete parameters hold the index into the reference's DiscreteValues list.
 * Slot order = declaration order in the reference's create*() tables (effects/<unit>.go).
 * Name lookup, range checks and error strings live in the host mirror, not in the oracle.
 */
typedef struct gdgo_unit gdgo_unit;
gdgo_unit *gdgo_unit_create(int unit_type);
void gdgo_unit_destroy(gdgo_unit *u);
int gdgo_unit_type(const gdgo_unit *u);
int gdgo_unit_param_count(const gdgo_unit *u);
int gdgo_unit_set_param(gdgo_unit *u, int idx, int32_t value);
int32_t gdgo_unit_get_param(const gdgo_unit *u, int idx);
/* power amp: hand over the compiled composite filter taps (effects/poweramp.go:25-127 stays
 * on the caller's side); n == 0 is the reference's Empty filter (zeros out);
 * calling this resets the FIR state exactly like the reference's recompile does. */
int gdgo_unit_set_fir(gdgo_unit *u, const double *taps, int n);
void gdgo_unit_process(gdgo_unit *u, const double *in, double *out, int n, uint32_t sample_rate);

/* ---- signal/signal.go ---------------------------------------------------------------- */
typedef struct gdgo_chain gdgo_chain;
gdgo_chain *gdgo_chain_create(void);
void gdgo_chain_destroy(gdgo_chain *c);
int gdgo_chain_append_unit(gdgo_chain *c, int unit_type);   /* returns slot id, new units start bypassed */
int gdgo_chain_remove_unit(gdgo_chain *c, int id);
int gdgo_chain_move_up(gdgo_chain *c, int id);
int gdgo_chain_move_down(gdgo_chain *c, int id);
int gdgo_chain_set_bypass(gdgo_chain *c, int id, int bypass);
int gdgo_chain_length(const gdgo_chain *c);
gdgo_unit *gdgo_chain_unit(gdgo_chain *c, int id);
void gdgo_chain_process(gdgo_chain *c, const double *in, int n_in, double *out, int n_out, uint32_t sample_rate);

/* ---- tuner/tuner.go ------------------------------------------------------------------ */
typedef struct gdgo_tuner gdgo_tuner;
typedef struct { double frequency; int32_t note_index; int8_t cents; } gdgo_tuner_result;
gdgo_tuner *gdgo_tuner_create(void);
void gdgo_tuner_destroy(gdgo_tuner *t);
void gdgo_tuner_process(gdgo_tuner *t, const double *samples, int n, uint32_t sample_rate);
int gdgo_tuner_analyze(gdgo_tuner *t, gdgo_tuner_result *res);
int gdgo_tuner_note_count(void);
const char *gdgo_tuner_note_name(int idx);
double gdgo_tuner_note_frequency(int idx);
/* access to the raw autocorrelation of the last analysis (test aid) */
const double *gdgo_tuner_correlation(const gdgo_tuner *t, int *n);

/* ---- spatializer/spatializer.go ------------------------------------------------------ */
typedef struct gdgo_spatializer gdgo_spatializer;
gdgo_spatializer *gdgo_spatializer_create(uint32_t input_channels);
void gdgo_spatializer_destroy(gdgo_spatializer *s);
int gdgo_spatializer_set_azimuth(gdgo_spatializer *s, uint32_t ch, double azimuth);
int gdgo_spatializer_set_distance(gdgo_spatializer *s, uint32_t ch, double distance);
int gdgo_spatializer_set_level(gdgo_spatializer *s, uint32_t ch, double level);
void gdgo_spatializer_set_sample_rate(gdgo_spatializer *s, uint32_t rate);
/* inputs: n_in pointers to n-sample buffers; aux may be NULL; out_left/out_right n samples */
void gdgo_spatializer_process(gdgo_spatializer *s, const double *const *inputs, int n_in, int n,
                              const double *aux, double *out_left, double *out_right);

/* ---- wave/wave.go sample codecs ------------------------------------------ */
enum { GDGO_FMT_LPCM8 = 0, GDGO_FMT_LPCM16, GDGO_FMT_LPCM24, GDGO_FMT_LPCM32, GDGO_FMT_IEEE32, GDGO_FMT_IEEE64 };
int gdgo_wave_bytes_per_sample(int fmt);
int gdgo_wave_encode(int fmt, const double *samples, size_t n, uint8_t *data);
int gdgo_wave_decode(int fmt, const uint8_t *data, size_t n, double *samples);

/* ---- level/level.go channel meter ---------------------------------------- */
typedef struct { int enabled; double current_value, peak_value; uint64_t sample_counter; } gdgo_meter;
void gdgo_meter_init(gdgo_meter *m);
void gdgo_meter_set_enabled(gdgo_meter *m, int enabled);
void gdgo_meter_process(gdgo_meter *m, const double *buffer, size_t n, uint32_t sample_rate);
void gdgo_meter_analyze(const gdgo_meter *m, int32_t *level, int32_t *peak);

/* ---- metronome/metronome.go ------------------------------------------------ */
typedef struct { uint32_t sample_counter, tick_counter, beats_per_period, bpm_speed, sample_rate; } gdgo_metronome;
void gdgo_metronome_init(gdgo_metronome *m);
void gdgo_metronome_process(gdgo_metronome *m, const double *tick, int n_tick, const double *tock, int n_tock, double *out, int n);

#ifdef __cplusplus
}
#endif
#endif
