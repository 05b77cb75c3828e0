/*
 * gdg_internal.h -- structures shared by the host side (api_*.cpp, ctx.h) and the HIP kernels.
 * Not part of the ABI.
 */
#ifndef GDG_INTERNAL_H
#define GDG_INTERNAL_H

#include <hip/hip_runtime.h>
#include <stdint.h>

#define GDG_MAX_FRAMES 8192          /* controller/controller.go:36 BLOCK_SIZE; one frame must fit the LDS */
#define GDG_MIN_FIR_FRAMES 64

/* ------------------------------------------------------------------------------------------------
 * FIR (power amp) -- uniformly partitioned overlap-save convolution, partition = frame.
 *
 * Per FIR unit and channel, in HBM:
 *   prev  [P]            last frame's input (overlap-save history)
 *   fdl   [K][P] cplx    frequency-domain delay line, slot s holds the packed half spectrum
 *                        (P complex128; bin 0 carries (Re X[0], Re X[P])) of [x_{t-1} | x_t]
 *   H     [K][P] cplx    packed half spectra of the IR partitions, pre-scaled by 1/(2P)
 *   Y     [P]    cplx    accumulated product spectrum (scratch between MAC and inverse)
 *   pos   int            frame counter; slot of the current frame = pos % K
 * ---------------------------------------------------------------------------------------------- */
/* A plan's descriptors hold absolute pointers; those into the caller's buffers are flagged, and every launch carries the distance
 * (in samples) from the buffers the plan was built on to the ones of this call -- walking through a file does not rebuild the plan. */
#define GDG_SRC_IS_INPUT 1
#define GDG_DST_IS_OUTPUT 2
#define GDG_DST_UNUSED 4             /* FIR unit whose output frame is consumed by a chained forward transform only (fir_inv_kernel, CHAIN) */
struct gdg_shift { long long in, out; };

struct gdg_fir_chan {
    const double *src;       /* current frame, `hop` samples */
    double *dst;             /* output frame, `hop` samples */
    double *prev;            /* [2][P]: the previous frame, ping-pong by frame parity */
    double2 *fdl;
    const double2 *H;
    double2 *Y;
    int *pos;
    int K;
    int R;                   /* slots of the delay-line ring: K, or K + W - 1 when the context runs windows of W blocks (time blocking) */
    int flags;               /* GDG_SRC_IS_INPUT / GDG_DST_IS_OUTPUT */
    int hop;                 /* samples per frame; == P for power-of-two frames, < P otherwise (the transform of
                              * [previous | current | zeros] still has 2P points, P = nextpow2(hop)) */
};

/* one forward transform job: [a | b | zeros] -> out.  IR spectra: a = one partition zero padded to P samples, b = NULL (zeros),
 * hop = 0.  Re-partitioning of a live delay line (frame size change): a, b = two consecutive frames of `hop` samples. */
struct gdg_fir_irjob {
    const double *a;
    double2 *out;            /* P complex */
    const double *b;
    int hop;                 /* 0: a holds P samples, b is not read */
    int pad;
};

/* one raw inverse transform job (frame size change): packed half spectrum Y of [first | second | zeros] -> the two frames of
 * `hop` samples, unclipped, scaled by `scale`; `first` may be NULL */
struct gdg_fir_rawjob {
    const double2 *Y;
    double *first, *second;
    int hop, pad;
};

/* process-wide launch-shape knobs of fir.hip / the tuner (their launchers have no context); set through gdg_ctx_set_option */
enum { GDG_KNOB_FFT_HALF_LDS = 0, GDG_KNOB_FWD_PER_CHANNEL, GDG_KNOB_WAVE_FFT, GDG_KNOB_MAC_VARIANT, GDG_KNOB_TUNER_PARTS, GDG_KNOB_COUNT };
int gdg_knob_get(int which);
void gdg_knob_set(int which, int value);

/* launchers implemented in fir.hip; all return hipError_t */
hipError_t gdg_fir_tables_create(int P, double2 **d_tw, double2 **d_tw2);
hipError_t gdg_launch_fir_fwd(int P, int hop, const gdg_fir_chan *d_chans, int n_chans, const double2 *d_tw, const double2 *d_tw2, gdg_shift shift, hipStream_t s);
/* k_lo: the partitions k = K - 1 .. k_lo are summed (descending: the order of every multiply-accumulate kernel).  0 = the whole sum;
 * 1 = everything but the newest partition, i.e. what can be computed BEFORE the frame exists (the premac of small shards, api_process.cpp) */
hipError_t gdg_launch_fir_mac(int P, const gdg_fir_chan *d_chans, int n_chans, int shared_spectra, hipStream_t s, int k_lo = 0, int lds_bytes = 0);
/* time blocking: a window of W (2, 4, 8 or 16) consecutive 8192-sample frames per channel; what = 0 forward transforms, 1 multiply-accumulate
 * (reads every spectrum once for the W frames), 2 inverse transforms, 3 history + frame counter.  chans[].src / dst: frame 0 of the
 * window, frame j at + j * 8192; chans[].Y holds W spectra; chans[].R >= K + W - 1. */
hipError_t gdg_launch_fir_window(int W, const gdg_fir_chan *d_chans, int n_chans, int shared_spectra, const double2 *d_tw, const double2 *d_tw2,
                                 int what, gdg_shift shift, hipStream_t s);
/* d_next_chans != NULL (P == 8192, frames of 8192): descriptor i is the power amp that follows chans[i] in the same channel's chain; its
 * forward transform (history + delay-line slot) is produced by this launch and gdg_launch_fir_fwd is NOT called for it */
/* time blocking with adjacent power amps: what = 2 of d_chans and what = 0 of d_next_chans in one launch (needs gdg_fir_window_chain_ok) */
int gdg_fir_window_chain_ok(int n_chans, int W);
hipError_t gdg_launch_fir_window_chain(int W, const gdg_fir_chan *d_chans, const gdg_fir_chan *d_next_chans, int n_chans, const double2 *d_tw,
                                       const double2 *d_tw2, gdg_shift shift, hipStream_t s);
hipError_t gdg_launch_fir_inv(int P, const gdg_fir_chan *d_chans, int n_chans, const double2 *d_tw, const double2 *d_tw2, int fused, gdg_shift shift, hipStream_t s,
                              const gdg_fir_chan *d_next_chans = nullptr, hipEvent_t ev_begin = nullptr, hipEvent_t ev_end = nullptr);
hipError_t gdg_launch_fir_ir(int P, const gdg_fir_irjob *d_jobs, int n_jobs, double scale, const double2 *d_tw, const double2 *d_tw2, hipStream_t s);
hipError_t gdg_launch_fir_raw_inv(int P, const gdg_fir_rawjob *d_jobs, int n_jobs, double scale, const double2 *d_tw, const double2 *d_tw2, hipStream_t s);

/* ------------------------------------------------------------------------------------------------
 * Segments -- the per-sample units between FIR units, fused into one launch (seg.hip).
 * One workgroup per channel; the frame lives in LDS; per-unit state lives in HBM:
 *   ds   [GDG_DS_LEN] doubles   small state (envelopes, capacitor voltages, LFO phase ...)
 *   is   [GDG_IS_LEN] ints      FSM state, ring write positions
 *   hist [..] doubles           input-history / all-pass rings (length depends on the sample rate)
 * dp / jp carry constants derived on the host from the resolved parameters in exactly the
 * reference's arithmetic (e.g. 10^(dB/20), 1 - exp(-2 pi f / sr)), so that only per-sample
 * transcendental calls differ between host libm and device ocml.  Layout per unit type: seg.hip.
 * ---------------------------------------------------------------------------------------------- */
#define GDG_DS_LEN 32
#define GDG_IS_LEN 8
#define GDG_DP_LEN 32
#define GDG_JP_LEN 8

struct gdg_seg_unit {
    int type;
    int pad;
    int ip[8];                /* resolved parameters (numeric value / discrete index) */
    int jp[GDG_JP_LEN];       /* derived integers */
    double dp[GDG_DP_LEN];    /* derived doubles */
    double *ds;
    int *is;
    double *hist;
    const double *tab;        /* scan tables of the unit's constant-coefficient recurrences (powers of the 8-sample chunk map), built on the
                               * host at plan time -- they depend on the coefficients only -- and shared by all units with the same ones */
};

/* scan tables (seg.hip: lin_scan / lin2_scan); layouts shared with the host builder in api_plan.cpp.  A thread's chunk has GDG_CHK samples:
 * 8 in the general segment kernel (1024 threads per frame), 16 in the two-per-CU kernel for the batch block size (512 threads,
 * seg.hip compiled with -DSEG_FAST); the layouts below take the chunk size as a parameter because the host builds both. */
#ifndef GDG_CHK
#define GDG_CHK 8                 /* samples per thread at the batch block size (8192 / 1024) */
#endif
#define GDG_CHK_FAST 16           /* ... of the 512-thread kernel */
#define LT_W_(c) 0                /* [c]  dot weights */
#define LT_ST_(c) (c)             /* [10] A^(2^k), k = 0..9 */
#define LT_PA_(c) ((c) + 10)      /* [16] A^(q + 1), q = lane & 15 */
#define LT_PB_(c) ((c) + 26)      /* [32] A^(lane - 31), lane = 32..63, at index lane - 32 */
#define LT_PC_(c) ((c) + 58)      /* [64] A^lane */
#define LT_SIZE_(c) ((c) + 122)
#define L2_W_(c) 0                /* [c][2]  dot weights (gH_i, gL_i) */
#define L2_ST_(c) (2 * (c))       /* [10][3] P^(2^k) */
#define L2_PA_(c) (2 * (c) + 30)  /* [16][3] */
#define L2_PB_(c) (2 * (c) + 78)  /* [32][3] */
#define L2_PC_(c) (2 * (c) + 174) /* [64][3] */
#define L2_SIZE_(c) (2 * (c) + 366)
#define LT_W LT_W_(GDG_CHK)
#define LT_ST LT_ST_(GDG_CHK)
#define LT_PA LT_PA_(GDG_CHK)
#define LT_PB LT_PB_(GDG_CHK)
#define LT_PC LT_PC_(GDG_CHK)
#define LT_SIZE LT_SIZE_(GDG_CHK)
#define L2_W L2_W_(GDG_CHK)
#define L2_ST L2_ST_(GDG_CHK)
#define L2_PA L2_PA_(GDG_CHK)
#define L2_PB L2_PB_(GDG_CHK)
#define L2_PC L2_PC_(GDG_CHK)
#define L2_SIZE L2_SIZE_(GDG_CHK)

struct gdg_seg_chan {
    const double *src;
    double *dst;
    double *scratch;          /* one frame of per-channel global scratch */
    int unit_begin;
    int unit_count;
    int flags;                /* GDG_SRC_IS_INPUT / GDG_DST_IS_OUTPUT */
    int wave_mask;            /* WAVE: bit u (u < 15) = unit u of the segment carries state from frame to frame (it meets its predecessor frame; units
                               * from the 16th on always do); bit 16 + u = it reads that state through sc1 loads only (no acquire fence at its
                               * hand-off); bit 31 = some unit of the segment stores that state with plain stores: hand-offs write the XCD's L2 back */
    int *wave;                /* [8 x unit_count] frame counters of the units of this channel's segment (seg.hip, WAVE); zero between launches */
};

/* oversampling tables shared by all channels (device memory) */
struct gdg_os_tables {
    const double *taps2;      /* 77  */
    const double *taps4;      /* 155 */
    const double *lanczos2;   /* [1][6] weights of the half-sample phase */
    const double *lanczos4;   /* [3][6] weights of phases 1/4, 2/4, 3/4 */
    /* the decimator taps again, phase-major and zero padded for the register-blocked decimator of seg.hip:
     * tp[r][e] = taps[F (e - GDG_OS_PADLO(F)) - r] where that index exists, else 0 */
    const double *tapsP2;     /* [2][GDG_OS_NE(2)] */
    const double *tapsP4;     /* [4][GDG_OS_NE(4)] */
};
#define GDG_OS_TAPS(F) ((F) == 2 ? 77 : 155)
#define GDG_OS_BACK(F) ((GDG_OS_TAPS(F) - 1 + (F) - 1) / (F))      /* 38 / 39 */
#define GDG_OS_R(F) ((F) == 2 ? 4 : 2)                              /* consecutive outputs per thread */
#define GDG_OS_NC 48                                                /* slots per phase walked by a thread (>= BACK + R, multiple of 8) */
#define GDG_OS_PADLO(F) (GDG_OS_NC - 1 - GDG_OS_BACK(F))
#define GDG_OS_NE(F) (GDG_OS_NC + GDG_OS_R(F) - 1 + 1)              /* table entries per phase (+1: even) */

/* epoch: a number no other WAVE launch of the context carries (marks that must not be mistaken for an earlier launch's).
 * d_wave_ticket != NULL and n_frames > 1: a workgroup per FRAME and channel, the frames of a channel meeting unit by unit (seg.hip, WAVE) --
 * for windows of channel counts that leave most of the chip idle; the counter (zero before the first launch, zero after every launch)
 * belongs to this launch slot alone */
hipError_t gdg_launch_seg(const gdg_seg_chan *d_chans, int n_chans, const gdg_seg_unit *d_units, int frames, int n_frames, gdg_shift shift,
                          gdg_os_tables os, int *d_error, hipStream_t s, int *d_wave_ticket = nullptr, int epoch = 0, int ahead = 0,
                          const int *d_ahead_list = nullptr, int n_ahead = 0);
/* one frame per launch (n_frames == 1), general kernel: d_ahead_list names n_ahead reverbs (indices into d_units) of LATER segment steps of the
 * call whose wet path this launch makes with extra workgroups (seg.hip REVERB_AHEAD); ahead != 0: reverbs whose descriptor says so (ip[7])
 * only mix -- an earlier launch of the call made theirs */
/* the same for segments that only hold units the two-per-CU kernel runs (gdg_segf_supported) on frames of 8192 samples */
hipError_t gdg_launch_segf(const gdg_seg_chan *d_chans, int n_chans, const gdg_seg_unit *d_units, int frames, int n_frames, gdg_shift shift,
                           gdg_os_tables os, int *d_error, hipStream_t s, int *d_wave_ticket = nullptr, int epoch = 0, int ahead = 0,
                           const int *d_ahead_list = nullptr, int n_ahead = 0);
int gdg_segf_supported(int unit_type);
/* per-frame calls of few channels: a channel's frame on TWO workgroups (seg.hip compiled with -DSEG_TILE; same bits as gdg_launch_seg).  The
 * segments may hold the unit types gdg_segt_supported says, without oversampling, and at most 32 exchange ids in all (gdg_segt_exchanges per
 * unit); d_ticket: a counter that is zero between launches; d_xch: gdg_segt_xch_words() 8-byte words per descriptor, zero before first use;
 * epoch: a number no earlier launch used on these words; d_ahead_list / n_ahead as in gdg_launch_seg */
hipError_t gdg_launch_segt(const gdg_seg_chan *d_chans, int n_chans, const gdg_seg_unit *d_units, gdg_shift shift, gdg_os_tables os, int *d_error,
                           hipStream_t s, int *d_ticket, int epoch, unsigned long long *d_xch, const int *d_ahead_list, int n_ahead);
int gdg_segt_supported(int unit_type);
int gdg_segt_exchanges(int unit_type);
size_t gdg_segt_xch_words(void);
/* an oversampled shaper (overdrive / distortion / excess at 2 x or 4 x) as a launch of its own, one workgroup per (channel, frame, tile): the
 * descriptors are segment descriptors whose unit_begin names the shaper; d_flags: one int per channel, any value but `epoch` (seg.hip) */
hipError_t gdg_launch_os_tiles(int factor, const gdg_seg_chan *d_chans, int n_chans, const gdg_seg_unit *d_units, int frames, int n_frames,
                               gdg_shift shift, gdg_os_tables os, int *d_flags, int epoch, int *d_error, hipStream_t s,
                               const gdg_seg_chan *d_pre_chans = nullptr, int *d_arrive = nullptr);
/* d_pre_chans (n_frames == 1): descriptor i = the segment step in front of d_chans[i]'s shaper, a lone compressor -- the tiles' workgroups run it
 * themselves (that step is then not launched); d_arrive: one int per channel, zero between launches */
hipError_t gdg_launch_os_debug(int factor, const double *d_in, int n, double *d_hist, double *d_up, double *d_down, gdg_os_tables os, hipStream_t s);
/* 1 when seg.hip implements the unit type */
int gdg_seg_supported(int unit_type);

/* ------------------------------------------------------------------------------------------------
 * Spatializer (spat.hip): per-channel constants derived on the host (spatializer.go:170-240).
 * mode 0: no inter-aural delay, 1: left ear delayed (delayTime > 0), 2: right ear delayed.
 * ---------------------------------------------------------------------------------------------- */
struct gdg_spat_chan {
    double fac_left, fac_right, w_early, w_late;
    int mode, early, late, pad;
};
/* one launch: mix of `frames` samples (right = left + out_stride) reading the history d_hist_read, and the new history (last H inputs of
 * every channel) into d_hist_write -- two different buffers, swapped by the caller from block to block */
hipError_t gdg_launch_spatializer(const gdg_spat_chan *d_chans, int nch, const double *d_in, int in_stride, const double *d_hist_read,
                                  double *d_hist_write, int H, double *d_out_lr, int out_stride, int frames, hipStream_t s);

/* ------------------------------------------------------------------------------------------------
 * Tuner (tuner.hip): one 96000-sample ring per channel; analysis = 262144-point real FFT
 * autocorrelation as a four-step FFT over HBM (tuner/tuner.go:379-577).
 * ---------------------------------------------------------------------------------------------- */
#define GDG_TUNER_RING 96000          /* tuner/tuner.go:16 */
#define GDG_TUNER_FFT 262144          /* nextpow2(2 * 96000), tuner.go:388-390 */
/* seq: the analysis this record belongs to -- written LAST, at system scope, so that a host that finds the current number in a record of
 * mapped, coherent host memory may read the record without waiting for the stream (api_tuner_spat.cpp) */
struct gdg_tuner_out { double frequency; int note_index; int cents; unsigned seq; int pad; };
hipError_t gdg_launch_tuner_enqueue(double *d_rings, int nch, int wp, const double *d_samples, int stride, int frames, hipStream_t s);
hipError_t gdg_tuner_tables_create(double2 **d_tw_n, double2 **d_tw_m);
int gdg_tuner_short_ok(double sample_rate, double lowest_note_frequency);
/* parts > 1: a channel's blocks over `parts` workgroups, partial sums through d_partial ([nch][parts][4096] complex) */
int gdg_tuner_short_parts(int nch);
hipError_t gdg_launch_tuner_short(const double *d_rings, int nch, int wp, double sample_rate, const double2 *tw4096, const double2 *tw2_4096,
                                  const double *d_note_freqs, int n_notes, gdg_tuner_out *d_out, double2 *d_partial, int parts, hipStream_t s, unsigned seq = 0);
hipError_t gdg_launch_tuner_analyze(const double *d_rings, int nch, int wp, double sample_rate, double2 *d_work,
                                    const double2 *d_tw_n, const double2 *d_tw_m, const double2 *d_tw512, const double2 *d_tw256,
                                    const double *d_note_freqs, int n_notes, gdg_tuner_out *d_out, hipStream_t s, unsigned seq = 0);

/* ------------------------------------------------------------------------------------------------
 * io.hip: the data formats either side of the path: wave sample codecs, resample.Time,
 * level meters.  Planar float64 [channels][per] on the sample side, interleaved little-endian bytes on
 * the file side.
 * ---------------------------------------------------------------------------------------------- */
#define GDG_METER_SEG 8192
struct gdg_meter_rec { double current, peak; unsigned long long counter; int enabled, pad; };
hipError_t gdg_launch_wave_decode(int fmt, const void *d_bytes, size_t per, unsigned channels, double *d_out, hipStream_t s);
hipError_t gdg_launch_wave_encode(int fmt, const double *d_in, size_t per, unsigned channels, void *d_bytes, hipStream_t s);
/* rows of one strided array -> compact encoded rows (row_len % 4 == 0) */
hipError_t gdg_launch_wave_encode_rows(int fmt, const double *d_in, size_t row_stride, size_t row_len, unsigned n_rows, void *d_bytes, hipStream_t s);
/* many mono pieces in one launch (the batch run's streamed upload): piece r = `count` samples of format `fmt` at `src` -> dst */
struct gdg_decode_row { const unsigned char *src; double *dst; unsigned count; int fmt; };
hipError_t gdg_launch_wave_decode_rows(const gdg_decode_row *d_rows, int n_rows, unsigned max_count, hipStream_t s);
hipError_t gdg_launch_resample_time(const double *d_in, int n, double dx, double *d_out, int n_out, hipStream_t s);
hipError_t gdg_launch_meter(const double *d_rows, size_t stride, int n_ports, int n, gdg_meter_rec *d_state,
                            double decay, unsigned long long hold, hipStream_t s);

/* dst_a[i] += src[i]; dst_b[i] += src[i]  (the aux input of the spatializer, spatializer.go:300-310) */
hipError_t gdg_launch_add_aux(double *d_a, double *d_b, const double *d_src, int n, hipStream_t s);
hipError_t gdg_launch_accumulate(double *d_dst, const double *d_src, int n, hipStream_t s);      /* dst[i] += src[i] */
hipError_t gdg_launch_metronome(const double *d_tick, unsigned n_tick, const double *d_tock, unsigned n_tock, double *d_out, int n,
                                unsigned sc0, unsigned tc0, unsigned spb, unsigned beats, unsigned j0, hipStream_t s);

/* compile.hip: power-amp filter compilation */
hipError_t gdg_launch_filter_reduce(const double *d_taps, int n, unsigned order, double2 *work_a, double2 *work_b, double2 *work_pos, double *d_out,
                                    hipStream_t s);
void gdg_filter_reduce_sizes(int n, unsigned order, size_t *work_points, size_t *pos_points);
/* d_partial: 257 doubles of scratch */
hipError_t gdg_launch_normalize_scale_add(const double *d_src, int n, double compensation, double level, double *d_partial, double *d_composite,
                                          hipStream_t s);

#endif
