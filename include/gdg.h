/*
 * gdg.h -- C-ABI of libgdg.so: the MI355X-native (HIP, gfx950) batch implementation of
 * go-dsp-guitar's per-channel effects pipeline.
 *
 * This is the drop-in boundary.  A cgo shim (go-dsp-guitar_amd/go/, shown in INTEGRATION.md)
 * keeps the reference's Go interfaces effects.Unit (effects/effects.go:83-91) and
 * signal.Chain (signal/signal.go:21-36) and forwards to the entry points below; name lookup,
 * range checks and error strings stay on the host side, only resolved integers, taps and
 * sample buffers cross the ABI.  Plain pointers and sizes only; no C++ or torch types.
 *
 * One context = one GPU = one shard of channels.  Channels are independent, so an N-GPU
 * host creates N contexts and never needs a collective.
 *
 * Conventions
 *   - every function returns GDG_OK (0) or a negative GDG_ERR_* code; gdg_last_error() gives
 *     a message for the last failure on that context;
 *   - a context is NOT internally synchronised: one call at a time per context (the reference's N worker goroutines meet in a
 *     rendezvous above the ABI -- go-dsp-guitar_amd/go/signal, host/gdg_host.cpp -- and the last arrival makes the one call;
 *     control-plane setters take the same lock).  Different contexts (different GPUs) are independent;
 *   - there is NO CPU fallback: without a usable HIP device gdg_ctx_create fails with
 *     GDG_ERR_NO_DEVICE, and a chain that contains something the HIP path cannot run fails
 *     with GDG_ERR_UNSUPPORTED instead of silently computing elsewhere;
 *   - unit_type values are the reference's UNIT_* iota (effects/effects.go:21-43);
 *   - parameter indices follow the declaration order of each unit's create*() table in
 *     effects/<unit>.go; numeric parameters carry their int32 value, discrete parameters the
 *     index into the reference's DiscreteValues list (e.g. oversampling: 0 "- NONE -", 1 "2", 2 "4").
 */
#ifndef GDG_H
#define GDG_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GDG_OK                0
#define GDG_ERR_INVALID      -1   /* bad handle, index or argument */
#define GDG_ERR_UNSUPPORTED  -2   /* valid in the reference, not runnable on the HIP path (yet) */
#define GDG_ERR_HIP          -3   /* a HIP runtime call failed */
#define GDG_ERR_NO_DEVICE    -4   /* no usable gfx950 device */
#define GDG_ERR_NOMEM        -5

enum gdg_unit_type {              /* effects/effects.go:21-43 */
    GDG_UNIT_SIGNALGENERATOR = 0, GDG_UNIT_NOISEGATE, GDG_UNIT_BANDPASS, GDG_UNIT_AUTOWAH,
    GDG_UNIT_AUTOYOY, GDG_UNIT_COMPRESSOR, GDG_UNIT_OCTAVER, GDG_UNIT_EXCESS, GDG_UNIT_FUZZ,
    GDG_UNIT_OVERDRIVE, GDG_UNIT_DISTORTION, GDG_UNIT_TONESTACK, GDG_UNIT_CHORUS,
    GDG_UNIT_FLANGER, GDG_UNIT_PHASER, GDG_UNIT_TREMOLO, GDG_UNIT_RINGMODULATOR,
    GDG_UNIT_DELAY, GDG_UNIT_REVERB, GDG_UNIT_POWERAMP, GDG_UNIT_CABINET, GDG_UNIT_COUNT
};

#define GDG_MAX_PARAMS 8

typedef struct gdg_ctx gdg_ctx;

/* ---- library / context ----------------------------------------------------------------------- */

/* "gdg <version> gfx950 hip" */
const char *gdg_version(void);

/* Number of HIP devices visible to the process (0 when there is none / no driver). */
int gdg_device_count(void);

/*
 * Create the shard of n_channels channels on HIP device `device`.  max_frames bounds the
 * frames argument of gdg_process* (the reference's batch loop uses BLOCK_SIZE = 8192,
 * controller/controller.go:36).  Replaces: N x signal.CreateChain (controller.go:3267-3269),
 * spatializer.Create (:3273) and tuner.Create (:3279) for this shard.
 */
int gdg_ctx_create(int n_channels, int max_frames, int device, gdg_ctx **out);
int gdg_ctx_destroy(gdg_ctx *ctx);
const char *gdg_last_error(const gdg_ctx *ctx);
int gdg_ctx_channels(const gdg_ctx *ctx);
/*
 * Power amps whose composite filters are identical (same taps, same partition size) share ONE copy of the IR spectra in
 * HBM; the spectrum multiply-accumulate then streams it from L2 / MALL for all but the first channel (d < 1).
 * On by default (env GDG_SHARE_IR_SPECTRA=0 or enable = 0 turns it off for power amps prepared afterwards).  Results are
 * bit-identical either way.
 */
int gdg_ctx_share_ir_spectra(gdg_ctx *ctx, int enable);
/*
 * Launch-shape options of a context -- everything that used to be an environment variable of the process that loads the library.  The
 * variables still exist as DEBUG overrides of the defaults, read once when the context is made; a value set here wins.  Results never depend
 * on an option beyond the last bits (which association a sum takes); unknown keys and values out of range are GDG_ERR_INVALID.
 *   key                          values      meaning (default)
 *   fir_fused                    -1, 0, 1    spectrum multiply-accumulate inside the inverse transform's kernel: by channel count / never / always (-1)
 *   fir_split_max_channels       >= 0        with fir_fused = -1: launches of up to this many channels take the bin-tiled multiply-accumulate (192: with the
 *                                            sums made ahead, fir_premac, it is the faster shape up to there when a channel has two power amps)
 *   fir_split_max_channels_one_amp >= 0      ... and when the call has ONE power amp per channel the smaller of the two applies (112: with one amp the
 *                                            fused kernel wins at 128 channels, profiles/shape_sweep_r06.txt)
 *   fir_chain_adjacent_amps      0, 1        a power amp's inverse transform also makes the forward transform of the amp behind it (1)
 *   fir_premac                   0, 1        per-frame calls of few channels (the split launch shape): when a call ends, the sums of the NEXT frame's
 *                                            convolution over the partitions that are already in the delay line (7 of 8 at 65536 taps) are launched on a
 *                                            stream of their own and run beside the segments; the next call adds the newest term.  Speculative: any library
 *                                            call but a process call, gdg_ctx_synchronize and gdg_ctx_stream drops them.  Same bits either way (1)
 *   fir_premac_min_partitions    >= 1        ... for launches of at least this many partitions (channels x ceil(taps / 8192)) (384 = 48 channels x 65536
 *                                            taps: below, the two cross-stream hops cost more than they hide)
 *   fir_premac_min_partitions_two_amps >= 1  ... and when a channel has two or more power amps, the smaller of the two (320 = 40 channels x 65536 taps)
 *   stat_premac_launches_used    >= 0        a counter, not a setting: inverse-transform launches so far that continued sums made ahead (tests read it to
 *                                            see that the path under test is the one that ran; setting it sets the count) (0)
 *   fir_premac_lds_bytes         -1 .. 65536 ... whose workgroups ask for this much LDS they never touch, so that they land on the CUs the segments leave
 *                                            idle instead of among the segments' waves; -1: 16384 below 120 channels, 49152 from there, 0 when a channel
 *                                            has fewer than 5 or more than 32 partitions (-1)
 *   share_ir_spectra             0, 1        = gdg_ctx_share_ir_spectra (1)
 *   seg_two_per_cu               0, 1        segments of in-place units on 8192-sample frames take the 512-thread kernel, two workgroups per CU (1)
 *   seg_two_per_cu_min_channels  >= 0        ... from this many channels per call on (128)
 *   seg_wave_max_channels        >= 0        windows (gdg_ctx_set_window) of up to this many channels per call: one workgroup per FRAME and channel, the
 *                                            frames of a channel meeting unit by unit -- fills the chip when the channels alone do not (448; 0: never)
 *   seg_wave_release_max_channels >= 0       ... but segments holding a unit whose state leaves the CU through plain stores (flanger, phaser, delay,
 *                                            fuzz, auto-yoy, auto-wah, band pass, octaver, noise gate: every hand-off writes the XCD's L2 back) only up
 *                                            to this many channels (112)
 *   seg_tile_max_channels        >= 0        per-frame calls of up to this many channels: a segment made of compressor, shapers without oversampling, tone
 *                                            stack, cabinet and chorus runs with a channel's 8192-sample frame on TWO workgroups -- the units of such a
 *                                            call are compute bound on one CU while most of the chip idles.  The scans keep the general kernel's
 *                                            association (a scan's sixteen wave totals meet in one place, eight of them through HBM): same bits.  Applies while the launch's
 *                                            workgroups -- 2 x channels + one per reverb whose wet path it carries -- stay within 224 of the 256 CUs (112; 0: never)
 *   seg_os_tiles_max_channels    >= 0        calls of up to this many channels run every 2 x / 4 x oversampled shaper as a launch of its own, one workgroup
 *                                            per (channel, frame, tile of 4096 / 2048 samples) instead of one per channel (192; 0: never)
 *   seg_os_tiles_prefix          0, 1        ... and when the step in front of such a launch is a lone compressor in every channel (compressor > 4 x overdrive:
 *                                            BASELINE config 3), a per-frame call runs that compressor inside the tiles' workgroups instead of launching it (1)
 *   seg_reverb_ahead_max_channels >= 0       per-frame calls of up to this many channels: the call's first segment launch also makes, with extra
 *                                            workgroups beside the channels' own, the wet path of every reverb of its LATER segment steps -- tapped
 *                                            sums and all-passes need nothing of the frame itself when every tap lies at least a frame back
 *                                            (8192-sample frames, rates from 42.7 kHz) -- and the reverb behind the power amps only mixes.  The limit
 *                                            applies to calls that also sum convolution terms ahead (fir_premac: its launches want the same idle CUs);
 *                                            without them up to 127 channels.  Same bits either way (72; 0: never)
 *   wave_spin_limit_ms           1 .. 600000 how long a workgroup of an in-launch hand-off (windows of few channels, tiles of an oversampled shaper) waits
 *                                            for its predecessor before the launch gives up: the wait ends, the context's error word is set and the next
 *                                            gdg_ctx_synchronize (or batch run) returns GDG_ERR_HIP -- the device never hangs.  RESULTS OF WINDOW CALLS ARE
 *                                            VALID ONLY AFTER A gdg_ctx_synchronize THAT RETURNED GDG_OK; after such an error the units' state is undefined
 *                                            (gdg_unit_reset them), the context itself stays usable (1000)
 *   debug_stall_unit             -1, handle  test hook: in the next windows' first frame this unit withholds its hand-off, so that the bounded wait can
 *                                            be seen to expire (-1)
 *   plan_patch                   0, 1        parameter changes patch the device descriptors in place instead of rebuilding the plan (1)
 *   scan_tables_max              >= 1        scan tables (one per distinct coefficient set) kept before a plan rebuild drops them all (1024)
 *   pcie_groups                  0 .. 16     channel groups of the host-buffer calls, 0 = by channel count (0)
 *   device_groups_default        0 .. 16     what gdg_ctx_set_overlap(ctx, 0) means, 0 = one group (0)
 *   copy_threads                 1 .. 256    host copy workers of the host-buffer paths and the batch run (8)
 *   numa                         0, 1, 2     copy workers on the CPUs, pinned slabs from the memory, of a NUMA node: 2 the node the caller runs on when
 *                                            they are made, 1 the device's node, 0 wherever the scheduler and hipHostMalloc put them (2; set it before
 *                                            the first host-buffer call -- slabs that exist stay where they are)
 *   tuner_poll_results           0, 1        gdg_tuner_analyze reads the result records (mapped host memory, each ending with its analysis number) as soon
 *                                            as they carry this call's number instead of waiting for the stream to drain (1)
 *   tuner_long_transform         0, 1        every tuner analysis through the reference's 262144-point transform pair (0)
 *   profile_attach               0, 1        the fused convolution launch records its own begin / end events (1)
 * Process-wide (the transforms' and the tuner's launchers have no context; set them before the first call that uses them):
 *   fft_half_lds_mask            0 .. 63     which 8192-point transforms run through ONE LDS buffer, two workgroups per CU (14)
 *   fir_forward_per_channel      0, 1        a window's forward transforms as one workgroup per channel from a chip's worth of channels on (1)
 *   fir_forward_wave_local       0 .. 3      8 x 1024 forward transform with wave-local sub-transforms (1)
 *   fir_mac_variant              0 .. 15     tile shape of the stand-alone multiply-accumulate (0)
 *   tuner_parts                  0 .. 24     workgroups per channel of the short-lag analysis, 0 = by channel count (0)
 * gdg_option_count / gdg_option_name enumerate the keys.
 */
int gdg_ctx_set_option(gdg_ctx *ctx, const char *key, long long value);
int gdg_ctx_get_option(gdg_ctx *ctx, const char *key, long long *value);
int gdg_option_count(void);
const char *gdg_option_name(int index);
/*
 * Where a PCI device hangs, from sysfs (what option "numa" uses): *node = <sysfs_root>/bus/pci/devices/<pci_bus_id>/numa_node (-1 when the
 * platform does not say), cpus[0 .. min(capacity, *n_cpus)) = the CPUs of <sysfs_root>/devices/system/node/node<N>/cpulist.  sysfs_root is
 * "/sys" outside tests.  No device needed.
 */
int gdg_numa_probe(const char *sysfs_root, const char *pci_bus_id, int *node, int *cpus, int capacity, int *n_cpus);
/* The hipStream_t all of this context's work is enqueued on (as void*), for event timing. */
void *gdg_ctx_stream(const gdg_ctx *ctx);
/* Block until everything enqueued so far has finished. */
int gdg_ctx_synchronize(gdg_ctx *ctx);
/* Give device memory the context no longer uses back to the device (entirely free chunks of the per-unit state arena, all but one spare).
 * Freeing device memory waits for the whole device, so the library never does it inside a process call; it happens here and whenever a plan
 * is rebuilt.  Blocks. */
int gdg_ctx_trim(gdg_ctx *ctx);

/* ---- effects units: effects.CreateUnit / Set*Value / state ----------------------------------- */

/* effects.CreateUnit(unitType) (effects/effects.go:443-516); parameters start at the reference's defaults. */
int gdg_unit_create(gdg_ctx *ctx, int channel, int unit_type, int *handle);
int gdg_unit_destroy(gdg_ctx *ctx, int handle);
/* Resolved value of one parameter (the host side has already done effects.go:144-384's checks).  Effective from the next process call, like the
 * reference's setter (effects/effects.go:283-345: a store under a mutex).  Cheap on a live context: the call itself stores the value; the next
 * process call re-derives that unit's constants and patches its descriptor on the device in place -- the launch plan is only rebuilt by changes of
 * a chain's layout (gdg_chain_set), of the frame size or rate, or by new filter taps.
 * "Cheap" has exceptions, all at the NEXT process call: (1) a value that moves the unit to another kernel -- any change of an oversampling
 * factor, a reverb leaving the in-place shape -- rebuilds the plan (~0.5 ms for 512 channels); (2) more than `scan_tables_max` distinct coefficient
 * sets since the last plan (a caller sweeping a tone stack through a thousand settings) rebuilds it once to drop the table cache; (3) a unit
 * whose constants cannot be derived (prepare fails) rebuilds it to report the error; (4) a value that re-makes a history the way the reference
 * does (a longer delay, a new band-pass order) waits for the stream and may take a new arena chunk: one device malloc and one fill of up to
 * 1 GiB, waited for.  Device memory is never FREED on this path (gdg_ctx_trim). */
int gdg_unit_set_param(gdg_ctx *ctx, int handle, int param_index, int32_t value);
int gdg_unit_get_param(gdg_ctx *ctx, int handle, int param_index, int32_t *value);
/*
 * Power amp only: the compiled composite FIR (what effects/poweramp.go:25-127 compile()
 * returns; Normalize/Reduce/Add stay on the host).  n_taps == 0 is filter.Empty (zeros out,
 * filter/filter.go:366-367).  Like the reference's recompile (poweramp.go:132-181) this
 * replaces the filter and therefore resets the convolution state.
 */
int gdg_unit_set_fir(gdg_ctx *ctx, int handle, const double *taps, int n_taps);
/* Zero all DSP state of a unit (what re-creating the unit does in the reference). */
int gdg_unit_reset(gdg_ctx *ctx, int handle);

/*
 * poweramp.compile on the device (effects/poweramp.go:25-127).  Slot i of the power amp holds the taps of
 * impulse response i at the current sample rate (filter.ImpulseResponses.CreateFilter(...).Coefficients(); NULL or length 0 =
 * "- NONE -"), its gain compensation FACTOR (filter/filter.go:127-138) and its `level_i` parameter in dB.  Per slot:
 * Reduce(target_order) when target_order > 0 and the slot is longer (filter.go:520-604), Normalize, Multiply(level); the
 * slots are then added in order (filter.go:167-236) and the composite becomes the unit's FIR exactly as gdg_unit_set_fir
 * would set it (including the state reset).  gdg_unit_get_fir reads the composite back (taps may be NULL to query n_taps).
 */
int gdg_unit_compile_fir(gdg_ctx *ctx, int handle, int n_filters, const double *const *taps, const int *lengths,
                         const double *gain_compensation, const int32_t *levels_db, uint32_t target_order);
int gdg_unit_get_fir(gdg_ctx *ctx, int handle, double *taps, int capacity, int *n_taps);

/*
 * signal.Chain slot list of one channel (signal/signal.go:52-157): handles in processing
 * order with their bypass flags.  Bypassed slots are skipped and do not advance their state
 * (signal.go:390-401).  State stays with the unit handle, not with the slot index.
 */
int gdg_chain_set(gdg_ctx *ctx, int channel, const int *handles, const uint8_t *bypass, int n);

/* ---- processing: signal.Chain.Process for all channels of the shard at once ------------------- */

/*
 * One block of `frames` samples for every channel (what controller.process() fans out to its
 * N workers, controller/controller.go:2682-2705).  in[c] / out[c] are host buffers of `frames`
 * float64 each; the call stages them through pinned memory, runs the batch and blocks until
 * out is written.  in[c] is not modified.
 * `frames` may change from call to call (1 .. max_frames): like filter.Process (filter/filter.go:370-428, tail and transform
 * sizes depend on the filter length only) a power amp carries its convolution state across the change -- its delay line is
 * re-partitioned once per change.  A (frames, filter length) pair on which the reference itself panics (frames not a power
 * of two and a nextpow2(L)-sized block starting beyond the frame, filter.go:443-453) is rejected with GDG_ERR_UNSUPPORTED.
 */
int gdg_process(gdg_ctx *ctx, const double *const *in, double *const *out, int frames, uint32_t sample_rate);

/*
 * Same for a subset of the shard's channels: in[i] / out[i] belong to channel channels[i]; the
 * chains of all other channels are left untouched (their state does not advance).  This is what
 * the host shim's rendezvous falls back to when fewer than N Chain.Process calls are in flight.
 */
int gdg_process_subset(gdg_ctx *ctx, const int *channels, int n, const double *const *in, double *const *out,
                       int frames, uint32_t sample_rate);

/*
 * Staged variant for hosts that may not hand their own pointers to C (cgo's pointer rules):
 * gdg_staging_buffers returns two pinned host slabs (hipHostMalloc) whose row c (row_stride
 * float64 apart) belongs to channel c; every worker copies its frame into its input row,
 * gdg_process_staged runs the listed channels and fills their output rows.
 */
int gdg_staging_buffers(gdg_ctx *ctx, double **in, double **out, int *row_stride);
int gdg_process_staged(gdg_ctx *ctx, const int *channels, int n, int frames, uint32_t sample_rate);

/*
 * Same, device-resident: d_in / d_out are device pointers to [n_channels][frames] float64
 * (row-major, row stride = frames).  Enqueued on gdg_ctx_stream() and NOT synchronised;
 * d_in == d_out is not allowed.
 */
int gdg_process_device(gdg_ctx *ctx, const double *d_in, double *d_out, int frames, uint32_t sample_rate);

/*
 * Time blocking for callers that hold several consecutive frames of every channel -- the batch run, whose files live in HBM
 * (controller.go:3076-3107 walks them 8192 samples at a time only because its buffers are that long).  A window of W frames per
 * channel and call: the units' state runs through the W frames in order, and every power amp reads its IR spectra and delay line
 * ONCE for all W frames (2 K + W - 1 spectrum reads instead of 2 K W; the sums keep their order and their arithmetic: the output is
 * bit-identical to W calls of gdg_process_device).
 * gdg_ctx_set_window: W in {1, 2, 4, 8, 16}; needs max_frames == 8192; may be called at any time, live convolution state moves into
 * the larger delay-line ring (K + W - 1 slots) like on a frame-size change.
 * gdg_process_window_device: d_in / d_out = [n_channels][row_stride] float64, frame j of channel c at c * row_stride + j * 8192;
 * frames_in_window in {1, 2, 4, 8, 16}, <= W (the tail of a file).  Enqueued on gdg_ctx_stream(), not synchronised.
 */
int gdg_ctx_set_window(gdg_ctx *ctx, int frames_per_call);
/*
 * Channel groups of the device-resident calls (gdg_process_device, gdg_process_window_device) -- OPT-IN.  By default every call's
 * kernels run on gdg_ctx_stream(): work a caller enqueues on that stream after the call is ordered after them.  With groups > 1 the
 * channels are cut into `groups` contiguous groups whose kernels run on streams of their own and are NOT joined at the end of the
 * call, so one group's latency-bound segment kernel overlaps another group's HBM-bound convolution, also across calls (+7-10 % at 512
 * channels with two groups).  The price is the ordering contract: the context's stream is ordered after the groups only by the next
 * library call of any other kind (including gdg_ctx_stream() and gdg_ctx_synchronize()) -- fetch the stream AFTER the process call if
 * you enqueue your own work behind it.  groups: 1 ... 16; 0 = back to the default (one group, unless env GDG_DEVICE_GROUPS names a
 * count).
 */
int gdg_ctx_set_overlap(gdg_ctx *ctx, int groups);
int gdg_process_window_device(gdg_ctx *ctx, const double *d_in, double *d_out, size_t row_stride, int frames_in_window, uint32_t sample_rate);

/* Device memory helpers for callers that have no HIP runtime of their own (e.g. the Go shim). */
int gdg_device_alloc(gdg_ctx *ctx, size_t bytes, void **d_ptr);
int gdg_device_free(gdg_ctx *ctx, void *d_ptr);
int gdg_copy_to_device(gdg_ctx *ctx, void *d_dst, const void *h_src, size_t bytes);
int gdg_copy_to_host(gdg_ctx *ctx, void *h_dst, const void *d_src, size_t bytes);
/* n_rows rows of row_len float64 from one strided device array to another (strides in float64), enqueued on the context's stream,
 * not synchronised: cuts an 8192-frame block out of whole files resident in HBM and puts a processed block back
 * (controller/controller.go:3088-3099) */
int gdg_copy_rows_device(gdg_ctx *ctx, double *d_dst, size_t dst_stride, const double *d_src, size_t src_stride, size_t row_len, size_t n_rows);

/* ---- the transforms underneath the power amp, stand-alone ------------------------------------ */

/*
 * fft.RealFourier / fft.RealInverseFourier (fft/fft.go:744-856, :863-990) as fir.hip computes them (packed-real Stockham
 * transforms in registers + LDS): n real samples <-> n / 2 + 1 complex bins (re, im interleaved; the other half is the
 * conjugate mirror the reference also stores).  n = 1 (the identity, fft/fft.go:765-768) or a power of two from 2 to 16384 (n <= 64: a small radix-2 kernel of
 * the same packed-real scheme, which is what lets the reference's own eight-point known answers, fft/fft_test.go:237-271, run on the
 * HIP transforms).  Forward unscaled, inverse scaled by 1 / n
 * (SCALING_DEFAULT); like the reference the inverse reads only the real parts of bins 0 and n / 2.  Host buffers, blocking.
 */
int gdg_fft_real(gdg_ctx *ctx, const double *samples, int n, double *spectrum);
int gdg_fft_real_inverse(gdg_ctx *ctx, const double *spectrum, int n, double *samples);

/*
 * Debug entry: oversampling.OversamplerDecimator (oversampling/oversampling.go:27-30) as the 2x / 4x units' LDS tiles compute it --
 * Oversample (:54-115: Lanczos-3 polyphase, 8 inputs of history) followed by Decimate of ITS output (:126-184: 77 / 155-tap
 * anti-aliasing filter, clip, stride-f pick, x 0.944...), with no waveshaper in between: exactly the sequence
 * oversampling/oversampling_test.go:84-128 checks, so the reference's own vectors run on the HIP tiles.
 * factor 2 or 4; n <= 8192 samples in; `state` = the object's state across calls, 8 + (77 or 155) - 1 doubles, zeros = a fresh
 * object (in / out); oversampled: factor * n samples out (may be NULL); decimated: n samples out.  Host buffers, blocking.
 */
int gdg_debug_oversample_decimate(gdg_ctx *ctx, int factor, const double *in, int n, double *state, double *oversampled, double *decimated);

/* ---- per-kernel timing on the context's stream (HIP events), for bench.py's roofline ---------- */

enum gdg_kernel_kind {
    GDG_K_FIR_FWD = 0,   /* forward real FFT of the input partition -> frequency-domain delay line */
    GDG_K_FIR_MAC,       /* sum over partitions of FDL x IR spectra (the HBM-bound kernel) */
    GDG_K_FIR_INV,       /* inverse real FFT, clip, write */
    GDG_K_SEGMENT,       /* fused per-sample units between FIR units */
    GDG_K_TUNER,
    GDG_K_SPATIALIZER,
    GDG_K_WAVE,          /* wave sample codecs */
    GDG_K_RESAMPLE,      /* resample.Time */
    GDG_K_METER,         /* level meters */
    GDG_K_FIR_MAC_CHAIN, /* GDG_K_FIR_MAC of a power amp that is followed by another one: the same kernel also makes the next amp's forward
                          * transform (fir_inv_kernel CHAIN); kept apart so that GDG_K_FIR_MAC times the plain kernel only */
    GDG_K_COUNT
};
/* enable == 1: bracket every kernel launch with a HIP event pair from now on (costs a few microseconds per launch);
 * enable == 1 << (kind + 1) (or-able): only the launches of those kernel kinds; 0: off.
 * The fused convolution launch (GDG_K_FIR_MAC / GDG_K_FIR_MAC_CHAIN) carries its two events itself: they hold the kernel's own begin and
 * end timestamps, and nothing is inserted into the stream around it. */
int gdg_profile_enable(gdg_ctx *ctx, int enable);
/* Bracket only every `every`-th process call (gdg_process_device / _window_device / host-buffer calls) while profiling is on: an event
 * pair costs a few microseconds AND keeps the bracketed kernel from overlapping its neighbours' ramp-up and tail, which is 5 % of a
 * 0.6 ms step when the dominant kernel of every step is bracketed.  1 = every call (default). */
int gdg_profile_sample(gdg_ctx *ctx, int every);
/* Drain the recorded pairs: total milliseconds and launch count of one kernel kind; resets it. */
int gdg_profile_read(gdg_ctx *ctx, int kind, double *total_ms, int *launches);

/* ---- tuner: tuner.Process / tuner.Analyze, one tuner per channel of the shard ------------------ */

typedef struct {
    double frequency;     /* tuner.Result.Frequency() */
    int32_t note_index;   /* index into the 61-note table (tuner/tuner.go:79-324), -1 = "Unknown" */
    int8_t cents;         /* tuner.Result.Cents() (truncated, tuner.go:557) */
} gdg_tuner_result;

/* tuner.Process for every channel: enqueue `frames` samples per channel into the 96000-sample rings. */
int gdg_tuner_enqueue(gdg_ctx *ctx, const double *const *samples, int frames, uint32_t sample_rate);
int gdg_tuner_enqueue_device(gdg_ctx *ctx, const double *d_samples, int frames, uint32_t sample_rate);
/* the same from the pinned INPUT slab of gdg_staging_buffers (row c = channel c): for hosts that may not hand over their own
 * pointers (the Go overlay of tuner.Tuner copies in[tunerChannel] into row 0 of a one-channel context) */
int gdg_tuner_enqueue_staged(gdg_ctx *ctx, int frames, uint32_t sample_rate);
/* One channel's whole ring at once: `n` must be the ring's length, 96000 (tuner/tuner.go:16 NUM_SAMPLES; anything else is GDG_ERR_INVALID),
 * samples oldest first -- what circular.Buffer.Retrieve hands out (tuner.go:392-399).  For a host that keeps the ring itself and only
 * analyses on the device (the Go overlay of tuner.Tuner: Process stays a host-side enqueue under the reference's lock): one upload per
 * analysis instead of twelve staged blocks.  Host buffer, blocks until it is consumed. */
int gdg_tuner_replace(gdg_ctx *ctx, int channel, const double *samples, int n, uint32_t sample_rate);
/* tuner.Analyze for every channel; results has n_channels entries.  Blocks. */
int gdg_tuner_analyze(gdg_ctx *ctx, gdg_tuner_result *results);
const char *gdg_tuner_note_name(int note_index);

/* ---- spatializer: partial N -> 2 mixdown of this shard ----------------------------------------- */

int gdg_spatializer_set_position(gdg_ctx *ctx, int channel, double azimuth, double distance, double level);
/* spatializer.SetSampleRate (rebuilds the history buffers; keeps the reference's 96000 quirk). */
int gdg_spatializer_set_sample_rate(gdg_ctx *ctx, uint32_t rate);
/*
 * spatializer.Process over the shard's channels WITHOUT the aux input: the host adds the
 * partial left/right pairs of all shards and then the aux buffer (spatializer.go:300-310).
 */
int gdg_spatialize(gdg_ctx *ctx, const double *const *in, double *out_left, double *out_right, int frames);
int gdg_spatialize_device(gdg_ctx *ctx, const double *d_in, double *d_out_lr, int frames);
/*
 * The same for hosts with the cgo pointer rules.  from_outputs == 0: the inputs are the rows of the pinned INPUT slab
 * (gdg_staging_buffers).  from_outputs != 0: the inputs are the chain outputs of the last gdg_process_staged, which are
 * still on the device -- controller.process() mixes exactly those (controller.go:2744-2761), so nothing is uploaded again.
 * out_left / out_right: `frames` float64 each, host memory.
 */
int gdg_spatialize_staged(gdg_ctx *ctx, int from_outputs, double *out_left, double *out_right, int frames);

/* ---- data formats either side of the path ---------------------------------------- */

/* wave/wave.go:51-60 sample formats x bit depths the reference reads and writes */
enum gdg_wave_format { GDG_FMT_LPCM8 = 0, GDG_FMT_LPCM16, GDG_FMT_LPCM24, GDG_FMT_LPCM32, GDG_FMT_IEEE32, GDG_FMT_IEEE64, GDG_FMT_COUNT };
int gdg_wave_bytes_per_sample(int format);      /* 0 for an unknown format */
/*
 * bytesToSamples + samplesToChannels (wave/wave.go:790-838, :237-270): the data section of a RIFF/WAVE file
 * (interleaved, little endian, `channels` x `samples_per_channel` samples) -> planar float64
 * [channels][samples_per_channel].  Bit exact with the reference.  Header parsing stays with the host.
 */
int gdg_wave_decode(gdg_ctx *ctx, int format, const void *bytes, size_t samples_per_channel, unsigned channels, double *samples);
int gdg_wave_decode_device(gdg_ctx *ctx, int format, const void *d_bytes, size_t samples_per_channel, unsigned channels, double *d_samples);
/* channelsToSamples + samplesToBytes (wave/wave.go:173-232, :737-785): the inverse, including the clipping rules. */
int gdg_wave_encode(gdg_ctx *ctx, int format, const double *samples, size_t samples_per_channel, unsigned channels, void *bytes);
int gdg_wave_encode_device(gdg_ctx *ctx, int format, const double *d_samples, size_t samples_per_channel, unsigned channels, void *d_bytes);

/* resample.Time (resample/resample.go:72-103): Lanczos-3 rate conversion; the length rule is :72-87. */
int gdg_resample_time_length(int input_length, uint32_t source_rate, uint32_t target_rate);
int gdg_resample_time(gdg_ctx *ctx, const double *samples, int n, uint32_t source_rate, uint32_t target_rate, double *out, int n_out);
int gdg_resample_time_device(gdg_ctx *ctx, const double *d_samples, int n, uint32_t source_rate, uint32_t target_rate, double *d_out, int n_out);

/*
 * level.Meter (level/level.go): n_ports independent channel meters (the reference runs 2N+3 of them:
 * inputs, outputs, master L/R, metronome).  gdg_meter_configure (re)creates them disabled and cleared;
 * process = level.go:147-210 for every enabled port over one buffer each; analyze = level.go:100-145
 * (integer dB, -200 floor).
 */
int gdg_meter_configure(gdg_ctx *ctx, int n_ports);
int gdg_meter_set_enabled(gdg_ctx *ctx, int port, int enabled);        /* port < 0: all ports (level.go:260-279) */
int gdg_meter_process(gdg_ctx *ctx, const double *const *buffers, int frames, uint32_t sample_rate);
int gdg_meter_process_device(gdg_ctx *ctx, const double *d_rows, size_t row_stride, int frames, uint32_t sample_rate);
int gdg_meter_analyze(gdg_ctx *ctx, int32_t *levels, int32_t *peaks);
/* raw meter state of one port (current value, held peak, hold counter) for parity tests */
int gdg_meter_state(gdg_ctx *ctx, int port, double *current, double *peak, uint64_t *counter);

/*
 * metronome.Metronome (metronome/metronome.go): SetTick / SetTock (NULL = no sound), SetBeatsPerPeriod / SetSpeed /
 * SetSampleRate (none of them touches the counters, as in the reference), Process (:63-131) into one output buffer.
 */
int gdg_metronome_set_tick(gdg_ctx *ctx, const double *coefficients, int n);
int gdg_metronome_set_tock(gdg_ctx *ctx, const double *coefficients, int n);
int gdg_metronome_configure(gdg_ctx *ctx, uint32_t beats_per_period, uint32_t bpm_speed, uint32_t sample_rate);
int gdg_metronome_process(gdg_ctx *ctx, double *out, int frames);
int gdg_metronome_process_device(gdg_ctx *ctx, double *d_out, int frames);

/* ---- the batch run: controller.processFiles between "the files are read" and "the files are written" ------------- */

/*
 * One input of the batch run (controller/controller.go:2884-2990): the data section of a RIFF/WAVE file -- `channels` interleaved
 * channels of `samples_per_channel` samples in `format` -- of which channel `channel` feeds the input.  bytes == NULL or
 * samples_per_channel == 0 is the reference's "leaving channel empty" (silence; its rate does not matter).
 */
typedef struct {
    const void *bytes;
    size_t samples_per_channel;
    int format;                 /* enum gdg_wave_format */
    uint32_t sample_rate;
    unsigned channels, channel;
} gdg_batch_input;

typedef struct {
    uint32_t target_rate;       /* the session rate every input is resampled to (resample.Time, controller.go:2991-3003) */
    int out_format;             /* enum gdg_wave_format of the N + 3 outputs (the "lpcm" / "float" + bit depth prompts, :2821-2880) */
    int metronome_to_master;    /* metrMasterOutput: the metronome is the spatializer's aux input (:2744-2761) */
    int run_meters;             /* levelMeterEnabled: meters over the 2N + 3 ports configured with gdg_meter_configure (:2707-2781) */
    int tuner_enqueue;          /* != 0: every block also goes into the tuner rings (tuner.Process, :2668-2672) */
} gdg_batch_options;

/* samples of every output: the longest resampled input, rounded up to a multiple of BLOCK_SIZE = 8192 (controller.go:3005-3045) */
int gdg_batch_length(gdg_ctx *ctx, const gdg_batch_input *inputs, int n_inputs, uint32_t target_rate, size_t *samples);
/*
 * The whole batch on the device: decode -> resample.Time (inputs whose rate differs from the target) -> zero-pad -> for every
 * 8192-frame block: N x Chain.Process, metronome, spatializer (+ aux), meters -> encode.  n_inputs must equal the context's channel
 * count and max_frames must be >= 8192.  out_bytes: N + 3 host buffers (out_0 .. out_{N-1}, master_left, master_right, metronome,
 * controller.go:3123-3219; NULL = "skipping output") of gdg_batch_length() * gdg_wave_bytes_per_sample(out_format) bytes each.
 * Only file bytes cross PCIe: the samples stay in HBM from decode to encode.  Chains, spatializer positions, metronome and meters
 * are whatever was configured on the context; their state carries on from earlier calls, like the reference's.
 * The block loop runs in steps of up to gdg_ctx_set_window() blocks (a long run opens with a quarter and a half window, its tail
 * runs in halves down to one block: step sizes change the time blocking, never a sample).  The host side of the call -- gathering the
 * next steps' input bytes, scattering a finished step's output bytes -- runs beside the device on the calling thread, on the
 * context's copy workers (option copy_threads; two sets) and, for runs of more than three steps of four blocks or more, on ONE
 * helper thread that lives for the call; inputs and out_bytes are only read / written during the call.
 */
int gdg_batch_run(gdg_ctx *ctx, const gdg_batch_input *inputs, int n_inputs, const gdg_batch_options *options, void *const *out_bytes);
/*
 * The batch run of ONE SHARD of a job whose channels are split over several contexts / GPUs (contiguous channel blocks,
 * no collective).  The master mix is the sum over ALL channels, then the aux input, then the encoder's clip
 * (spatializer/spatializer.go:300-310, controller/controller.go:3123-3219); encoding a shard's partial mix would clip and truncate
 * before the sum.  So a shard hands out its partial sums as float64 and the caller finishes the master on any one context:
 *   for every shard g (in parallel, one context each):  gdg_batch_run_shard(ctx_g, inputs of g's channels, ..., out_bytes_g, &shard_g)
 *   then once:  gdg_batch_finish_master(ctx_0, fmt, {left_g}, {right_g}, G, aux, samples, rate, meters, master_left, master_right)
 * gdg_batch_run_shard = gdg_batch_run except: out_bytes holds the shard's n_inputs chain outputs only; master_left / master_right
 * receive gdg_batch_length() float64 samples each (this shard's channels mixed, NO aux, not clipped); the metronome runs on the
 * shard that passes metronome_bytes (its encoded track, the N + 3rd file) and / or metronome (the float64 track = the master's aux
 * input when metrMasterOutput is set) -- exactly one shard should; options->metronome_to_master must be 0 here (GDG_ERR_INVALID
 * otherwise: the aux input joins the master once, as `aux` of gdg_batch_finish_master); with run_meters the context carries 2 n + 3 ports of which a
 * shard feeds its inputs, its outputs and, if it runs it, the metronome.
 * gdg_batch_finish_master: master = ((p_0 + p_1) + ... + p_{G-1}) + aux per side, summed and encoded on ctx's device (aux may be
 * NULL; left_bytes / right_bytes NULL = "skipping output").  The shards' partial sums are associated differently from the single
 * context's sum over all channels (groups of 16 channels per shard, then shard order), so the sharded master equals gdg_batch_run's
 * to ~1e-16 relative, not bit for bit: a sample that sits on a 24- or 32-bit code boundary may come out one code apart.
 * run_meters != 0 feeds the two LAST ports of ctx's meters with the
 * finished master, block by block.
 */
typedef struct {
    double *master_left, *master_right;   /* host, gdg_batch_length() float64 each */
    void *metronome_bytes;                /* host, gdg_batch_length() encoded samples, or NULL */
    double *metronome;                    /* host, gdg_batch_length() float64, or NULL */
    size_t job_samples;                   /* samples of every output of the JOB (the longest gdg_batch_length over the shards: the
                                           * reference pads every channel to the longest input, controller.go:3005-3045); 0 = this
                                           * shard's own length */
} gdg_batch_shard_out;
int gdg_batch_run_shard(gdg_ctx *ctx, const gdg_batch_input *inputs, int n_inputs, const gdg_batch_options *options, void *const *out_bytes,
                        const gdg_batch_shard_out *shard);
int gdg_batch_finish_master(gdg_ctx *ctx, int out_format, const double *const *left, const double *const *right, int n_shards, const double *aux,
                            size_t samples, uint32_t sample_rate, int run_meters, void *left_bytes, void *right_bytes);
/* The device buffers of a batch run (the decoded inputs are the large part: N x length x 8 bytes) stay with the context for the next
 * run of the same or a smaller size; this frees them. */
int gdg_batch_release(gdg_ctx *ctx);

#ifdef __cplusplus


}
#endif
#endif
