/*
 * api_tuner_spat.cpp -- tuner (tuner.Process / tuner.Analyze) and spatializer (spatializer.Process) glue.
 * Part of the host side of libgdg.so (the C-ABI of include/gdg.h on top of the HIP kernels; see ctx.h for the map).
 * There is no CPU compute path here: every sample is produced by a HIP kernel.
 */
#include "ctx.h"

/* ---- tuner: tuner.Process / tuner.Analyze for every channel of the shard ------------------------------------------ */

static int ensure_tuner(gdg_ctx *ctx) {
    if (ctx->d_tuner_ring) return GDG_OK;
    size_t ring_bytes = (size_t)ctx->nch * GDG_TUNER_RING * sizeof(double);
    HIP_TRY(ctx, hipMalloc((void **)&ctx->d_tuner_ring, ring_bytes));
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_tuner_ring, 0, ring_bytes, ctx->stream));
    HIP_TRY(ctx, hipMalloc((void **)&ctx->d_note_freqs, GDG_NOTE_COUNT * sizeof(double)));
    HIP_TRY(ctx, hipMemcpy(ctx->d_note_freqs, GDG_NOTE_FREQS, GDG_NOTE_COUNT * sizeof(double), hipMemcpyHostToDevice));
    /* the results (16 bytes per channel) are written by the last kernel straight into pinned, device-mapped host memory: no copy command behind
     * the kernels, the caller only waits for the stream (a DMA of a few hundred bytes cost 8-10 us of a 62 us call at 32 channels) */
    /* ... and COHERENT (fine grained): every record ends with the number of its analysis, stored last at system scope, and the caller polls those
     * words instead of waiting for the runtime to notice that the stream is idle (round 6: ~8 of the 20 us a 32-channel call spent outside its kernels) */
    HIP_TRY(ctx, hipHostMalloc((void **)&ctx->h_tuner_out, (size_t)ctx->nch * sizeof(gdg_tuner_out), hipHostMallocMapped | hipHostMallocCoherent));
    memset(ctx->h_tuner_out, 0, (size_t)ctx->nch * sizeof(gdg_tuner_out));
    HIP_TRY(ctx, hipHostGetDevicePointer((void **)&ctx->d_tuner_out, ctx->h_tuner_out, 0));
    ctx->tuner_wp = 0;
    return GDG_OK;
}

int tuner_enqueue_rows(gdg_ctx *ctx, const double *d_samples, size_t stride, int frames, uint32_t sample_rate) {
    int rc = ensure_tuner(ctx);
    if (rc != GDG_OK) return rc;
    if (stride > 0x7fffffff) return fail(ctx, GDG_ERR_INVALID, "row stride %zu too long", stride);
    { ProfScope ps(ctx, GDG_K_TUNER); HIP_TRY(ctx, gdg_launch_tuner_enqueue(ctx->d_tuner_ring, ctx->nch, ctx->tuner_wp, d_samples, (int)stride, frames, ctx->stream)); }
    if (frames < GDG_TUNER_RING) ctx->tuner_wp = (ctx->tuner_wp + frames) % GDG_TUNER_RING;
    ctx->tuner_sr = sample_rate;          /* tuner.go:582-587 */
    return GDG_OK;
}

int gdg_tuner_enqueue_device(gdg_ctx *ctx, const double *d_samples, int frames, uint32_t sample_rate) {
    if (!ctx || !d_samples || frames < 0) return GDG_ERR_INVALID;
    enter(ctx);
    return tuner_enqueue_rows(ctx, d_samples, (size_t)frames, frames, sample_rate);
}


int gdg_tuner_enqueue(gdg_ctx *ctx, const double *const *samples, int frames, uint32_t sample_rate) {
    if (!ctx || !samples) return GDG_ERR_INVALID;
    if (frames < 0 || frames > ctx->max_frames) return fail(ctx, GDG_ERR_INVALID, "frames %d out of range (max %d)", frames, ctx->max_frames);
    enter(ctx);
    int rc = ensure_staging(ctx);
    if (rc != GDG_OK) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (int c = 0; c < ctx->nch; c++) memcpy(ctx->h_stage_in + (size_t)c * frames, samples[c], (size_t)frames * sizeof(double));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_stage_in, ctx->h_stage_in, (size_t)ctx->nch * frames * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    rc = gdg_tuner_enqueue_device(ctx, ctx->d_stage_in, frames, sample_rate);
    if (rc != GDG_OK) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return GDG_OK;
}

int gdg_tuner_enqueue_staged(gdg_ctx *ctx, int frames, uint32_t sample_rate) {
    if (!ctx) return GDG_ERR_INVALID;
    if (frames < 0 || frames > ctx->max_frames) return fail(ctx, GDG_ERR_INVALID, "frames %d out of range (max %d)", frames, ctx->max_frames);
    enter(ctx);
    int rc = ensure_staging(ctx);
    if (rc != GDG_OK) return rc;
    if (frames == 0) return GDG_OK;
    /* pinned rows (stride max_frames) -> compact device rows */
    HIP_TRY(ctx, hipMemcpy2DAsync(ctx->d_stage_in, (size_t)frames * sizeof(double), ctx->h_stage_in, (size_t)ctx->max_frames * sizeof(double),
                                  (size_t)frames * sizeof(double), (size_t)ctx->nch, hipMemcpyHostToDevice, ctx->stream));
    rc = gdg_tuner_enqueue_device(ctx, ctx->d_stage_in, frames, sample_rate);
    if (rc != GDG_OK) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return GDG_OK;
}

/* circular.Buffer.Retrieve -> the device ring in ONE call: the n = 96000 samples of a channel's ring, oldest first, replace the whole ring */
int gdg_tuner_replace(gdg_ctx *ctx, int channel, const double *samples, int n, uint32_t sample_rate) {
    if (!ctx || !samples) return GDG_ERR_INVALID;
    if (channel < 0 || channel >= ctx->nch) return fail(ctx, GDG_ERR_INVALID, "channel %d out of range", channel);
    if (n != GDG_TUNER_RING) return fail(ctx, GDG_ERR_INVALID, "%d samples do not replace a ring of %d (tuner/tuner.go:16 NUM_SAMPLES)", n, GDG_TUNER_RING);
    enter(ctx);
    int rc = ensure_tuner(ctx);
    if (rc != GDG_OK) return rc;
    /* the oldest sample sits at the write position (shared by the context's channels): two pieces around the ring's end */
    double *ring = ctx->d_tuner_ring + (size_t)channel * GDG_TUNER_RING;
    const int wp = ctx->tuner_wp, head = GDG_TUNER_RING - wp;
    HIP_TRY(ctx, hipMemcpyAsync(ring + wp, samples, (size_t)head * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    if (wp > 0) HIP_TRY(ctx, hipMemcpyAsync(ring, samples + head, (size_t)wp * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));           /* the caller's buffer is free again */
    ctx->tuner_sr = sample_rate;
    return GDG_OK;
}

int gdg_tuner_analyze(gdg_ctx *ctx, gdg_tuner_result *results) {
    if (!ctx || !results) return GDG_ERR_INVALID;
    enter(ctx);
    int rc = ensure_tuner(ctx);
    if (rc != GDG_OK) return rc;
    const int force_long = ctx->tuner_long;
    const unsigned seq = ++ctx->tuner_seq ? ctx->tuner_seq : ++ctx->tuner_seq;          /* never 0: what a fresh record holds */
    if (!force_long && gdg_tuner_short_ok((double)ctx->tuner_sr, GDG_NOTE_FREQS[0])) {
        /* every standard rate: block-wise autocorrelation for the lags the analysis can look at; the ring is read once */
        double2 *tw4096, *tw2_4096;
        rc = fir_tables(ctx, 4096, &tw4096, &tw2_4096);
        if (rc != GDG_OK) return rc;
        const int parts = gdg_tuner_short_parts(ctx->nch);
        if (parts > 1 && parts > ctx->tuner_part_cap) {
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            hipFree(ctx->d_tuner_part);
            ctx->d_tuner_part = nullptr;
            HIP_TRY(ctx, hipMalloc((void **)&ctx->d_tuner_part, (size_t)ctx->nch * (size_t)std::max(parts, 8) * 4096 * sizeof(double2)));
            ctx->tuner_part_cap = std::max(parts, 8);
        }
        ProfScope ps(ctx, GDG_K_TUNER);
        HIP_TRY(ctx, gdg_launch_tuner_short(ctx->d_tuner_ring, ctx->nch, ctx->tuner_wp, (double)ctx->tuner_sr, tw4096, tw2_4096,
                                            ctx->d_note_freqs, GDG_NOTE_COUNT, ctx->d_tuner_out, ctx->d_tuner_part, parts, ctx->stream, seq));
    } else {
        /* rates above ~252 kHz: the window reaches past lag 4096 -- the reference's own scheme, a 262144-point transform pair */
        if (!ctx->d_tuner_work) {
            /* two complex work arrays of 131072 points per channel (2 x 2 MiB) */
            HIP_TRY(ctx, hipMalloc((void **)&ctx->d_tuner_work, (size_t)ctx->nch * 2 * (GDG_TUNER_FFT / 2) * sizeof(double2)));
            HIP_TRY(ctx, gdg_tuner_tables_create(&ctx->d_tuner_twn, &ctx->d_tuner_twm));
        }
        double2 *tw512, *tw256, *unused;
        rc = fir_tables(ctx, 512, &tw512, &unused);
        if (rc == GDG_OK) rc = fir_tables(ctx, 256, &tw256, &unused);
        if (rc != GDG_OK) return rc;
        ProfScope ps(ctx, GDG_K_TUNER);
        HIP_TRY(ctx, gdg_launch_tuner_analyze(ctx->d_tuner_ring, ctx->nch, ctx->tuner_wp, (double)ctx->tuner_sr, ctx->d_tuner_work,
                                              ctx->d_tuner_twn, ctx->d_tuner_twm, tw512, tw256, ctx->d_note_freqs, GDG_NOTE_COUNT,
                                              ctx->d_tuner_out, ctx->stream, seq));
    }
    /* The results are written by the last kernel straight into mapped, coherent host memory, each record's analysis number last (system-scope
     * release): the caller polls those words -- a record that carries this call's number is complete -- and falls back to waiting for the stream
     * when they do not show up within a few milliseconds (or when the kernels' events are being collected: profiling reads them at the sync). */
    gdg_tuner_out *host = ctx->h_tuner_out;
    bool seen = false;
    if (!ctx->profiling && ctx->tuner_poll) {
        const auto t0 = std::chrono::steady_clock::now();
        int c = 0, spins = 0;
        for (;;) {
            while (c < ctx->nch && __atomic_load_n(&host[c].seq, __ATOMIC_ACQUIRE) == seq) c++;
            if (c == ctx->nch) { seen = true; break; }
            __builtin_ia32_pause();
            if ((++spins & 255) == 0 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(5)) break;
        }
    }
    if (!seen) HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (int c = 0; c < ctx->nch; c++) {
        results[c].frequency = host[c].frequency;
        results[c].note_index = host[c].note_index;
        results[c].cents = (int8_t)host[c].cents;
    }
    return GDG_OK;
}

const char *gdg_tuner_note_name(int note_index) { return (note_index >= 0 && note_index < GDG_NOTE_COUNT) ? GDG_NOTE_NAMES[note_index] : "Unknown"; }

/* ---- spatializer: spatializer.Process over the shard ---------------------------------------------------------------- */

#define SPAT_GROUP_DELAY 6.3e-4                   /* spatializer/spatializer.go:23 */
#define SPAT_DEFAULT_RATE 96000                   /* spatializer.go:20; the delay computation never leaves this rate */

static int ensure_spatializer(gdg_ctx *ctx) {
    if (ctx->d_sp_hist) return GDG_OK;
    ctx->sp_hist_len = (int)ceil((double)ctx->sp_hist_sr * SPAT_GROUP_DELAY);
    if (ctx->sp_hist_len > 1024)       /* the mix kernel's limit (spat.hip): rates beyond 1.6 MHz */
        return fail(ctx, GDG_ERR_UNSUPPORTED, "spatializer history of %d samples (rate %u Hz): at most 1024 (rates up to 1 625 000 Hz)", ctx->sp_hist_len, ctx->sp_hist_sr);
    size_t hist_bytes = 2 * (size_t)ctx->nch * (size_t)ctx->sp_hist_len * sizeof(double);
    HIP_TRY(ctx, hipMalloc((void **)&ctx->d_sp_hist, hist_bytes));
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_sp_hist, 0, hist_bytes, ctx->stream));
    ctx->sp_hist_cur = 0;
    if (!ctx->d_sp_chan) {
        HIP_TRY(ctx, hipMalloc((void **)&ctx->d_sp_chan, (size_t)ctx->nch * sizeof(gdg_spat_chan)));
        HIP_TRY(ctx, hipMalloc((void **)&ctx->d_sp_out, 2 * (size_t)ctx->max_frames * sizeof(double)));
    }
    ctx->sp_dirty = true;
    return GDG_OK;
}

int gdg_spatializer_set_position(gdg_ctx *ctx, int channel, double azimuth, double distance, double level) {
    if (!ctx) return GDG_ERR_INVALID;
    if (channel < 0 || channel >= ctx->nch) return fail(ctx, GDG_ERR_INVALID, "Cannot set azimuth for channel %d: Only %d channels exist.", channel, ctx->nch);
    if (distance < 0.0 || distance > 10.0) return fail(ctx, GDG_ERR_INVALID, "Failed to set distance: Value must be within [0, 10].");
    if (level < 0.0 || level > 1.0) return fail(ctx, GDG_ERR_INVALID, "Failed to set level: Value must be within [0, 1].");
    ctx->sp_az[(size_t)channel] = azimuth;
    ctx->sp_dist[(size_t)channel] = distance;
    ctx->sp_level[(size_t)channel] = level;
    ctx->sp_dirty = true;
    return GDG_OK;
}

int gdg_spatializer_set_sample_rate(gdg_ctx *ctx, uint32_t rate) {
    if (!ctx) return GDG_ERR_INVALID;
    enter(ctx);
    /* spatializer.go:418-431: new (zeroed) history buffers of ceil(rate * 6.3e-4) samples; this.sampleRate stays 96000 */
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    hipFree(ctx->d_sp_hist);
    ctx->d_sp_hist = nullptr;
    ctx->sp_hist_sr = rate;
    return ensure_spatializer(ctx);
}

static int upload_spat_chans(gdg_ctx *ctx) {
    std::vector<gdg_spat_chan> host((size_t)ctx->nch);
    const double sample_rate = (double)SPAT_DEFAULT_RATE;
    const int H = ctx->sp_hist_len;
    for (int c = 0; c < ctx->nch; c++) {
        /* spatializer.go:170-240, statement by statement */
        double azimuth = GO_MATH_DEGREE_TO_RADIANS * ctx->sp_az[(size_t)c];
        double distance = ctx->sp_dist[(size_t)c], level = ctx->sp_level[(size_t)c];
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
        double xr_sq = x_right * x_right;
        double dist_right = sqrt(xr_sq + y_sq);
        double pre_right = 1.0 / dist_right;
        if (pre_right > 1.0) pre_right = 1.0;
        double dist_diff = dist_left - dist_right;
        double delay_time = GO_GROUP_DELAY_OVER_EFFECTIVE_DISTANCE * dist_diff;
        double delay_samples = fabs(delay_time) * sample_rate;
        double early = floor(delay_samples), late = ceil(delay_samples);
        int early_i = (int)early, late_i = (int)late;
        if (early_i >= H) early_i = H - 1;
        if (late_i >= H) late_i = H - 1;
        gdg_spat_chan &d = host[(size_t)c];
        d.fac_left = level * pre_left;
        d.fac_right = level * pre_right;
        d.w_early = 1.0 - (delay_samples - early);
        d.w_late = 1.0 - (late - delay_samples);
        d.mode = (delay_time == 0.0) ? 0 : (delay_time > 0.0 ? 1 : 2);
        d.early = early_i;
        d.late = late_i;
        d.pad = 0;
    }
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(ctx->d_sp_chan, host.data(), host.size() * sizeof(gdg_spat_chan), hipMemcpyHostToDevice));
    ctx->sp_dirty = false;
    return GDG_OK;
}

/* the one launch of a block: mix + history for the next block (the two history buffers swap) */
static int launch_spatializer(gdg_ctx *ctx, const double *d_in, int in_stride, double *d_left, int out_stride, int frames) {
    const size_t one = (size_t)ctx->nch * (size_t)ctx->sp_hist_len;
    const double *rd = ctx->d_sp_hist + (size_t)ctx->sp_hist_cur * one;
    double *wr = ctx->d_sp_hist + (size_t)(ctx->sp_hist_cur ^ 1) * one;
    ProfScope ps(ctx, GDG_K_SPATIALIZER);
    HIP_TRY(ctx, gdg_launch_spatializer(ctx->d_sp_chan, ctx->nch, d_in, in_stride, rd, wr, ctx->sp_hist_len, d_left, out_stride, frames, ctx->stream));
    ctx->sp_hist_cur ^= 1;
    return GDG_OK;
}

int gdg_spatialize_device(gdg_ctx *ctx, const double *d_in, double *d_out_lr, int frames) {
    if (!ctx || !d_in || !d_out_lr) return GDG_ERR_INVALID;
    if (frames <= 0 || frames > ctx->max_frames) return fail(ctx, GDG_ERR_INVALID, "frames %d out of range (max %d)", frames, ctx->max_frames);
    enter(ctx);
    int rc = ensure_spatializer(ctx);
    if (rc != GDG_OK) return rc;
    if (ctx->sp_dirty) { rc = upload_spat_chans(ctx); if (rc != GDG_OK) return rc; }
    return launch_spatializer(ctx, d_in, frames, d_out_lr, frames, frames);
}

/* one frame out of rows of any stride (the batch run's windows): left to d_left, right to d_left + out_stride */
int spatialize_rows(gdg_ctx *ctx, const double *d_in, int in_stride, double *d_left, int out_stride, int frames) {
    int rc = ensure_spatializer(ctx);
    if (rc != GDG_OK) return rc;
    if (ctx->sp_dirty) { rc = upload_spat_chans(ctx); if (rc != GDG_OK) return rc; }
    return launch_spatializer(ctx, d_in, in_stride, d_left, out_stride, frames);
}

int gdg_spatialize(gdg_ctx *ctx, const double *const *in, double *out_left, double *out_right, int frames) {
    if (!ctx || !in || !out_left || !out_right) return GDG_ERR_INVALID;
    if (frames <= 0 || frames > ctx->max_frames) return fail(ctx, GDG_ERR_INVALID, "frames %d out of range (max %d)", frames, ctx->max_frames);
    enter(ctx);
    int rc = ensure_staging(ctx);
    if (rc == GDG_OK) rc = ensure_spatializer(ctx);
    if (rc != GDG_OK) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (int c = 0; c < ctx->nch; c++) memcpy(ctx->h_stage_in + (size_t)c * frames, in[c], (size_t)frames * sizeof(double));
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_stage_in, ctx->h_stage_in, (size_t)ctx->nch * frames * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    rc = gdg_spatialize_device(ctx, ctx->d_stage_in, ctx->d_sp_out, frames);
    if (rc != GDG_OK) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->h_stage_out, ctx->d_sp_out, 2 * (size_t)frames * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    memcpy(out_left, ctx->h_stage_out, (size_t)frames * sizeof(double));
    memcpy(out_right, ctx->h_stage_out + frames, (size_t)frames * sizeof(double));
    return GDG_OK;
}

int gdg_spatialize_staged(gdg_ctx *ctx, int from_outputs, double *out_left, double *out_right, int frames) {
    if (!ctx || !out_left || !out_right) return GDG_ERR_INVALID;
    if (frames <= 0 || frames > ctx->max_frames) return fail(ctx, GDG_ERR_INVALID, "frames %d out of range (max %d)", frames, ctx->max_frames);
    enter(ctx);
    int rc = ensure_staging(ctx);
    if (rc == GDG_OK) rc = ensure_spatializer(ctx);
    if (rc != GDG_OK) return rc;
    if (ctx->sp_dirty) { rc = upload_spat_chans(ctx); if (rc != GDG_OK) return rc; }
    const double *d_rows = ctx->d_stage_out;                 /* what the last gdg_process / gdg_process_staged left on the device */
    int stride = ctx->stage_out_stride;
    if (from_outputs && stride <= 0)
        return fail(ctx, GDG_ERR_INVALID, "no complete block of chain outputs on the device (the last host-buffer call did not cover all %d channels)", ctx->nch);
    if (from_outputs && frames != ctx->stage_out_frames)
        return fail(ctx, GDG_ERR_INVALID, "the chain outputs on the device are blocks of %d frames, %d were asked for", ctx->stage_out_frames, frames);
    if (!from_outputs) {
        stride = ctx->max_frames;
        HIP_TRY(ctx, hipMemcpy2DAsync(ctx->d_stage_in, (size_t)ctx->max_frames * sizeof(double), ctx->h_stage_in, (size_t)ctx->max_frames * sizeof(double),
                                      (size_t)frames * sizeof(double), (size_t)ctx->nch, hipMemcpyHostToDevice, ctx->stream));
        d_rows = ctx->d_stage_in;
    }
    rc = launch_spatializer(ctx, d_rows, stride, ctx->d_sp_out, frames, frames);
    if (rc != GDG_OK) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(out_left, ctx->d_sp_out, (size_t)frames * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipMemcpyAsync(out_right, ctx->d_sp_out + frames, (size_t)frames * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return GDG_OK;
}

/* ================================================================================================
 * Data formats either side of the path: wave codecs, resample.Time, level meters
 * ============================================================================================== */
