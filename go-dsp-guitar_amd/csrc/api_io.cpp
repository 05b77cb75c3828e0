/*
 * api_io.cpp -- either side of the path: wave sample codecs, resample.Time, level meters, power-amp compilation, metronome.
 * Part of the host side of libgdg.so (the C-ABI of include/gdg.h on top of the HIP kernels; see ctx.h for the map).
 * There is no CPU compute path here: every sample is produced by a HIP kernel.
 */
#include "ctx.h"

int ensure_io(gdg_ctx *ctx, int which, size_t bytes) {
    if (ctx->io_cap[which] >= bytes) return GDG_OK;
    if (ctx->d_io[which]) { hipStreamSynchronize(ctx->stream); hipFree(ctx->d_io[which]); ctx->d_io[which] = nullptr; ctx->io_cap[which] = 0; }
    size_t cap = bytes + bytes / 4 + 4096;
    if (hipMalloc(&ctx->d_io[which], cap) != hipSuccess) return fail(ctx, GDG_ERR_NOMEM, "cannot allocate %zu bytes of io scratch", cap);
    ctx->io_cap[which] = cap;
    return GDG_OK;
}

int gdg_wave_bytes_per_sample(int format) {
    static const int w[GDG_FMT_COUNT] = { 1, 2, 3, 4, 4, 8 };
    return (format >= 0 && format < GDG_FMT_COUNT) ? w[format] : 0;
}

int gdg_wave_decode_device(gdg_ctx *ctx, int format, const void *d_bytes, size_t per, unsigned channels, double *d_samples) {
    if (!ctx) return GDG_ERR_INVALID;
    if (!gdg_wave_bytes_per_sample(format)) return fail(ctx, GDG_ERR_UNSUPPORTED, "unknown sample format %d", format);
    if (channels == 0) return fail(ctx, GDG_ERR_INVALID, "channel count must be positive");
    if (per == 0) return GDG_OK;
    if (!d_bytes || !d_samples) return GDG_ERR_INVALID;
    enter(ctx);
    ProfScope ps(ctx, GDG_K_WAVE);
    HIP_TRY(ctx, gdg_launch_wave_decode(format, d_bytes, per, channels, d_samples, ctx->stream));
    return GDG_OK;
}

int gdg_wave_encode_device(gdg_ctx *ctx, int format, const double *d_samples, size_t per, unsigned channels, void *d_bytes) {
    if (!ctx) return GDG_ERR_INVALID;
    if (!gdg_wave_bytes_per_sample(format)) return fail(ctx, GDG_ERR_UNSUPPORTED, "unknown sample format %d", format);
    if (channels == 0) return fail(ctx, GDG_ERR_INVALID, "channel count must be positive");
    if (per == 0) return GDG_OK;
    if (!d_bytes || !d_samples) return GDG_ERR_INVALID;
    enter(ctx);
    ProfScope ps(ctx, GDG_K_WAVE);
    HIP_TRY(ctx, gdg_launch_wave_encode(format, d_samples, per, channels, d_bytes, ctx->stream));
    return GDG_OK;
}

int gdg_wave_decode(gdg_ctx *ctx, int format, const void *bytes, size_t per, unsigned channels, double *samples) {
    if (!ctx) return GDG_ERR_INVALID;
    int w = gdg_wave_bytes_per_sample(format);
    if (!w) return fail(ctx, GDG_ERR_UNSUPPORTED, "unknown sample format %d", format);
    if (channels == 0) return fail(ctx, GDG_ERR_INVALID, "channel count must be positive");
    size_t n = per * channels;
    if (n == 0) return GDG_OK;
    if (!bytes || !samples) return GDG_ERR_INVALID;
    enter(ctx);
    int rc = ensure_io(ctx, 0, n * w);
    if (rc == GDG_OK) rc = ensure_io(ctx, 1, n * sizeof(double));
    if (rc != GDG_OK) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_io[0], bytes, n * w, hipMemcpyHostToDevice, ctx->stream));
    rc = gdg_wave_decode_device(ctx, format, ctx->d_io[0], per, channels, static_cast<double *>(ctx->d_io[1]));
    if (rc != GDG_OK) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(samples, ctx->d_io[1], n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return GDG_OK;
}

int gdg_wave_encode(gdg_ctx *ctx, int format, const double *samples, size_t per, unsigned channels, void *bytes) {
    if (!ctx) return GDG_ERR_INVALID;
    int w = gdg_wave_bytes_per_sample(format);
    if (!w) return fail(ctx, GDG_ERR_UNSUPPORTED, "unknown sample format %d", format);
    if (channels == 0) return fail(ctx, GDG_ERR_INVALID, "channel count must be positive");
    size_t n = per * channels;
    if (n == 0) return GDG_OK;
    if (!bytes || !samples) return GDG_ERR_INVALID;
    enter(ctx);
    int rc = ensure_io(ctx, 0, n * w);
    if (rc == GDG_OK) rc = ensure_io(ctx, 1, n * sizeof(double));
    if (rc != GDG_OK) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_io[1], samples, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    rc = gdg_wave_encode_device(ctx, format, static_cast<const double *>(ctx->d_io[1]), per, channels, ctx->d_io[0]);
    if (rc != GDG_OK) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(bytes, ctx->d_io[0], n * w, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return GDG_OK;
}

/* resample/resample.go:72-87 */
int gdg_resample_time_length(int input_length, uint32_t source_rate, uint32_t target_rate) {
    if (input_length < 0 || source_rate == 0 || target_rate == 0) return -1;
    double expansion = (double)target_rate / (double)source_rate;
    double out_len_f = (double)input_length * expansion;
    double out_len_floor = floor(out_len_f);
    int out_len = (int)out_len_floor;
    if (out_len_floor == out_len_f) out_len--;
    return out_len < 0 ? 0 : out_len;
}

int gdg_resample_time_device(gdg_ctx *ctx, const double *d_samples, int n, uint32_t source_rate, uint32_t target_rate, double *d_out, int n_out) {
    if (!ctx) return GDG_ERR_INVALID;
    if (source_rate == 0 || target_rate == 0 || n < 0) return fail(ctx, GDG_ERR_INVALID, "invalid rates or length");
    if (n_out != gdg_resample_time_length(n, source_rate, target_rate))
        return fail(ctx, GDG_ERR_INVALID, "output length %d does not follow the reference's length rule (%d)", n_out,
                    gdg_resample_time_length(n, source_rate, target_rate));
    if (n_out == 0) return GDG_OK;
    if (!d_samples || !d_out) return GDG_ERR_INVALID;
    enter(ctx);
    double dx = (double)source_rate / (double)target_rate;       /* resample.go:88-90 */
    ProfScope ps(ctx, GDG_K_RESAMPLE);
    HIP_TRY(ctx, gdg_launch_resample_time(d_samples, n, dx, d_out, n_out, ctx->stream));
    return GDG_OK;
}

int gdg_resample_time(gdg_ctx *ctx, const double *samples, int n, uint32_t source_rate, uint32_t target_rate, double *out, int n_out) {
    if (!ctx) return GDG_ERR_INVALID;
    if (n_out == 0 && n >= 0 && source_rate && target_rate && gdg_resample_time_length(n, source_rate, target_rate) == 0) return GDG_OK;
    if (!samples || !out || n <= 0 || n_out < 0) return fail(ctx, GDG_ERR_INVALID, "invalid buffers");
    enter(ctx);
    int rc = ensure_io(ctx, 0, (size_t)n * sizeof(double));
    if (rc == GDG_OK) rc = ensure_io(ctx, 1, (size_t)(n_out > 0 ? n_out : 1) * sizeof(double));
    if (rc != GDG_OK) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(ctx->d_io[0], samples, (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    rc = gdg_resample_time_device(ctx, static_cast<const double *>(ctx->d_io[0]), n, source_rate, target_rate, static_cast<double *>(ctx->d_io[1]), n_out);
    if (rc != GDG_OK) return rc;
    if (n_out > 0) HIP_TRY(ctx, hipMemcpyAsync(out, ctx->d_io[1], (size_t)n_out * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return GDG_OK;
}

/* ---- level meters ----------------------------------------------------------------------------------------- */
#define METER_PEAK_HOLD_SECONDS 2       /* level/level.go:12 */
#define METER_TIME_CONSTANT 1.7         /* level/level.go:13 */
#define METER_MIN_LEVEL (-200.0)        /* level/level.go:14 */

int gdg_meter_configure(gdg_ctx *ctx, int n_ports) {
    if (!ctx || n_ports < 0) return GDG_ERR_INVALID;
    enter(ctx);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (ctx->d_meter) { hipFree(ctx->d_meter); ctx->d_meter = nullptr; }
    ctx->n_meter = 0;
    if (n_ports == 0) return GDG_OK;
    if (hipMalloc(&ctx->d_meter, (size_t)n_ports * sizeof(gdg_meter_rec)) != hipSuccess) return fail(ctx, GDG_ERR_NOMEM, "cannot allocate meter state");
    HIP_TRY(ctx, hipMemset(ctx->d_meter, 0, (size_t)n_ports * sizeof(gdg_meter_rec)));
    ctx->n_meter = n_ports;
    return GDG_OK;
}

int gdg_meter_set_enabled(gdg_ctx *ctx, int port, int enabled) {
    if (!ctx) return GDG_ERR_INVALID;
    if (port >= ctx->n_meter) return fail(ctx, GDG_ERR_INVALID, "meter port %d out of range (%d configured)", port, ctx->n_meter);
    if (ctx->n_meter == 0) return GDG_OK;
    enter(ctx);
    std::vector<gdg_meter_rec> st(ctx->n_meter);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(st.data(), ctx->d_meter, st.size() * sizeof(gdg_meter_rec), hipMemcpyDeviceToHost));
    int lo = port < 0 ? 0 : port, hi = port < 0 ? ctx->n_meter : port + 1;
    for (int p = lo; p < hi; p++) {
        if ((enabled != 0) == (st[p].enabled != 0)) continue;           /* level.go:264: only a change acts */
        if (!enabled) { st[p].current = 0.0; st[p].peak = 0.0; st[p].counter = 0; }
        st[p].enabled = enabled != 0;
    }
    HIP_TRY(ctx, hipMemcpy(ctx->d_meter, st.data(), st.size() * sizeof(gdg_meter_rec), hipMemcpyHostToDevice));
    return GDG_OK;
}

/* ports [port0, port0 + n_ports) over one buffer each (rows of d_rows) */
int meter_rows(gdg_ctx *ctx, const double *d_rows, size_t row_stride, int port0, int n_ports, int frames, uint32_t sample_rate) {
    double sr = (double)sample_rate;                                   /* level.go:166-171 */
    unsigned long long hold = (unsigned long long)(METER_PEAK_HOLD_SECONDS * sr);
    double decay = pow(10.0, -1.0 / (METER_TIME_CONSTANT * sr));
    int seg = GDG_METER_SEG;
    if ((unsigned long long)seg > hold) seg = (int)hold;             /* the kernel's single-record argument needs n <= hold */
    ProfScope ps(ctx, GDG_K_METER);
    for (int off = 0; off < frames; off += seg) {
        int n = frames - off < seg ? frames - off : seg;
        HIP_TRY(ctx, gdg_launch_meter(d_rows + off, row_stride, n_ports, n, ctx->d_meter + port0, decay, hold, ctx->stream));
    }
    return GDG_OK;
}

int gdg_meter_process_device(gdg_ctx *ctx, const double *d_rows, size_t row_stride, int frames, uint32_t sample_rate) {
    if (!ctx) return GDG_ERR_INVALID;
    if (ctx->n_meter == 0 || frames == 0) return GDG_OK;
    if (!d_rows || frames < 0 || sample_rate == 0) return fail(ctx, GDG_ERR_INVALID, "invalid meter input");
    enter(ctx);
    return meter_rows(ctx, d_rows, row_stride, 0, ctx->n_meter, frames, sample_rate);
}

int gdg_meter_process(gdg_ctx *ctx, const double *const *buffers, int frames, uint32_t sample_rate) {
    if (!ctx) return GDG_ERR_INVALID;
    if (ctx->n_meter == 0 || frames == 0) return GDG_OK;
    if (!buffers || frames < 0) return GDG_ERR_INVALID;
    enter(ctx);
    int rc = ensure_io(ctx, 1, (size_t)ctx->n_meter * frames * sizeof(double));
    if (rc != GDG_OK) return rc;
    double *d = static_cast<double *>(ctx->d_io[1]);
    for (int p = 0; p < ctx->n_meter; p++) {
        if (!buffers[p]) return fail(ctx, GDG_ERR_INVALID, "meter buffer %d is null", p);
        HIP_TRY(ctx, hipMemcpyAsync(d + (size_t)p * frames, buffers[p], (size_t)frames * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
    }
    rc = gdg_meter_process_device(ctx, d, (size_t)frames, frames, sample_rate);
    if (rc != GDG_OK) return rc;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return GDG_OK;
}

static int32_t to_decibels_int(double value) {                         /* level.go:100-118 */
    double level = 20.0 * log10(value);
    if (std::isnan(level) || level < METER_MIN_LEVEL) level = METER_MIN_LEVEL;
    return (int32_t)round(level);
}

int gdg_meter_analyze(gdg_ctx *ctx, int32_t *levels, int32_t *peaks) {
    if (!ctx || !levels || !peaks) return GDG_ERR_INVALID;
    if (ctx->n_meter == 0) return GDG_OK;
    enter(ctx);
    std::vector<gdg_meter_rec> st(ctx->n_meter);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(st.data(), ctx->d_meter, st.size() * sizeof(gdg_meter_rec), hipMemcpyDeviceToHost));
    for (int p = 0; p < ctx->n_meter; p++) { levels[p] = to_decibels_int(st[p].current); peaks[p] = to_decibels_int(st[p].peak); }
    return GDG_OK;
}

int gdg_meter_state(gdg_ctx *ctx, int port, double *current, double *peak, uint64_t *counter) {
    if (!ctx || port < 0 || port >= ctx->n_meter) return GDG_ERR_INVALID;
    enter(ctx);
    gdg_meter_rec st;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    HIP_TRY(ctx, hipMemcpy(&st, ctx->d_meter + port, sizeof(st), hipMemcpyDeviceToHost));
    if (current) *current = st.current;
    if (peak) *peak = st.peak;
    if (counter) *counter = st.counter;
    return GDG_OK;
}

/* ---- power-amp filter compilation on the device (effects/poweramp.go:25-127) ----------------------------------------------- */

int gdg_unit_compile_fir(gdg_ctx *ctx, int handle, int n_filters, const double *const *taps, const int *lengths, const double *gain_compensation,
                         const int32_t *levels_db, uint32_t target_order) {
    Unit *u = get_unit(ctx, handle);
    if (!u) return fail(ctx, GDG_ERR_INVALID, "bad unit handle %d", handle);
    if (u->type != GDG_UNIT_POWERAMP) return fail(ctx, GDG_ERR_INVALID, "unit %d is not a power amp", handle);
    if (n_filters < 0 || (n_filters > 0 && (!taps || !lengths || !gain_compensation || !levels_db))) return fail(ctx, GDG_ERR_INVALID, "bad filter list");
    enter(ctx);
    /* lengths after Reduce, composite length = the longest (filter.go:167-236 Add pads with zeros) */
    size_t max_in = 0, max_out = 0, work_points = 0, pos_points = 0;
    for (int i = 0; i < n_filters; i++) {
        if (!taps[i] || lengths[i] <= 0) continue;                  /* "- NONE -" slot (poweramp.go:78) */
        size_t n = (size_t)lengths[i];
        size_t out = (target_order > 0 && n > (size_t)target_order) ? (size_t)target_order : n;
        if (out != n) {
            size_t w, p;
            gdg_filter_reduce_sizes(lengths[i], target_order, &w, &p);
            if (w > work_points) work_points = w;
            if (p > pos_points) pos_points = p;
        }
        if (n > max_in) max_in = n;
        if (out > max_out) max_out = out;
    }
    std::vector<double> composite(max_out, 0.0);
    if (max_out > 0) {
        /* temporaries from the context's arena (seven hipMalloc / hipFree pairs were 0.7 of a compile's 1.6 ms); every slot has its own
         * upload buffer, so the slots' uploads and kernels queue up behind one another without a host-side wait per slot */
        double *d_in[2] = { nullptr, nullptr }, *d_red = nullptr, *d_comp = nullptr, *d_partial = nullptr;
        double2 *d_wa = nullptr, *d_wb = nullptr, *d_wp = nullptr;
        auto take = [&](void **p, size_t bytes) { return ctx->arena.alloc(p, bytes) == hipSuccess; };
        bool ok = take((void **)&d_in[0], max_in * sizeof(double)) && take((void **)&d_in[1], max_in * sizeof(double));
        ok = ok && take((void **)&d_red, max_out * sizeof(double));
        ok = ok && take((void **)&d_comp, max_out * sizeof(double));
        ok = ok && take((void **)&d_partial, 257 * sizeof(double));
        if (work_points) {
            ok = ok && take((void **)&d_wa, work_points * sizeof(double2));
            ok = ok && take((void **)&d_wb, work_points * sizeof(double2));
            ok = ok && take((void **)&d_wp, pos_points * sizeof(double2));
        }
        hipError_t e = ok ? hipMemsetAsync(d_comp, 0, max_out * sizeof(double), ctx->stream) : hipErrorOutOfMemory;
        int slot = 0;
        for (int i = 0; e == hipSuccess && i < n_filters; i++) {
            if (!taps[i] || lengths[i] <= 0) continue;
            const int n = lengths[i];
            double *d_up = d_in[slot++ & 1];
            /* pageable source: the call returns when the taps have left the caller's buffer; the copy itself is ordered on the stream
             * behind the kernels that read this upload buffer two slots ago */
            e = hipMemcpyAsync(d_up, taps[i], (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream);
            const double *d_cur = d_up;
            int n_cur = n;
            if (e == hipSuccess && target_order > 0 && (size_t)n > (size_t)target_order) {        /* poweramp.go:88-90 */
                e = gdg_launch_filter_reduce(d_up, n, target_order, d_wa, d_wb, d_wp, d_red, ctx->stream);
                d_cur = d_red;
                n_cur = (int)target_order;
            }
            /* Normalize, Multiply(level), Add (poweramp.go:92-94, :108-118) */
            if (e == hipSuccess)
                e = gdg_launch_normalize_scale_add(d_cur, n_cur, gain_compensation[i], decibels_to_factor(levels_db[i]), d_partial, d_comp, ctx->stream);
        }
        if (e == hipSuccess) e = hipMemcpyAsync(composite.data(), d_comp, max_out * sizeof(double), hipMemcpyDeviceToHost, ctx->stream);
        hipError_t e_sync = hipStreamSynchronize(ctx->stream);       /* everything above has run: the temporaries can go back */
        if (e == hipSuccess) e = e_sync;
        ctx->arena.release(d_in[0]); ctx->arena.release(d_in[1]); ctx->arena.release(d_red); ctx->arena.release(d_comp); ctx->arena.release(d_partial);
        ctx->arena.release(d_wa); ctx->arena.release(d_wb); ctx->arena.release(d_wp);
        if (e != hipSuccess) return fail(ctx, e == hipErrorOutOfMemory ? GDG_ERR_NOMEM : GDG_ERR_HIP, "filter compilation failed: %s", hipGetErrorString(e));
    }
    return gdg_unit_set_fir(ctx, handle, composite.data(), (int)composite.size());
}

int gdg_unit_get_fir(gdg_ctx *ctx, int handle, double *taps, int capacity, int *n_taps) {
    Unit *u = get_unit(ctx, handle);
    if (!u) return fail(ctx, GDG_ERR_INVALID, "bad unit handle %d", handle);
    if (u->type != GDG_UNIT_POWERAMP) return fail(ctx, GDG_ERR_INVALID, "unit %d is not a power amp", handle);
    if (n_taps) *n_taps = (int)u->taps.size();
    if (taps) {
        if (capacity < (int)u->taps.size()) return fail(ctx, GDG_ERR_INVALID, "buffer too small for %zu taps", u->taps.size());
        if (!u->taps.empty()) memcpy(taps, u->taps.data(), u->taps.size() * sizeof(double));
    }
    return GDG_OK;
}

/* ---- metronome (metronome/metronome.go) ------------------------------------------------------------------------------------ */

static int set_sound(gdg_ctx *ctx, double **d_buf, uint32_t *n_buf, const double *coeffs, int n) {
    if (n < 0) return fail(ctx, GDG_ERR_INVALID, "bad sound length");
    enter(ctx);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (*d_buf) { hipFree(*d_buf); *d_buf = nullptr; }
    *n_buf = 0;
    if (!coeffs) return GDG_OK;                                    /* SetTick(name, nil): no sound */
    /* a non-nil empty slice is an allocated sound of length 0: keep a one-element allocation so the pointer stays non-null */
    if (hipMalloc((void **)d_buf, (size_t)(n > 0 ? n : 1) * sizeof(double)) != hipSuccess) return fail(ctx, GDG_ERR_NOMEM, "cannot allocate the metronome sound");
    if (n > 0) HIP_TRY(ctx, hipMemcpy(*d_buf, coeffs, (size_t)n * sizeof(double), hipMemcpyHostToDevice));
    *n_buf = (uint32_t)n;
    return GDG_OK;
}

int gdg_metronome_set_tick(gdg_ctx *ctx, const double *coefficients, int n) {
    if (!ctx) return GDG_ERR_INVALID;
    return set_sound(ctx, &ctx->d_tick, &ctx->n_tick, coefficients, n);
}

int gdg_metronome_set_tock(gdg_ctx *ctx, const double *coefficients, int n) {
    if (!ctx) return GDG_ERR_INVALID;
    return set_sound(ctx, &ctx->d_tock, &ctx->n_tock, coefficients, n);
}

int gdg_metronome_configure(gdg_ctx *ctx, uint32_t beats_per_period, uint32_t bpm_speed, uint32_t sample_rate) {
    if (!ctx) return GDG_ERR_INVALID;
    if (bpm_speed == 0) return fail(ctx, GDG_ERR_INVALID, "metronome speed must be positive");     /* the reference would divide by zero */
    ctx->met_beats = beats_per_period;
    ctx->met_bpm = bpm_speed;
    ctx->met_sr = sample_rate;
    return GDG_OK;
}

int gdg_metronome_process_device(gdg_ctx *ctx, double *d_out, int frames) {
    if (!ctx || (frames > 0 && !d_out) || frames < 0) return GDG_ERR_INVALID;
    if (frames == 0) return GDG_OK;
    enter(ctx);
    const uint32_t sc0 = ctx->met_sample_counter, tc0 = ctx->met_tick_counter;
    const uint32_t spb = (60u * ctx->met_sr) / ctx->met_bpm;                    /* metronome.go:79, uint32 arithmetic */
    const uint32_t beats = ctx->met_beats == 0 ? 1u : ctx->met_beats;           /* :84-86 */
    /* sample j0 is the first whose increment reaches samples_per_beat (:122-125) */
    const uint32_t j0 = (sc0 + 1u >= spb) ? 0u : (spb - 1u - sc0);
    HIP_TRY(ctx, gdg_launch_metronome(ctx->d_tick, ctx->n_tick, ctx->d_tock, ctx->n_tock, d_out, frames, sc0, tc0, spb, beats, j0, ctx->stream));
    /* counters after the buffer */
    const uint32_t n = (uint32_t)frames;
    if (n - 1u < j0) { ctx->met_sample_counter = sc0 + n; }
    else {
        uint32_t m = n - j0 - 1u;                                               /* samples after the first reset */
        uint32_t resets = 1u + (spb ? m / spb : m);
        ctx->met_sample_counter = spb ? m % spb : 0u;
        ctx->met_tick_counter = ((tc0 + 1u) % beats + (resets - 1u) % beats) % beats;
    }
    return GDG_OK;
}

int gdg_metronome_process(gdg_ctx *ctx, double *out, int frames) {
    if (!ctx || (frames > 0 && !out) || frames < 0) return GDG_ERR_INVALID;
    if (frames == 0) return GDG_OK;
    enter(ctx);
    int rc = ensure_io(ctx, 1, (size_t)frames * sizeof(double));
    if (rc != GDG_OK) return rc;
    rc = gdg_metronome_process_device(ctx, static_cast<double *>(ctx->d_io[1]), frames);
    if (rc != GDG_OK) return rc;
    HIP_TRY(ctx, hipMemcpyAsync(out, ctx->d_io[1], (size_t)frames * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return GDG_OK;
}

/* ================================================================================================
 * The batch run (controller.processFiles, controller/controller.go:2809-3219, without prompts and file I/O)
 * ============================================================================================== */
