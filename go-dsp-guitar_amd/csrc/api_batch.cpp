/*
 * api_batch.cpp -- controller.processFiles on the device: gdg_batch_run, its sharded form and the master mix.
 * Part of the host side of libgdg.so (the C-ABI of include/gdg.h on top of the HIP kernels; see ctx.h for the map).
 * There is no CPU compute path here: every sample is produced by a HIP kernel.
 */
#include "ctx.h"
#include <future>

#define GDG_BLOCK_SIZE 8192           /* controller/controller.go:36 */

int gdg_batch_length(gdg_ctx *ctx, const gdg_batch_input *inputs, int n_inputs, uint32_t target_rate, size_t *samples) {
    if (!ctx || !inputs || !samples || n_inputs <= 0) return GDG_ERR_INVALID;
    size_t max_len = 0;
    for (int i = 0; i < n_inputs; i++) {
        const gdg_batch_input &in = inputs[i];
        size_t len = (in.bytes && in.samples_per_channel) ? in.samples_per_channel : 0;
        if (len > 0x7fffffff) return fail(ctx, GDG_ERR_INVALID, "input %d is too long", i);
        if (len > 0 && in.sample_rate != target_rate) {                       /* controller.go:2993-2999 */
            int r = gdg_resample_time_length((int)len, in.sample_rate, target_rate);
            len = r > 0 ? (size_t)r : 0;
        }
        if (len > max_len) max_len = len;
    }
    if (max_len % GDG_BLOCK_SIZE) max_len = GDG_BLOCK_SIZE * (max_len / GDG_BLOCK_SIZE + 1);       /* :3014-3016 */
    *samples = max_len;
    return GDG_OK;
}

/* slot i of the batch run's device buffers with at least `bytes` */
static int batch_buffer(gdg_ctx *ctx, int i, size_t bytes, void **out) {
    if (bytes > ctx->batch_dev_cap[i]) {
        hipFree(ctx->batch_dev[i]);
        ctx->batch_dev[i] = nullptr;
        ctx->batch_dev_cap[i] = 0;
        if (hipMalloc(&ctx->batch_dev[i], bytes) != hipSuccess) return fail(ctx, GDG_ERR_NOMEM, "the batch run cannot allocate %zu bytes on the device", bytes);
        ctx->batch_dev_cap[i] = bytes;
    }
    *out = ctx->batch_dev[i];
    return GDG_OK;
}

int gdg_batch_release(gdg_ctx *ctx) {
    if (!ctx) return GDG_ERR_INVALID;
    enter(ctx);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    for (int i = 0; i < 6; i++) { hipFree(ctx->batch_dev[i]); ctx->batch_dev[i] = nullptr; ctx->batch_dev_cap[i] = 0; }
    return GDG_OK;
}

static int ensure_batch_pipe(gdg_ctx *ctx, size_t half_bytes, size_t up_half_bytes) {
    if (!ctx->batch_stream) {
        HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->batch_stream, hipStreamNonBlocking));
        HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->batch_up_stream, hipStreamNonBlocking));
        HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->batch_begin, hipEventDisableTiming));
        for (int h = 0; h < 2; h++) {
            HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->batch_ready[h], hipEventDisableTiming));
            HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->batch_moved[h], hipEventDisableTiming));
            HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->batch_up_ready[h], hipEventDisableTiming));
            for (int c = 0; c < 4; c++) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->batch_chunk[h][c], hipEventDisableTiming));
        }
    }
    if (up_half_bytes > ctx->h_up_cap) {
        for (int h = 0; h < 2; h++) {
            if (ctx->h_up[h]) hipHostFree(ctx->h_up[h]);
            ctx->h_up[h] = nullptr;
        }
        ctx->h_up_cap = 0;
        for (int h = 0; h < 2; h++) HIP_TRY(ctx, pinned_alloc(ctx, (void **)&ctx->h_up[h], up_half_bytes));
        ctx->h_up_cap = up_half_bytes;
    }
    if (half_bytes > ctx->h_batch_cap) {
        for (int h = 0; h < 2; h++) {
            if (ctx->h_batch[h]) hipHostFree(ctx->h_batch[h]);
            ctx->h_batch[h] = nullptr;
        }
        ctx->h_batch_cap = 0;
        for (int h = 0; h < 2; h++) HIP_TRY(ctx, pinned_alloc(ctx, (void **)&ctx->h_batch[h], half_bytes));
        ctx->h_batch_cap = half_bytes;
    }
    return GDG_OK;
}

/* host memcpy pieces (dst, src, bytes), spread over the copy threads */
struct BatchPiece { unsigned char *dst; const unsigned char *src; size_t bytes; };
static void move_pieces(gdg_ctx *ctx, const std::vector<BatchPiece> &pieces, int which = 0) {
    size_t total = 0;
    for (auto &p : pieces) total += p.bytes;
    copy_rows_parallel(ctx, 0, pieces.size(), [&](size_t i) { memcpy(pieces[i].dst, pieces[i].src, pieces[i].bytes); },
                       pieces.empty() ? 0 : total / pieces.size(), which);
}

/*
 * Phases (all device work on the context's stream; PCIe on the copy stream through two pinned halves):
 *   1. the file bytes of all inputs, packed into one arena, go up in half-sized chunks: the copy threads gather chunk k + 1 while
 *      the DMA engine moves chunk k; then one decode (+ resample.Time) per input into its row of the [N][length] inputs.
 *   2. per block: copy in, tuner, N x Chain.Process, metronome, spatializer, meters, encode the block's N + 3 rows in one launch;
 *      the encoded block (N + 3 rows x 8192 x width bytes) goes down on the copy stream while the next block computes, and the
 *      copy threads scatter it into the caller's N + 3 buffers.
 */
static int batch_run_impl(gdg_ctx *ctx, const gdg_batch_input *inputs, int n_inputs, const gdg_batch_options *opt, void *const *out_bytes,
                          const gdg_batch_shard_out *shard) {
    if (!ctx || !inputs || !opt || !out_bytes) return GDG_ERR_INVALID;
    if (n_inputs != ctx->nch) return fail(ctx, GDG_ERR_INVALID, "the batch has %d inputs, the context %d channels", n_inputs, ctx->nch);
    if (ctx->max_frames < GDG_BLOCK_SIZE)
        return fail(ctx, GDG_ERR_INVALID, "the batch loop runs blocks of %d frames, the context allows %d", GDG_BLOCK_SIZE, ctx->max_frames);
    const int out_width = gdg_wave_bytes_per_sample(opt->out_format);
    if (!out_width) return fail(ctx, GDG_ERR_UNSUPPORTED, "unknown sample format %d", opt->out_format);
    if (opt->target_rate == 0) return fail(ctx, GDG_ERR_INVALID, "sample rate must be positive");
    const int N = n_inputs, NO = N + 3, B = GDG_BLOCK_SIZE, ports = 2 * N + 3;
    /* One shard of a job split over several contexts: the master mix is the sum over ALL channels, then the aux input,
     * then the encoder's clip (spatializer.go:300-310, controller.go:3123-3219) -- so a shard hands out its PARTIAL sums as float64
     * and gdg_batch_finish_master adds the shards' partials in shard order, then aux, then encodes.  The metronome runs on the shard
     * that is given somewhere to put it. */
    const bool sharded = shard != nullptr;
    if (sharded && (!shard->master_left || !shard->master_right)) return fail(ctx, GDG_ERR_INVALID, "a shard needs buffers for its partial master mix");
    const bool run_metro = !sharded || shard->metronome_bytes || shard->metronome;
    const int enc_rows = sharded ? N + (shard->metronome_bytes ? 1 : 0) : NO;      /* rows that leave the device encoded */
    const int f64_rows = sharded ? 2 + (shard->metronome ? 1 : 0) : 0;             /* rows that leave it as float64 */
    /* inputs that are mono and already at the target rate are STREAMED: their bytes go up step by step while the block loop runs;
     * the others (a channel picked out of an interleaved file, resample.Time over the whole file) go up before the loop */
    std::vector<size_t> arena_off((size_t)N, 0);
    std::vector<char> streamed((size_t)N, 0);
    size_t arena_bytes = 0, src_cap = 0, up_sample_bytes = 0;
    int n_streamed = 0;
    for (int i = 0; i < N; i++) {
        const gdg_batch_input &in = inputs[i];
        if (!in.bytes || !in.samples_per_channel) continue;
        const int w = gdg_wave_bytes_per_sample(in.format);
        if (!w) return fail(ctx, GDG_ERR_UNSUPPORTED, "input %d: unknown sample format %d", i, in.format);
        if (in.channels == 0 || in.channel >= in.channels) return fail(ctx, GDG_ERR_INVALID, "input %d: channel %u of %u", i, in.channel, in.channels);
        if (in.sample_rate == 0) return fail(ctx, GDG_ERR_INVALID, "input %d: sample rate must be positive", i);
        const size_t count = in.samples_per_channel * in.channels;
        if (in.sample_rate == opt->target_rate && in.channels == 1) {
            streamed[(size_t)i] = 1;
            n_streamed++;
            up_sample_bytes += (size_t)w;
            continue;
        }
        arena_off[(size_t)i] = arena_bytes;
        arena_bytes += (count * (size_t)w + 15) & ~(size_t)15;
        if (count > src_cap) src_cap = count;
    }
    size_t length = 0;
    int rc = gdg_batch_length(ctx, inputs, n_inputs, opt->target_rate, &length);
    if (rc != GDG_OK) return rc;
    if (sharded && shard->job_samples) {
        if (shard->job_samples < length || shard->job_samples % GDG_BLOCK_SIZE)
            return fail(ctx, GDG_ERR_INVALID, "the job's %zu samples: at least this shard's %zu and a multiple of %d", shard->job_samples, length, GDG_BLOCK_SIZE);
        length = shard->job_samples;
    }
    if (length == 0) return GDG_OK;                                            /* every output has 0 samples */
    if (opt->run_meters && ctx->n_meter != ports)
        return fail(ctx, GDG_ERR_INVALID, "level meters: %d ports configured, the batch needs 2 N + 3 = %d (a shard: its N inputs, its N outputs, metronome, left, right)",
                    ctx->n_meter, ports);
    enter(ctx);
    const int W = ctx->window;                                                 /* frames per step (gdg_ctx_set_window; 1 = the reference's loop) */
    const size_t ws = (size_t)W * B;                                           /* row stride of the window, the same for every step */
    const size_t enc_bytes = (((size_t)enc_rows * ws * (size_t)out_width + 15) & ~(size_t)15) + (size_t)f64_rows * ws * sizeof(double);   /* one window on its way down */
    const size_t half = std::max(enc_bytes, (size_t)8 << 20);
    if (length > 0x7fffffff) return fail(ctx, GDG_ERR_INVALID, "files of %zu samples are too long", length);
    /* one step of the streamed inputs: the piece descriptors, then every piece on a 16-byte boundary */
    const size_t up_rows_bytes = ((size_t)n_streamed * sizeof(gdg_decode_row) + 255) & ~(size_t)255;
    const size_t up_half = n_streamed ? up_rows_bytes + up_sample_bytes * (size_t)W * B + 16 * (size_t)n_streamed : 0;
    rc = ensure_batch_pipe(ctx, half, up_half);
    if (rc != GDG_OK) return rc;
    double *d_inputs = nullptr, *d_win = nullptr, *d_src = nullptr;
    unsigned char *d_arena = nullptr, *d_enc = nullptr, *d_up = nullptr;
    static int trace = -1;
    if (trace < 0) { const char *e = getenv("GDG_BATCH_TRACE"); trace = e ? atoi(e) : 0; }
    auto now_ms = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    const double t_begin = now_ms();
    auto body = [&]() -> int {
        int r;
        if ((r = batch_buffer(ctx, 0, (size_t)N * length * sizeof(double), (void **)&d_inputs)) != GDG_OK) return r;
        /* one window of the N + 3 outputs, rows in the output files' order (out_0 .. out_{N-1}, master left, master right, metronome,
         * controller.go:3123-3219); the inputs are read where they lie */
        if ((r = batch_buffer(ctx, 1, (size_t)NO * ws * sizeof(double), (void **)&d_win)) != GDG_OK) return r;
        if ((r = batch_buffer(ctx, 2, 2 * enc_bytes, (void **)&d_enc)) != GDG_OK) return r;
        if (arena_bytes && (r = batch_buffer(ctx, 3, arena_bytes, (void **)&d_arena)) != GDG_OK) return r;
        if (up_half && (r = batch_buffer(ctx, 4, 2 * up_half, (void **)&d_up)) != GDG_OK) return r;
        if (src_cap && (r = batch_buffer(ctx, 5, src_cap * sizeof(double), (void **)&d_src)) != GDG_OK) return r;
        /* the zero padding (:3018-3045): only what no decode / resample will write -- the tail of every row behind its file's samples, the
         * whole row of an empty input (zeroing all N x length samples first cost 1.5 ms of a 60 ms run at 512 x 1 Mi samples) */
        for (int i = 0; i < N; i++) {
            const gdg_batch_input &in = inputs[i];
            size_t covered = 0;
            if (in.bytes && in.samples_per_channel) {
                covered = in.samples_per_channel;
                if (in.sample_rate != opt->target_rate) {
                    int n_out = gdg_resample_time_length((int)in.samples_per_channel, in.sample_rate, opt->target_rate);
                    covered = n_out > 0 ? (size_t)n_out : 0;
                }
                if (covered > length) covered = length;
            }
            if (covered < length)
                HIP_TRY(ctx, hipMemsetAsync(d_inputs + (size_t)i * length + covered, 0, (length - covered) * sizeof(double), ctx->stream));
        }

        /* 1a. the arena goes up */
        int used[2] = { 0, 0 };
        int next_input = 0;
        for (size_t k = 0, lo = 0; lo < arena_bytes; k++, lo += half) {
            const size_t hi = std::min(arena_bytes, lo + half);
            const int h = (int)(k & 1);
            if (used[h]) HIP_TRY(ctx, hipEventSynchronize(ctx->batch_moved[h]));
            std::vector<BatchPiece> pieces;
            while (next_input < N && (!inputs[next_input].bytes || !inputs[next_input].samples_per_channel || streamed[(size_t)next_input])) next_input++;
            for (int i = next_input; i < N; i++) {
                const gdg_batch_input &in = inputs[i];
                if (!in.bytes || !in.samples_per_channel || streamed[(size_t)i]) continue;
                const size_t a = arena_off[(size_t)i], nb = in.samples_per_channel * in.channels * (size_t)gdg_wave_bytes_per_sample(in.format);
                if (a >= hi) break;
                if (a + nb <= lo) { if (i == next_input) next_input++; continue; }
                size_t s0 = std::max(a, lo), s1 = std::min(a + nb, hi);
                for (size_t q = s0; q < s1; q += (size_t)1 << 20)                 /* pieces of <= 1 MiB */
                    pieces.push_back({ ctx->h_batch[h] + (q - lo), static_cast<const unsigned char *>(in.bytes) + (q - a), std::min(s1 - q, (size_t)1 << 20) });
            }
            move_pieces(ctx, pieces);
            HIP_TRY(ctx, hipMemcpyAsync(d_arena + lo, ctx->h_batch[h], hi - lo, hipMemcpyHostToDevice, ctx->batch_stream));
            HIP_TRY(ctx, hipEventRecord(ctx->batch_moved[h], ctx->batch_stream));
            used[h] = 1;
        }
        for (int h = 0; h < 2; h++) if (used[h]) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->batch_moved[h], 0));
        /* 1b. decode (+ resample.Time) every input into its row */
        for (int i = 0; i < N; i++) {
            const gdg_batch_input &in = inputs[i];
            if (!in.bytes || !in.samples_per_channel || streamed[(size_t)i]) continue;        /* "leaving channel empty" / comes with its step */
            const size_t per = in.samples_per_channel;
            double *row = d_inputs + (size_t)i * length;
            if ((r = gdg_wave_decode_device(ctx, in.format, d_arena + arena_off[(size_t)i], per, in.channels, d_src)) != GDG_OK) return r;
            const double *chan = d_src + (size_t)in.channel * per;           /* planar: samplesToChannels */
            if (in.sample_rate == opt->target_rate)
                HIP_TRY(ctx, hipMemcpyAsync(row, chan, per * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
            else {
                int n_out = gdg_resample_time_length((int)per, in.sample_rate, opt->target_rate);
                if (n_out > 0 && (r = gdg_resample_time_device(ctx, chan, (int)per, in.sample_rate, opt->target_rate, row, n_out)) != GDG_OK) return r;
            }
        }
        /* the pinned halves change direction: every upload has been consumed by the DMA engine (events above), nothing else reads them */
        for (int h = 0; h < 2; h++) if (used[h]) HIP_TRY(ctx, hipEventSynchronize(ctx->batch_moved[h]));

        /* 2. the block loop, controller.go:3076-3107 around controller.process (:2648-2783), `w` blocks per step */
        if (ctx->all_channels.empty()) for (int c = 0; c < ctx->nch; c++) ctx->all_channels.push_back(c);
        struct Step { size_t off; int w; };
        std::vector<Step> steps;
        /* A long run opens with a quarter window and a half window: the device has nothing to do until the first step's bytes are gathered and uploaded (16 blocks
         * of 512 files: 1.9 + 2.4 ms), and what it computes first comes down and is scattered while nothing else waits for the host.  The blocks
         * that step leaves over make the tail shorter the same way (W/2, W/4: the last download and scatter are a quarter step's).  Window sizes
         * only change the time blocking, never a sample (tests/test_gpu_window.py). */
        size_t off0 = 0;
        if (W >= 8 && length >= (size_t)3 * W * B) {                             /* W/4, W/2, then whole windows: each step's upload fits behind the step before */
            steps.push_back({ 0, W / 4 });
            steps.push_back({ (size_t)(W / 4) * B, W / 2 });
            off0 = (size_t)(W / 4 + W / 2) * B;
        }
        for (size_t off = off0; off < length;) {
            int w = W;
            while ((size_t)w * B > length - off) w >>= 1;                        /* the tail: windows of W/2, W/4 .. 1 */
            steps.push_back({ off, w });
            off += (size_t)w * B;
        }
        /* A step comes down in `chunks` pieces of whole rows (the float64 rows of a shard ride with the last one), an event behind each: the
         * scatter of piece c runs while piece c + 1 is on the bus -- the run's tail (last download, then last scatter) and its head are that
         * much shorter; in between the device sets the pace either way. */
        auto chunks_of = [&](size_t i) { return (steps[i].w >= 4 && enc_rows >= 8) ? 4 : 1; };
        auto chunk_rows = [&](size_t i, int c) { return (size_t)enc_rows * (size_t)c / (size_t)chunks_of(i); };      /* first encoded row of piece c */
        auto scatter = [&](size_t i) -> int {                                    /* step i's bytes from its pinned half into the files */
            const unsigned char *src = ctx->h_batch[i & 1];
            const size_t wb = (size_t)steps[i].w * B, row_bytes = wb * out_width, at = steps[i].off * out_width;
            const size_t f64_at = ((size_t)enc_rows * row_bytes + 15) & ~(size_t)15;
            const int K = chunks_of(i);
            for (int c = 0; c < K; c++) {
                HIP_TRY(ctx, hipEventSynchronize(ctx->batch_chunk[i & 1][c]));    /* piece c has landed */
                const size_t o0 = chunk_rows(i, c), o1 = (c + 1 == K) ? (size_t)enc_rows + (size_t)f64_rows : chunk_rows(i, c + 1);
                copy_rows_parallel(ctx, o0, o1, [&](size_t o) {
                    if (o < (size_t)enc_rows) {
                        /* NULL: "skipping output" (:3143); a shard's row N is the metronome track */
                        void *dst = (sharded && o == (size_t)N) ? shard->metronome_bytes : out_bytes[o];
                        if (dst) memcpy(static_cast<unsigned char *>(dst) + at, src + o * row_bytes, row_bytes);
                    } else {
                        const size_t k = o - (size_t)enc_rows;
                        double *dst = k == 0 ? shard->master_left : (k == 1 ? shard->master_right : shard->metronome);
                        memcpy(dst + steps[i].off, src + f64_at + k * wb * sizeof(double), wb * sizeof(double));
                    }
                }, row_bytes);
            }
            return GDG_OK;
        };
        /* the streamed inputs of step i: gathered into a pinned half by the copy threads, moved and decoded on the upload stream while
         * the block loop is busy with the steps before */
        HIP_TRY(ctx, hipEventRecord(ctx->batch_begin, ctx->stream));             /* rows zeroed, whole-file inputs decoded */
        if (n_streamed) HIP_TRY(ctx, hipStreamWaitEvent(ctx->batch_up_stream, ctx->batch_begin, 0));
        int up_used[2] = { 0, 0 };
        auto stage = [&](size_t i, int pool = 0) -> int {                        /* pool 1: from the helper thread, with the upload side's copy workers */
            if (!n_streamed || i >= steps.size()) return GDG_OK;
            const int h = (int)(i & 1);
            if (up_used[h]) HIP_TRY(ctx, hipEventSynchronize(ctx->batch_up_ready[h]));     /* step i - 2 has left this half */
            unsigned char *hb = ctx->h_up[h], *db = d_up + (size_t)h * up_half;
            gdg_decode_row *rows = reinterpret_cast<gdg_decode_row *>(hb);
            std::vector<BatchPiece> pieces;
            size_t cur = up_rows_bytes;
            int n_rows = 0;
            unsigned max_count = 0;
            const size_t a = steps[i].off, span = (size_t)steps[i].w * B;
            for (int c = 0; c < N; c++) {
                if (!streamed[(size_t)c]) continue;
                const gdg_batch_input &in = inputs[c];
                if (a >= in.samples_per_channel) continue;                       /* the file ended in an earlier step: zeros */
                const size_t cnt = std::min(in.samples_per_channel - a, span), width = (size_t)gdg_wave_bytes_per_sample(in.format);
                rows[n_rows++] = gdg_decode_row{ db + cur, d_inputs + (size_t)c * length + a, (unsigned)cnt, in.format };
                const unsigned char *src = static_cast<const unsigned char *>(in.bytes) + a * width;
                for (size_t q = 0; q < cnt * width; q += (size_t)1 << 20)
                    pieces.push_back({ hb + cur + q, src + q, std::min(cnt * width - q, (size_t)1 << 20) });
                if (cnt > max_count) max_count = (unsigned)cnt;
                cur += (cnt * width + 15) & ~(size_t)15;
            }
            up_used[h] = 1;
            if (n_rows) {
                move_pieces(ctx, pieces, pool);
                HIP_TRY(ctx, hipMemcpyAsync(db, hb, cur, hipMemcpyHostToDevice, ctx->batch_up_stream));
                HIP_TRY(ctx, gdg_launch_wave_decode_rows(reinterpret_cast<const gdg_decode_row *>(db), n_rows, max_count, ctx->batch_up_stream));
            }
            HIP_TRY(ctx, hipEventRecord(ctx->batch_up_ready[h], ctx->batch_up_stream));
            return GDG_OK;
        };
        /* step i on the compute stream: the block loop's work for its w blocks, then the encoder into the step's half of `enc` */
        auto enqueue_compute = [&](size_t i) -> int {
            const size_t off = steps[i].off;
            const int w = steps[i].w, h = (int)(i & 1), wb = w * B;               /* this step fills the first wb samples of the window's rows */
            if (n_streamed) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->batch_up_ready[h], 0));
            const double *d_in = d_inputs + off;
            double *d_master = d_win + (size_t)N * ws, *d_metro = d_master + 2 * ws;
            unsigned char *enc = d_enc + h * enc_bytes;
            if (opt->tuner_enqueue)
                for (int j = 0; j < w; j++) if ((r = tuner_enqueue_rows(ctx, d_in + (size_t)j * B, length, B, opt->target_rate)) != GDG_OK) return r;
            if ((r = process_rows(ctx, ctx->all_channels, d_in, d_win, B, opt->target_rate, (int)length, false, 1, nullptr, nullptr, w, (int)ws)) != GDG_OK) return r;
            if (run_metro && (r = gdg_metronome_process_device(ctx, d_metro, wb)) != GDG_OK) return r;
            /* the step's w frames are consecutive in their rows: ONE mix over w x 8192 samples gives the samples of w calls (a frame's first
             * samples find their delayed neighbours in the frame before instead of in the history, which holds the same values) */
            if ((r = spatialize_rows(ctx, d_win, (int)ws, d_master, (int)ws, wb)) != GDG_OK) return r;
            /* a shard's master rows stay partial sums: the aux input is added once, after the shards' sums (gdg_batch_finish_master) */
            if (opt->metronome_to_master && !sharded) HIP_TRY(ctx, gdg_launch_add_aux(d_master, d_master + ws, d_metro, wb, ctx->stream));
            if (opt->run_meters) {                                               /* ports: inputs | outputs | metronome | left, right (:2707-2777) */
                if ((r = meter_rows(ctx, d_in, length, 0, N, wb, opt->target_rate)) != GDG_OK) return r;
                if ((r = meter_rows(ctx, d_win, ws, N, N, wb, opt->target_rate)) != GDG_OK) return r;
                if (run_metro && (r = meter_rows(ctx, d_metro, ws, 2 * N, 1, wb, opt->target_rate)) != GDG_OK) return r;
                if (!sharded && (r = meter_rows(ctx, d_master, ws, 2 * N + 1, 2, wb, opt->target_rate)) != GDG_OK) return r;     /* a shard's master ports: gdg_batch_finish_master */
            }
            if (i >= 2) HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->batch_moved[h], 0));     /* step i - 2 has left enc */
            {
                ProfScope ps(ctx, GDG_K_WAVE);
                const size_t row_bytes = (size_t)wb * out_width;
                if (!sharded) HIP_TRY(ctx, gdg_launch_wave_encode_rows(opt->out_format, d_win, ws, (size_t)wb, (unsigned)NO, enc, ctx->stream));
                else {
                    HIP_TRY(ctx, gdg_launch_wave_encode_rows(opt->out_format, d_win, ws, (size_t)wb, (unsigned)N, enc, ctx->stream));
                    if (shard->metronome_bytes)
                        HIP_TRY(ctx, gdg_launch_wave_encode_rows(opt->out_format, d_metro, ws, (size_t)wb, 1u, enc + (size_t)N * row_bytes, ctx->stream));
                    unsigned char *f64 = enc + (((size_t)enc_rows * row_bytes + 15) & ~(size_t)15);
                    HIP_TRY(ctx, hipMemcpy2DAsync(f64, (size_t)wb * sizeof(double), d_master, ws * sizeof(double), (size_t)wb * sizeof(double), 2,
                                                  hipMemcpyDeviceToDevice, ctx->stream));
                    if (shard->metronome)
                        HIP_TRY(ctx, hipMemcpyAsync(f64 + 2 * (size_t)wb * sizeof(double), d_metro, (size_t)wb * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
                }
            }
            HIP_TRY(ctx, hipEventRecord(ctx->batch_ready[h], ctx->stream));
            return GDG_OK;
        };
        /* ... and its way down on the download stream, into the step's pinned half (which step i - 2 must have left: scatter(i - 2) is done) */
        auto enqueue_down = [&](size_t i) -> int {
            const int h = (int)(i & 1), wb = steps[i].w * B;
            unsigned char *enc = d_enc + h * enc_bytes;
            HIP_TRY(ctx, hipStreamWaitEvent(ctx->batch_stream, ctx->batch_ready[h], 0));
            const size_t row_bytes = (size_t)wb * out_width;
            const size_t down = sharded ? (((size_t)enc_rows * row_bytes + 15) & ~(size_t)15) + (size_t)f64_rows * wb * sizeof(double) : (size_t)NO * row_bytes;
            const int K = chunks_of(i);
            for (int c = 0; c < K; c++) {
                const size_t b0 = chunk_rows(i, c) * row_bytes, b1 = (c + 1 == K) ? down : chunk_rows(i, c + 1) * row_bytes;
                if (b1 > b0) HIP_TRY(ctx, hipMemcpyAsync(ctx->h_batch[h] + b0, enc + b0, b1 - b0, hipMemcpyDeviceToHost, ctx->batch_stream));
                HIP_TRY(ctx, hipEventRecord(ctx->batch_chunk[h][c], ctx->batch_stream));
            }
            HIP_TRY(ctx, hipEventRecord(ctx->batch_moved[h], ctx->batch_stream));
            return GDG_OK;
        };
        /* The compute stream is kept TWO steps ahead of the files.  Round 2 enqueued step i + 1 only after step i - 1 had been scattered into
         * the caller's buffers, which closed a loop of compute -> download -> scatter over two steps: (5.7 + 4.0 + 3.1) / 2 = 6.4 ms per step
         * of 16 blocks where the device needs 5.7 (GDG_BATCH_TRACE).  Now step i + 2 is enqueued as soon as step i's download has finished (before
         * its bytes are scattered), while step i + 1 is already queued behind step i on the device. */
        if (trace) fprintf(stderr, "[batch] set-up %.2f ms\n", now_ms() - t_begin);
        /* Gathering step i + 2's input bytes (1.9 ms of 16 blocks x 512 files) and scattering step i's output bytes (3.0 ms) were one thread's
         * work, one after the other: 4.8 ms per step beside the device's 4.7 -- the host set the pace half of the time.  With windows of four
         * blocks or more the gather runs on a helper thread with copy workers of its own (copy_pool_up), one step further ahead (step i + 3 while
         * step i is scattered; its pinned half and its device half were step i + 1's, whose upload and decode are long done -- stage() waits
         * for their event): the host's step is the scatter alone and the device sets the pace. */
        const bool helper = n_streamed && ws >= (size_t)4 * B && steps.size() > 3;
        std::future<int> staged;
        struct Join { std::future<int> &f; ~Join() { if (f.valid()) f.wait(); } } join_on_exit{ staged };     /* stage() captures this frame by reference */
        /* the upload side's copy workers are made HERE, by the caller's thread, and the helper thread goes where they go: with option "numa" = 2
         * workers are bound to the node of the thread that makes them, and a helper the scheduler happened to start on the other socket would
         * put the gather's workers a socket away from the caller's buffers and the pinned halves */
        const std::vector<int> *helper_cpus = nullptr;
        if (helper) { ensure_copy_pool(ctx, 1); numa_target(ctx, &helper_cpus); }
        auto on_helper = [&](size_t first, size_t last) {                        /* stage(first .. last), one after the other, on the helper thread */
            if (first >= steps.size()) return;
            try {
                staged = std::async(std::launch::async, [&, first, last]() -> int {
                    if (helper_cpus) numa_bind_thread(*helper_cpus);
                    if (hipSetDevice(ctx->device) != hipSuccess) return GDG_ERR_HIP;
                    for (size_t k = first; k <= last && k < steps.size(); k++) { const int rr = stage(k, 1); if (rr != GDG_OK) return rr; }
                    return GDG_OK;
                });
            } catch (...) {                                                    /* no thread to be had: gather here, as the short runs do */
                int rr = GDG_OK;
                for (size_t k = first; k <= last && k < steps.size() && rr == GDG_OK; k++) rr = stage(k, 0);
                std::promise<int> done;
                done.set_value(rr);
                staged = done.get_future();
            }
        };
        auto stage_async = [&](size_t i) { on_helper(i, i); };
        if (helper) {
            /* the head: step 0 is gathered here; steps 1 and 2 on the helper meanwhile, so that the first whole window's bytes are on the bus
             * while the quarter and the half window compute */
            if ((r = stage(0)) != GDG_OK) return r;
            on_helper(1, 2);
            if ((r = enqueue_compute(0)) != GDG_OK || (r = enqueue_down(0)) != GDG_OK) return r;
            if ((r = staged.get()) != GDG_OK) return r;
            if ((r = enqueue_compute(1)) != GDG_OK || (r = enqueue_down(1)) != GDG_OK) return r;
        } else {
            for (size_t i = 0; i < 2 && i < steps.size(); i++) {
                if ((r = stage(i)) != GDG_OK) return r;
                if ((r = enqueue_compute(i)) != GDG_OK || (r = enqueue_down(i)) != GDG_OK) return r;
            }
        }
        if (trace) fprintf(stderr, "[batch] steps 0 and 1 staged and enqueued at %.2f ms\n", now_ms() - t_begin);
        for (size_t i = 0; i < steps.size(); i++) {
            const double t_it = now_ms();
            if (helper) {
                if (staged.valid() && (r = staged.get()) != GDG_OK) return r;     /* step i + 2's inputs are on their way up (i = 0: since the head) */
                stage_async(i + 3);
            } else if ((r = stage(i + 2)) != GDG_OK) return r;                   /* while steps i, i + 1 run: the inputs of step i + 2 go up ... */
            const double t_st = now_ms();
            const double t_wait = t_st;
            /* step i + 2 needs step i's half of `enc` (the compute stream waits for its download itself) but not its pinned half: it goes onto
             * the compute stream BEFORE the scatter, so the loop compute -> download -> compute spans 5.7 + 4.0 ms per two steps and the device,
             * not the host, sets the pace */
            if (i + 2 < steps.size() && (r = enqueue_compute(i + 2)) != GDG_OK) return r;
            const double t_enq = now_ms();
            if ((r = scatter(i)) != GDG_OK) return r;                            /* ... step i comes down and goes into the files, piece by piece */
            if (i + 2 < steps.size() && (r = enqueue_down(i + 2)) != GDG_OK) return r;     /* its pinned half is free again */
            if (trace) fprintf(stderr, "[batch] step %zu: stage %zu %.2f | (%.2f) | enqueue %zu %.2f | download + scatter %.2f  (at %.2f ms)\n", i, i + 2,
                               t_st - t_it, t_wait - t_st, i + 2, t_enq - t_wait, now_ms() - t_enq, now_ms() - t_begin);
        }
        return check_device_error(ctx);
    };
    rc = body();
    hipStreamSynchronize(ctx->batch_up_stream);
    hipStreamSynchronize(ctx->batch_stream);
    hipStreamSynchronize(ctx->stream);
    /* the device buffers stay with the context for the next batch (gdg_batch_release) */
    return rc;
}

int gdg_batch_run(gdg_ctx *ctx, const gdg_batch_input *inputs, int n_inputs, const gdg_batch_options *opt, void *const *out_bytes) {
    return batch_run_impl(ctx, inputs, n_inputs, opt, out_bytes, nullptr);
}

int gdg_batch_run_shard(gdg_ctx *ctx, const gdg_batch_input *inputs, int n_inputs, const gdg_batch_options *opt, void *const *out_bytes,
                        const gdg_batch_shard_out *shard) {
    if (!shard) return GDG_ERR_INVALID;
    /* a shard's master mix is a PARTIAL sum: the aux input joins the master once, in gdg_batch_finish_master (its `aux` = the float64
     * metronome track of the shard that ran it).  A set flag here would be silently dropped -- refuse it instead. */
    if (ctx && opt && opt->metronome_to_master)
        return fail(ctx, GDG_ERR_INVALID, "gdg_batch_run_shard: metronome_to_master must be 0 -- a shard's master mix is a partial sum; pass the metronome's float64 "
                    "track (gdg_batch_shard_out.metronome of the shard that runs it) as `aux` to gdg_batch_finish_master");
    return batch_run_impl(ctx, inputs, n_inputs, opt, out_bytes, shard);
}

/* master = ((p_0 + p_1) + ... + p_{G-1}) + aux per side, then the encoder (its clip included) -- all on this context's device; the host
 * only moves the G partial pairs up and the two encoded rows down, in pieces of <= 2^20 samples through the context's io scratch */
int gdg_batch_finish_master(gdg_ctx *ctx, int out_format, const double *const *left, const double *const *right, int n_shards, const double *aux,
                            size_t samples, uint32_t sample_rate, int run_meters, void *left_bytes, void *right_bytes) {
    if (!ctx || !left || !right || n_shards <= 0) return GDG_ERR_INVALID;
    const int width = gdg_wave_bytes_per_sample(out_format);
    if (!width) return fail(ctx, GDG_ERR_UNSUPPORTED, "unknown sample format %d", out_format);
    for (int g = 0; g < n_shards; g++) if (!left[g] || !right[g]) return fail(ctx, GDG_ERR_INVALID, "shard %d has no partial master mix", g);
    if (run_meters && (ctx->n_meter < 2 || sample_rate == 0)) return fail(ctx, GDG_ERR_INVALID, "master meters: the context's last two ports, at a positive rate");
    if (samples == 0) return GDG_OK;
    enter(ctx);
    const size_t piece = (size_t)1 << 20;
    int rc = ensure_io(ctx, 1, 3 * piece * sizeof(double));                     /* [left | right | incoming partial or aux] */
    if (rc == GDG_OK) rc = ensure_io(ctx, 0, 2 * piece * (size_t)width);
    if (rc != GDG_OK) return rc;
    double *d_l = static_cast<double *>(ctx->d_io[1]), *d_r = d_l + piece, *d_p = d_r + piece;
    unsigned char *d_enc = static_cast<unsigned char *>(ctx->d_io[0]);
    for (size_t at = 0; at < samples; at += piece) {
        const size_t n = std::min(piece, samples - at);
        HIP_TRY(ctx, hipMemcpyAsync(d_l, left[0] + at, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(d_r, right[0] + at, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        for (int g = 1; g < n_shards; g++) {
            HIP_TRY(ctx, hipMemcpyAsync(d_p, left[g] + at, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
            HIP_TRY(ctx, gdg_launch_accumulate(d_l, d_p, (int)n, ctx->stream));
            HIP_TRY(ctx, hipMemcpyAsync(d_p, right[g] + at, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
            HIP_TRY(ctx, gdg_launch_accumulate(d_r, d_p, (int)n, ctx->stream));
        }
        if (aux) {                                                               /* spatializer.go:300-310 */
            HIP_TRY(ctx, hipMemcpyAsync(d_p, aux + at, n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
            HIP_TRY(ctx, gdg_launch_add_aux(d_l, d_r, d_p, (int)n, ctx->stream));
        }
        if (run_meters) {
            for (size_t o = 0; o < n; o += GDG_BLOCK_SIZE)                       /* block by block, like the loop that fed the other ports */
                if ((rc = meter_rows(ctx, d_l + o, piece, ctx->n_meter - 2, 2, (int)std::min((size_t)GDG_BLOCK_SIZE, n - o), sample_rate)) != GDG_OK) return rc;
        }
        HIP_TRY(ctx, gdg_launch_wave_encode(out_format, d_l, n, 1, d_enc, ctx->stream));
        HIP_TRY(ctx, gdg_launch_wave_encode(out_format, d_r, n, 1, d_enc + piece * (size_t)width, ctx->stream));
        if (left_bytes) HIP_TRY(ctx, hipMemcpyAsync(static_cast<unsigned char *>(left_bytes) + at * width, d_enc, n * width, hipMemcpyDeviceToHost, ctx->stream));
        if (right_bytes) HIP_TRY(ctx, hipMemcpyAsync(static_cast<unsigned char *>(right_bytes) + at * width, d_enc + piece * (size_t)width, n * width, hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    return GDG_OK;
}
