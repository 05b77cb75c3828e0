/*
 * api_process.cpp -- a process call: parameter patches, the launch sequence (process_rows), the entry points, pinned staging and the copy workers.
 * Part of the host side of libgdg.so (the C-ABI of include/gdg.h on top of the HIP kernels; see ctx.h for the map).
 * There is no CPU compute path here: every sample is produced by a HIP kernel.
 */
#include "ctx.h"

/* ---- processing --------------------------------------------------------------------------------------- */

/* Parameter changes since the plan was built (gdg_unit_set_param): re-derive the constants of exactly those units -- with every side
 * effect the reference ties to the new value: histories re-made for a new delay, capacitors zeroed for a new band-pass order ... --
 * and overwrite their descriptors in the device blob, one small copy each, ordered on the context's stream behind the launches
 * that still read the old ones. */
static int apply_patches(gdg_ctx *ctx, int frames, uint32_t sample_rate) {
    join_groups(ctx);                          /* free-running channel groups may still read the descriptors */
    {
        const long limit = ctx->scan_tables_max;
        if ((long)ctx->scan_tabs.size() > (limit < 1 ? 1 : limit)) { ctx->dirty = true; return GDG_OK; }       /* the rebuild trims the cache */
    }
    size_t lo = (size_t)-1, hi = 0;
    for (int h : ctx->patch_units) {
        Unit *u = get_unit(ctx, h);
        const int slot = (size_t)h < ctx->plan_unit_slot.size() ? ctx->plan_unit_slot[(size_t)h] : -1;
        if (!u || slot < 0) { ctx->dirty = true; return GDG_OK; }
        const bool fast = ctx->plan_unit_fast[(size_t)h] != 0;
        /* e.g. oversampling switched on or off: the segment may change kernels -- the rebuild decides (and patched == rebuilt stays true bit for bit) */
        if ((segf_unit_ok(*u, frames, sample_rate) ? 1 : 0) != ctx->plan_unit_fast_ok[(size_t)h]) { ctx->dirty = true; return GDG_OK; }
        gdg_seg_unit du;
        int rc = prepare_unit(ctx, *u, frames, sample_rate, du, fast ? GDG_CHK_FAST : GDG_CHK);
        if (rc != GDG_OK) { ctx->dirty = true; return rc; }
        const size_t off = ctx->units_offset + (size_t)slot * sizeof(gdg_seg_unit);
        if (du.type == GDG_UNIT_REVERB) du.ip[7] = reinterpret_cast<const gdg_seg_unit *>(ctx->blob.data() + off)->ip[7];      /* the plan's decision (build_plan: wet path made by an earlier launch) */
        memcpy(ctx->blob.data() + off, &du, sizeof(du));
        lo = std::min(lo, off); hi = std::max(hi, off + sizeof(du));
    }
    /* a few knobs: one small copy each; a preset change over many channels: ONE copy of the span they cover (descriptors in between are
     * rewritten with the bytes they already hold).  `blob` is pageable on purpose: the runtime has copied such a source into its staging
     * buffer when the call returns, so the next patch may rewrite the same bytes at once (a pinned blob would need a fence per patch). */
    if (ctx->patch_units.size() > 4) {
        HIP_TRY(ctx, hipMemcpyAsync(ctx->d_blob + lo, ctx->blob.data() + lo, hi - lo, hipMemcpyHostToDevice, ctx->stream));
    } else {
        for (int h : ctx->patch_units) {
            const size_t off = ctx->units_offset + (size_t)ctx->plan_unit_slot[(size_t)h] * sizeof(gdg_seg_unit);
            HIP_TRY(ctx, hipMemcpyAsync(ctx->d_blob + off, ctx->blob.data() + off, sizeof(gdg_seg_unit), hipMemcpyHostToDevice, ctx->stream));
        }
    }
    ctx->patch_units.clear();
    return GDG_OK;
}



/* group g of G over n active channels = [b[g], b[g + 1]): equal shares (first i with floor(i G / n) == g) */
static std::vector<size_t> equal_group_bounds(size_t n, int G) {
    std::vector<size_t> b((size_t)G + 1);
    for (int g = 0; g <= G; g++) b[(size_t)g] = ((size_t)g * n + (size_t)G - 1) / (size_t)G;
    return b;
}

/* Groups of the host-buffer calls: EQUAL shares by default (two of them from 128 channels on, pcie_groups).  The first group's upload
 * and the last group's download are the two transfers nothing overlaps, so small outer and large inner groups looked attractive
 * (1 : 2 : 1, 1 : 3 : 3 : 1) -- measured slower than two equal groups (profiles/host_path_weights_r03.txt) and kept only as an experiment
 * knob: env GDG_PCIE_WEIGHTS="1,3,3,1" sets weights AND the group count.  A malformed list (an entry that is not a positive integer)
 * is refused as a whole, with one line on stderr -- never half applied. */
static std::vector<int> parse_pcie_weights(const char *e) {
    std::vector<int> w;
    if (!e || !*e) return w;
    for (const char *p = e;;) {
        char *end = nullptr;
        long v = strtol(p, &end, 10);
        while (end && (*end == ' ' || *end == '\t')) end++;
        if (end == p || v <= 0 || v > 1000000 || (end && *end && *end != ',')) {
            fprintf(stderr, "libgdg: GDG_PCIE_WEIGHTS=\"%s\" is not a comma-separated list of positive integers: ignored (equal groups)\n", e);
            return std::vector<int>();
        }
        w.push_back((int)v);
        if (!*end) break;
        p = end + 1;
        if (!*p) { fprintf(stderr, "libgdg: GDG_PCIE_WEIGHTS=\"%s\" ends in a comma: ignored (equal groups)\n", e); return std::vector<int>(); }
    }
    if (w.size() > 16) { fprintf(stderr, "libgdg: GDG_PCIE_WEIGHTS names %zu groups, at most 16: ignored (equal groups)\n", w.size()); w.clear(); }
    return w;
}
static std::vector<size_t> pcie_group_bounds(size_t n, int *G_io) {
    static std::vector<int> forced = parse_pcie_weights(getenv("GDG_PCIE_WEIGHTS"));
    int G = *G_io;
    std::vector<int> w = forced;
    if (!w.empty()) G = (int)w.size();
    if ((size_t)G > n) { G = (int)n; w.clear(); }
    if (G < 1) G = 1;
    *G_io = G;
    if (w.empty()) return equal_group_bounds(n, G);
    size_t total = 0, acc = 0;
    for (int v : w) total += (size_t)v;
    std::vector<size_t> b((size_t)G + 1, 0);
    for (int g = 0; g < G; g++) {
        acc += (size_t)w[(size_t)g];
        b[(size_t)g + 1] = std::max(b[(size_t)g] + 1, std::min(n - (size_t)(G - 1 - g), (acc * n + total / 2) / total));      /* never empty */
    }
    b[(size_t)G] = n;
    return b;
}

/* the side stream of the sums made ahead of the next frame (premac) and its events */
static int ensure_side_stream(gdg_ctx *ctx) {
    if (ctx->premac_stream) return GDG_OK;
    HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->premac_stream, hipStreamNonBlocking));
    HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_fir_done, hipEventDisableTiming));
    HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_premac, hipEventDisableTiming));
    return GDG_OK;
}

int process_rows(gdg_ctx *ctx, const std::vector<int> &active, const double *d_in, double *d_out, int frames, uint32_t sample_rate,
                        int stride, bool rows_by_channel, int groups, const GroupHook *before, const GroupHook *after,
                        int window, int stride_out, const std::vector<size_t> *group_bounds_in) {
    if (stride == 0) stride = frames;
    if (stride_out == 0) stride_out = stride;
    if (d_in == d_out) return fail(ctx, GDG_ERR_INVALID, "in-place processing is not supported");
    if (frames <= 0 || frames > ctx->max_frames) return fail(ctx, GDG_ERR_INVALID, "frames %d out of range (max %d)", frames, ctx->max_frames);
    if (sample_rate == 0) return fail(ctx, GDG_ERR_INVALID, "sample rate must be positive");
    hipSetDevice(ctx->device);
    struct ProfPhase {          /* this call's launches are bracketed or not as a whole (gdg_profile_sample) */
        gdg_ctx *c;
        explicit ProfPhase(gdg_ctx *ctx_) : c(ctx_) { c->prof_now = c->prof_every <= 1 || !c->profiling || (c->prof_calls++ % (unsigned)c->prof_every) == 0; }
        ~ProfPhase() { c->prof_now = true; }
    } prof_phase(ctx);
    const int G = groups < 1 ? 1 : groups;
    /* device-resident calls: the groups are not joined at the end of the call, so one group's kernels overlap the other's across calls
     * (the join happens when anything else touches the context: enter()) */
    const bool free_run = G > 1 && !before && !after;
    if (!free_run) join_groups(ctx);      /* (a change of the group count rebuilds the plan, and build_plan joins every stream there is) */
    const int P2 = fir_transform_size(frames);
    /* sums made ahead by the previous call (premac) are this call's if nothing has touched the context since and the plan still fits */
    bool use_pre = ctx->premac_valid && window == 1 && G == 1;
    std::vector<size_t> bounds;
    if (group_bounds_in && (int)group_bounds_in->size() == G + 1) bounds = *group_bounds_in;
    else bounds = equal_group_bounds(active.size(), G);
    /* the plan holds pointers into the buffers it was built on; other buffers of the same shape are reached by a shift */
    const bool plan_fits = !ctx->dirty && ctx->plan_frames == frames && ctx->plan_sr == sample_rate && ctx->plan_active == active && ctx->plan_stride == stride &&
                           ctx->plan_stride_out == stride_out && ctx->plan_by_channel == rows_by_channel && ctx->plan_groups == G && ctx->plan_bounds == bounds;
    if (plan_fits && !ctx->patch_units.empty()) {
        int rc = apply_patches(ctx, frames, sample_rate);            /* knob moves: the affected descriptors only (may fall back to dirty) */
        if (rc != GDG_OK) return rc;
    }
    if (!plan_fits) use_pre = false;
    if (!use_pre) join_premac(ctx, false);       /* an unused premac still writes Y: this call's launches go behind it */
    ctx->premac_valid = false;                   /* consumed by this call or dropped; the call's end makes the next one */
    if (ctx->dirty || ctx->plan_frames != frames || ctx->plan_sr != sample_rate ||
        ctx->plan_active != active || ctx->plan_stride != stride || ctx->plan_stride_out != stride_out || ctx->plan_by_channel != rows_by_channel ||
        ctx->plan_groups != G || ctx->plan_bounds != bounds) {
        int rc = build_plan(ctx, active, d_in, d_out, frames, sample_rate, stride, stride_out, rows_by_channel, G, bounds);
        ctx->plan_bounds = bounds;
        ctx->plan_stride = stride;
        ctx->plan_stride_out = stride_out;
        ctx->plan_by_channel = rows_by_channel;
        if (rc != GDG_OK) {
            /* filters whose spectra were allocated but not transformed start over at the next plan */
            for (auto &sp : ctx->pending_ir) for (auto &u : ctx->units) if (u.alive && u.H == sp) { u.H.reset(); u.fir_dirty = true; u.fir_live = false; }
            ctx->pending_ir.clear();
            ctx->dirty = true;
            return rc;
        }
        ctx->plan_active = active;
    }
    const gdg_seg_unit *d_units = reinterpret_cast<const gdg_seg_unit *>(ctx->d_blob + ctx->units_offset);
    const gdg_shift shift = { (long long)(((intptr_t)d_in - (intptr_t)ctx->plan_in) / (intptr_t)sizeof(double)),
                              (long long)(((intptr_t)d_out - (intptr_t)ctx->plan_out) / (intptr_t)sizeof(double)) };
    double2 *tw = nullptr, *tw2 = nullptr;
    for (auto &st : ctx->steps)
        if (st.is_fir && st.n) { int rc = fir_tables(ctx, fir_transform_size(frames), &tw, &tw2); if (rc != GDG_OK) return rc; break; }
    if (G > 1) {
        /* HIP streams share a few hardware queues (two on this runtime: profiles/groups_overlap_r04.txt); an idle premac stream left over from
         * one-group calls takes a slot and the two group streams end up behind one another (64 channels: 149 -> 274 us per step) */
        if (ctx->premac_stream) {
            join_premac(ctx, false);
            HIP_TRY(ctx, hipStreamSynchronize(ctx->premac_stream));
            hipStreamDestroy(ctx->premac_stream); hipEventDestroy(ctx->ev_fir_done); hipEventDestroy(ctx->ev_premac);
            ctx->premac_stream = nullptr; ctx->ev_fir_done = nullptr; ctx->ev_premac = nullptr;
        }
        while ((int)ctx->gstreams.size() < G) {
            hipStream_t s = nullptr;
            hipEvent_t e = nullptr;
            HIP_TRY(ctx, hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
            HIP_TRY(ctx, hipEventCreateWithFlags(&e, hipEventDisableTiming));
            ctx->gstreams.push_back(s);
            ctx->gjoin.push_back(e);
        }
        if (!ctx->gfork) HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->gfork, hipEventDisableTiming));
        HIP_TRY(ctx, hipEventRecord(ctx->gfork, ctx->stream));            /* the plan upload and earlier work on the main stream */
    }
    /* premac: which step is the call's last power amp, and is there anything to sum ahead */
    bool premac_here = false;
    size_t premac_after = 0;
    if (window == 1 && G == 1 && P2 == GDG_MAX_FRAMES && !ctx->profiling && !before && !after) {
        for (size_t sj = 0; sj < ctx->steps.size(); sj++) {
            if (!ctx->steps[sj].is_fir || !ctx->steps[sj].n) continue;
            premac_after = sj;
            premac_here = premac_here || ctx->steps[sj].premac_ok;
        }
    }
    for (int g = 0; g < G; g++) {
        hipStream_t s = G > 1 ? ctx->gstreams[(size_t)g] : ctx->stream;
        if (G > 1) HIP_TRY(ctx, hipStreamWaitEvent(s, ctx->gfork, 0));
        if (before) HIP_TRY(ctx, (*before)(g, s));
        for (size_t si = 0; si < ctx->steps.size(); si++) {
            const StepDesc &st = ctx->steps[si];
            int first = st.group_range[(size_t)g].first, n = st.group_range[(size_t)g].second;
            if (n == 0) continue;
            if (st.is_fir) {
                const gdg_fir_chan *d = reinterpret_cast<const gdg_fir_chan *>(ctx->d_blob + st.offset) + first;
                if (window > 1) {
                    /* `window` frames per channel: every spectrum is read once for all of them (fir.hip, "Time blocking"); with adjacent
                     * power amps the inverse transforms of one make the forward transforms of the next (one launch, no frame round trip) */
                    const int sh = st.shared_spectra ? 1 : 0;
                    const bool chain_ok = gdg_fir_window_chain_ok(n, window) != 0;
                    const bool chained_w = chain_ok && si > 0 && ctx->steps[si - 1].chain_next;
                    const bool chains_w = chain_ok && st.chain_next;
                    if (!chained_w) { ProfScope ps(ctx, GDG_K_FIR_FWD, s); HIP_TRY(ctx, gdg_launch_fir_window(window, d, n, sh, tw, tw2, 0, shift, s)); }
                    { ProfScope ps(ctx, GDG_K_FIR_MAC, s); HIP_TRY(ctx, gdg_launch_fir_window(window, d, n, sh, tw, tw2, 1, shift, s)); }
                    {
                        ProfScope ps(ctx, GDG_K_FIR_INV, s);
                        if (chains_w) {
                            const gdg_fir_chan *d_next = reinterpret_cast<const gdg_fir_chan *>(ctx->d_blob + ctx->steps[si + 1].offset) + first;
                            HIP_TRY(ctx, gdg_launch_fir_window_chain(window, d, d_next, n, tw, tw2, shift, s));
                        } else HIP_TRY(ctx, gdg_launch_fir_window(window, d, n, sh, tw, tw2, 2, shift, s));
                        HIP_TRY(ctx, gdg_launch_fir_window(window, d, n, sh, tw, tw2, 3, shift, s));
                    }
                    continue;
                }
                const bool chained = si > 0 && ctx->steps[si - 1].chain_next;      /* the previous power amp's inverse made this one's spectrum */
                if (!chained) { ProfScope ps(ctx, GDG_K_FIR_FWD, s); HIP_TRY(ctx, gdg_launch_fir_fwd(P2, frames, d, n, tw, tw2, shift, s)); }
                const gdg_fir_chan *d_next = st.chain_next ? reinterpret_cast<const gdg_fir_chan *>(ctx->d_blob + ctx->steps[si + 1].offset) + first : nullptr;
                const bool fused = ctx->fir_fused < 0 ? (n > fir_split_limit(ctx)) : (ctx->fir_fused != 0);
                if (fused) {
                    /* multiply-accumulate fused into the inverse transform's first stage (reported as the MAC kernel; its chained
                     * variant, which also makes the next amp's forward transform, under a kind of its own) */
                    ProfScope ps(ctx, d_next ? GDG_K_FIR_MAC_CHAIN : GDG_K_FIR_MAC, s, ctx->prof_attach);
                    HIP_TRY(ctx, gdg_launch_fir_inv(P2, d, n, tw, tw2, st.shared_spectra ? 2 : 1, shift, s, d_next, ps.attached ? ps.a : nullptr, ps.attached ? ps.b : nullptr));
                } else if (use_pre && st.premac_ok) {
                    /* the terms k >= 1 are in Y already (the previous call's premac): the newest term + the inverse transform */
                    if (ctx->premac_outstanding) { HIP_TRY(ctx, hipStreamWaitEvent(s, ctx->ev_premac, 0)); ctx->premac_outstanding = false; }
                    ProfScope ps(ctx, GDG_K_FIR_INV, s);
                    HIP_TRY(ctx, gdg_launch_fir_inv(P2, d, n, tw, tw2, 4, shift, s, d_next));
                    ctx->stat_premac_used++;
                } else {
                    { ProfScope ps(ctx, GDG_K_FIR_MAC, s); HIP_TRY(ctx, gdg_launch_fir_mac(P2, d, n, st.shared_spectra ? 1 : 0, s)); }
                    { ProfScope ps(ctx, GDG_K_FIR_INV, s); HIP_TRY(ctx, gdg_launch_fir_inv(P2, d, n, tw, tw2, 0, shift, s, d_next)); }
                }
                /* behind the call's LAST power amp the next frame's sums can start (premac, below): mark the place in the stream */
                if (premac_here && si == premac_after) {
                    { int rc = ensure_side_stream(ctx); if (rc != GDG_OK) return rc; }
                    HIP_TRY(ctx, hipEventRecord(ctx->ev_fir_done, s));
                }
            } else if (st.absorbed_per_frame && window == 1 && G == 1) {
                continue;                          /* the oversampled shaper's launch behind this step runs its compressor (os_tiles_kernel, pre_chans) */
            } else if (st.os_factor) {
                /* an oversampled shaper of few channels: a workgroup per (channel, frame, tile) instead of one per channel */
                const gdg_seg_chan *d = reinterpret_cast<const gdg_seg_chan *>(ctx->d_blob + st.offset) + first;
                ProfScope ps(ctx, GDG_K_SEGMENT, s);
                const int epoch = (ctx->wave_epoch = (ctx->wave_epoch % 0x3ffffff) + 1);
                /* a per-frame call: the lone compressor in front of the shaper runs inside this launch (the step itself was skipped above) */
                const gdg_seg_chan *d_pre = (window == 1 && G == 1 && st.os_prefix_step >= 0)
                                            ? reinterpret_cast<const gdg_seg_chan *>(ctx->d_blob + ctx->steps[(size_t)st.os_prefix_step].offset) + first : nullptr;
                HIP_TRY(ctx, gdg_launch_os_tiles(st.os_factor, d, n, d_units, frames, window, shift, ctx->os, ctx->d_wave + st.os_flags + first, epoch, ctx->d_error, s,
                                                 d_pre, ctx->d_wave + st.os_arrive + first));
            } else {
                const gdg_seg_chan *d = reinterpret_cast<const gdg_seg_chan *>(ctx->d_blob + st.offset) + first;
                ProfScope ps(ctx, GDG_K_SEGMENT, s);
                /* one launch per window: a channel's workgroup walks its frames in order, the units' state runs through them */
                /* ... unless the channels are few: then a workgroup per frame, the frames of a channel meeting unit by unit (seg.hip, WAVE) */
                /* (by the CALL's channel count, not the group's: two groups of 256 channels fill the chip like one launch of 512) */
                const int wave_max = st.wave_release ? std::min(ctx->seg_wave_max, ctx->seg_wave_release_max) : ctx->seg_wave_max;
                int *tickets = (window > 1 && (int)active.size() <= wave_max && st.wave_tickets >= 0) ? ctx->d_wave + st.wave_tickets + g : nullptr;
                const int epoch = tickets ? (ctx->wave_epoch = (ctx->wave_epoch % 0x3ffffff) + 1) : 0;          /* epoch * 32 + frame fits an int */
                /* option debug_stall_unit: in a WAVE launch frame 0 of that unit's channel withholds the unit's counter (seg.hip) */
                int stall = 0;
                if (tickets && ctx->debug_stall_unit >= 0 && (size_t)ctx->debug_stall_unit < ctx->plan_unit_slot.size() &&
                    ctx->plan_unit_slot[(size_t)ctx->debug_stall_unit] >= 0) stall = 1 + ctx->plan_unit_slot[(size_t)ctx->debug_stall_unit];
                if (st.fast) HIP_TRY(ctx, gdg_launch_segf(d, n, d_units, frames, window, shift, ctx->os, ctx->d_error, s, tickets, epoch, stall));
                else if (tickets) HIP_TRY(ctx, gdg_launch_seg(d, n, d_units, frames, window, shift, ctx->os, ctx->d_error, s, tickets, epoch, stall));
                else if (st.tile_ok && window == 1 && G == 1 && st.wave_tickets >= 0 && ctx->d_tile_xch && 2 * n + st.ahead_n <= GDG_TILE_WORKGROUP_BUDGET) {
                    /* a per-frame call of few channels: a channel's frame on two workgroups (seg.hip SEG_TILE; the bits of the general kernel),
                     * the reverbs' wet paths of later steps beside them as in the general launch below */
                    const int *d_ahead = st.ahead_n > 0 ? reinterpret_cast<const int *>(ctx->d_blob + st.ahead_offset) : nullptr;
                    const int tile_epoch = (ctx->wave_epoch = (ctx->wave_epoch % 0x3ffffff) + 1);
                    HIP_TRY(ctx, gdg_launch_segt(d, n, d_units, shift, ctx->os, ctx->d_error, s, ctx->d_wave + st.wave_tickets + g, tile_epoch, ctx->d_tile_xch + (size_t)first * gdg_segt_xch_words(),
                                                 d_ahead, d_ahead ? st.ahead_n : 0));
                } else {
                    /* one frame per launch: the launch also makes the wet paths of the reverbs of LATER segment steps (extra workgroups beside
                     * the channels'; seg.hip REVERB_AHEAD), and reverbs whose wet path an earlier launch of this call made only mix */
                    const bool piggy = window == 1 && G == 1;
                    const int *d_ahead = piggy && st.ahead_n > 0 ? reinterpret_cast<const int *>(ctx->d_blob + st.ahead_offset) : nullptr;
                    HIP_TRY(ctx, gdg_launch_seg(d, n, d_units, frames, window, shift, ctx->os, ctx->d_error, s, tickets, epoch, piggy ? 1 : 0, d_ahead, d_ahead ? st.ahead_n : 0));
                }
            }
        }
        if (after) HIP_TRY(ctx, (*after)(g, s));
        if (premac_here) {
            /* every launch of the call is in the context's stream: now the side stream's share (the host must not keep the main stream
             * waiting for its next kernel while it enqueues these: 6 us per step) */
            HIP_TRY(ctx, hipStreamWaitEvent(ctx->premac_stream, ctx->ev_fir_done, 0));
            for (auto &sx : ctx->steps) {
                if (!sx.is_fir || !sx.premac_ok || !sx.n) continue;
                const gdg_fir_chan *dx = reinterpret_cast<const gdg_fir_chan *>(ctx->d_blob + sx.offset);
                /* ... on the CUs the segments leave idle (premac_lds: api_plan.cpp) */
                HIP_TRY(ctx, gdg_launch_fir_mac(P2, dx, sx.n, sx.shared_spectra ? 1 : 0, ctx->premac_stream, 1, sx.premac_lds));
            }
            HIP_TRY(ctx, hipEventRecord(ctx->ev_premac, ctx->premac_stream));
            ctx->premac_valid = true;
            ctx->premac_outstanding = true;
        }
        if (free_run) ctx->groups_pending = true;
        else if (G > 1) {
            HIP_TRY(ctx, hipEventRecord(ctx->gjoin[(size_t)g], s));
            HIP_TRY(ctx, hipStreamWaitEvent(ctx->stream, ctx->gjoin[(size_t)g], 0));
        }
    }
    return GDG_OK;
}

/* how many channel groups a host-buffer call over n channels is split into (option "pcie_groups" overrides) */
static int pcie_groups(const gdg_ctx *ctx, int n) {
    const int forced = ctx->pcie_groups_forced;
    int g = forced > 0 ? forced : (n >= 128 ? 2 : 1);        /* measured: profiles/host_path_rate_r01.txt */
    if (g > n) g = n;
    return g < 1 ? 1 : (g > 16 ? 16 : g);
}

/* channel groups of the device-resident calls: the groups' kernels run on streams of their own, are NOT joined at the end of the call
 * and overlap (one group's latency-bound segment kernel with the other's HBM-bound convolution).  Opt-in only -- gdg_ctx_set_overlap(G > 1)
 * or env GDG_DEVICE_GROUPS -- because a caller that caches gdg_ctx_stream() and enqueues its own work behind a process call is only
 * ordered after the call's kernels when they run on that stream: the default is ONE group on the context's stream.
 * Measured (profiles/device_groups_r02.txt): two groups gain 7-10 % from 512 channels on, nothing below, four lose. */
static int device_groups(const gdg_ctx *ctx) {
    const int forced = ctx->device_groups_env;
    const int n = ctx->nch;
    int g = ctx->overlap_groups > 0 ? ctx->overlap_groups : (forced > 0 ? forced : 1);
    if (g > n) g = n;
    return g < 1 ? 1 : (g > 16 ? 16 : g);
}

int gdg_process_device(gdg_ctx *ctx, const double *d_in, double *d_out, int frames, uint32_t sample_rate) {
    if (!ctx || !d_in || !d_out) return GDG_ERR_INVALID;
    if (ctx->all_channels.empty()) for (int c = 0; c < ctx->nch; c++) ctx->all_channels.push_back(c);
    return process_rows(ctx, ctx->all_channels, d_in, d_out, frames, sample_rate, 0, false, device_groups(ctx));
}

int gdg_ctx_set_overlap(gdg_ctx *ctx, int groups) {
    if (!ctx) return GDG_ERR_INVALID;
    if (groups < 0 || groups > 16) return fail(ctx, GDG_ERR_INVALID, "%d channel groups: 0 (automatic) to 16", groups);
    enter(ctx);
    ctx->overlap_groups = groups;
    return GDG_OK;
}

int gdg_ctx_set_window(gdg_ctx *ctx, int frames_per_call) {
    if (!ctx) return GDG_ERR_INVALID;
    const int W = frames_per_call;
    if (W != 1 && W != 2 && W != 4 && W != 8 && W != 16) return fail(ctx, GDG_ERR_INVALID, "window of %d frames: 1, 2, 4, 8 or 16", W);
    if (W > 1 && ctx->max_frames != GDG_MAX_FRAMES)
        return fail(ctx, GDG_ERR_UNSUPPORTED, "windows are made of %d-sample frames, the context allows %d", GDG_MAX_FRAMES, ctx->max_frames);
    if (W == ctx->window) return GDG_OK;
    enter(ctx);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    const size_t stride = (size_t)W * (size_t)ctx->max_frames, bytes = (size_t)ctx->nch * stride * sizeof(double);
    double *w0 = nullptr, *w1 = nullptr;
    if (hipMalloc((void **)&w0, bytes) != hipSuccess || hipMalloc((void **)&w1, bytes) != hipSuccess) {
        hipFree(w0);
        return fail(ctx, GDG_ERR_NOMEM, "cannot allocate the window's intermediate frames");
    }
    hipFree(ctx->d_w0); hipFree(ctx->d_w1);
    ctx->d_w0 = w0; ctx->d_w1 = w1;
    ctx->w_stride = stride;
    ctx->window = W;           /* the power amps' delay lines follow at their next frame (prepare_fir: ring of K + W - 1 slots) */
    ctx->dirty = true;
    return GDG_OK;
}

int gdg_process_window_device(gdg_ctx *ctx, const double *d_in, double *d_out, size_t row_stride, int frames_in_window, uint32_t sample_rate) {
    if (!ctx || !d_in || !d_out) return GDG_ERR_INVALID;
    const int W = frames_in_window;
    if (W < 1 || W > ctx->window) return fail(ctx, GDG_ERR_INVALID, "window of %d frames, the context is set up for %d (gdg_ctx_set_window)", W, ctx->window);
    if (W != 1 && W != 2 && W != 4 && W != 8 && W != 16) return fail(ctx, GDG_ERR_INVALID, "window of %d frames: 1, 2, 4, 8 or 16", W);
    if (row_stride < (size_t)W * (size_t)ctx->max_frames || row_stride > 0x7fffffff)
        return fail(ctx, GDG_ERR_INVALID, "row stride %zu is shorter than the window (%d x %d)", row_stride, W, ctx->max_frames);
    if (ctx->all_channels.empty()) for (int c = 0; c < ctx->nch; c++) ctx->all_channels.push_back(c);
    return process_rows(ctx, ctx->all_channels, d_in, d_out, ctx->max_frames, sample_rate, (int)row_stride, false, device_groups(ctx), nullptr, nullptr, W);
}

int check_device_error(gdg_ctx *ctx) {
    int e = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&e, ctx->d_error, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (e != 0) {
        hipMemsetAsync(ctx->d_error, 0, sizeof(int), ctx->stream);
        if (e == 0x57415645) {                  /* seg.hip GDG_WAVE_TIMEOUT_CODE: a frame waited ~1 s for its predecessor's counter */
            ctx->dirty = true;                  /* the next plan starts from fresh counters */
            if (ctx->d_wave) hipMemsetAsync(ctx->d_wave, 0, ctx->d_wave_cap * sizeof(int), ctx->stream);
            return fail(ctx, GDG_ERR_HIP, "a segment launch timed out waiting for a hand-off between its workgroups -- a window's frame counter, a tile's carry "
                                          "(the results of that call are invalid; reset the units)");
        }
        return fail(ctx, GDG_ERR_UNSUPPORTED, "segment kernel met unit type %d without a HIP implementation", e - 1);
    }
    return GDG_OK;
}

int gdg_ctx_trim(gdg_ctx *ctx) {
    if (!ctx) return GDG_ERR_INVALID;
    enter(ctx);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    ctx->arena.trim_pending = true;
    ctx->arena.trim();
    return GDG_OK;
}

int gdg_ctx_synchronize(gdg_ctx *ctx) {
    if (!ctx) return GDG_ERR_INVALID;
    enter(ctx, true);
    return check_device_error(ctx);
}

int ensure_staging(gdg_ctx *ctx) {
    if (ctx->d_stage_in) return GDG_OK;
    size_t bytes = (size_t)std::max(ctx->nch, 2) * (size_t)ctx->max_frames * sizeof(double);
    HIP_TRY(ctx, hipMalloc((void **)&ctx->d_stage_in, bytes));
    HIP_TRY(ctx, hipMalloc((void **)&ctx->d_stage_out, bytes));
    HIP_TRY(ctx, pinned_alloc(ctx, (void **)&ctx->h_stage_in, bytes));
    HIP_TRY(ctx, pinned_alloc(ctx, (void **)&ctx->h_stage_out, bytes));
    return GDG_OK;
}


/* A forked child inherits the pool object but none of its threads: joining or detaching std::thread handles of threads that do not exist in
 * this process is undefined -- the child leaves the object alone (a few hundred bytes, once). */
void destroy_copy_pool(CopyPool *p) { if (p && p->usable()) delete p; }

static CopyPool &copy_pool(gdg_ctx *ctx, int which = 0) {
    if (which == 1) {
        if (!ctx->copy_pool_up) {
            int threads = ctx->copy_threads;
            unsigned hw = std::thread::hardware_concurrency();
            if (hw > 0 && threads > (int)hw) threads = (int)hw;
            if (threads < 1) threads = 1;
            const std::vector<int> *cpus;
            numa_target(ctx, &cpus);
            ctx->copy_pool_up = new CopyPool(threads - 1, *cpus);
        }
        return *ctx->copy_pool_up;
    }
    if (!ctx->copy_pool) {
        int threads = ctx->copy_threads;
        unsigned hw = std::thread::hardware_concurrency();
        if (hw > 0 && threads > (int)hw) threads = (int)hw;
        if (threads < 1) threads = 1;
        const std::vector<int> *cpus;
        numa_target(ctx, &cpus);
        ctx->copy_pool = new CopyPool(threads - 1, *cpus);
    }
    return *ctx->copy_pool;
}

void ensure_copy_pool(gdg_ctx *ctx, int which) { (void)copy_pool(ctx, which); }

/* option "numa": the workers are re-made (bound or not) at the next host-buffer call; pinned slabs that exist stay where they are (the
 * staging slabs' addresses are in the caller's hands), those made afterwards follow the new mode -- set it before the first call */
int numa_rebind(gdg_ctx *ctx, int mode) {
    (void)mode;
    if (ctx->copy_pool) { destroy_copy_pool(ctx->copy_pool); ctx->copy_pool = nullptr; }
    if (ctx->copy_pool_up) { destroy_copy_pool(ctx->copy_pool_up); ctx->copy_pool_up = nullptr; }
    return GDG_OK;
}

/* rows [a, b) of a host-side staging copy, spread over the copy workers (at least ~1 MiB per thread) */
void copy_rows_parallel(gdg_ctx *ctx, size_t a, size_t b, const std::function<void(size_t)> &copy_row, size_t row_bytes, int which) {
    size_t n = b > a ? b - a : 0;
    if (n == 0) return;
    CopyPool &pool = copy_pool(ctx, which);
    size_t T = std::min(pool.slots(), n * row_bytes / (1u << 20) + 1);
    if (T <= 1 || n < 2 || !pool.usable()) { for (size_t i = a; i < b; i++) copy_row(i); return; }
    pool.run(T, [&](size_t t) { for (size_t i = a + n * t / T; i < a + n * (t + 1) / T; i++) copy_row(i); });
}

int gdg_process_subset(gdg_ctx *ctx, const int *channels, int n, const double *const *in, double *const *out, int frames, uint32_t sample_rate) {
    if (!ctx || !in || !out || !channels) return GDG_ERR_INVALID;
    if (n <= 0 || n > ctx->nch) return fail(ctx, GDG_ERR_INVALID, "bad channel count %d", n);
    if (frames <= 0 || frames > ctx->max_frames) return fail(ctx, GDG_ERR_INVALID, "frames %d out of range (max %d)", frames, ctx->max_frames);
    std::vector<int> active(channels, channels + n);
    std::vector<char> seen((size_t)ctx->nch, 0);
    for (int c : active) {
        if (c < 0 || c >= ctx->nch || seen[(size_t)c]) return fail(ctx, GDG_ERR_INVALID, "bad or repeated channel %d", c);
        seen[(size_t)c] = 1;
    }
    enter(ctx);
    int rc = ensure_staging(ctx);
    if (rc != GDG_OK) return rc;
    /* rows travel compactly ([i][frames]); G channel groups: group g's rows are staged and uploaded on stream g while the
     * earlier groups already compute, and copied back to the caller while the later groups still run */
    int G = pcie_groups(ctx, n);
    const std::vector<size_t> gb = pcie_group_bounds((size_t)n, &G);
    const size_t row = (size_t)frames;
    auto lo = [&](int g) { return gb[(size_t)g]; };
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));                           /* the staging slabs may still be in use */
    GroupHook before = [&](int g, hipStream_t s) -> hipError_t {
        size_t a = lo(g), b = lo(g + 1);
        copy_rows_parallel(ctx, a, b, [&](size_t i) { memcpy(ctx->h_stage_in + i * row, in[i], row * sizeof(double)); }, row * sizeof(double));
        if (b == a) return hipSuccess;
        return hipMemcpyAsync(ctx->d_stage_in + a * row, ctx->h_stage_in + a * row, (b - a) * row * sizeof(double), hipMemcpyHostToDevice, s);
    };
    GroupHook after = [&](int g, hipStream_t s) -> hipError_t {
        size_t a = lo(g), b = lo(g + 1);
        if (b == a) return hipSuccess;
        return hipMemcpyAsync(ctx->h_stage_out + a * row, ctx->d_stage_out + a * row, (b - a) * row * sizeof(double), hipMemcpyDeviceToHost, s);
    };
    ctx->stage_out_stride = 0;
    rc = process_rows(ctx, active, ctx->d_stage_in, ctx->d_stage_out, frames, sample_rate, 0, false, G, &before, &after, 1, 0, &gb);
    if (rc != GDG_OK) return rc;
    {   /* all channels, in order: the compact rows are a complete block (gdg_spatialize_staged may mix it without an upload) */
        bool complete = n == ctx->nch;
        for (int i = 0; complete && i < n; i++) complete = active[(size_t)i] == i;
        if (complete) { ctx->stage_out_stride = frames; ctx->stage_out_frames = frames; }
    }
    for (int g = 0; g < G; g++) {
        if (G > 1) HIP_TRY(ctx, hipStreamSynchronize(ctx->gstreams[(size_t)g]));
        else HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        copy_rows_parallel(ctx, lo(g), lo(g + 1), [&](size_t i) { memcpy(out[i], ctx->h_stage_out + i * row, row * sizeof(double)); }, row * sizeof(double));
    }
    return check_device_error(ctx);
}

int gdg_process(gdg_ctx *ctx, const double *const *in, double *const *out, int frames, uint32_t sample_rate) {
    if (!ctx) return GDG_ERR_INVALID;
    if (ctx->all_channels.empty()) for (int c = 0; c < ctx->nch; c++) ctx->all_channels.push_back(c);
    return gdg_process_subset(ctx, ctx->all_channels.data(), ctx->nch, in, out, frames, sample_rate);
}

/* pinned host slabs for callers that must not hand Go (or other managed) pointers to C: row c = channel c */
int gdg_staging_buffers(gdg_ctx *ctx, double **in, double **out, int *row_stride) {
    if (!ctx || !in || !out || !row_stride) return GDG_ERR_INVALID;
    enter(ctx);
    int rc = ensure_staging(ctx);
    if (rc != GDG_OK) return rc;
    *in = ctx->h_stage_in;
    *out = ctx->h_stage_out;
    *row_stride = ctx->max_frames;
    return GDG_OK;
}

int gdg_process_staged(gdg_ctx *ctx, const int *channels, int n, int frames, uint32_t sample_rate) {
    if (!ctx || !channels) return GDG_ERR_INVALID;
    if (n <= 0 || n > ctx->nch) return fail(ctx, GDG_ERR_INVALID, "bad channel count %d", n);
    if (frames <= 0 || frames > ctx->max_frames) return fail(ctx, GDG_ERR_INVALID, "frames %d out of range (max %d)", frames, ctx->max_frames);
    std::vector<int> active(channels, channels + n);
    std::vector<char> seen((size_t)ctx->nch, 0);
    for (int c : active) {
        if (c < 0 || c >= ctx->nch || seen[(size_t)c]) return fail(ctx, GDG_ERR_INVALID, "bad or repeated channel %d", c);
        seen[(size_t)c] = 1;
    }
    enter(ctx);
    int rc = ensure_staging(ctx);
    if (rc != GDG_OK) return rc;
    const size_t stride = (size_t)ctx->max_frames;
    /* G channel groups on their own streams: uploads, kernels and downloads of different groups overlap.  One strided copy
     * per run of consecutive channels inside a group (512 single-row copies would cost ~10 us each). */
    int G = pcie_groups(ctx, n);
    const std::vector<size_t> gb = pcie_group_bounds((size_t)n, &G);
    auto lo = [&](int g) { return gb[(size_t)g]; };
    auto copy_runs = [&](int g, double *dst, const double *src, hipMemcpyKind kind, hipStream_t s) -> hipError_t {
        for (size_t i = lo(g); i < lo(g + 1);) {
            size_t j = i + 1;
            while (j < lo(g + 1) && active[j] == active[j - 1] + 1) j++;
            size_t off = (size_t)active[i] * stride;
            hipError_t e = hipMemcpy2DAsync(dst + off, stride * sizeof(double), src + off, stride * sizeof(double),
                                            (size_t)frames * sizeof(double), j - i, kind, s);
            if (e != hipSuccess) return e;
            i = j;
        }
        return hipSuccess;
    };
    GroupHook before = [&](int g, hipStream_t s) { return copy_runs(g, ctx->d_stage_in, ctx->h_stage_in, hipMemcpyHostToDevice, s); };
    GroupHook after = [&](int g, hipStream_t s) { return copy_runs(g, ctx->h_stage_out, ctx->d_stage_out, hipMemcpyDeviceToHost, s); };
    ctx->stage_out_stride = 0;
    rc = process_rows(ctx, active, ctx->d_stage_in, ctx->d_stage_out, frames, sample_rate, ctx->max_frames, true, G, &before, &after, 1, 0, &gb);
    if (rc != GDG_OK) return rc;
    if (n == ctx->nch) { ctx->stage_out_stride = ctx->max_frames; ctx->stage_out_frames = frames; }     /* rows by channel: every channel took part */
    return check_device_error(ctx);
}
