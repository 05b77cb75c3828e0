/*
 * api_ctx.cpp -- context life cycle, options, NUMA placement, units and chains, profiling, device-memory helpers, the stand-alone transforms.
 * Part of the host side of libgdg.so (the C-ABI of include/gdg.h on top of the HIP kernels; see ctx.h for the map).
 * There is no CPU compute path here: every sample is produced by a HIP kernel.
 */
#include "ctx.h"

/* ---- parameter tables (defaults), effects/<unit>.go create*() ------------------------------------ */
static const int g_param_count[GDG_UNIT_COUNT] = { 6, 3, 3, 5, 4, 3, 7, 3, 7, 6, 4, 4, 2, 2, 3, 3, 1, 3, 1, 1, 1 };
static const int32_t g_param_default[GDG_UNIT_COUNT][GDG_MAX_PARAMS] = {
    { 100, 0, 0, 440, 100, 0 }, { -20, -40, 50 }, { 0, 300, 3000 }, { 1, -40, -10, 300, 6000 },
    { 1, -40, -10, 100 }, { 1, 30, -20 }, { 1, -20, -20, -20, -20, -20, -20 }, { 0, 0, 0 },
    { 1, 50, 0, 0, 100, 0, 0 }, { 0, 0, 100, 0, 1, 0 }, { 0, 0, 0, 0 }, { 0, -2, -5, -5 },
    { 100, 30 }, { 100, 10 }, { 100, 10, 45 }, { 100, 50, -10 }, { 100 }, { 200, -5, -5 }, { 50 }, { 14 }, { 0 },
};


const char *gdg_version(void) { return "gdg 0.1 gfx950 hip"; }

int gdg_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}


/* ---- NUMA: where the device hangs, from sysfs ------------------------------------------------------------------------------- */
/* "0-63,128-191" -> the CPU numbers; false on anything else */
static bool parse_cpulist(const char *text, std::vector<int> &cpus) {
    cpus.clear();
    const char *p = text;
    while (*p && *p != '\n') {
        char *end = nullptr;
        long a = strtol(p, &end, 10);
        if (end == p || a < 0) return false;
        long b = a;
        p = end;
        if (*p == '-') { b = strtol(p + 1, &end, 10); if (end == p + 1 || b < a) return false; p = end; }
        if (b - a > 65536) return false;
        for (long c = a; c <= b; c++) cpus.push_back((int)c);
        if (*p == ',') p++;
        else if (*p && *p != '\n') return false;
    }
    return true;
}
static bool read_text(const std::string &path, char *buf, size_t cap) {
    FILE *f = fopen(path.c_str(), "r");
    if (!f) return false;
    size_t n = fread(buf, 1, cap - 1, f);
    fclose(f);
    buf[n] = 0;
    return n > 0;
}
/* node of the PCI function `pci_bus_id` ("0000:05:00.0", any case) and that node's CPUs under `sysfs_root` ("/sys") */
int gdg_numa_probe(const char *sysfs_root, const char *pci_bus_id, int *node, int *cpus, int capacity, int *n_cpus) {
    if (!sysfs_root || !pci_bus_id || !node) return GDG_ERR_INVALID;
    *node = -1;
    if (n_cpus) *n_cpus = 0;
    std::string id(pci_bus_id);
    for (auto &ch : id) ch = (char)tolower((unsigned char)ch);
    char buf[4096];
    if (!read_text(std::string(sysfs_root) + "/bus/pci/devices/" + id + "/numa_node", buf, sizeof(buf))) return GDG_OK;       /* unknown: not an error */
    char *end = nullptr;
    long n = strtol(buf, &end, 10);
    if (end == buf || n < 0) return GDG_OK;                 /* "-1": the platform does not say (one node, or a VM) */
    std::vector<int> list;
    if (!read_text(std::string(sysfs_root) + "/devices/system/node/node" + std::to_string(n) + "/cpulist", buf, sizeof(buf)) || !parse_cpulist(buf, list) || list.empty())
        return GDG_OK;
    *node = (int)n;
    if (n_cpus) *n_cpus = (int)list.size();
    if (cpus) for (int i = 0; i < capacity && i < (int)list.size(); i++) cpus[i] = list[(size_t)i];
    return GDG_OK;
}
static void numa_discover(gdg_ctx *ctx) {
    char id[64] = { 0 };
    if (hipDeviceGetPCIBusId(id, (int)sizeof(id), ctx->device) != hipSuccess) return;
    int node = -1, n = 0;
    std::vector<int> cpus(4096);
    if (gdg_numa_probe("/sys", id, &node, cpus.data(), (int)cpus.size(), &n) != GDG_OK || node < 0) return;
    cpus.resize((size_t)std::min(n, (int)cpus.size()));
    ctx->numa_node = node;
    ctx->numa_cpus = cpus;
    for (int n = 0; n < 64; n++) {                      /* every node's CPU list, for "the caller's node" */
        char buf[4096];
        std::vector<int> list;
        if (!read_text("/sys/devices/system/node/node" + std::to_string(n) + "/cpulist", buf, sizeof(buf)) || !parse_cpulist(buf, list)) break;
        ctx->node_cpus.push_back(list);
    }
}
/* the node and CPUs option "numa" points at right now: the device's (1) or the calling thread's (2); node < 0: bind nothing */
int numa_target(const gdg_ctx *ctx, const std::vector<int> **cpus) {
    static const std::vector<int> none;
    *cpus = &none;
    if (ctx->numa_mode == 1 && ctx->numa_node >= 0) { *cpus = &ctx->numa_cpus; return ctx->numa_node; }
    if (ctx->numa_mode == 2) {
        const int cpu = sched_getcpu();
        for (size_t n = 0; n < ctx->node_cpus.size(); n++)
            if (std::find(ctx->node_cpus[n].begin(), ctx->node_cpus[n].end(), cpu) != ctx->node_cpus[n].end()) { *cpus = &ctx->node_cpus[n]; return (int)n; }
    }
    return -1;
}
/* the calling thread onto the device's node (copy workers) */
void numa_bind_thread(const std::vector<int> &cpus) {
    if (cpus.empty()) return;
    cpu_set_t *set = CPU_ALLOC(4096);
    if (!set) return;
    const size_t bytes = CPU_ALLOC_SIZE(4096);
    CPU_ZERO_S(bytes, set);
    for (int c : cpus) if (c >= 0 && c < 4096) CPU_SET_S(c, bytes, set);
    pthread_setaffinity_np(pthread_self(), bytes, set);      /* a cpuset that forbids those CPUs leaves the thread where it was */
    CPU_FREE(set);
}
/* pinned host memory from the device's node: the pages are taken (and pinned) inside hipHostMalloc, under the calling thread's memory policy */
hipError_t pinned_alloc(gdg_ctx *ctx, void **p, size_t bytes) {
    const std::vector<int> *unused;
    const int node = numa_target(ctx, &unused);
    const bool bind = node >= 0 && node < 1024;
    if (bind) {
        unsigned long mask[16] = { 0 };
        mask[node / (8 * sizeof(unsigned long))] |= 1ul << (node % (8 * sizeof(unsigned long)));
        /* the caller's own policy (numactl --membind / --interleave, set_mempolicy by the host program) is put back afterwards; a thread whose
         * policy cannot be read keeps it: no binding then */
        int old_mode = 0;
        unsigned long old_mask[16] = { 0 };
        const bool saved = syscall(SYS_get_mempolicy, &old_mode, old_mask, 1024ul + 1, nullptr, 0ul) == 0;
        const bool policy = saved && syscall(SYS_set_mempolicy, 1 /* MPOL_PREFERRED */, mask, 1024ul + 1) == 0;
        hipError_t e = hipHostMalloc(p, bytes, policy ? hipHostMallocNumaUser : hipHostMallocDefault);
        if (policy && syscall(SYS_set_mempolicy, old_mode, old_mode == 0 ? nullptr : old_mask, old_mode == 0 ? 0ul : 1024ul + 1) != 0)
            syscall(SYS_set_mempolicy, 0 /* MPOL_DEFAULT */, nullptr, 0ul);
        if (e == hipSuccess) return e;
        (void)hipGetLastError();
    }
    return hipHostMalloc(p, bytes, hipHostMallocDefault);
}

/* ---- options: everything that decides a launch shape, behind ONE entry point (gdg_ctx_set_option) ----------------------------------
 * The environment variable of an option is read once, when the context is made, as a debug override of its default -- a library's behaviour
 * should not depend on the environment of whoever loads it.  `knob` options are process-wide (their launchers have no context). */
struct OptionDef {
    const char *key, *env;
    long long lo, hi;
    int knob;                                   /* >= 0: gdg_knob_set / gdg_knob_get (process-wide) */
    int gdg_ctx::*field;
    bool gdg_ctx::*flag;
    bool replans;                               /* a change invalidates the cached plan */
};
static const OptionDef g_options[] = {
    /* the convolution */
    { "fir_fused", "GDG_FIR_FUSED", -1, 1, -1, &gdg_ctx::fir_fused, nullptr, true },                  /* -1: by channel count (fir_split_max) */
    { "fir_split_max_channels", "GDG_FIR_SPLIT_MAX", 0, 1 << 20, -1, &gdg_ctx::fir_split_max, nullptr, true },
    { "fir_split_max_channels_one_amp", "GDG_FIR_SPLIT_MAX_ONE_AMP", 0, 1 << 20, -1, &gdg_ctx::fir_split_max_single, nullptr, true },
    { "fir_chain_adjacent_amps", "GDG_FIR_CHAIN", 0, 1, -1, nullptr, &gdg_ctx::fir_chain, true },
    { "fir_premac", "GDG_FIR_PREMAC", 0, 1, -1, &gdg_ctx::fir_premac, nullptr, true },
    { "fir_premac_min_partitions", "GDG_FIR_PREMAC_MIN", 1, 1 << 24, -1, &gdg_ctx::fir_premac_min, nullptr, true },
    { "fir_premac_min_partitions_two_amps", "GDG_FIR_PREMAC_MIN_TWO", 1, 1 << 24, -1, &gdg_ctx::fir_premac_min_two, nullptr, true },
    { "stat_premac_launches_used", "GDG_STAT_PREMAC_USED", 0, 0x7fffffff, -1, &gdg_ctx::stat_premac_used, nullptr, false },
    { "fir_premac_lds_bytes", "GDG_FIR_PREMAC_LDS", -1, 65536, -1, &gdg_ctx::fir_premac_lds, nullptr, true },
    { "share_ir_spectra", "GDG_SHARE_IR_SPECTRA", 0, 1, -1, nullptr, &gdg_ctx::share_spectra, false },
    { "fft_half_lds_mask", "GDG_FFT_HALF_LDS", 0, 63, GDG_KNOB_FFT_HALF_LDS, nullptr, nullptr, false },
    { "fir_forward_per_channel", "GDG_FWD_PER_CHANNEL", 0, 1, GDG_KNOB_FWD_PER_CHANNEL, nullptr, nullptr, false },
    { "fir_forward_wave_local", "GDG_WAVE_FFT", 0, 3, GDG_KNOB_WAVE_FFT, nullptr, nullptr, false },
    { "fir_mac_variant", "GDG_MAC_VARIANT", 0, 15, GDG_KNOB_MAC_VARIANT, nullptr, nullptr, false },
    /* the segments */
    { "seg_two_per_cu", "GDG_SEG_FAST", 0, 1, -1, nullptr, &gdg_ctx::seg_fast, true },
    { "seg_two_per_cu_min_channels", "GDG_SEG_FAST_MIN", 0, 1 << 20, -1, &gdg_ctx::seg_fast_min, nullptr, true },
    { "seg_wave_max_channels", "GDG_SEG_WAVE_MAX", 0, 1 << 20, -1, &gdg_ctx::seg_wave_max, nullptr, true },
    { "seg_wave_release_max_channels", "GDG_SEG_WAVE_RELEASE_MAX", 0, 1 << 20, -1, &gdg_ctx::seg_wave_release_max, nullptr, true },
    { "seg_tile_max_channels", "GDG_SEG_TILE_MAX", 0, 1 << 20, -1, &gdg_ctx::seg_tile_max, nullptr, true },
    { "seg_os_tiles_max_channels", "GDG_SEG_OS_TILES_MAX", 0, 1 << 20, -1, &gdg_ctx::seg_os_tiles_max, nullptr, true },
    { "seg_os_tiles_prefix", "GDG_SEG_OS_PREFIX", 0, 1, -1, nullptr, &gdg_ctx::seg_os_prefix, true },
    { "seg_reverb_ahead_max_channels", "GDG_SEG_REVERB_AHEAD_MAX", 0, 1 << 20, -1, &gdg_ctx::seg_reverb_ahead_max, nullptr, true },
    { "wave_spin_limit_ms", "GDG_WAVE_SPIN_LIMIT_MS", 1, 600000, -1, &gdg_ctx::wave_spin_ms, nullptr, false },
    { "debug_stall_unit", "GDG_DEBUG_STALL_UNIT", -1, 1 << 30, -1, &gdg_ctx::debug_stall_unit, nullptr, false },
    { "plan_patch", "GDG_PLAN_PATCH", 0, 1, -1, nullptr, &gdg_ctx::plan_patch, false },
    { "scan_tables_max", "GDG_SCAN_TABLES_MAX", 1, 1 << 20, -1, &gdg_ctx::scan_tables_max, nullptr, false },
    /* host paths, tuner, profiling */
    { "pcie_groups", "GDG_PCIE_GROUPS", 0, 16, -1, &gdg_ctx::pcie_groups_forced, nullptr, false },
    { "device_groups_default", "GDG_DEVICE_GROUPS", 0, 16, -1, &gdg_ctx::device_groups_env, nullptr, false },
    { "copy_threads", "GDG_COPY_THREADS", 1, 256, -1, &gdg_ctx::copy_threads, nullptr, false },
    { "numa", "GDG_NUMA", 0, 2, -1, &gdg_ctx::numa_mode, nullptr, false },
    { "tuner_parts", "GDG_TUNER_PARTS", 0, 24, GDG_KNOB_TUNER_PARTS, nullptr, nullptr, false },
    { "tuner_poll_results", "GDG_TUNER_POLL", 0, 1, -1, &gdg_ctx::tuner_poll, nullptr, false },
    { "tuner_long_transform", "GDG_TUNER_LONG", 0, 1, -1, &gdg_ctx::tuner_long, nullptr, false },
    { "profile_attach", "GDG_PROFILE_ATTACH", 0, 1, -1, nullptr, &gdg_ctx::prof_attach, false },
};
static const OptionDef *find_option(const char *key) {
    if (!key) return nullptr;
    for (const OptionDef &o : g_options) if (strcmp(o.key, key) == 0) return &o;
    return nullptr;
}
static void option_store(gdg_ctx *ctx, const OptionDef &o, long long v) {
    if (o.knob >= 0) gdg_knob_set(o.knob, (int)v);
    else if (o.field) ctx->*(o.field) = (int)v;
    else ctx->*(o.flag) = v != 0;
}
static void options_from_env(gdg_ctx *ctx) {
    for (const OptionDef &o : g_options) {
        if (o.knob >= 0) continue;                      /* gdg_knob_get reads its variable itself, once */
        const char *e = getenv(o.env);
        if (!e) continue;
        long long v = atoll(e);
        if (o.flag) v = v != 0;
        if (v < o.lo) v = o.lo;
        if (v > o.hi) v = o.hi;
        option_store(ctx, o, v);
    }
}

int gdg_ctx_create(int n_channels, int max_frames, int device, gdg_ctx **out) {
    if (!out) return GDG_ERR_INVALID;
    *out = nullptr;
    if (n_channels <= 0 || max_frames <= 0) return GDG_ERR_INVALID;
    if (max_frames > GDG_MAX_FRAMES) return GDG_ERR_UNSUPPORTED;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) return GDG_ERR_NO_DEVICE;
    if (device < 0 || device >= n) return GDG_ERR_NO_DEVICE;
    if (hipSetDevice(device) != hipSuccess) return GDG_ERR_NO_DEVICE;
    gdg_ctx *ctx = new gdg_ctx();
    ctx->nch = n_channels;
    ctx->max_frames = max_frames;
    ctx->w_stride = (size_t)max_frames;
    ctx->device = device;
    ctx->chains.resize((size_t)n_channels);
    options_from_env(ctx);                     /* debug overrides of the options' defaults (gdg_ctx_set_option) */
    numa_discover(ctx);
    ctx->sp_az.assign((size_t)n_channels, 0.0);
    ctx->sp_dist.assign((size_t)n_channels, 0.0);
    ctx->sp_level.assign((size_t)n_channels, 1.0);
    size_t row = (size_t)n_channels * (size_t)max_frames * sizeof(double);
    bool ok = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking) == hipSuccess;
    ctx->arena.stream = ctx->stream;
    ctx->arena.defer_trim = true;               /* hipFree waits for the device: chunks go back in build_plan / gdg_ctx_trim, never inside a patch (arena.h) */
    ok = ok && hipMalloc((void **)&ctx->d_w0, row) == hipSuccess;
    ok = ok && hipMalloc((void **)&ctx->d_w1, row) == hipSuccess;
    ok = ok && hipMalloc((void **)&ctx->d_scratch, row) == hipSuccess;
    ok = ok && hipMalloc((void **)&ctx->d_error, 4 * sizeof(int)) == hipSuccess;       /* [error word | wave_spin_limit_ms | spare] */
    ok = ok && hipMemset(ctx->d_error, 0, 4 * sizeof(int)) == hipSuccess;
    ok = ok && hipMemcpy(ctx->d_error + 1, &ctx->wave_spin_ms, sizeof(int), hipMemcpyHostToDevice) == hipSuccess;
    /* oversampling tables: 77 + 155 taps, 6 + 18 Lanczos-3 weights (resample.go:36-66 evaluated once per phase) */
    const size_t os_base = 77 + 155 + 6 + 18;
    std::vector<double> tab(os_base + 2 * GDG_OS_NE(2) + 4 * GDG_OS_NE(4), 0.0);
    for (int k = 0; k < 39; k++) { tab[k] = GDG_AA2_HALF[k]; tab[76 - k] = GDG_AA2_HALF[k]; }
    for (int k = 0; k < 78; k++) { tab[77 + k] = GDG_AA4_HALF[k]; tab[77 + 154 - k] = GDG_AA4_HALF[k]; }
    for (int q = 0; q < 6; q++) tab[232 + q] = lanczos_kernel((double)(2 - q) + 0.5, 3.0);
    for (int r = 1; r < 4; r++)
        for (int q = 0; q < 6; q++) tab[238 + (r - 1) * 6 + q] = lanczos_kernel((double)(2 - q) + 0.25 * (double)r, 3.0);
    for (int F = 2; F <= 4; F += 2) {
        /* phase-major, zero padded copies for the register-blocked decimator (seg.hip os_decimate) */
        double *tp = tab.data() + os_base + (F == 4 ? 2 * GDG_OS_NE(2) : 0);
        const double *taps = tab.data() + (F == 4 ? 77 : 0);
        for (int r = 0; r < F; r++)
            for (int e = 0; e < GDG_OS_NE(F); e++) {
                int b = e - GDG_OS_PADLO(F), k = F * b - r;
                tp[r * GDG_OS_NE(F) + e] = (b >= 0 && k >= 0 && k < GDG_OS_TAPS(F)) ? taps[k] : 0.0;
            }
    }
    ok = ok && hipMalloc((void **)&ctx->d_os, tab.size() * sizeof(double)) == hipSuccess;
    ok = ok && hipMemcpy(ctx->d_os, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice) == hipSuccess;
    if (!ok) { gdg_ctx_destroy(ctx); return GDG_ERR_HIP; }
    ctx->os.taps2 = ctx->d_os;
    ctx->os.taps4 = ctx->d_os + 77;
    ctx->os.lanczos2 = ctx->d_os + 232;
    ctx->os.lanczos4 = ctx->d_os + 238;
    ctx->os.tapsP2 = ctx->d_os + os_base;
    ctx->os.tapsP4 = ctx->d_os + os_base + 2 * GDG_OS_NE(2);
    *out = ctx;
    return GDG_OK;
}

/* the caller has waited for every launch that used the unit */
static void free_unit(gdg_ctx *ctx, Unit &u) {
    DevArena &a = ctx->arena;
    a.release(u.d_ds); a.release(u.d_is); a.release(u.d_hist);
    a.release(u.d_prev); a.release(u.d_fdl); a.release(u.d_Y); a.release(u.d_pos);
    u = Unit();                     /* drops the unit's reference to its (possibly shared) IR spectra */
}

int gdg_ctx_destroy(gdg_ctx *ctx) {
    if (!ctx) return GDG_ERR_INVALID;
    enter(ctx);
    if (ctx->stream) hipStreamSynchronize(ctx->stream);
    if (ctx->premac_stream) { hipStreamSynchronize(ctx->premac_stream); hipStreamDestroy(ctx->premac_stream); hipEventDestroy(ctx->ev_fir_done); hipEventDestroy(ctx->ev_premac); }
    for (auto &u : ctx->units) if (u.alive) free_unit(ctx, u);
    ctx->spectra.clear();
    if (const char *e = getenv("GDG_ARENA_TRACE")) if (atoi(e))
        fprintf(stderr, "[arena] %d channels: %zu chunks, %.1f MiB (peak %.1f MiB, %zu chunks given back), %zu blocks live, zero fills issued %zu, avoided %zu\n",
                ctx->nch, ctx->arena.chunks_held(), (double)ctx->arena.total / 1048576.0, (double)ctx->arena.peak_total / 1048576.0, ctx->arena.trimmed,
                ctx->arena.live.size(), ctx->arena.fills, ctx->arena.fills_saved);
    ctx->arena.destroy();
    for (void *p : ctx->user_allocs) hipFree(p);
    for (auto &kv : ctx->fir_tables) { hipFree(kv.second.first); hipFree(kv.second.second); }
    for (auto &p : ctx->prof) { hipEventDestroy(p.a); hipEventDestroy(p.b); }
    for (auto e : ctx->event_pool) hipEventDestroy(e);
    hipFree(ctx->d_w0); hipFree(ctx->d_w1); hipFree(ctx->d_scratch); hipFree(ctx->d_error); hipFree(ctx->d_wave); hipFree(ctx->d_tile_xch);
    hipFree(ctx->d_stage_in); hipFree(ctx->d_stage_out); hipFree(ctx->d_blob); hipFree(ctx->d_os);
    hipFree(ctx->d_tuner_ring); hipFree(ctx->d_sp_hist);
    hipFree(ctx->d_note_freqs); /* d_tuner_out is the device view of h_tuner_out */ hipFree(ctx->d_tuner_work); hipFree(ctx->d_tuner_part); hipFree(ctx->d_tuner_twn); hipFree(ctx->d_tuner_twm);
    hipFree(ctx->d_sp_chan); hipFree(ctx->d_sp_out); hipFree(ctx->d_io[0]); hipFree(ctx->d_io[1]); hipFree(ctx->d_meter); hipFree(ctx->d_tick); hipFree(ctx->d_tock);
    for (auto st : ctx->gstreams) hipStreamDestroy(st);
    for (auto e : ctx->gjoin) hipEventDestroy(e);
    if (ctx->gfork) hipEventDestroy(ctx->gfork);
    for (int h = 0; h < 2; h++) {
        if (ctx->h_batch[h]) hipHostFree(ctx->h_batch[h]);
        if (ctx->batch_ready[h]) hipEventDestroy(ctx->batch_ready[h]);
        if (ctx->batch_moved[h]) hipEventDestroy(ctx->batch_moved[h]);
        for (int c = 0; c < 4; c++) if (ctx->batch_chunk[h][c]) hipEventDestroy(ctx->batch_chunk[h][c]);
    }
    if (ctx->batch_stream) hipStreamDestroy(ctx->batch_stream);
    for (int h = 0; h < 2; h++) {
        if (ctx->h_up[h]) hipHostFree(ctx->h_up[h]);
        if (ctx->batch_up_ready[h]) hipEventDestroy(ctx->batch_up_ready[h]);
    }
    if (ctx->batch_begin) hipEventDestroy(ctx->batch_begin);
    for (int i = 0; i < 6; i++) hipFree(ctx->batch_dev[i]);
    if (ctx->batch_up_stream) hipStreamDestroy(ctx->batch_up_stream);
    if (ctx->h_tuner_out) hipHostFree(ctx->h_tuner_out);
    if (ctx->h_stage_in) hipHostFree(ctx->h_stage_in);
    if (ctx->h_stage_out) hipHostFree(ctx->h_stage_out);
    if (ctx->stream) hipStreamDestroy(ctx->stream);
    destroy_copy_pool(ctx->copy_pool);
    destroy_copy_pool(ctx->copy_pool_up);
    delete ctx;
    return GDG_OK;
}

const char *gdg_last_error(const gdg_ctx *ctx) { return ctx ? ctx->err.c_str() : "null context"; }
int gdg_ctx_channels(const gdg_ctx *ctx) { return ctx ? ctx->nch : 0; }

int gdg_ctx_share_ir_spectra(gdg_ctx *ctx, int enable) {
    if (!ctx) return GDG_ERR_INVALID;
    ctx->share_spectra = enable != 0;       /* affects power amps prepared from now on */
    return GDG_OK;
}
int gdg_ctx_set_option(gdg_ctx *ctx, const char *key, long long value) {
    if (!ctx) return GDG_ERR_INVALID;
    const OptionDef *o = find_option(key);
    if (!o) return fail(ctx, GDG_ERR_INVALID, "unknown option \"%s\"", key ? key : "(null)");
    if (value < o->lo || value > o->hi) return fail(ctx, GDG_ERR_INVALID, "option %s = %lld: %lld to %lld", key, value, o->lo, o->hi);
    enter(ctx);                                 /* free-running groups join before a launch shape changes under them */
    if (strcmp(key, "copy_threads") == 0 && value != ctx->copy_threads) {
        if (ctx->copy_pool) { destroy_copy_pool(ctx->copy_pool); ctx->copy_pool = nullptr; }
        if (ctx->copy_pool_up) { destroy_copy_pool(ctx->copy_pool_up); ctx->copy_pool_up = nullptr; }
    }
    if (strcmp(key, "numa") == 0 && value != ctx->numa_mode) { int rc = numa_rebind(ctx, (int)value); if (rc != GDG_OK) return rc; }
    option_store(ctx, *o, value);
    if (o->replans) ctx->dirty = true;
    if (strcmp(key, "wave_spin_limit_ms") == 0)       /* the kernels read it from the context's error block (seg.hip wave_spin_expired) */
        HIP_TRY(ctx, hipMemcpyAsync(ctx->d_error + 1, &ctx->wave_spin_ms, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
    return GDG_OK;
}

int gdg_ctx_get_option(gdg_ctx *ctx, const char *key, long long *value) {
    if (!ctx || !value) return GDG_ERR_INVALID;
    const OptionDef *o = find_option(key);
    if (!o) return fail(ctx, GDG_ERR_INVALID, "unknown option \"%s\"", key ? key : "(null)");
    if (o->knob >= 0) *value = gdg_knob_get(o->knob);
    else if (o->field) *value = ctx->*(o->field);
    else *value = (ctx->*(o->flag)) ? 1 : 0;
    return GDG_OK;
}

int gdg_option_count(void) { return (int)(sizeof(g_options) / sizeof(g_options[0])); }
const char *gdg_option_name(int index) { return (index >= 0 && index < gdg_option_count()) ? g_options[index].key : NULL; }

void *gdg_ctx_stream(const gdg_ctx *ctx) {
    if (!ctx) return nullptr;
    join_groups(const_cast<gdg_ctx *>(ctx));           /* work enqueued on the returned stream from here on follows everything already submitted */
    join_premac(const_cast<gdg_ctx *>(ctx), true);
    return (void *)ctx->stream;
}

/* ---- units ------------------------------------------------------------------------------------- */

Unit *get_unit(gdg_ctx *ctx, int handle) {
    if (!ctx || handle < 0 || handle >= (int)ctx->units.size() || !ctx->units[(size_t)handle].alive) return nullptr;
    return &ctx->units[(size_t)handle];
}

int gdg_unit_create(gdg_ctx *ctx, int channel, int unit_type, int *handle) {
    if (!ctx || !handle) return GDG_ERR_INVALID;
    if (channel < 0 || channel >= ctx->nch) return fail(ctx, GDG_ERR_INVALID, "channel %d out of range", channel);
    if (unit_type < 0 || unit_type >= GDG_UNIT_COUNT) return fail(ctx, GDG_ERR_INVALID, "Failed to create effects unit.");
    enter(ctx);
    size_t h = 0;
    while (h < ctx->units.size() && ctx->units[h].alive) h++;
    if (h == ctx->units.size()) ctx->units.emplace_back();
    Unit &u = ctx->units[h];
    u = Unit();
    u.alive = true;
    u.type = unit_type;
    u.channel = channel;
    memcpy(u.params, g_param_default[unit_type], sizeof(u.params));
    hipError_t e = ctx->arena.alloc_zeroed((void **)&u.d_ds, GDG_DS_LEN * sizeof(double), ctx->stream);
    if (e == hipSuccess) e = ctx->arena.alloc_zeroed((void **)&u.d_is, GDG_IS_LEN * sizeof(int), ctx->stream);
    if (e != hipSuccess) {
        hipStreamSynchronize(ctx->stream);
        free_unit(ctx, u);          /* the slot goes back to "not alive"; nothing leaks */
        return fail(ctx, GDG_ERR_HIP, "gdg_unit_create: %s", hipGetErrorString(e));
    }
    *handle = (int)h;
    return GDG_OK;
}

int gdg_unit_destroy(gdg_ctx *ctx, int handle) {
    Unit *u = get_unit(ctx, handle);
    if (!u) return fail(ctx, GDG_ERR_INVALID, "bad unit handle %d", handle);
    enter(ctx);
    hipStreamSynchronize(ctx->stream);
    for (auto &chain : ctx->chains)
        chain.erase(std::remove_if(chain.begin(), chain.end(), [&](const Slot &s) { return s.handle == handle; }), chain.end());
    free_unit(ctx, *u);
    ctx->dirty = true;
    return GDG_OK;
}

int gdg_unit_set_param(gdg_ctx *ctx, int handle, int param_index, int32_t value) {
    Unit *u = get_unit(ctx, handle);
    if (!u) return fail(ctx, GDG_ERR_INVALID, "bad unit handle %d", handle);
    if (param_index < 0 || param_index >= g_param_count[u->type]) return fail(ctx, GDG_ERR_INVALID, "bad parameter index %d", param_index);
    if (u->params[param_index] != value) {
        u->params[param_index] = value;
        /* The reference's setter is a mutex and one store (effects/effects.go:283-345).  Here a parameter reaches the device as ONE
         * gdg_seg_unit of the plan's descriptor blob: the next process call re-derives that unit's constants and patches them in place
         * (apply_patches) -- chain shape, launches and every other descriptor stay.  A power amp's parameters only matter through its
         * taps (gdg_unit_set_fir), and a unit that is not in the plan (bypassed, or in no chain) has nothing on the device to update.
         * Layout changes (gdg_chain_set), frame size, rate and new filters still rebuild the plan. */
        /* a shaper's oversampling factor decides which LAUNCHES the plan holds (a segment cut at the unit, os_tiles_kernel<2 / 4>): a new plan */
        const bool shaper = u->type == GDG_UNIT_OVERDRIVE || u->type == GDG_UNIT_DISTORTION || u->type == GDG_UNIT_EXCESS;
        const bool os_changed = shaper && param_index == (u->type == GDG_UNIT_OVERDRIVE ? 5 : (u->type == GDG_UNIT_DISTORTION ? 3 : 2));
        if (ctx->dirty || !ctx->plan_patch || os_changed) ctx->dirty = true;
        else if (u->type != GDG_UNIT_POWERAMP && (size_t)handle < ctx->plan_unit_slot.size() && ctx->plan_unit_slot[(size_t)handle] >= 0) {
            if (std::find(ctx->patch_units.begin(), ctx->patch_units.end(), handle) == ctx->patch_units.end()) ctx->patch_units.push_back(handle);
        }
    }
    return GDG_OK;
}

int gdg_unit_get_param(gdg_ctx *ctx, int handle, int param_index, int32_t *value) {
    Unit *u = get_unit(ctx, handle);
    if (!u || !value) return fail(ctx, GDG_ERR_INVALID, "bad unit handle %d", handle);
    if (param_index < 0 || param_index >= g_param_count[u->type]) return fail(ctx, GDG_ERR_INVALID, "bad parameter index %d", param_index);
    *value = u->params[param_index];
    return GDG_OK;
}

int gdg_unit_set_fir(gdg_ctx *ctx, int handle, const double *taps, int n_taps) {
    Unit *u = get_unit(ctx, handle);
    if (!u) return fail(ctx, GDG_ERR_INVALID, "bad unit handle %d", handle);
    if (u->type != GDG_UNIT_POWERAMP) return fail(ctx, GDG_ERR_INVALID, "unit %d is not a power amp", handle);
    if (n_taps < 0 || (n_taps > 0 && !taps)) return fail(ctx, GDG_ERR_INVALID, "bad taps");
    u->taps.assign(taps, taps + n_taps);
    u->fir_dirty = true;            /* new filter => fresh state (effects/poweramp.go:132-181) */
    u->fir_live = false;
    ctx->dirty = true;
    return GDG_OK;
}

static int zero_unit_state(gdg_ctx *ctx, Unit &u) {
    HIP_TRY(ctx, hipMemsetAsync(u.d_ds, 0, GDG_DS_LEN * sizeof(double), ctx->stream));
    HIP_TRY(ctx, hipMemsetAsync(u.d_is, 0, GDG_IS_LEN * sizeof(int), ctx->stream));
    if (u.d_hist) HIP_TRY(ctx, hipMemsetAsync(u.d_hist, 0, u.hist_len * sizeof(double), ctx->stream));
    u.fir_dirty = true;
    u.fir_live = false;
    return GDG_OK;
}

int gdg_unit_reset(gdg_ctx *ctx, int handle) {
    Unit *u = get_unit(ctx, handle);
    if (!u) return fail(ctx, GDG_ERR_INVALID, "bad unit handle %d", handle);
    enter(ctx);
    ctx->dirty = true;
    return zero_unit_state(ctx, *u);
}

int gdg_chain_set(gdg_ctx *ctx, int channel, const int *handles, const uint8_t *bypass, int n) {
    if (!ctx) return GDG_ERR_INVALID;
    if (channel < 0 || channel >= ctx->nch) return fail(ctx, GDG_ERR_INVALID, "channel %d out of range", channel);
    if (n < 0 || (n > 0 && (!handles || !bypass))) return fail(ctx, GDG_ERR_INVALID, "bad chain");
    std::vector<Slot> chain;
    for (int i = 0; i < n; i++) {
        Unit *u = get_unit(ctx, handles[i]);
        if (!u) return fail(ctx, GDG_ERR_INVALID, "bad unit handle %d in chain", handles[i]);
        if (u->channel != channel) return fail(ctx, GDG_ERR_INVALID, "unit %d belongs to channel %d", handles[i], u->channel);
        for (auto &s : chain) if (s.handle == handles[i]) return fail(ctx, GDG_ERR_INVALID, "unit %d appears twice", handles[i]);
        chain.push_back(Slot{ handles[i], bypass[i] != 0 });
    }
    ctx->chains[(size_t)channel] = chain;
    ctx->dirty = true;
    return GDG_OK;
}


/* ---- debug: the oversampler / decimator tiles on their own ------------------------------------------------------------ */

int gdg_debug_oversample_decimate(gdg_ctx *ctx, int factor, const double *in, int n, double *state, double *oversampled, double *decimated) {
    if (!ctx || !in || !state || !decimated) return GDG_ERR_INVALID;
    if (factor != 2 && factor != 4) return fail(ctx, GDG_ERR_INVALID, "oversampling factor %d: 2 or 4", factor);
    if (n <= 0 || n > GDG_MAX_FRAMES) return fail(ctx, GDG_ERR_INVALID, "%d samples: 1 to %d", n, GDG_MAX_FRAMES);
    enter(ctx);
    const size_t n_state = 8 + (size_t)GDG_OS_TAPS(factor) - 1, n_up = (size_t)factor * (size_t)n;
    double *d = nullptr;                                      /* [in | state | up | down] */
    HIP_TRY(ctx, hipMalloc((void **)&d, ((size_t)n + n_state + n_up + (size_t)n) * sizeof(double)));
    double *d_in = d, *d_state = d + n, *d_up = d_state + n_state, *d_down = d_up + n_up;
    int rc = GDG_OK;
    auto body = [&]() -> int {
        HIP_TRY(ctx, hipMemcpyAsync(d_in, in, (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(d_state, state, n_state * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, gdg_launch_os_debug(factor, d_in, n, d_state, d_up, d_down, ctx->os, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(state, d_state, n_state * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        if (oversampled) HIP_TRY(ctx, hipMemcpyAsync(oversampled, d_up, n_up * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(decimated, d_down, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        return GDG_OK;
    };
    rc = body();
    hipStreamSynchronize(ctx->stream);
    hipFree(d);
    return rc;
}

/* ---- profiling -------------------------------------------------------------------------------------- */

hipEvent_t take_event(gdg_ctx *ctx) {
    if (!ctx->event_pool.empty()) { hipEvent_t e = ctx->event_pool.back(); ctx->event_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    hipEventCreate(&e);
    return e;
}

/* attached = true: the launch inside the scope takes the two events itself (hipExtLaunchKernelGGL: the kernel's own begin / end timestamps);
 * otherwise the events are recorded on the stream before and after whatever the scope launches */

int gdg_profile_enable(gdg_ctx *ctx, int enable) {
    if (!ctx) return GDG_ERR_INVALID;
    ctx->profiling = enable < 0 ? 0u : (unsigned)enable;
    return GDG_OK;
}

int gdg_profile_sample(gdg_ctx *ctx, int every) {
    if (!ctx || every < 1) return GDG_ERR_INVALID;
    ctx->prof_every = every;
    ctx->prof_calls = 0;
    return GDG_OK;
}

int gdg_profile_read(gdg_ctx *ctx, int kind, double *total_ms, int *launches) {
    if (!ctx || kind < 0 || kind >= GDG_K_COUNT) return GDG_ERR_INVALID;
    enter(ctx, true);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    double total = 0.0;
    int n = 0;
    std::vector<ProfEvent> keep;
    for (auto &p : ctx->prof) {
        if (p.kind != kind) { keep.push_back(p); continue; }
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, p.a, p.b) == hipSuccess) { total += ms; n++; }
        ctx->event_pool.push_back(p.a);
        ctx->event_pool.push_back(p.b);
    }
    ctx->prof.swap(keep);
    if (total_ms) *total_ms = total;
    if (launches) *launches = n;
    return GDG_OK;
}


/* ---- device memory helpers --------------------------------------------------------------------------------- */

int gdg_device_alloc(gdg_ctx *ctx, size_t bytes, void **d_ptr) {
    if (!ctx || !d_ptr) return GDG_ERR_INVALID;
    enter(ctx);
    HIP_TRY(ctx, hipMalloc(d_ptr, bytes));
    ctx->user_allocs.push_back(*d_ptr);
    return GDG_OK;
}

int gdg_device_free(gdg_ctx *ctx, void *d_ptr) {
    if (!ctx) return GDG_ERR_INVALID;
    enter(ctx);
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    auto it = std::find(ctx->user_allocs.begin(), ctx->user_allocs.end(), d_ptr);
    if (it != ctx->user_allocs.end()) ctx->user_allocs.erase(it);
    HIP_TRY(ctx, hipFree(d_ptr));
    return GDG_OK;
}

int gdg_copy_to_device(gdg_ctx *ctx, void *d_dst, const void *h_src, size_t bytes) {
    if (!ctx) return GDG_ERR_INVALID;
    enter(ctx);
    HIP_TRY(ctx, hipMemcpyAsync(d_dst, h_src, bytes, hipMemcpyHostToDevice, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return GDG_OK;
}

int gdg_copy_to_host(gdg_ctx *ctx, void *h_dst, const void *d_src, size_t bytes) {
    if (!ctx) return GDG_ERR_INVALID;
    enter(ctx);
    HIP_TRY(ctx, hipMemcpyAsync(h_dst, d_src, bytes, hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    return GDG_OK;
}

/* fft.RealFourier / fft.RealInverseFourier (fft/fft.go:744-856, :863-990) as the FIR path computes them: the packed-real transforms
 * of fir.hip, stand-alone, so that the HIP FFT has known-answer tests of its own.  n real samples <->
 * n / 2 + 1 complex bins (re, im interleaved), n = 1 or a power of two from 2 to 16384.  Forward unscaled, inverse scaled by 1 / n, the
 * reference's SCALING_DEFAULT. */
static int fft_size_ok(gdg_ctx *ctx, int n) {
    if (n < 2 || n > 2 * GDG_MAX_FRAMES || (n & (n - 1)) != 0)
        return fail(ctx, GDG_ERR_INVALID, "transform size %d: a power of two from 2 to %d", n, 2 * GDG_MAX_FRAMES);
    return GDG_OK;
}

int gdg_fft_real(gdg_ctx *ctx, const double *samples, int n, double *spectrum) {
    if (!ctx || !samples || !spectrum) return GDG_ERR_INVALID;
    if (n == 1) { spectrum[0] = samples[0]; spectrum[1] = 0.0; return GDG_OK; }          /* fft.go:765-768: one element is its own transform */
    int rc = fft_size_ok(ctx, n);
    if (rc != GDG_OK) return rc;
    enter(ctx);
    const int P = n / 2;
    double2 *tw, *tw2;
    rc = fir_tables(ctx, P, &tw, &tw2);
    if (rc != GDG_OK) return rc;
    double *d_x = nullptr;
    double2 *d_out = nullptr;
    gdg_fir_irjob *d_job = nullptr;
    auto body = [&]() -> int {
        HIP_TRY(ctx, hipMalloc((void **)&d_x, (size_t)n * sizeof(double)));
        HIP_TRY(ctx, hipMalloc((void **)&d_out, (size_t)P * sizeof(double2)));
        HIP_TRY(ctx, hipMalloc((void **)&d_job, sizeof(gdg_fir_irjob)));
        gdg_fir_irjob job;
        memset(&job, 0, sizeof(job));
        job.a = d_x; job.b = d_x + P; job.hop = P; job.out = d_out;
        HIP_TRY(ctx, hipMemcpyAsync(d_x, samples, (size_t)n * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(d_job, &job, sizeof(job), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, gdg_launch_fir_ir(P, d_job, 1, 1.0, tw, tw2, ctx->stream));
        std::vector<double2> packed((size_t)P);
        HIP_TRY(ctx, hipMemcpyAsync(packed.data(), d_out, (size_t)P * sizeof(double2), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        /* bin 0 of the packed half spectrum carries (Re X[0], Re X[P]) */
        spectrum[0] = packed[0].x; spectrum[1] = 0.0;
        for (int k = 1; k < P; k++) { spectrum[2 * k] = packed[(size_t)k].x; spectrum[2 * k + 1] = packed[(size_t)k].y; }
        spectrum[2 * P] = packed[0].y; spectrum[2 * P + 1] = 0.0;
        return GDG_OK;
    };
    rc = body();
    hipFree(d_x); hipFree(d_out); hipFree(d_job);
    return rc;
}

int gdg_fft_real_inverse(gdg_ctx *ctx, const double *spectrum, int n, double *samples) {
    if (!ctx || !samples || !spectrum) return GDG_ERR_INVALID;
    if (n == 1) { samples[0] = spectrum[0]; return GDG_OK; }
    int rc = fft_size_ok(ctx, n);
    if (rc != GDG_OK) return rc;
    enter(ctx);
    const int P = n / 2;
    double2 *tw, *tw2;
    rc = fir_tables(ctx, P, &tw, &tw2);
    if (rc != GDG_OK) return rc;
    double *d_x = nullptr;
    double2 *d_Y = nullptr;
    gdg_fir_rawjob *d_job = nullptr;
    auto body = [&]() -> int {
        HIP_TRY(ctx, hipMalloc((void **)&d_x, (size_t)n * sizeof(double)));
        HIP_TRY(ctx, hipMalloc((void **)&d_Y, (size_t)P * sizeof(double2)));
        HIP_TRY(ctx, hipMalloc((void **)&d_job, sizeof(gdg_fir_rawjob)));
        std::vector<double2> packed((size_t)P);
        packed[0] = make_double2(spectrum[0], spectrum[2 * P]);          /* like fft.go:899-906 only Re X[0], Re X[n/2] are used */
        for (int k = 1; k < P; k++) packed[(size_t)k] = make_double2(spectrum[2 * k], spectrum[2 * k + 1]);
        gdg_fir_rawjob job;
        memset(&job, 0, sizeof(job));
        job.Y = d_Y; job.first = d_x; job.second = d_x + P; job.hop = P;
        HIP_TRY(ctx, hipMemcpyAsync(d_Y, packed.data(), (size_t)P * sizeof(double2), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(d_job, &job, sizeof(job), hipMemcpyHostToDevice, ctx->stream));
        HIP_TRY(ctx, gdg_launch_fir_raw_inv(P, d_job, 1, 1.0 / (double)n, tw, tw2, ctx->stream));
        HIP_TRY(ctx, hipMemcpyAsync(samples, d_x, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, ctx->stream));
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        return GDG_OK;
    };
    rc = body();
    hipFree(d_x); hipFree(d_Y); hipFree(d_job);
    return rc;
}

/* strided device-to-device copy of n_rows rows of row_len float64, enqueued on the context's stream: what the batch loop's
 * copy(inputBuffers[i], input[offsetStart:offsetEnd]) / copy(output[offsetStart:offsetEnd], outputBuffers[i]) become when the
 * whole files live in HBM (controller/controller.go:3088-3099) */
int gdg_copy_rows_device(gdg_ctx *ctx, double *d_dst, size_t dst_stride, const double *d_src, size_t src_stride, size_t row_len, size_t n_rows) {
    if (!ctx || !d_dst || !d_src) return GDG_ERR_INVALID;
    if (row_len > dst_stride || row_len > src_stride) return fail(ctx, GDG_ERR_INVALID, "row length %zu exceeds a row stride", row_len);
    if (row_len == 0 || n_rows == 0) return GDG_OK;
    enter(ctx);
    HIP_TRY(ctx, hipMemcpy2DAsync(d_dst, dst_stride * sizeof(double), d_src, src_stride * sizeof(double), row_len * sizeof(double), n_rows,
                                  hipMemcpyDeviceToDevice, ctx->stream));
    return GDG_OK;
}
