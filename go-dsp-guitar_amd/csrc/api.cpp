/*
 * api.cpp -- host side of libgdg.so: the C-ABI of include/gdg.h on top of the HIP kernels.
 *
 * What happens here: unit/chain bookkeeping, derivation of the per-unit constants in the
 * reference's arithmetic, device state management (with the reference's "re-make => zero"
 * semantics), compilation of the per-channel chains into batched launches
 * [segment | FIR | segment | FIR ...] across all channels of the shard, and launch.
 * There is no CPU compute path in this file: every sample is produced by a HIP kernel.
 */
#include "../../include/gdg.h"
#include "gdg_internal.h"
#include "aa_taps.h"
#include "go_consts.h"
#include "notes.h"

#include <math.h>
#include <stdarg.h>
#include <stdio.h>
#include <string.h>

#include <unistd.h>
#include <pthread.h>
#include <sched.h>
#include <sys/syscall.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <functional>
#include <mutex>
#include <memory>
#include <map>
#include <string>
#include <thread>
#include <vector>

#include "arena.h"

#define NUM_FILTERS 8

/* ---- parameter tables (defaults), effects/<unit>.go create*() ------------------------------------ */
static const int g_param_count[GDG_UNIT_COUNT] = { 6, 3, 3, 5, 4, 3, 7, 3, 7, 6, 4, 4, 2, 2, 3, 3, 1, 3, 1, 1, 1 };
static const int32_t g_param_default[GDG_UNIT_COUNT][GDG_MAX_PARAMS] = {
    { 100, 0, 0, 440, 100, 0 }, { -20, -40, 50 }, { 0, 300, 3000 }, { 1, -40, -10, 300, 6000 },
    { 1, -40, -10, 100 }, { 1, 30, -20 }, { 1, -20, -20, -20, -20, -20, -20 }, { 0, 0, 0 },
    { 1, 50, 0, 0, 100, 0, 0 }, { 0, 0, 100, 0, 1, 0 }, { 0, 0, 0, 0 }, { 0, -2, -5, -5 },
    { 100, 30 }, { 100, 10 }, { 100, 10, 45 }, { 100, 50, -10 }, { 100 }, { 200, -5, -5 }, { 50 }, { 14 }, { 0 },
};

/* DevArena: arena.h over the HIP runtime */
struct HipArenaBackend {
    using err_t = hipError_t;
    using stream_t = hipStream_t;
    static err_t ok() { return hipSuccess; }
    static err_t malloc(void **p, size_t n) { return hipMalloc(p, n); }
    static void free(void *p) { (void)hipFree(p); }
    static err_t fill_zero(void *p, size_t n, stream_t st) { return hipMemsetAsync(p, 0, n, st); }
    static err_t wait(stream_t st) { return hipStreamSynchronize(st); }
    static void wait_device() { (void)hipDeviceSynchronize(); }
};
using DevArena = ArenaT<HipArenaBackend>;

/* IR spectra of one (taps, partition size) pair; power amps with identical composite filters share one copy in HBM
 * (the MAC then streams it from L2 / MALL for all but the first channel: SURVEY.md 8d, d < 1) */
struct SharedSpectra {
    std::vector<double> taps;
    int P = 0, K = 0, hop = 0;
    double2 *d_H = nullptr;
    DevArena *arena = nullptr;
    ~SharedSpectra() { if (d_H && arena) arena->release(d_H); }
};

struct Unit {
    bool alive = false;
    int type = 0, channel = 0;
    int32_t params[GDG_MAX_PARAMS] = { 0 };
    /* segment state */
    double *d_ds = nullptr;
    int *d_is = nullptr;
    double *d_hist = nullptr;
    size_t hist_len = 0;
    long long hist_key = -1;          /* what the current history layout was built for */
    int os_frames[2] = { -1, -1 };    /* frame size the 2x / 4x oversampler last saw */
    int bp_half_order = -1;
    /* FIR */
    std::vector<double> taps;
    bool fir_dirty = true;
    bool fir_live = false;
    int fir_P = 0, fir_K = 0, fir_hop = 0;     /* transform half size (power of two), partitions, samples per frame */
    int fir_R = 0;                             /* delay-line ring slots: fir_K + window - 1 */
    uint32_t fir_sr = 0;
    double *d_prev = nullptr;
    double2 *d_fdl = nullptr, *d_Y = nullptr;
    std::shared_ptr<SharedSpectra> H;
    int *d_pos = nullptr;
};

struct Slot { int handle; bool bypass; };
#define GDG_WAVE_STEPS 64             /* segment steps of a plan that can run as WAVE launches (a chain with more power amps than that walks) */
#define GDG_WAVE_GROUPS 16            /* = the most channel groups of a call (every group's launch of a step draws its own tickets) */

struct StepDesc {
    bool is_fir;
    int n;
    size_t offset;                    /* byte offset of its descriptor array inside the plan blob */
    bool shared_spectra = false;      /* FIR step: some channels read the same IR spectra */
    bool chain_next = false;          /* FIR step whose every channel feeds another power amp next (the following step): that amp's forward
                                       * transform rides on this step's inverse (fir_inv_kernel CHAIN) in per-frame calls */
    bool fast = false;                /* segment step: every unit of every channel works in place on 8192-sample frames -> the two-per-CU kernel (segf) */
    bool premac_ok = false;           /* FIR step: split shape (few channels), 8192-sample frames, every channel with K >= 2: the terms k >= 1 can be summed ahead */
    int wave_tickets = -1;            /* segment step: first of its GDG_WAVE_GROUPS ticket counters in d_wave (seg.hip, WAVE), -1: none */
    std::vector<std::pair<int, int>> group_range;     /* per channel group: (first descriptor, count) */
};

struct ProfEvent { int kind; hipEvent_t a, b; };
class CopyPool;
static void destroy_copy_pool(CopyPool *p);

struct gdg_ctx {
    int nch = 0, max_frames = 0, device = 0;
    hipStream_t stream = nullptr;
    mutable std::string err;
    std::vector<Unit> units;
    std::vector<std::vector<Slot>> chains;
    bool dirty = true;
    /* cached plan */
    int plan_frames = 0;
    uint32_t plan_sr = 0;
    const double *plan_in = nullptr;
    double *plan_out = nullptr;
    std::vector<int> plan_active, all_channels;
    int plan_stride = 0, plan_stride_out = 0;
    bool plan_by_channel = false;
    std::vector<StepDesc> steps;
    std::vector<int> plan_unit_slot;           /* unit handle -> index of its descriptor in the plan's array of gdg_seg_unit, -1: not in the plan */
    std::vector<char> plan_unit_fast;          /* ... and whether its segment runs on the two-per-CU kernel (scan tables for 16-sample chunks) */
    std::vector<char> plan_unit_fast_ok;       /* ... and whether the unit itself could (segf_unit_ok at plan time): a change of that rebuilds the plan */
    bool seg_fast = true;                      /* GDG_SEG_FAST=0: every segment on the general kernel (A/B measurements, bit-identity tests) */
    int seg_fast_min = 257;                    /* GDG_SEG_FAST_MIN: fewest channels of a call that take the two-per-CU kernel.  Up to a chip's worth of
                                                * channels (256 CUs) the general kernel's 1024 threads per channel run in ONE round and finish a frame
                                                * sooner (64 channels: 158 vs 163 us per step, 128: 214 vs 217, 256: 301 vs 305; W = 16: 77 / 102 / 147 vs
                                                * 83 / 108 / 152 us per frame); beyond that it needs a second round and the two-per-CU kernel wins
                                                * (512: 75-81 vs 60-65 us per segment launch; profiles/fast_min_ab_r04.txt) */
    /* Windows of few channels (seg.hip, WAVE): up to this many channels per launch a window's segment launch puts every FRAME of a channel on a
     * workgroup of its own, the frames meeting unit by unit through counters in HBM -- a GPU's share of the 512-channel job on eight GPUs is 64
     * channels, and one workgroup per channel walking the window leaves 3/4 of the CUs idle (64 channels, W = 16: 417 us per segment launch,
     * 52 of the 77 us per frame).  0: never.  gdg_ctx_set_option("seg_wave_max_channels"), env GDG_SEG_WAVE_MAX. */
    int seg_wave_max = 192;                    /* 64 / 128 / 192 / 256 channels, W = 16: 77 / 102 / 128 / 146 us per frame walking, 51 / 87 / 125 / 160 in flight */
    int scan_tables_max = 1024;                /* scan tables kept before a plan rebuild drops them all (a caller sweeping a parameter) */
    int pcie_groups_forced = 0;                /* channel groups of the host-buffer calls; 0: by channel count */
    int device_groups_env = 0;                 /* GDG_DEVICE_GROUPS: the debug override of gdg_ctx_set_overlap(0) */
    int copy_threads = 8;                      /* host copy workers of the host-buffer paths (made on first use) */
    int tuner_long = 0;                        /* 1: every analysis through the 262144-point transform pair (A/B, tests) */
    /* NUMA placement of the host paths (option "numa").  The CPU side of a host-buffer call is copying between the CALLER's pageable buffers
     * and the pinned slabs; the GPU's DMA engines reach either socket's memory at PCIe speed.  So the default (2) puts the copy workers and
     * the pinned slabs on the node the caller runs on when they are made; 1 puts them on the device's node (deterministic per GPU, but a
     * caller on the other socket then has every byte read across the socket link: batch run 56 -> 73 ms); 0 leaves both to the scheduler and
     * to hipHostMalloc's default (profiles/host_path_numa_r05.txt: both sockets within 1 % for the batch run and 4 % for the staged call at 2,
     * 30 % / 1 % apart at 1, 0 % / 10 % at 0). */
    int numa_mode = 2;                         /* 0: nothing; 1: the device's node; 2: the node the CALLER runs on when the pool / a slab is made */
    int numa_node = -1;                        /* /sys/bus/pci/devices/<bus id>/numa_node of the device, -1: unknown or a one-node host */
    std::vector<int> numa_cpus;                /* /sys/devices/system/node/node<N>/cpulist */
    std::vector<std::vector<int>> node_cpus;   /* every node's CPUs (mode 2) */
    /* Small shards, per-frame calls (one GPU's share of a job split over several): the convolution's multiply-accumulate is the one kernel
     * of the step that does not depend on the frame for 7/8 of its work -- the terms k = K - 1 .. 1 of Y = sum_k FDL[pos - k] H[k] only need
     * frames that are already in the delay line.  So when a call ends, that part of the NEXT frame's sum is launched on a stream of its own
     * (the "premac": fir_mac_kernel with k_lo = 1 into Y) and runs beside the call's last segment and the next call's first (64 workgroups
     * on a 256-CU chip); the next call's inverse kernel adds the newest term and transforms (fir_inv_kernel FUSED = 4): 2 spectra per channel
     * on the critical path instead of 2 K.  Every multiply-accumulate kernel sums k DESCENDING, so the split sum has the bits of the whole one.
     * The premac is speculative: any library call but a process call drops it (the next call then runs the whole sum). */
    int fir_premac = 1;                        /* option "fir_premac": 0 never */
    int fir_premac_min = 384;                  /* fewest partitions (sum of K over a launch's channels) worth it: the two cross-stream hops and the
                                                * three extra spectra of the inverse kernel cost ~20 us per step -- the multiply-accumulate of 48 x 8
                                                * partitions takes that long (16 x 8: 113.7 -> 123.4 us per step with it, 64 x 4 (config 3): 137 -> 141) */
    hipStream_t premac_stream = nullptr;
    hipEvent_t ev_fir_done = nullptr, ev_premac = nullptr;
    bool premac_valid = false;                 /* Y of every premac step holds the terms k >= 1 of the plan's NEXT frame */
    bool premac_outstanding = false;           /* ... and the context's stream has not been ordered behind that launch yet */
    int *d_wave = nullptr;                     /* [GDG_WAVE_STEPS x GDG_WAVE_GROUPS ticket counters | one counter per unit in a segment]: zero between launches */
    size_t d_wave_cap = 0;
    std::vector<int> patch_units;              /* units whose parameters changed since the plan was built: their descriptors are patched in place */
    bool plan_patch = true;                    /* GDG_PLAN_PATCH=0: every parameter change rebuilds the whole plan (A/B measurements) */
    std::vector<unsigned char> blob;
    unsigned char *d_blob = nullptr;
    size_t d_blob_cap = 0;
    size_t units_offset = 0;
    /* buffers */
    double *d_w0 = nullptr, *d_w1 = nullptr, *d_scratch = nullptr;
    int window = 1;                            /* frames per channel and call of gdg_process_window_device (time blocking) */
    size_t w_stride = 0;                       /* row stride of d_w0 / d_w1: window * max_frames */
    double *d_stage_in = nullptr, *d_stage_out = nullptr;
    double *h_stage_in = nullptr, *h_stage_out = nullptr;
    int stage_out_stride = 0;         /* > 0: d_stage_out holds one complete block of chain outputs, row c = channel c, this stride */
    int stage_out_frames = 0;         /* ... of this many frames per row */
    int *d_error = nullptr;
    DevArena arena;                   /* per-unit state (see DevArena) */
    std::map<std::vector<double>, double *> scan_tabs;     /* scan tables by their coefficients: one copy per distinct set (scan_tables) */
    std::vector<void *> user_allocs;  /* gdg_device_alloc blocks the caller has not freed (released with the context) */
    /* tables */
    std::map<int, std::pair<double2 *, double2 *>> fir_tables;
    std::multimap<uint64_t, std::weak_ptr<SharedSpectra>> spectra;     /* content hash -> live IR spectra */
    std::vector<std::shared_ptr<SharedSpectra>> pending_ir;           /* spectra allocated by the plan being built, transformed together (flush_ir) */
    bool share_spectra = true;
    /* FIR launch shape.  -1 (default): by channel count -- the fused kernel (one workgroup per channel: multiply-accumulate
     * straight into the inverse transform) needs >= ~128 channels to fill the 256 CUs; below that the multiply-accumulate runs
     * as its own bin-tiled kernel (32 workgroups per channel) followed by the inverse (profiles/channels_sweep_r02.txt).
     * GDG_FIR_FUSED=0 / 1 forces one shape (A/B measurements). */
    int fir_fused = -1;
    int fir_split_max = 96;           /* GDG_FIR_SPLIT_MAX: largest launch (channels) that takes the split shape */
    bool fir_chain = true;            /* GDG_FIR_CHAIN=0: adjacent power amps keep separate launches (A/B measurements, bit-identity tests) */
    double *d_os = nullptr;
    gdg_os_tables os;
    /* profiling */
    unsigned profiling = 0;                  /* bit 0: everything; bit k+1: kernel kind k */
    int prof_every = 1;                      /* gdg_profile_sample: bracket every n-th process call only */
    unsigned long long prof_calls = 0;
    bool prof_now = true;
    bool prof_attach = true;                 /* the fused convolution kernel takes its events itself (kernel timestamps); GDG_PROFILE_ATTACH=0: recorded around it */
    std::vector<ProfEvent> prof;
    std::vector<hipEvent_t> event_pool;
    /* tuner / spatializer */
    double *d_tuner_ring = nullptr;
    int tuner_wp = 0;
    uint32_t tuner_sr = 0;
    double *d_note_freqs = nullptr;
    gdg_tuner_out *d_tuner_out = nullptr, *h_tuner_out = nullptr;      /* results on the device / in pinned host memory */
    double2 *d_tuner_work = nullptr, *d_tuner_twn = nullptr, *d_tuner_twm = nullptr;
    double2 *d_tuner_part = nullptr;           /* partial sums of a short-lag analysis split over several workgroups per channel */
    std::vector<double> sp_az, sp_dist, sp_level;
    uint32_t sp_hist_sr = 96000;
    double *d_sp_hist = nullptr;               /* [2][nch][sp_hist_len]: read this block / written for the next (sp_hist_cur) */
    int sp_hist_len = 0, sp_hist_cur = 0;
    gdg_spat_chan *d_sp_chan = nullptr;
    double *d_sp_out = nullptr;
    bool sp_dirty = true;
    /* io (wave codecs, resample.Time, level meters) */
    void *d_io[2] = { nullptr, nullptr };
    size_t io_cap[2] = { 0, 0 };
    gdg_meter_rec *d_meter = nullptr;
    int n_meter = 0;
    /* the batch run's PCIe side: two pinned halves, a copy stream and events (ensure_batch_pipe) */
    unsigned char *h_batch[2] = { nullptr, nullptr };
    size_t h_batch_cap = 0;
    hipStream_t batch_stream = nullptr;
    hipEvent_t batch_ready[2] = { nullptr, nullptr }, batch_moved[2] = { nullptr, nullptr };
    /* ... and the streamed upload of the inputs that need no resampling: two more pinned halves, a stream, events */
    unsigned char *h_up[2] = { nullptr, nullptr };
    size_t h_up_cap = 0;
    hipStream_t batch_up_stream = nullptr;
    hipEvent_t batch_up_ready[2] = { nullptr, nullptr }, batch_begin = nullptr;
    /* the batch run's device buffers (inputs, window, encoded steps, arena, upload halves, planar scratch): kept from call to call,
     * grown when a batch needs more -- allocating and mapping gigabytes per call cost more than the run (gdg_batch_release frees them) */
    void *batch_dev[6] = { nullptr, nullptr, nullptr, nullptr, nullptr, nullptr };
    size_t batch_dev_cap[6] = { 0, 0, 0, 0, 0, 0 };
    /* channel groups of the host-buffer paths: group g's upload, kernels and download run on stream g, so one group's
     * PCIe transfers overlap the other groups' kernels (channels are independent) */
    int plan_groups = 1;
    std::vector<size_t> plan_bounds;           /* first active index of every channel group (+ the end) the plan was built for */
    int overlap_groups = 0;                    /* 0: automatic (device_groups) */
    bool groups_pending = false;               /* group streams hold work the context's stream has not been ordered after */
    std::vector<hipStream_t> gstreams;
    std::vector<hipEvent_t> gjoin;
    hipEvent_t gfork = nullptr;
    CopyPool *copy_pool = nullptr;             /* host copy workers of the host-buffer paths, made on first use */
    /* metronome (metronome/metronome.go): sounds in HBM, the two counters on the host */
    double *d_tick = nullptr, *d_tock = nullptr;
    uint32_t n_tick = 0, n_tock = 0;
    uint32_t met_sample_counter = 0, met_tick_counter = 0, met_beats = 4, met_bpm = 120, met_sr = 96000;
};

static int fail(const gdg_ctx *ctx, int code, const char *fmt, ...) {
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) ctx->err = buf;
    return code;
}

/* Device-resident calls may leave their channel groups running on streams of their own (process_rows, `free_run`); whatever
 * touches the context next -- any entry point -- first makes the context's stream wait for them. */
static void join_groups(gdg_ctx *ctx);
/* the context's stream behind the premac launch; `keep` = the caller changes no state (synchronize, stream, profiling): the sums stay usable */
static void join_premac(gdg_ctx *ctx, bool keep) {
    if (ctx->premac_outstanding) {
        hipStreamWaitEvent(ctx->stream, ctx->ev_premac, 0);
        ctx->premac_outstanding = false;
    }
    if (!keep) ctx->premac_valid = false;
}
static void enter(gdg_ctx *ctx, bool read_only = false) {
    hipSetDevice(ctx->device);
    join_groups(ctx);
    join_premac(ctx, read_only);
}

#define HIP_TRY(ctx, call)                                                                          \
    do {                                                                                            \
        hipError_t e_ = (call);                                                                     \
        if (e_ != hipSuccess) return fail(ctx, GDG_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)

static double decibels_to_factor(int32_t decibels) {          /* effects/effects.go:389-394 */
    double e = 0.05 * (double)decibels;
    return pow(10.0, e);
}

static double lanczos_kernel(double x, double a) {             /* resample/resample.go:10-31 */
    if (x == 0) return 1.0;
    if ((-a < x) && (x < a)) {
        double pi_x = M_PI * x;
        double pi_xa = pi_x / a;
        double pi_x_squared = pi_x * pi_x;
        double prod = sin(pi_x) * sin(pi_xa);
        double arg = a * prod;
        return arg / pi_x_squared;
    }
    return 0.0;
}

extern "C" {

const char *gdg_version(void) { return "gdg 0.1 gfx950 hip"; }

int gdg_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

static void join_groups(gdg_ctx *ctx) {
    if (!ctx->groups_pending) return;
    for (size_t g = 0; g < ctx->gstreams.size() && g < ctx->gjoin.size(); g++) {
        hipEventRecord(ctx->gjoin[g], ctx->gstreams[g]);
        hipStreamWaitEvent(ctx->stream, ctx->gjoin[g], 0);
    }
    ctx->groups_pending = false;
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
static int numa_target(const gdg_ctx *ctx, const std::vector<int> **cpus) {
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
static void numa_bind_thread(const std::vector<int> &cpus) {
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
static hipError_t pinned_alloc(gdg_ctx *ctx, void **p, size_t bytes) {
    const std::vector<int> *unused;
    const int node = numa_target(ctx, &unused);
    const bool bind = node >= 0 && node < 1024;
    if (bind) {
        unsigned long mask[16] = { 0 };
        mask[node / (8 * sizeof(unsigned long))] |= 1ul << (node % (8 * sizeof(unsigned long)));
        const bool policy = syscall(SYS_set_mempolicy, 1 /* MPOL_PREFERRED */, mask, 1024ul + 1) == 0;
        hipError_t e = hipHostMalloc(p, bytes, policy ? hipHostMallocNumaUser : hipHostMallocDefault);
        if (policy) syscall(SYS_set_mempolicy, 0 /* MPOL_DEFAULT */, nullptr, 0ul);
        if (e == hipSuccess) return e;
        (void)hipGetLastError();
    }
    return hipHostMalloc(p, bytes, hipHostMallocDefault);
}
static int numa_rebind(gdg_ctx *ctx, int mode);

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
    { "fir_chain_adjacent_amps", "GDG_FIR_CHAIN", 0, 1, -1, nullptr, &gdg_ctx::fir_chain, true },
    { "fir_premac", "GDG_FIR_PREMAC", 0, 1, -1, &gdg_ctx::fir_premac, nullptr, true },
    { "fir_premac_min_partitions", "GDG_FIR_PREMAC_MIN", 1, 1 << 24, -1, &gdg_ctx::fir_premac_min, nullptr, true },
    { "share_ir_spectra", "GDG_SHARE_IR_SPECTRA", 0, 1, -1, nullptr, &gdg_ctx::share_spectra, false },
    { "fft_half_lds_mask", "GDG_FFT_HALF_LDS", 0, 63, GDG_KNOB_FFT_HALF_LDS, nullptr, nullptr, false },
    { "fir_forward_per_channel", "GDG_FWD_PER_CHANNEL", 0, 1, GDG_KNOB_FWD_PER_CHANNEL, nullptr, nullptr, false },
    { "fir_forward_wave_local", "GDG_WAVE_FFT", 0, 3, GDG_KNOB_WAVE_FFT, nullptr, nullptr, false },
    { "fir_mac_variant", "GDG_MAC_VARIANT", 0, 15, GDG_KNOB_MAC_VARIANT, nullptr, nullptr, false },
    /* the segments */
    { "seg_two_per_cu", "GDG_SEG_FAST", 0, 1, -1, nullptr, &gdg_ctx::seg_fast, true },
    { "seg_two_per_cu_min_channels", "GDG_SEG_FAST_MIN", 0, 1 << 20, -1, &gdg_ctx::seg_fast_min, nullptr, true },
    { "seg_wave_max_channels", "GDG_SEG_WAVE_MAX", 0, 1 << 20, -1, &gdg_ctx::seg_wave_max, nullptr, false },
    { "plan_patch", "GDG_PLAN_PATCH", 0, 1, -1, nullptr, &gdg_ctx::plan_patch, false },
    { "scan_tables_max", "GDG_SCAN_TABLES_MAX", 1, 1 << 20, -1, &gdg_ctx::scan_tables_max, nullptr, false },
    /* host paths, tuner, profiling */
    { "pcie_groups", "GDG_PCIE_GROUPS", 0, 16, -1, &gdg_ctx::pcie_groups_forced, nullptr, false },
    { "device_groups_default", "GDG_DEVICE_GROUPS", 0, 16, -1, &gdg_ctx::device_groups_env, nullptr, false },
    { "copy_threads", "GDG_COPY_THREADS", 1, 256, -1, &gdg_ctx::copy_threads, nullptr, false },
    { "numa", "GDG_NUMA", 0, 2, -1, &gdg_ctx::numa_mode, nullptr, false },
    { "tuner_parts", "GDG_TUNER_PARTS", 0, 24, GDG_KNOB_TUNER_PARTS, nullptr, nullptr, false },
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
    ok = ok && hipMalloc((void **)&ctx->d_error, sizeof(int)) == hipSuccess;
    ok = ok && hipMemset(ctx->d_error, 0, sizeof(int)) == hipSuccess;
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
    hipFree(ctx->d_w0); hipFree(ctx->d_w1); hipFree(ctx->d_scratch); hipFree(ctx->d_error); hipFree(ctx->d_wave);
    hipFree(ctx->d_stage_in); hipFree(ctx->d_stage_out); hipFree(ctx->d_blob); hipFree(ctx->d_os);
    hipFree(ctx->d_tuner_ring); hipFree(ctx->d_sp_hist);
    hipFree(ctx->d_note_freqs); hipFree(ctx->d_tuner_out); hipFree(ctx->d_tuner_work); hipFree(ctx->d_tuner_part); hipFree(ctx->d_tuner_twn); hipFree(ctx->d_tuner_twm);
    hipFree(ctx->d_sp_chan); hipFree(ctx->d_sp_out); hipFree(ctx->d_io[0]); hipFree(ctx->d_io[1]); hipFree(ctx->d_meter); hipFree(ctx->d_tick); hipFree(ctx->d_tock);
    for (auto st : ctx->gstreams) hipStreamDestroy(st);
    for (auto e : ctx->gjoin) hipEventDestroy(e);
    if (ctx->gfork) hipEventDestroy(ctx->gfork);
    for (int h = 0; h < 2; h++) {
        if (ctx->h_batch[h]) hipHostFree(ctx->h_batch[h]);
        if (ctx->batch_ready[h]) hipEventDestroy(ctx->batch_ready[h]);
        if (ctx->batch_moved[h]) hipEventDestroy(ctx->batch_moved[h]);
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
    if (strcmp(key, "copy_threads") == 0 && ctx->copy_pool && value != ctx->copy_threads) { destroy_copy_pool(ctx->copy_pool); ctx->copy_pool = nullptr; }
    if (strcmp(key, "numa") == 0 && value != ctx->numa_mode) { int rc = numa_rebind(ctx, (int)value); if (rc != GDG_OK) return rc; }
    option_store(ctx, *o, value);
    if (o->replans) ctx->dirty = true;
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

static Unit *get_unit(gdg_ctx *ctx, int handle) {
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
        if (ctx->dirty || !ctx->plan_patch) ctx->dirty = true;
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

/* ---- plan ----------------------------------------------------------------------------------------- */

static int ensure_hist(gdg_ctx *ctx, Unit &u, size_t len, long long key) {
    if (u.hist_key == key && u.hist_len == len) return GDG_OK;
    if (u.d_hist) { HIP_TRY(ctx, hipStreamSynchronize(ctx->stream)); ctx->arena.release(u.d_hist); u.d_hist = nullptr; }
    u.hist_len = len;
    u.hist_key = key;
    if (len > 0) {
        HIP_TRY(ctx, ctx->arena.alloc_zeroed((void **)&u.d_hist, len * sizeof(double), ctx->stream));
    }
    return GDG_OK;
}

static int zero_is(gdg_ctx *ctx, Unit &u, int first, int count) {
    HIP_TRY(ctx, hipMemsetAsync(u.d_is + first, 0, (size_t)count * sizeof(int), ctx->stream));
    return GDG_OK;
}

/* ---- scan tables of the constant-coefficient recurrences (seg.hip: lin_scan, lin2_scan) ----------------------------------------------
 * Powers of the 8-sample chunk map of a one-pole section / follower (scalar A = keep^8) or of a high-pass feeding a low-pass (2 x 2
 * lower-triangular P = M^8), by binary exponentiation in plain FP64 multiplications -- the arithmetic the kernel itself used while
 * it still built them on every call.  They depend on the coefficients only, i.e. on parameters and the sample rate: built when a plan
 * is built, one copy in HBM per distinct coefficient set (512 channels with the same tone-stack settings share one). */
static double pow_u(double A, int e) {
    double r = 1.0, b = A;
    for (int k = 0; k < 10; k++) { if (e & (1 << k)) r *= b; b *= b; }
    return r;
}
/* chk = samples per thread of the kernel that will read the table: 8 (seg_kernel) or 16 (the two-per-CU kernel, GDG_CHK_FAST) */
static void lin_tab_host(bool maxop, double a, double keep, double *tab, int chk) {
    const double k2 = keep * keep, k4 = k2 * k2, A8 = k4 * k4, A = chk == 16 ? A8 * A8 : A8;
    for (int lane = 0; lane < 64; lane++) {
        tab[LT_PC_(chk) + lane] = pow_u(A, lane);
        if (lane < 16) tab[LT_PA_(chk) + lane] = pow_u(A, lane + 1);
        if (lane >= 32) tab[LT_PB_(chk) + lane - 32] = pow_u(A, lane - 31);
        if (lane < 10) tab[LT_ST_(chk) + lane] = pow_u(A, 1 << lane);
        if (lane < chk) tab[LT_W_(chk) + lane] = (maxop ? 1.0 : a) * pow_u(keep, chk - 1 - lane);
    }
}
struct Tri { double a, b, c; };                                /* [[a, 0], [b, c]] */
static Tri tri_mul(const Tri &x, const Tri &y) { Tri r = { x.a * y.a, (x.b * y.a) + (x.c * y.b), x.c * y.c }; return r; }
static Tri tri_pow(Tri M, int e) {
    Tri r = { 1.0, 0.0, 1.0 };
    for (int k = 0; k < 10; k++) { if (e & (1 << k)) r = tri_mul(r, M); M = tri_mul(M, M); }
    return r;
}
static void tri_store(double *p, const Tri &t) { p[0] = t.a; p[1] = t.b; p[2] = t.c; }
/* per sample (h, l) <- M (h, l) + (aH, aL) x, M = [[1-aH, 0], [-aL, 1-aL]] */
static void lin2_tab_host(double aH, double aL, double *tab, int chk) {
    const Tri M = { 1.0 - aH, -aL, 1.0 - aL };
    const Tri P = tri_pow(M, chk);
    for (int lane = 0; lane < 64; lane++) {
        tri_store(tab + L2_PC_(chk) + 3 * lane, tri_pow(P, lane));
        if (lane < 16) tri_store(tab + L2_PA_(chk) + 3 * lane, tri_pow(P, lane + 1));
        if (lane >= 32) tri_store(tab + L2_PB_(chk) + 3 * (lane - 32), tri_pow(P, lane - 31));
        if (lane < 10) tri_store(tab + L2_ST_(chk) + 3 * lane, tri_pow(P, 1 << lane));
        if (lane < chk) {
            Tri G = tri_pow(M, chk - 1 - lane);
            tab[L2_W_(chk) + 2 * lane] = G.a * aH;
            tab[L2_W_(chk) + 2 * lane + 1] = (G.b * aH) + (G.c * aL);
        }
    }
}
/* the tables of `key` (a tag + the coefficients): the device copy, BUILT and uploaded on first use only -- 512 channels with the same
 * tone-stack setting asked for the same 12 KB table 512 times, and building it (4 bands x ~200 triangular matrix powers) before the
 * look-up cost 13 us each: 6.6 of the 6.2-6.8 ms a plan of 512 channels took to rebuild */
static int scan_tables(gdg_ctx *ctx, const std::vector<double> &key, size_t n_doubles, const std::function<void(double *)> &build, const double **out) {
    auto it = ctx->scan_tabs.find(key);
    if (it == ctx->scan_tabs.end()) {
        std::vector<double> tab(n_doubles, 0.0);
        build(tab.data());
        double *d = nullptr;
        HIP_TRY(ctx, ctx->arena.alloc((void **)&d, tab.size() * sizeof(double)));
        HIP_TRY(ctx, hipMemcpy(d, tab.data(), tab.size() * sizeof(double), hipMemcpyHostToDevice));
        it = ctx->scan_tabs.emplace(key, d).first;
    }
    *out = it->second;
    return GDG_OK;
}
/* [follower | coupling capacitor] of fuzz / octaver, or the follower alone (compressor): follow 0 = peak (max-affine), 1 = level */
static int follower_tables(gdg_ctx *ctx, int follow, double d_inv, double d, bool with_cap, const double **out, int chk) {
    std::vector<double> key = { 1.0 + 0.001 * chk, (double)follow, d_inv, d, with_cap ? 1.0 : 0.0 };
    return scan_tables(ctx, key, (size_t)(with_cap ? 2 : 1) * LT_SIZE_(chk), [&](double *tab) {
        if (follow == 0) lin_tab_host(true, 0.0, d_inv, tab, chk);
        else lin_tab_host(false, d, d_inv, tab, chk);
        if (with_cap) lin_tab_host(false, d, 1.0 - d, tab + LT_SIZE_(chk), chk);
    }, out);
}

/* May this unit run on the two-per-CU segment kernel (seg.hip compiled with SEG_FAST: 8192-sample frames, one LDS frame buffer, every
 * unit in place)?  By type (gdg_segf_supported), then by what the in-place variants assume: no oversampling (the oversampled shapers
 * stage a whole output frame in a second buffer); the reverb with every tap at least a frame back (rates from 42.7 kHz) and all-pass
 * rings of at most 3072 / 1024 values for the two short ones (rates up to 226 kHz). */
static bool segf_unit_ok(const Unit &u, int frames, uint32_t sample_rate) {
    if (frames != GDG_MAX_FRAMES || !gdg_segf_supported(u.type)) return false;
    const int32_t *p = u.params;
    const double sr = (double)sample_rate;
    switch (u.type) {
    case GDG_UNIT_OVERDRIVE: return p[5] == 0;
    case GDG_UNIT_DISTORTION: return p[3] == 0;
    case GDG_UNIT_EXCESS: return p[2] == 0;
    case GDG_UNIT_REVERB:
        return (uint32_t)round(0.19196 * sr) >= (uint32_t)frames && (int)round(0.01348 * sr) - 1 <= 3072 && (int)round(0.00452 * sr) - 1 <= 1024 &&
               (int)round(0.00452 * sr) - 1 >= 1;
    default: return true;
    }
}

/* Fill the device-side description of one non-FIR unit; (re)build its history for this rate / frame size. */
static int prepare_unit(gdg_ctx *ctx, Unit &u, int frames, uint32_t sample_rate, gdg_seg_unit &d, int chk = GDG_CHK) {
    memset(&d, 0, sizeof(d));
    d.type = u.type;
    for (int i = 0; i < GDG_MAX_PARAMS; i++) d.ip[i] = u.params[i];
    const int32_t *p = u.params;
    const double sr = (double)sample_rate;
    int rc = GDG_OK;
    switch (u.type) {
    case GDG_UNIT_COMPRESSOR: {
        d.dp[0] = decibels_to_factor(p[1]);
        d.dp[1] = decibels_to_factor(p[2]);
        d.dp[2] = exp(-20.0 / sr);
        d.dp[3] = 1.0 - d.dp[2];
        rc = follower_tables(ctx, p[0], d.dp[2], d.dp[3], false, &d.tab, chk);
        break;
    }
    case GDG_UNIT_OVERDRIVE:
    case GDG_UNIT_DISTORTION:
    case GDG_UNIT_EXCESS: {
        int os_idx;
        if (u.type == GDG_UNIT_OVERDRIVE) {
            d.dp[0] = decibels_to_factor(p[0] + p[1]);
            d.dp[1] = 0.01 * (double)p[2];
            d.dp[2] = 1.0 - d.dp[1];
            d.dp[3] = decibels_to_factor(p[3]);
            d.ip[4] = p[4];
            os_idx = p[5];
        } else if (u.type == GDG_UNIT_DISTORTION) {
            d.dp[0] = decibels_to_factor(p[0] + p[1]);
            d.dp[3] = decibels_to_factor(p[2]);
            os_idx = p[3];
        } else {
            d.dp[0] = decibels_to_factor(p[0]);
            d.dp[3] = decibels_to_factor(p[1]);
            os_idx = p[2];
        }
        int f = (os_idx == 1) ? 2 : (os_idx == 2) ? 4 : 1;
        d.jp[0] = f;
        if (f > 1) {
            /* one history block per oversampler object (oversamplerTwo / oversamplerFour keep separate state) */
            const size_t len2 = 8 + 76, len4 = 8 + 154;
            rc = ensure_hist(ctx, u, len2 + len4, 1);
            if (rc != GDG_OK) return rc;
            double *base = u.d_hist + (f == 2 ? 0 : len2);
            int which = (f == 2) ? 0 : 1;
            if (u.os_frames[which] != frames) {
                /* bufferPreUpsampling is re-made when the frame size changes (oversampling.go:86-89) */
                if (u.os_frames[which] >= 0) HIP_TRY(ctx, hipMemsetAsync(base, 0, 8 * sizeof(double), ctx->stream));
                u.os_frames[which] = frames;
            }
            d.hist = base;
        }
        break;
    }
    case GDG_UNIT_TONESTACK: {
        static const double freqs[5] = { 20.0, 300.0, 3000.0, 6000.0, 20000.0 };
        double m2pi_sr = -GO_MATH_TWO_PI / sr;
        for (int j = 0; j < 4; j++) {
            d.dp[j] = decibels_to_factor(p[j]);
            d.dp[4 + j] = 1.0 - exp(m2pi_sr * freqs[j]);
            d.dp[8 + j] = 1.0 - exp(m2pi_sr * freqs[j + 1]);
        }
        std::vector<double> key = { 2.0 + 0.001 * chk };
        for (int j = 0; j < 4; j++) { key.push_back(d.dp[4 + j]); key.push_back(d.dp[8 + j]); }
        rc = scan_tables(ctx, key, 4 * L2_SIZE_(chk), [&](double *tab) { for (int j = 0; j < 4; j++) lin2_tab_host(d.dp[4 + j], d.dp[8 + j], tab + j * L2_SIZE_(chk), chk); }, &d.tab);
        break;
    }
    case GDG_UNIT_CABINET: {
        static const double f[7] = { 300.0, 120.0, 80.0, 3000.0, 4000.0, 5000.0, 6000.0 };
        double m2pi_sr = -GO_MATH_TWO_PI / sr;
        for (int j = 0; j < 7; j++) d.dp[j] = 1.0 - exp(m2pi_sr * f[j]);
        std::vector<double> key = { 3.0 + 0.001 * chk };
        for (int j = 0; j < 7; j++) key.push_back(d.dp[j]);
        rc = scan_tables(ctx, key, 7 * LT_SIZE_(chk), [&](double *tab) { for (int j = 0; j < 7; j++) lin_tab_host(false, d.dp[j], 1.0 - d.dp[j], tab + j * LT_SIZE_(chk), chk); }, &d.tab);
        break;
    }
    case GDG_UNIT_CHORUS: {
        double depth = 0.1 * (double)p[0];
        if (depth < 0.0) depth = 0.0; else if (depth > 10.0) depth = 10.0;
        d.dp[0] = depth;
        d.dp[1] = GO_MATH_PI_THOUSANDTH * (double)p[1];
        d.dp[2] = sr;
        int C = (int)floor((0.05 * sr) + 0.5);
        d.jp[0] = C;
        /* the history ring holds the reference's C samples PLUS one frame (the frame is appended BEFORE the delays are read, so every
         * tap -- in the frame or before it -- is one ring access), rounded up to a power of two (index masks) + one guard cell that
         * mirrors cell 0 (a sample pair never wraps).  Re-made (zeroed) when the reference re-makes its buffer: when C changes. */
        size_t cp = 1;
        while (cp < (size_t)C + (size_t)ctx->max_frames) cp <<= 1;
        d.jp[1] = (int)(cp - 1);
        if (u.hist_key != C) rc = zero_is(ctx, u, 0, 1);
        if (rc == GDG_OK) rc = ensure_hist(ctx, u, cp + 1, C);
        d.hist = u.d_hist;
        break;
    }
    case GDG_UNIT_FLANGER:
    case GDG_UNIT_PHASER: {
        double depth = 0.01 * (double)p[0];
        if (depth < 0.0) depth = 0.0; else if (depth > 1.0) depth = 1.0;
        d.dp[0] = depth;
        d.dp[1] = GO_MATH_TWO_PI_HUNDREDTH * (double)p[1];
        d.dp[2] = sr;
        d.dp[3] = 1.0 / sr;
        d.dp[4] = 0.5;
        d.dp[5] = 0.5;
        if (u.type == GDG_UNIT_PHASER) {
            double radians = GO_MATH_DEGREE_TO_RADIANS * (double)p[2];
            d.dp[5] = 0.5 * sin(radians);
            d.dp[4] = 1.0 - fabs(d.dp[5]);
        }
        int C = (int)floor((0.002 * sr) + 0.5);
        d.jp[0] = C;
        if (u.hist_key != C) rc = zero_is(ctx, u, 0, 1);
        if (rc == GDG_OK) rc = ensure_hist(ctx, u, (size_t)C, C);
        d.hist = u.d_hist;
        break;
    }
    case GDG_UNIT_DELAY: {
        double seconds = 0.001 * (double)p[0];
        int D = (int)floor((seconds * sr) + 0.5);
        d.dp[0] = decibels_to_factor(p[1]);
        d.dp[1] = decibels_to_factor(p[2]);
        d.jp[0] = D;
        if (u.hist_key != D) rc = zero_is(ctx, u, 0, 1);
        if (rc == GDG_OK) rc = ensure_hist(ctx, u, (size_t)D, D);
        d.hist = u.d_hist;
        break;
    }
    case GDG_UNIT_RINGMODULATOR: {
        double angular = GO_MATH_TWO_PI * (double)p[0];
        d.dp[0] = angular / sr;
        break;
    }
    case GDG_UNIT_TREMOLO: {
        double frequency = 0.1 * (double)p[0];
        double period_f = sr / frequency;
        uint32_t period = (uint32_t)period_f;
        double phase = 0.01 * (double)p[1];
        uint32_t unatt = (uint32_t)(period_f * phase);
        uint32_t att = period - unatt;
        d.dp[0] = decibels_to_factor(p[2]);
        d.jp[0] = (int)unatt;
        d.jp[1] = (int)att;
        break;
    }
    case GDG_UNIT_SIGNALGENERATOR: {
        d.dp[0] = (0.01 * (double)p[0]) * decibels_to_factor(p[1]);
        double fac_signal_gain = decibels_to_factor(p[5]);
        d.dp[1] = (0.01 * (double)p[4]) * fac_signal_gain;
        d.dp[2] = GO_MATH_TWO_PI * ((double)p[3] / sr);
        break;
    }
    case GDG_UNIT_REVERB: {
        static const double ap_delays[3] = { 0.04204, 0.01348, 0.00452 };
        static const double tap_times[4] = { 0.19196, 0.19996, 0.21596, 0.23204 };
        double wet = 0.01 * (double)p[0];
        d.dp[0] = 1.0 - wet;
        d.dp[1] = 0.5 * wet;
        uint32_t max_index = 0;
        for (int i = 0; i < 4; i++) {
            uint32_t t = (uint32_t)round(tap_times[i] * sr);
            d.jp[i] = (int)t;
            if (t > max_index) max_index = t;
        }
        /* the delay line holds the longest tap PLUS one frame of the batch block size: the in-place reverb of the two-per-CU kernel appends
         * the frame before it reads the taps (seg.hip); the general kernel only sees a longer ring */
        d.jp[4] = (int)max_index + GDG_MAX_FRAMES;
        size_t len = (size_t)max_index + GDG_MAX_FRAMES;
        for (int i = 0; i < 3; i++) {
            int D = (int)round(ap_delays[i] * sr);
            d.jp[5 + i] = D;
            len += (size_t)(D > 1 ? D - 1 : 0);
        }
        /* the reference rebuilds every reverb buffer when the sample rate changes (reverb.go:207-271) */
        if (u.hist_key != (long long)sample_rate) rc = zero_is(ctx, u, 0, 4);
        if (rc == GDG_OK) rc = ensure_hist(ctx, u, len, (long long)sample_rate);
        d.hist = u.d_hist;
        break;
    }
    case GDG_UNIT_FUZZ: {
        int f = (p[6] == 1) ? 2 : (p[6] == 2) ? 4 : 1;
        d.jp[0] = f;
        d.dp[0] = 0.01 * (double)p[1];
        d.dp[1] = decibels_to_factor(p[2] + p[3]);
        d.dp[2] = 0.01 * (double)p[4];
        d.dp[3] = 1.0 - d.dp[2];
        d.dp[4] = decibels_to_factor(p[5]);
        /* the follower and the coupling capacitor run at the OVERSAMPLED rate (fuzz.go:42-45, :167-168) */
        double inner_rate = (double)((uint32_t)f * sample_rate);
        d.dp[5] = exp(-20.0 / inner_rate);
        d.dp[6] = 1.0 - d.dp[5];
        rc = follower_tables(ctx, p[0], d.dp[5], d.dp[6], true, &d.tab, GDG_CHK);
        if (rc != GDG_OK) return rc;
        if (f > 1) {
            const size_t len2 = 8 + 76, len4 = 8 + 154;
            rc = ensure_hist(ctx, u, len2 + len4, 1);
            if (rc != GDG_OK) return rc;
            double *base = u.d_hist + (f == 2 ? 0 : len2);
            int which = (f == 2) ? 0 : 1;
            if (u.os_frames[which] != frames) {
                if (u.os_frames[which] >= 0) HIP_TRY(ctx, hipMemsetAsync(base, 0, 8 * sizeof(double), ctx->stream));
                u.os_frames[which] = frames;
            }
            d.hist = base;
        }
        break;
    }
    case GDG_UNIT_AUTOYOY: {
        int32_t level_a = p[1], level_b = p[2];
        double depth_a = 0.0, depth_b = 0.01 * (double)p[3];
        if (level_a > level_b) { std::swap(level_a, level_b); std::swap(depth_a, depth_b); }
        double la = (double)level_a, lb = (double)level_b;
        double sr_inv = 1.0 / sr;
        d.dp[0] = la; d.dp[1] = lb; d.dp[2] = depth_a; d.dp[3] = depth_b;
        d.dp[4] = (depth_b - depth_a) / (lb - la);
        d.dp[5] = exp(-20.0 * sr_inv);
        d.dp[6] = 1.0 - d.dp[5];
        d.dp[7] = sr;
        int C = (int)floor((0.01 * sr) + 0.5);
        d.jp[0] = C;
        if (u.hist_key != C) rc = zero_is(ctx, u, 0, 1);
        if (rc == GDG_OK) rc = ensure_hist(ctx, u, (size_t)C, C);
        d.hist = u.d_hist;
        break;
    }
    case GDG_UNIT_AUTOWAH: {
        int32_t level_a = p[1], level_b = p[2], freq_a = p[3], freq_b = p[4];
        if (level_a > level_b) { std::swap(level_a, level_b); std::swap(freq_a, freq_b); }
        double la = (double)level_a, lb = (double)level_b, fa = (double)freq_a, fb = (double)freq_b;
        d.dp[0] = la; d.dp[1] = lb; d.dp[2] = fa; d.dp[3] = fb;
        d.dp[4] = (fb - fa) / (lb - la);
        d.dp[5] = exp(-20.0 / sr);
        d.dp[6] = 1.0 - d.dp[5];
        d.dp[7] = sr;
        break;
    }
    case GDG_UNIT_BANDPASS: {
        static const int orders[4] = { 2, 4, 6, 8 };
        int half = (p[0] >= 0 && p[0] < 4) ? orders[p[0]] >> 1 : 0;
        int32_t fa = p[1], fb = p[2];
        if (fa > fb) std::swap(fa, fb);
        double m2pi_sr = -GO_MATH_TWO_PI / sr;
        d.dp[0] = 1.0 - exp(m2pi_sr * (double)fa);
        d.dp[1] = 1.0 - exp(m2pi_sr * (double)fb);
        {
            std::vector<double> key = { 4.0, d.dp[0], d.dp[1] };
            rc = scan_tables(ctx, key, L2_SIZE, [&](double *tab) { lin2_tab_host(d.dp[0], d.dp[1], tab, GDG_CHK); }, &d.tab);
            if (rc != GDG_OK) return rc;
        }
        d.jp[0] = half;
        if (u.bp_half_order != half) {
            /* bandpass.go:40-49: both capacitor slices are re-made when the order changes */
            if (u.bp_half_order >= 0) HIP_TRY(ctx, hipMemsetAsync(u.d_ds, 0, 8 * sizeof(double), ctx->stream));
            u.bp_half_order = half;
        }
        break;
    }
    case GDG_UNIT_OCTAVER: {
        for (int i = 0; i < 6; i++) d.dp[i] = decibels_to_factor(p[1 + i]);
        d.dp[6] = exp(-20.0 / sr);
        d.dp[7] = 1.0 - d.dp[6];
        rc = follower_tables(ctx, p[0], d.dp[6], d.dp[7], true, &d.tab, GDG_CHK);
        break;
    }
    case GDG_UNIT_NOISEGATE: {
        d.dp[0] = decibels_to_factor(p[0]);
        d.dp[1] = decibels_to_factor(p[1]);
        double hold_seconds = 0.001 * (double)p[2];
        d.jp[0] = (int)(uint32_t)floor((hold_seconds * sr) + 0.5);
        d.jp[1] = (p[0] < p[1]) ? 1 : 0;
        break;
    }
    default:
        return fail(ctx, GDG_ERR_UNSUPPORTED, "unit type %d has no HIP implementation yet", u.type);
    }
    if (rc != GDG_OK) return rc;
    d.ds = u.d_ds;
    d.is = u.d_is;
    return GDG_OK;
}

static int fir_tables(gdg_ctx *ctx, int P, double2 **tw, double2 **tw2) {
    auto it = ctx->fir_tables.find(P);
    if (it == ctx->fir_tables.end()) {
        double2 *a = nullptr, *b = nullptr;
        HIP_TRY(ctx, gdg_fir_tables_create(P, &a, &b));
        it = ctx->fir_tables.emplace(P, std::make_pair(a, b)).first;
    }
    *tw = it->second.first;
    *tw2 = it->second.second;
    return GDG_OK;
}

/* transform half size for a frame of `frames` samples: the next power of two, at least GDG_MIN_FIR_FRAMES */
static int fir_transform_size(int frames) {
    int P = GDG_MIN_FIR_FRAMES;
    while (P < frames) P <<= 1;
    return P;
}

/* filter.Process walks the frame in blocks of nextpow2(L) samples but counts them on nextpow2(N): when N is not a power of two a
 * block can start beyond the frame and the reference panics on the slice bounds (filter/filter.go:370-382, :443-453).  Such a
 * (frame size, filter length) pair is rejected instead of replicated. */
static bool reference_panics(int frames, int taps) {
    if (taps <= 0 || frames <= 0) return false;
    uint64_t n_power = 1, block = 1;
    while (n_power < (uint64_t)frames) n_power <<= 1;
    while (block < (uint64_t)taps) block <<= 1;
    uint64_t blocks = n_power / block + ((n_power % block) ? 1 : 0);
    return blocks > 0 && (blocks - 1) * block > (uint64_t)frames;
}

/* IR spectra of `taps` for frames of `hop` samples: reuse a live copy of the same taps at the same partition size, else build one */
static int fir_spectra(gdg_ctx *ctx, Unit &u, int hop, int P, int K) {
    const int L = (int)u.taps.size();
    size_t spec = (size_t)K * (size_t)P * sizeof(double2);
    uint64_t key = 1469598103934665603ull;                              /* FNV-1a style over the taps' 64-bit patterns, P, hop and L (byte-wise
                                                                         * it cost 0.5 ms per 65536-tap filter: half a second for 1024 of them) */
    {
        for (size_t i = 0; i < u.taps.size(); i++) { uint64_t w; memcpy(&w, &u.taps[i], sizeof(w)); key ^= w; key *= 1099511628211ull; key ^= key >> 29; }
        key ^= (uint64_t)P; key *= 1099511628211ull;
        key ^= (uint64_t)hop; key *= 1099511628211ull;
        key ^= (uint64_t)L; key *= 1099511628211ull;
    }
    u.H.reset();
    if (ctx->share_spectra) {
        auto range = ctx->spectra.equal_range(key);
        for (auto it = range.first; it != range.second;) {
            std::shared_ptr<SharedSpectra> sp = it->second.lock();
            if (!sp) { it = ctx->spectra.erase(it); continue; }
            if (sp->P == P && sp->hop == hop && sp->taps == u.taps) { u.H = sp; break; }      /* compared in full: a hash match alone is not trusted */
            ++it;
        }
    }
    if (u.H) return GDG_OK;
    auto sp = std::make_shared<SharedSpectra>();
    sp->taps = u.taps;
    sp->P = P;
    sp->K = K;
    sp->hop = hop;
    sp->arena = &ctx->arena;
    HIP_TRY(ctx, ctx->arena.alloc((void **)&sp->d_H, spec));
    if (L > 0) ctx->pending_ir.push_back(sp);                           /* transformed with the plan's other new filters: flush_ir */
    else HIP_TRY(ctx, hipMemsetAsync(sp->d_H, 0, spec, ctx->stream));   /* filter.Empty: one all-zero partition */
    u.H = sp;
    if (ctx->share_spectra) ctx->spectra.emplace(key, sp);
    return GDG_OK;
}

/* The IR spectra of every filter the plan under construction brought in, in a few large launches: a 512-channel context with private IRs
 * has 1024 of them, and one upload + launch + synchronise + release each cost 1.1 ms a piece (1.2 s before the first frame).  Filters of
 * the same transform size go together, at most ~256 MiB of zero-padded taps per round: partition k = taps [k hop, (k + 1) hop) padded to P. */
static int flush_ir_body(gdg_ctx *ctx, const std::vector<std::shared_ptr<SharedSpectra>> &todo);
static int flush_ir(gdg_ctx *ctx) {
    if (ctx->pending_ir.empty()) return GDG_OK;
    int rc = flush_ir_body(ctx, ctx->pending_ir);
    if (rc == GDG_OK) ctx->pending_ir.clear();          /* on failure the list stays: the caller marks these filters for a fresh start */
    return rc;
}
static int flush_ir_body(gdg_ctx *ctx, const std::vector<std::shared_ptr<SharedSpectra>> &todo) {
    std::map<int, std::vector<SharedSpectra *>> by_P;
    for (auto &sp : todo) by_P[sp->P].push_back(sp.get());
    for (auto &kv : by_P) {
        const int P = kv.first;
        double2 *tw, *tw2;
        int rc = fir_tables(ctx, P, &tw, &tw2);
        if (rc != GDG_OK) return rc;
        const size_t budget = ((size_t)256 << 20) / ((size_t)P * sizeof(double));       /* partitions per round */
        size_t at = 0;
        while (at < kv.second.size()) {
            size_t end = at, parts = 0;
            while (end < kv.second.size() && (parts == 0 || parts + (size_t)kv.second[end]->K <= budget)) parts += (size_t)kv.second[end++]->K;
            std::vector<double> padded(parts * (size_t)P, 0.0);
            std::vector<gdg_fir_irjob> jobs(parts);
            struct Temps {
                DevArena &a; hipStream_t st; void *p = nullptr, *q = nullptr;
                ~Temps() { hipStreamSynchronize(st); a.release(p); a.release(q); }
            } tmp{ ctx->arena, ctx->stream };
            HIP_TRY(ctx, ctx->arena.alloc(&tmp.p, padded.size() * sizeof(double)));
            HIP_TRY(ctx, ctx->arena.alloc(&tmp.q, jobs.size() * sizeof(gdg_fir_irjob)));
            double *d_taps = static_cast<double *>(tmp.p);
            size_t j = 0;
            for (size_t i = at; i < end; i++) {
                const SharedSpectra &sp = *kv.second[i];
                const int L = (int)sp.taps.size();
                for (int k = 0; k < sp.K; k++, j++) {
                    const int n = std::min(sp.hop, L - k * sp.hop);
                    if (n > 0) memcpy(padded.data() + j * (size_t)P, sp.taps.data() + (size_t)k * sp.hop, (size_t)n * sizeof(double));
                    memset(&jobs[j], 0, sizeof(gdg_fir_irjob));
                    jobs[j].a = d_taps + j * (size_t)P;
                    jobs[j].out = sp.d_H + (size_t)k * P;
                }
            }
            HIP_TRY(ctx, hipMemcpyAsync(d_taps, padded.data(), padded.size() * sizeof(double), hipMemcpyHostToDevice, ctx->stream));
            HIP_TRY(ctx, hipMemcpyAsync(tmp.q, jobs.data(), jobs.size() * sizeof(gdg_fir_irjob), hipMemcpyHostToDevice, ctx->stream));
            /* 1/(2P): the inverse real transform's scale, folded into the IR spectra */
            HIP_TRY(ctx, gdg_launch_fir_ir(P, static_cast<const gdg_fir_irjob *>(tmp.q), (int)parts, 1.0 / (2.0 * (double)P), tw, tw2, ctx->stream));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));                     /* `padded` and `jobs` are locals */
            at = end;
        }
    }
    return GDG_OK;
}

/* (Re)build the partitioned spectra of one power amp for frames of `hop` samples: IR partitions of `hop` taps, transforms of 2 P
 * points with P = fir_transform_size(hop) (hop == P for the power-of-two frame sizes).
 *   - new filter / new sample rate / reset: the convolution state starts from zero (poweramp.go:131-203);
 *   - frame size changed while the filter is live: the reference's tail and transform sizes depend on L only, so any sequence of
 *     frame sizes is one continuous convolution (filter.go:370-428).  Here the partition size follows the frame, so the delay
 *     line is RE-PARTITIONED: its slots are transformed back to the last (K + 1) hop input samples (raw inverse, ~1e-16), which
 *     are re-cut into frames of the new size and transformed into the new delay line.  Samples older than that only ever meet
 *     zero-padded taps, so the continuation is exact.  Happens once per change, never in the steady state. */
static int prepare_fir(gdg_ctx *ctx, Unit &u, int hop, uint32_t sample_rate) {
    const int P = fir_transform_size(hop);
    if (u.fir_sr != sample_rate) {
        /* poweramp.go:191-203: a sample-rate change recompiles the filter, i.e. fresh state */
        u.fir_sr = sample_rate;
        u.fir_dirty = true;
        u.fir_live = false;
    }
    const int W = (hop == GDG_MAX_FRAMES) ? ctx->window : 1;      /* time blocking exists for the batch block size only */
    if (!u.fir_dirty && u.fir_hop == hop && u.fir_R == u.fir_K + W - 1) return GDG_OK;
    int L = (int)u.taps.size();
    if (reference_panics(hop, L))
        return fail(ctx, GDG_ERR_UNSUPPORTED, "frame size %d with a %d-tap filter: the reference panics on this pair (filter/filter.go:443-453: a block of "
                    "nextpow2(L) samples starts beyond the frame); rejected, not replicated", hop, L);
    int K = (L + hop - 1) / hop;
    if (K < 1) K = 1;                 /* filter.Empty: one all-zero partition => zeros out */
    const int R = K + W - 1;
    /* a live delay line moves into the new layout when the frame size OR the ring size (gdg_ctx_set_window) changes */
    const bool carry = !u.fir_dirty && u.fir_live && (u.fir_hop != hop || u.fir_R != R) && L > 0 && u.d_fdl && u.d_pos;
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    /* the old state, kept until the new delay line is built */
    double *o_prev = u.d_prev; double2 *o_fdl = u.d_fdl, *o_Y = u.d_Y; int *o_pos = u.d_pos;
    const int K1 = u.fir_K, P1 = u.fir_P, hop1 = u.fir_hop, R1 = u.fir_R;
    DevArena &arena = ctx->arena;
    auto free_old = [&]() { arena.release(o_prev); arena.release(o_fdl); arena.release(o_Y); arena.release(o_pos); };
    u.d_prev = nullptr; u.d_fdl = nullptr; u.d_Y = nullptr; u.d_pos = nullptr;
    double *d_old_hist = nullptr, *d_new_hist = nullptr;
    void *d_jobs = nullptr;
    size_t old_len = 0;
    int rc = GDG_OK;
    auto body = [&]() -> int {
        double2 *tw, *tw2;
        if (carry) {
            /* 1. the last (K1 + 1) hop1 input samples out of the old delay line, oldest first */
            int pos = 0;
            HIP_TRY(ctx, hipMemcpy(&pos, o_pos, sizeof(int), hipMemcpyDeviceToHost));
            old_len = (size_t)(K1 + 1) * (size_t)hop1;
            HIP_TRY(ctx, arena.alloc((void **)&d_old_hist, old_len * sizeof(double)));
            std::vector<gdg_fir_rawjob> jobs((size_t)K1);
            for (int j = 0; j < K1; j++) {
                int m = K1 - 1 - j;                                   /* frame t - m, t = the latest one, sits in slot (pos - 1 - m) mod R1 */
                int slot = (((pos - 1 - m) % R1) + R1) % R1;
                jobs[(size_t)j].Y = o_fdl + (size_t)slot * P1;
                jobs[(size_t)j].first = (j == 0) ? d_old_hist : nullptr;
                jobs[(size_t)j].second = d_old_hist + (size_t)(j + 1) * hop1;
                jobs[(size_t)j].hop = hop1;
            }
            HIP_TRY(ctx, arena.alloc(&d_jobs, jobs.size() * sizeof(gdg_fir_rawjob)));
            HIP_TRY(ctx, hipMemcpyAsync(d_jobs, jobs.data(), jobs.size() * sizeof(gdg_fir_rawjob), hipMemcpyHostToDevice, ctx->stream));
            int r = fir_tables(ctx, P1, &tw, &tw2);
            if (r != GDG_OK) return r;
            HIP_TRY(ctx, gdg_launch_fir_raw_inv(P1, (const gdg_fir_rawjob *)d_jobs, K1, 1.0 / (2.0 * (double)P1), tw, tw2, ctx->stream));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            arena.release(d_jobs); d_jobs = nullptr;
        }
        size_t spec = (size_t)R * (size_t)P * sizeof(double2);
        HIP_TRY(ctx, arena.alloc_zeroed((void **)&u.d_prev, 2 * (size_t)P * sizeof(double), ctx->stream));
        HIP_TRY(ctx, arena.alloc_zeroed((void **)&u.d_fdl, spec, ctx->stream));
        HIP_TRY(ctx, arena.alloc((void **)&u.d_Y, (size_t)W * (size_t)P * sizeof(double2)));
        HIP_TRY(ctx, arena.alloc_zeroed((void **)&u.d_pos, sizeof(int), ctx->stream));
        int r = fir_spectra(ctx, u, hop, P, K);
        if (r != GDG_OK) return r;
        if (carry) {
            /* 2. the newest K hop samples, re-cut into K frames of the new size (zeros where the old line does not reach) */
            size_t new_len = (size_t)K * (size_t)hop;
            HIP_TRY(ctx, arena.alloc((void **)&d_new_hist, new_len * sizeof(double)));
            HIP_TRY(ctx, hipMemsetAsync(d_new_hist, 0, new_len * sizeof(double), ctx->stream));
            size_t n = std::min(old_len, new_len);
            HIP_TRY(ctx, hipMemcpyAsync(d_new_hist + (new_len - n), d_old_hist + (old_len - n), n * sizeof(double), hipMemcpyDeviceToDevice, ctx->stream));
            /* 3. slot f = spectrum of [frame f - 1 | frame f], f = 1 .. K - 1; the next frame goes to slot K mod R */
            if (K > 1) {
                std::vector<gdg_fir_irjob> jobs((size_t)(K - 1));
                for (int f = 1; f < K; f++) {
                    jobs[(size_t)(f - 1)].a = d_new_hist + (size_t)(f - 1) * hop;
                    jobs[(size_t)(f - 1)].b = d_new_hist + (size_t)f * hop;
                    jobs[(size_t)(f - 1)].hop = hop;
                    jobs[(size_t)(f - 1)].out = u.d_fdl + (size_t)f * P;
                }
                HIP_TRY(ctx, arena.alloc(&d_jobs, jobs.size() * sizeof(gdg_fir_irjob)));
                HIP_TRY(ctx, hipMemcpyAsync(d_jobs, jobs.data(), jobs.size() * sizeof(gdg_fir_irjob), hipMemcpyHostToDevice, ctx->stream));
                r = fir_tables(ctx, P, &tw, &tw2);
                if (r != GDG_OK) return r;
                HIP_TRY(ctx, gdg_launch_fir_ir(P, (const gdg_fir_irjob *)d_jobs, K - 1, 1.0, tw, tw2, ctx->stream));
            }
            /* 4. the overlap-save history = the newest frame, where the forward transform of frame counter K looks for it */
            HIP_TRY(ctx, hipMemcpyAsync(u.d_prev + (size_t)((K + 1) & 1) * P, d_new_hist + (size_t)(K - 1) * hop, (size_t)hop * sizeof(double),
                                        hipMemcpyDeviceToDevice, ctx->stream));
            HIP_TRY(ctx, hipMemcpyAsync(u.d_pos, &K, sizeof(int), hipMemcpyHostToDevice, ctx->stream));
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        }
        return GDG_OK;
    };
    rc = body();
    if (rc != GDG_OK || carry) hipStreamSynchronize(ctx->stream);          /* nothing in flight reads what is released below */
    arena.release(d_old_hist); arena.release(d_new_hist); arena.release(d_jobs);
    free_old();
    if (rc != GDG_OK) { u.fir_dirty = true; u.fir_live = false; return rc; }
    u.fir_P = P;
    u.fir_K = K;
    u.fir_R = R;
    u.fir_hop = hop;
    u.fir_dirty = false;
    u.fir_live = carry;
    return GDG_OK;
}

struct Op { bool is_fir; std::vector<int> handles; };

/* `active`: the channels taking part in this call; row i of d_in / d_out belongs to channel active[i] */
static int build_plan(gdg_ctx *ctx, const std::vector<int> &active, const double *d_in, double *d_out, int frames, uint32_t sample_rate,
                      int stride, int stride_out, bool rows_by_channel, int G, const std::vector<size_t> &bounds) {
    const int nch = ctx->nch;
    int ptrace = 0;
    { const char *e = getenv("GDG_PLAN_TRACE"); ptrace = e ? atoi(e) : 0; }        /* read per plan: a test switches it on */
    auto pnow = []() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); };
    double t_fir = 0.0, t_unit = 0.0;
    const double t_plan0 = pnow();
    join_groups(ctx);                 /* a new plan replaces descriptors (and possibly unit state) the group streams may still be reading */
    join_premac(ctx, false);          /* ... and the sums made ahead belong to the old plan's next frame */
    /* Scan tables live as long as some plan's descriptors point at them -- there is one plan, this one.  A caller that sweeps a parameter
     * through thousands of values would let the cache grow without bound (12 KB per tone-stack setting): past the limit everything is
     * dropped once the work in flight has drained, and this plan re-makes the few tables it needs. */
    {
        long limit = ctx->scan_tables_max;                         /* gdg_ctx_set_option "scan_tables_max": a test lowers it */
        if (limit < 1) limit = 1;
        if ((long)ctx->scan_tabs.size() > limit) {
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            for (auto &kv : ctx->scan_tabs) ctx->arena.release(kv.second);
            ctx->scan_tabs.clear();
        }
    }
    /* channel groups: contiguous runs of `active`, group g = [bounds[g], bounds[g + 1]) (equal shares unless the caller weights them) */
    std::vector<int> group_of((size_t)nch, 0);
    for (int g = 0; g < G; g++)
        for (size_t i = bounds[(size_t)g]; i < bounds[(size_t)g + 1]; i++) group_of[(size_t)active[i]] = g;
    ctx->plan_groups = G;
    /* per channel: ops placed on a common grid of slots: 2k = segment k, 2k+1 = FIR k */
    std::map<int, std::vector<std::pair<int, Op>>> by_slot;           /* slot -> (channel, op) */
    std::vector<int> n_ops((size_t)nch, 0);
    std::vector<int> row_of((size_t)nch, -1);
    for (size_t i = 0; i < active.size(); i++) row_of[(size_t)active[i]] = rows_by_channel ? active[i] : (int)i;
    bool any_fir = false;
    for (int c : active) {
        std::vector<int> seg;
        int k = 0, count = 0;
        for (auto &s : ctx->chains[(size_t)c]) {
            if (s.bypass) continue;                                   /* signal.go:390-401 */
            Unit &u = ctx->units[(size_t)s.handle];
            if (u.type == GDG_UNIT_POWERAMP) {
                if (!seg.empty()) { by_slot[2 * k].push_back({ c, Op{ false, seg } }); seg.clear(); count++; }
                by_slot[2 * k + 1].push_back({ c, Op{ true, { s.handle } } });
                count++;
                k++;
                any_fir = true;
            } else {
                if (!gdg_seg_supported(u.type)) return fail(ctx, GDG_ERR_UNSUPPORTED, "unit type %d has no HIP implementation yet", u.type);
                seg.push_back(s.handle);
            }
        }
        if (!seg.empty()) { by_slot[2 * k].push_back({ c, Op{ false, seg } }); count++; }
        if (count == 0) { by_slot[0].push_back({ c, Op{ false, {} } }); count = 1; }     /* empty chain: copy */
        n_ops[(size_t)c] = count;
    }
    (void)any_fir;
    /* counters of the WAVE launches: tickets per (segment step, channel group), then one frame counter per unit that sits in a segment */
    {
        const size_t need = (size_t)GDG_WAVE_STEPS * GDG_WAVE_GROUPS + 2 * ctx->units.size() + (size_t)nch + 64;
        if (need > ctx->d_wave_cap) {
            HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
            hipFree(ctx->d_wave);
            ctx->d_wave = nullptr;
            ctx->d_wave_cap = need * 2;
            HIP_TRY(ctx, hipMalloc((void **)&ctx->d_wave, ctx->d_wave_cap * sizeof(int)));
            HIP_TRY(ctx, hipMemsetAsync(ctx->d_wave, 0, ctx->d_wave_cap * sizeof(int), ctx->stream));
        }
    }
    size_t wave_next = (size_t)GDG_WAVE_STEPS * GDG_WAVE_GROUPS;
    int seg_steps = 0;
    /* blob layout: [step 0 descs][step 1 descs]...[seg units] */
    std::vector<gdg_seg_unit> seg_units;
    std::vector<std::vector<gdg_seg_chan>> seg_descs;
    std::vector<std::vector<gdg_fir_chan>> fir_descs;
    std::vector<int> done((size_t)nch, 0);
    std::vector<const double *> cur((size_t)nch);
    for (int c : active) cur[(size_t)c] = d_in + (size_t)row_of[(size_t)c] * stride;
    ctx->steps.clear();
    ctx->plan_unit_slot.assign(ctx->units.size(), -1);
    ctx->patch_units.clear();                  /* this plan reads every unit's current parameters */
    ctx->plan_unit_fast.assign(ctx->units.size(), 0);
    ctx->plan_unit_fast_ok.assign(ctx->units.size(), 0);
    for (auto &kv : by_slot) {
        bool is_fir = (kv.first & 1) != 0;
        std::vector<gdg_seg_chan> sd;
        std::vector<gdg_fir_chan> fd;
        /* a segment step goes to the two-per-CU kernel when EVERY unit of EVERY channel in it can (one launch per step) */
        bool step_fast = !is_fir && ctx->seg_fast && frames == GDG_MAX_FRAMES && (int)active.size() >= ctx->seg_fast_min;
        if (step_fast)
            for (auto &entry : kv.second)
                for (int h : entry.second.handles) if (!segf_unit_ok(ctx->units[(size_t)h], frames, sample_rate)) { step_fast = false; break; }
        for (auto &entry : kv.second) {
            int c = entry.first;
            Op &op = entry.second;
            bool last = (done[(size_t)c] + 1 == n_ops[(size_t)c]);
            double *dst;
            if (last) dst = d_out + (size_t)row_of[(size_t)c] * stride_out;
            else dst = ((done[(size_t)c] & 1) ? ctx->d_w1 : ctx->d_w0) + (size_t)c * ctx->w_stride;
            if (is_fir) {
                Unit &u = ctx->units[(size_t)op.handles[0]];
                const double tq = pnow();
                int rc = prepare_fir(ctx, u, frames, sample_rate);
                t_fir += pnow() - tq;
                if (rc != GDG_OK) return rc;
                gdg_fir_chan f;
                memset(&f, 0, sizeof(f));
                f.src = cur[(size_t)c]; f.dst = dst; f.prev = u.d_prev; f.fdl = u.d_fdl; f.H = u.H->d_H; f.Y = u.d_Y;
                f.pos = u.d_pos; f.K = u.fir_K; f.R = u.fir_R; f.hop = frames;
                f.flags = (done[(size_t)c] == 0 ? GDG_SRC_IS_INPUT : 0) | (last ? GDG_DST_IS_OUTPUT : 0);
                fd.push_back(f);
                u.fir_live = true;
            } else {
                gdg_seg_chan s;
                memset(&s, 0, sizeof(s));
                s.src = cur[(size_t)c]; s.dst = dst;
                s.flags = (done[(size_t)c] == 0 ? GDG_SRC_IS_INPUT : 0) | (last ? GDG_DST_IS_OUTPUT : 0);
                s.scratch = ctx->d_scratch + (size_t)c * ctx->max_frames;
                s.unit_begin = (int)seg_units.size();
                s.unit_count = (int)op.handles.size();
                s.wave = ctx->d_wave + wave_next;
                wave_next += 2 * op.handles.size();                     /* two counters per unit: the reverb meets its predecessor frame twice */
                {   /* which units meet their predecessor frame in a WAVE launch, and whether their stores are write-through there (seg.hip, wt) */
                    unsigned mask = 0;
                    for (size_t ui = 0; ui < op.handles.size(); ui++) {
                        const Unit &wu = ctx->units[(size_t)op.handles[ui]];
                        const bool shaper = wu.type == GDG_UNIT_OVERDRIVE || wu.type == GDG_UNIT_DISTORTION || wu.type == GDG_UNIT_EXCESS;
                        const int os_param = wu.type == GDG_UNIT_OVERDRIVE ? 5 : (wu.type == GDG_UNIT_DISTORTION ? 3 : 2);
                        if (shaper && wu.params[os_param] == 0) continue;                       /* memoryless: no state, no meeting */
                        if (ui < 31) mask |= 1u << ui;
                        const bool write_through = wu.type == GDG_UNIT_COMPRESSOR || wu.type == GDG_UNIT_TONESTACK || wu.type == GDG_UNIT_CABINET ||
                                                   wu.type == GDG_UNIT_CHORUS || (wu.type == GDG_UNIT_REVERB && !step_fast) || (shaper && !step_fast);
                        if (!write_through) mask |= 1u << 31;
                    }
                    s.wave_mask = (int)mask;
                }
                for (int h : op.handles) {
                    gdg_seg_unit du;
                    const double tq = pnow();
                    int rc = prepare_unit(ctx, ctx->units[(size_t)h], frames, sample_rate, du, step_fast ? GDG_CHK_FAST : GDG_CHK);
                    t_unit += pnow() - tq;
                    if (rc != GDG_OK) return rc;
                    /* both reverbs that append the frame BEFORE they tap (two-per-CU kernel; general kernel in a WAVE launch) rely on a delay
                     * line exactly one batch frame longer than the longest tap (seg.hip): a change of one side without the other stops here */
                    if (du.type == GDG_UNIT_REVERB && du.jp[4] != std::max(std::max(du.jp[0], du.jp[1]), std::max(du.jp[2], du.jp[3])) + GDG_MAX_FRAMES)
                        return fail(ctx, GDG_ERR_INVALID, "reverb delay line of %d cells, expected the longest tap + %d", du.jp[4], GDG_MAX_FRAMES);
                    ctx->plan_unit_slot[(size_t)h] = (int)seg_units.size();
                    ctx->plan_unit_fast[(size_t)h] = step_fast ? 1 : 0;
                    ctx->plan_unit_fast_ok[(size_t)h] = segf_unit_ok(ctx->units[(size_t)h], frames, sample_rate) ? 1 : 0;
                    seg_units.push_back(du);
                }
                sd.push_back(s);
            }
            cur[(size_t)c] = dst;
            done[(size_t)c]++;
        }
        StepDesc st;
        st.is_fir = is_fir;
        st.fast = step_fast;
        st.n = is_fir ? (int)fd.size() : (int)sd.size();
        st.offset = 0;
        if (!is_fir && seg_steps < GDG_WAVE_STEPS && G <= GDG_WAVE_GROUPS) st.wave_tickets = GDG_WAVE_GROUPS * seg_steps++;
        /* descriptors are in `active` order, so every channel group owns one contiguous run of them */
        st.group_range.assign((size_t)G, std::make_pair(0, 0));
        {
            int pos = 0;
            for (auto &entry : kv.second) {
                auto &r = st.group_range[(size_t)group_of[(size_t)entry.first]];
                if (r.second == 0) r.first = pos;
                r.second++;
                pos++;
            }
        }
        if (is_fir) {
            std::vector<const void *> hp;
            for (auto &f : fd) hp.push_back(f.H);
            std::sort(hp.begin(), hp.end());
            st.shared_spectra = std::adjacent_find(hp.begin(), hp.end()) != hp.end();
            /* the terms k >= 1 ahead of the frame (premac): the split launch shape of few channels, one group, batch frames, every channel K >= 2 */
            const bool split = ctx->fir_fused < 0 ? (st.n <= ctx->fir_split_max) : (ctx->fir_fused == 0);
            long partitions = 0;
            for (auto &f : fd) partitions += f.K;
            st.premac_ok = ctx->fir_premac != 0 && split && G == 1 && frames == GDG_MAX_FRAMES && partitions >= ctx->fir_premac_min;
            for (auto &f : fd) if (f.K < 2 || f.hop != frames) st.premac_ok = false;
        }
        ctx->steps.push_back(st);
        seg_descs.push_back(sd);
        fir_descs.push_back(fd);
    }
    {   /* the new filters' spectra, all together */
        const double tq = pnow();
        int rc = flush_ir(ctx);
        if (ptrace) fprintf(stderr, "[plan] prepare_fir %.1f ms, prepare_unit %.1f ms, flush_ir %.1f ms, so far %.1f ms\n", t_fir, t_unit, pnow() - tq, pnow() - t_plan0);
        if (rc != GDG_OK) return rc;
    }
    /* adjacent power amps (the benchmark chain: cabinet IR, then reverb IR): when EVERY channel of a FIR step hands its frame to the
     * next FIR step, that step's forward transform is produced by this step's inverse kernel -- no launch, no round trip of the frame */
    if (ctx->fir_chain && frames == GDG_MAX_FRAMES) {
        for (size_t i = 0; i + 1 < ctx->steps.size(); i++) {
            if (!ctx->steps[i].is_fir || !ctx->steps[i + 1].is_fir) continue;
            auto &a = fir_descs[i], &b = fir_descs[i + 1];
            bool ok = !a.empty() && a.size() == b.size() && ctx->steps[i].group_range == ctx->steps[i + 1].group_range;
            for (size_t k = 0; ok && k < a.size(); k++) ok = a[k].dst == b[k].src && !(a[k].flags & GDG_DST_IS_OUTPUT) && a[k].hop == frames && b[k].hop == frames;
            if (!ok) continue;
            ctx->steps[i].chain_next = true;
            for (auto &f : a) f.flags |= GDG_DST_UNUSED;      /* only the chained transform reads the frame (window-mode kernels ignore the flag) */
        }
    }
    /* serialise */
    ctx->blob.clear();
    auto append = [&](const void *p, size_t bytes) {
        size_t off = (ctx->blob.size() + 255) & ~(size_t)255;
        ctx->blob.resize(off + bytes);
        if (bytes) memcpy(ctx->blob.data() + off, p, bytes);
        return off;
    };
    for (size_t i = 0; i < ctx->steps.size(); i++) {
        if (ctx->steps[i].is_fir) ctx->steps[i].offset = append(fir_descs[i].data(), fir_descs[i].size() * sizeof(gdg_fir_chan));
        else ctx->steps[i].offset = append(seg_descs[i].data(), seg_descs[i].size() * sizeof(gdg_seg_chan));
    }
    ctx->units_offset = append(seg_units.data(), seg_units.size() * sizeof(gdg_seg_unit));
    if (ctx->blob.size() > ctx->d_blob_cap) {
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
        hipFree(ctx->d_blob);
        ctx->d_blob = nullptr;
        ctx->d_blob_cap = ctx->blob.size() * 2 + 4096;
        HIP_TRY(ctx, hipMalloc((void **)&ctx->d_blob, ctx->d_blob_cap));
    } else {
        /* the previous plan may still be in use by launches in flight */
        HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    }
    ctx->arena.trim();                         /* the stream is drained: the one place where giving spare chunks back stalls nobody */
    if (!ctx->blob.empty())
        HIP_TRY(ctx, hipMemcpyAsync(ctx->d_blob, ctx->blob.data(), ctx->blob.size(), hipMemcpyHostToDevice, ctx->stream));
    ctx->plan_frames = frames;
    ctx->plan_sr = sample_rate;
    ctx->plan_in = d_in;
    ctx->plan_out = d_out;
    ctx->dirty = false;
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

static hipEvent_t take_event(gdg_ctx *ctx) {
    if (!ctx->event_pool.empty()) { hipEvent_t e = ctx->event_pool.back(); ctx->event_pool.pop_back(); return e; }
    hipEvent_t e = nullptr;
    hipEventCreate(&e);
    return e;
}

/* attached = true: the launch inside the scope takes the two events itself (hipExtLaunchKernelGGL: the kernel's own begin / end timestamps);
 * otherwise the events are recorded on the stream before and after whatever the scope launches */
struct ProfScope {
    gdg_ctx *ctx; int kind; hipStream_t st; hipEvent_t a = nullptr, b = nullptr; bool on = false, attached = false;
    ProfScope(gdg_ctx *c, int k, hipStream_t s = nullptr, bool attach = false) : ctx(c), kind(k), st(s ? s : c->stream), attached(attach) {
        on = ctx->prof_now && ((ctx->profiling & 1u) || (ctx->profiling & (1u << (k + 1))));
        if (on) { a = take_event(ctx); b = take_event(ctx); if (!attached) hipEventRecord(a, st); }
    }
    ~ProfScope() {
        if (on) { if (!attached) hipEventRecord(b, st); ctx->prof.push_back(ProfEvent{ kind, a, b }); }
    }
};

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

static int ensure_staging(gdg_ctx *ctx);
static int check_device_error(gdg_ctx *ctx);

/* what a host-buffer entry point does around group g's kernels, on group g's stream (upload before, download after) */
typedef std::function<hipError_t(int g, hipStream_t s)> GroupHook;

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

static int process_rows(gdg_ctx *ctx, const std::vector<int> &active, const double *d_in, double *d_out, int frames, uint32_t sample_rate,
                        int stride = 0, bool rows_by_channel = false, int groups = 1, const GroupHook *before = nullptr, const GroupHook *after = nullptr,
                        int window = 1, int stride_out = 0, const std::vector<size_t> *group_bounds_in = nullptr) {
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
                const bool fused = ctx->fir_fused < 0 ? (n > ctx->fir_split_max) : (ctx->fir_fused != 0);
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
                } else {
                    { ProfScope ps(ctx, GDG_K_FIR_MAC, s); HIP_TRY(ctx, gdg_launch_fir_mac(P2, d, n, st.shared_spectra ? 1 : 0, s)); }
                    { ProfScope ps(ctx, GDG_K_FIR_INV, s); HIP_TRY(ctx, gdg_launch_fir_inv(P2, d, n, tw, tw2, 0, shift, s, d_next)); }
                }
                /* behind the call's LAST power amp the next frame's sums can start (premac, below): mark the place in the stream */
                if (premac_here && si == premac_after) {
                    if (!ctx->premac_stream) {
                        HIP_TRY(ctx, hipStreamCreateWithFlags(&ctx->premac_stream, hipStreamNonBlocking));
                        HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_fir_done, hipEventDisableTiming));
                        HIP_TRY(ctx, hipEventCreateWithFlags(&ctx->ev_premac, hipEventDisableTiming));
                    }
                    HIP_TRY(ctx, hipEventRecord(ctx->ev_fir_done, s));
                }
            } else {
                const gdg_seg_chan *d = reinterpret_cast<const gdg_seg_chan *>(ctx->d_blob + st.offset) + first;
                ProfScope ps(ctx, GDG_K_SEGMENT, s);
                /* one launch per window: a channel's workgroup walks its frames in order, the units' state runs through them */
                /* ... unless the channels are few: then a workgroup per frame, the frames of a channel meeting unit by unit (seg.hip, WAVE) */
                int *tickets = (window > 1 && n <= ctx->seg_wave_max && st.wave_tickets >= 0) ? ctx->d_wave + st.wave_tickets + g : nullptr;
                if (st.fast) HIP_TRY(ctx, gdg_launch_segf(d, n, d_units, frames, window, shift, ctx->os, ctx->d_error, s, tickets));
                else HIP_TRY(ctx, gdg_launch_seg(d, n, d_units, frames, window, shift, ctx->os, ctx->d_error, s, tickets));
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
                HIP_TRY(ctx, gdg_launch_fir_mac(P2, dx, sx.n, sx.shared_spectra ? 1 : 0, ctx->premac_stream, 1));
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

static int check_device_error(gdg_ctx *ctx) {
    int e = 0;
    HIP_TRY(ctx, hipMemcpyAsync(&e, ctx->d_error, sizeof(int), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
    if (e != 0) {
        hipMemsetAsync(ctx->d_error, 0, sizeof(int), ctx->stream);
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

static int ensure_staging(gdg_ctx *ctx) {
    if (ctx->d_stage_in) return GDG_OK;
    size_t bytes = (size_t)std::max(ctx->nch, 2) * (size_t)ctx->max_frames * sizeof(double);
    HIP_TRY(ctx, hipMalloc((void **)&ctx->d_stage_in, bytes));
    HIP_TRY(ctx, hipMalloc((void **)&ctx->d_stage_out, bytes));
    HIP_TRY(ctx, pinned_alloc(ctx, (void **)&ctx->h_stage_in, bytes));
    HIP_TRY(ctx, pinned_alloc(ctx, (void **)&ctx->h_stage_out, bytes));
    return GDG_OK;
}

/* Host copy workers.  One core moves pageable memory at ~10 GB/s, which made the 2 x 32 MiB of a 512-channel block cost 3 ms -- more
 * than the whole chain -- so staging copies are spread over a few threads (env GDG_COPY_THREADS, default 8).  The workers are
 * PERSISTENT: created on a context's first host-buffer call and parked on a condition variable between jobs (spawning and joining
 * std::threads on every call cost 60-100 us per call, twice per block).  One pool PER CONTEXT since round 4: with one process-wide
 * pool only one of G shards copying at the same time got the workers and the others copied on their caller's thread alone
 * (the reference's deployment is G shards in one process, controller.go:3262-3269).  Joined and freed with the context.
 * fork(): a child inherits the pool object but none of its threads; it finds another pid in the pool and copies inline. */
class CopyPool {
public:
    /* cpus: the workers' CPUs (the device's NUMA node), empty = wherever the scheduler puts them */
    explicit CopyPool(int workers, std::vector<int> cpus = {}) : cpus_(std::move(cpus)), pid_(getpid()) {
        for (int i = 0; i < workers; i++) threads_.emplace_back([this, i]() { numa_bind_thread(cpus_); loop((size_t)i + 1); });
    }
    ~CopyPool() {                               /* only in the process that made the pool (destroy_copy_pool) */
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true; gen_++;
        }
        cv_work_.notify_all();
        for (auto &t : threads_) t.join();
    }
    size_t slots() const { return threads_.size() + 1; }
    bool usable() const { return pid_ == getpid(); }
    /* slice t of T runs fn(t); the caller takes slice 0 and returns when all slices are done */
    void run(size_t T, const std::function<void(size_t)> &fn) {
        if (T <= 1) { fn(0); return; }
        std::unique_lock<std::mutex> job(job_mu_, std::try_to_lock);
        if (!job.owns_lock()) { for (size_t t = 0; t < T; t++) fn(t); return; }
        {
            std::lock_guard<std::mutex> lk(mu_);
            fn_ = &fn; T_ = T; pending_ = T - 1; gen_++;
        }
        cv_work_.notify_all();
        fn(0);
        std::unique_lock<std::mutex> lk(mu_);
        cv_done_.wait(lk, [this]() { return pending_ == 0; });
        fn_ = nullptr;
    }
private:
    void loop(size_t slot) {
        uint64_t seen = 0;
        for (;;) {
            const std::function<void(size_t)> *fn = nullptr;
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_work_.wait(lk, [&]() { return gen_ != seen; });
                if (stop_) return;
                seen = gen_;
                if (slot < T_) fn = fn_;
            }
            if (!fn) continue;
            (*fn)(slot);
            std::lock_guard<std::mutex> lk(mu_);
            if (--pending_ == 0) cv_done_.notify_one();
        }
    }
    std::vector<std::thread> threads_;
    std::mutex job_mu_, mu_;
    std::condition_variable cv_work_, cv_done_;
    const std::function<void(size_t)> *fn_ = nullptr;
    size_t T_ = 0, pending_ = 0;
    uint64_t gen_ = 0;
    bool stop_ = false;
    std::vector<int> cpus_;
    pid_t pid_;
};

/* A forked child inherits the pool object but none of its threads: joining or detaching std::thread handles of threads that do not exist in
 * this process is undefined -- the child leaves the object alone (a few hundred bytes, once). */
static void destroy_copy_pool(CopyPool *p) { if (p && p->usable()) delete p; }

static CopyPool &copy_pool(gdg_ctx *ctx) {
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

/* option "numa": the workers are re-made (bound or not) at the next host-buffer call; pinned slabs that exist stay where they are (the
 * staging slabs' addresses are in the caller's hands), those made afterwards follow the new mode -- set it before the first call */
static int numa_rebind(gdg_ctx *ctx, int mode) {
    (void)mode;
    if (ctx->copy_pool) { destroy_copy_pool(ctx->copy_pool); ctx->copy_pool = nullptr; }
    return GDG_OK;
}

/* rows [a, b) of a host-side staging copy, spread over the copy workers (at least ~1 MiB per thread) */
static void copy_rows_parallel(gdg_ctx *ctx, size_t a, size_t b, const std::function<void(size_t)> &copy_row, size_t row_bytes) {
    size_t n = b > a ? b - a : 0;
    if (n == 0) return;
    CopyPool &pool = copy_pool(ctx);
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

/* ---- tuner: tuner.Process / tuner.Analyze for every channel of the shard ------------------------------------------ */

static int ensure_tuner(gdg_ctx *ctx) {
    if (ctx->d_tuner_ring) return GDG_OK;
    size_t ring_bytes = (size_t)ctx->nch * GDG_TUNER_RING * sizeof(double);
    HIP_TRY(ctx, hipMalloc((void **)&ctx->d_tuner_ring, ring_bytes));
    HIP_TRY(ctx, hipMemsetAsync(ctx->d_tuner_ring, 0, ring_bytes, ctx->stream));
    HIP_TRY(ctx, hipMalloc((void **)&ctx->d_note_freqs, GDG_NOTE_COUNT * sizeof(double)));
    HIP_TRY(ctx, hipMemcpy(ctx->d_note_freqs, GDG_NOTE_FREQS, GDG_NOTE_COUNT * sizeof(double), hipMemcpyHostToDevice));
    HIP_TRY(ctx, hipMalloc((void **)&ctx->d_tuner_out, (size_t)ctx->nch * sizeof(gdg_tuner_out)));
    HIP_TRY(ctx, hipHostMalloc((void **)&ctx->h_tuner_out, (size_t)ctx->nch * sizeof(gdg_tuner_out), hipHostMallocDefault));
    ctx->tuner_wp = 0;
    return GDG_OK;
}

static int tuner_enqueue_rows(gdg_ctx *ctx, const double *d_samples, size_t stride, int frames, uint32_t sample_rate) {
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

static int ensure_staging(gdg_ctx *ctx);

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
    if (!force_long && gdg_tuner_short_ok((double)ctx->tuner_sr, GDG_NOTE_FREQS[0])) {
        /* every standard rate: block-wise autocorrelation for the lags the analysis can look at; the ring is read once */
        double2 *tw4096, *tw2_4096;
        rc = fir_tables(ctx, 4096, &tw4096, &tw2_4096);
        if (rc != GDG_OK) return rc;
        const int parts = gdg_tuner_short_parts(ctx->nch);
        if (parts > 1 && !ctx->d_tuner_part) HIP_TRY(ctx, hipMalloc((void **)&ctx->d_tuner_part, (size_t)ctx->nch * 8 * 4096 * sizeof(double2)));
        ProfScope ps(ctx, GDG_K_TUNER);
        HIP_TRY(ctx, gdg_launch_tuner_short(ctx->d_tuner_ring, ctx->nch, ctx->tuner_wp, (double)ctx->tuner_sr, tw4096, tw2_4096,
                                            ctx->d_note_freqs, GDG_NOTE_COUNT, ctx->d_tuner_out, ctx->d_tuner_part, parts, ctx->stream));
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
                                              ctx->d_tuner_out, ctx->stream));
    }
    /* pinned destination: the copy is a plain DMA behind the kernel (a pageable one goes through the runtime's staging path) */
    gdg_tuner_out *host = ctx->h_tuner_out;
    HIP_TRY(ctx, hipMemcpyAsync(host, ctx->d_tuner_out, (size_t)ctx->nch * sizeof(gdg_tuner_out), hipMemcpyDeviceToHost, ctx->stream));
    HIP_TRY(ctx, hipStreamSynchronize(ctx->stream));
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
static int spatialize_rows(gdg_ctx *ctx, const double *d_in, int in_stride, double *d_left, int out_stride, int frames) {
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

static int ensure_io(gdg_ctx *ctx, int which, size_t bytes) {
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
static int meter_rows(gdg_ctx *ctx, const double *d_rows, size_t row_stride, int port0, int n_ports, int frames, uint32_t sample_rate) {
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
static void move_pieces(gdg_ctx *ctx, const std::vector<BatchPiece> &pieces) {
    size_t total = 0;
    for (auto &p : pieces) total += p.bytes;
    copy_rows_parallel(ctx, 0, pieces.size(), [&](size_t i) { memcpy(pieces[i].dst, pieces[i].src, pieces[i].bytes); },
                       pieces.empty() ? 0 : total / pieces.size());
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
        for (size_t off = 0; off < length;) {
            int w = W;
            while ((size_t)w * B > length - off) w >>= 1;                        /* the tail: windows of W/2, W/4 .. 1 */
            steps.push_back({ off, w });
            off += (size_t)w * B;
        }
        auto scatter = [&](size_t i) {                                           /* step i's bytes from its pinned half into the files */
            const unsigned char *src = ctx->h_batch[i & 1];
            const size_t wb = (size_t)steps[i].w * B, row_bytes = wb * out_width, at = steps[i].off * out_width;
            const size_t f64_at = ((size_t)enc_rows * row_bytes + 15) & ~(size_t)15;
            copy_rows_parallel(ctx, 0, (size_t)enc_rows + (size_t)f64_rows, [&](size_t o) {
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
        };
        /* the streamed inputs of step i: gathered into a pinned half by the copy threads, moved and decoded on the upload stream while
         * the block loop is busy with the steps before */
        HIP_TRY(ctx, hipEventRecord(ctx->batch_begin, ctx->stream));             /* rows zeroed, whole-file inputs decoded */
        if (n_streamed) HIP_TRY(ctx, hipStreamWaitEvent(ctx->batch_up_stream, ctx->batch_begin, 0));
        int up_used[2] = { 0, 0 };
        auto stage = [&](size_t i) -> int {
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
                move_pieces(ctx, pieces);
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
            const size_t down = (((size_t)enc_rows * wb * out_width + 15) & ~(size_t)15) + (size_t)f64_rows * wb * sizeof(double);
            HIP_TRY(ctx, hipMemcpyAsync(ctx->h_batch[h], enc, sharded ? down : (size_t)NO * wb * out_width, hipMemcpyDeviceToHost, ctx->batch_stream));
            HIP_TRY(ctx, hipEventRecord(ctx->batch_moved[h], ctx->batch_stream));
            return GDG_OK;
        };
        /* The compute stream is kept TWO steps ahead of the files.  Round 2 enqueued step i + 1 only after step i - 1 had been scattered into
         * the caller's buffers, which closed a loop of compute -> download -> scatter over two steps: (5.7 + 4.0 + 3.1) / 2 = 6.4 ms per step
         * of 16 blocks where the device needs 5.7 (GDG_BATCH_TRACE).  Now step i + 2 is enqueued as soon as step i's download has finished (before
         * its bytes are scattered), while step i + 1 is already queued behind step i on the device. */
        if (trace) fprintf(stderr, "[batch] set-up %.2f ms\n", now_ms() - t_begin);
        for (size_t i = 0; i < 2 && i < steps.size(); i++) {
            if ((r = stage(i)) != GDG_OK) return r;
            if ((r = enqueue_compute(i)) != GDG_OK || (r = enqueue_down(i)) != GDG_OK) return r;
        }
        if (trace) fprintf(stderr, "[batch] steps 0 and 1 staged and enqueued at %.2f ms\n", now_ms() - t_begin);
        for (size_t i = 0; i < steps.size(); i++) {
            const double t_it = now_ms();
            if ((r = stage(i + 2)) != GDG_OK) return r;                          /* while steps i, i + 1 run: the inputs of step i + 2 go up ... */
            const double t_st = now_ms();
            HIP_TRY(ctx, hipEventSynchronize(ctx->batch_moved[i & 1]));          /* ... step i comes down ... */
            const double t_wait = now_ms();
            /* step i + 2 needs step i's half of `enc` (free now) but not its pinned half: it goes onto the compute stream BEFORE the scatter, so
             * the loop compute -> download -> compute spans 5.7 + 4.0 ms per two steps and the device, not the host, sets the pace */
            if (i + 2 < steps.size() && (r = enqueue_compute(i + 2)) != GDG_OK) return r;
            const double t_enq = now_ms();
            scatter(i);                                                          /* ... and goes into the files */
            if (i + 2 < steps.size() && (r = enqueue_down(i + 2)) != GDG_OK) return r;     /* its pinned half is free again */
            if (trace) fprintf(stderr, "[batch] step %zu: stage %zu %.2f | wait for the download %.2f | enqueue %zu %.2f | scatter %.2f  (at %.2f ms)\n", i, i + 2,
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

}  /* extern "C" */
