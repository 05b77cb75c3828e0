/*
 * ctx.h -- what the host side of libgdg.so shares between its source files: the context, the per-unit records, the plan's step
 * descriptors, the small helpers every entry point uses (fail / enter / HIP_TRY), the copy-worker pool and the profiling scope, and
 * the handful of functions one file offers the others.  Not part of the ABI (include/gdg.h is); everything declared here has hidden
 * visibility: libgdg.so exports the gdg_* entry points and nothing else.
 *
 *   api_ctx.cpp         context life cycle, options, NUMA placement, units and chains, profiling, device-memory helpers, stand-alone FFT
 *   api_plan.cpp        per-unit constants (prepare_unit), scan tables, IR spectra / delay lines (prepare_fir), the launch plan (build_plan)
 *   api_process.cpp     parameter patches, process_rows (the launch sequence of a call), the process entry points, staging + copy workers
 *   api_tuner_spat.cpp  tuner and spatializer glue
 *   api_io.cpp          wave codecs, resample.Time, level meters, power-amp compilation, metronome
 *   api_batch.cpp       the batch run (gdg_batch_run, its sharded form, the master mix)
 */
#ifndef GDG_CTX_H
#define GDG_CTX_H

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

#pragma GCC visibility push(hidden)

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
    int premac_lds = 0;               /* ... and the LDS its launch asks for without using it (api_plan.cpp) */
    bool premac_ok = false;           /* FIR step: split shape (few channels), 8192-sample frames, every channel with K >= 2: the terms k >= 1 can be summed ahead */
    int os_factor = 0;                /* 2 / 4: the step is ONE oversampled shaper per channel, run as a launch of its own (seg.hip os_tiles_kernel) */
    int os_flags = -1;                /* ... and its per-channel flags start here in d_wave */
    int os_arrive = -1;               /* ... its per-channel arrival counters (a compressor step absorbed into the launch, below) */
    int os_prefix_step = -1;          /* ... the step right in front of it when that is a lone compressor in every one of its channels: in per-frame calls the tiles'
                                       * workgroups run it themselves and the step is not launched (seg.hip os_tiles_kernel, pre_chans) */
    bool absorbed_per_frame = false;  /* segment step: a per-frame call skips it -- the oversampled shaper's launch behind it runs its compressor */
    int ahead_n = 0;                  /* segment step (general kernel, one frame per launch): it also makes the wet paths of this many reverbs of LATER steps ... */
    size_t ahead_offset = 0;          /* ... whose indices into the plan's unit array start here in the blob (seg.hip REVERB_AHEAD) */
    bool tile_ok = false;             /* segment step: every unit of every channel can run with the frame on two workgroups (seg.hip SEG_TILE) */
    bool wave_release = false;        /* segment step: some channel's segment hands its state on through a write-back of the XCD's L2 (wave_mask bit 31) */
    int wave_tickets = -1;            /* segment step: first of its GDG_WAVE_GROUPS ticket counters in d_wave (seg.hip, WAVE), -1: none */
    std::vector<std::pair<int, int>> group_range;     /* per channel group: (first descriptor, count) */
};

struct ProfEvent { int kind; hipEvent_t a, b; };
class CopyPool;

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
    int seg_fast_min = 128;                    /* GDG_SEG_FAST_MIN: fewest channels of a call that take the two-per-CU kernel.  Per-frame calls of up to a chip's
                                                * worth of channels finish a frame 1-3 % sooner on the general kernel (one round of 1024-thread workgroups; 128
                                                * channels 214 vs 217 us per step, 256: 301 vs 305; profiles/fast_min_ab_r04.txt) -- but windows of few channels
                                                * run a workgroup per frame (WAVE, below), and there twice the frames in flight are worth 7-13 % from 128 channels
                                                * on (W = 16: 96 channels 65.1 vs 63.0 us per frame, 128: 83.1 vs 76.8, 256: 159 vs 137.6; not from 96: per-frame
                                                * calls of 96 channels lose 13 % -- pairs of 512-thread workgroups land on one CU beside the premac).  ONE threshold for both
                                                * kinds of call: the two builds differ in the association of their scans (~1e-16), and a window has to give the
                                                * bits of the same frames called one by one. */
    /* Windows of few channels (seg.hip, WAVE): up to this many channels per launch a window's segment launch puts every FRAME of a channel on a
     * workgroup of its own, the frames meeting unit by unit through counters in HBM -- a GPU's share of the 512-channel job on eight GPUs is 64
     * channels, and one workgroup per channel walking the window leaves 3/4 of the CUs idle (64 channels, W = 16: 417 us per segment launch,
     * 52 of the 77 us per frame).  0: never.  gdg_ctx_set_option("seg_wave_max_channels"), env GDG_SEG_WAVE_MAX. */
    int seg_tile_max = 112;                    /* per-frame calls of up to this many channels: a segment of compressor / shapers / tone stack / cabinet / chorus (/ the mix of a
                                                * reverb made ahead) runs with a channel's frame on TWO workgroups (seg.hip SEG_TILE; same bits) -- as long as the launch's
                                                * workgroups, 2 x channels + the reverbs' extra ones, leave the chip room (GDG_TILE_WORKGROUP_BUDGET) */
#define GDG_TILE_WORKGROUP_BUDGET 224          /* of 256 CUs, one workgroup each.  Per step, tile kernel off / on (profiles/tile_ab_r06.txt, shape_sweep_r06.txt): bench chain 64
                                                * channels (192 workgroups) 142.9 -> 138.4 us, 72 (216) 148.8 -> 146.6, 80 (240) 155.9 -> 158.0; no reverb, 96 channels (192)
                                                * 111.5 -> 103.6; 96 kHz chain with a reverb, 96 channels (288) 110.9 -> 118.7 */
    unsigned long long *d_tile_xch = nullptr;  /* what crosses between the two workgroups of a channel: gdg_segt_xch_words() words per descriptor of a launch */
    size_t d_tile_xch_cap = 0;
    bool seg_os_prefix = true;                 /* a lone compressor in front of an oversampled shaper's own launch runs inside that launch in per-frame calls (option seg_os_tiles_prefix) */
    int seg_os_tiles_max = 192;                /* calls of up to this many channels run oversampled shapers as launches of their own, a workgroup per tile
                                                * (option "seg_os_tiles_max_channels"; 0: never) */
    int seg_wave_release_max = 112;            /* ... of segments with a unit whose state leaves the CU through plain stores (flanger, phaser, delay, fuzz, auto-yoy, auto-wah,
                                                * band pass, octaver, noise gate): every hand-off then writes the XCD's L2 back, and from ~128 channels on the walk
                                                * is faster (192 channels 84 vs 66 us per frame, 448: 202 vs 145; profiles/shape_sweep_r06.txt) */
    int seg_wave_max = 448;                    /* W = 16, us per frame, walk -> a workgroup per frame: 64 channels 77 -> 48, 128: 102 -> 77, 256: 146 -> 138, 384: 222 -> 206;
                                                * 512: 259 -> 263 (the walk wins once two workgroups per CU are all busy anyway) */
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
    int fir_premac_lds = -1;                   /* option "fir_premac_lds_bytes": -1 by channel count (api_process.cpp) */
    int fir_premac_min = 384;                  /* fewest partitions (sum of K over a launch's channels) worth it: the two cross-stream hops and the
                                                * three extra spectra of the inverse kernel cost ~20 us per step -- the multiply-accumulate of 48 x 8
                                                * partitions takes that long (16 x 8: 113.7 -> 123.4 us per step with it, 64 x 4 (config 3): 137 -> 141) */
    int fir_premac_min_two = 320;              /* ... with two or more power amps per channel the hops hide twice as much: 40 x 8 partitions per amp 120.6 -> 109.3 us
                                                * per step with it, 32 x 8: 112.4 -> 109.6, 24 x 8: 107.2 -> 108.6 (profiles/premac_loads_ab_r06.txt) */
    hipStream_t premac_stream = nullptr;
    hipEvent_t ev_fir_done = nullptr, ev_premac = nullptr;
    int stat_premac_used = 0;                  /* option "stat_premac_launches_used" (read it; tests): inverse launches that continued sums made ahead */
    bool premac_valid = false;                 /* Y of every premac step holds the terms k >= 1 of the plan's NEXT frame */
    bool premac_outstanding = false;           /* ... and the context's stream has not been ordered behind that launch yet */
    /* reverbs' wet paths ahead of the frame (seg.hip REVERB_AHEAD): made by extra workgroups of an EARLIER segment launch of the same call */
    int seg_reverb_ahead_max = 72;             /* most channels of a call that does it: 64 channels 156 -> 142 us per step, 96 channels 168 -> 175 (twice the
                                                * workgroups in the first segment launch, and the premac's share of the chip with them); with the premac's
                                                * loads fixed: 64 channels 135.1 -> 125.7, 72: 135.6 -> 134.7, 80: 137.7 -> 140.0, 88: 143.3 -> 147.4 */
    int wave_spin_ms = 1000;                   /* how long a frame waits for its predecessor's counter before the launch gives up (seg.hip wave_spin_expired; d_error[1]) */
    int debug_stall_unit = -1;                 /* test hook: this unit's first counter of a WAVE launch stays away, so that the bounded wait expires */
    int wave_epoch = 0;                        /* a number per WAVE launch (seg.hip: "done" marks carry it) */
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
    int fir_split_max = 192;          /* largest launch (channels) that takes the split shape.  Round 2 measured the two shapes equal at 128 channels
                                       * (220 us per step either way) and set 96; with the sums made ahead of the frame (premac, below) the split shape
                                       * took 207 us there and lost at 192 (252 vs 274), hence 128 until the premac stopped reading through the cache
                                       * (fir.hip, fir_mac_kernel): 128 channels 181 vs 223 fused, 160 215 vs 236, 192 247 vs 257, 208 266 vs 264
                                       * (profiles/shape_sweep_r06.txt, chain d) */
    int fir_split_max_single = 112;   /* ... when the call has ONE power amp per channel: the premac has half as much to hide, and at 128 channels the fused kernel wins
                                       * (one amp: 138.8 split vs 127.1 fused us per step, 96 channels 112 vs 125; profiles/shape_sweep_r06.txt) */
    int plan_fir_steps = 0;           /* power-amp steps of the current plan */
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
    gdg_tuner_out *d_tuner_out = nullptr, *h_tuner_out = nullptr;      /* results: pinned host memory the kernels write directly (d_ = its device-side address) */
    double2 *d_tuner_work = nullptr, *d_tuner_twn = nullptr, *d_tuner_twm = nullptr;
    int tuner_part_cap = 0;                    /* parts per channel d_tuner_part holds */
    unsigned tuner_seq = 0;                    /* number of the last analysis (every result record carries it: api_tuner_spat.cpp) */
    int tuner_poll = 1;                        /* option tuner_poll_results: the caller polls the records instead of waiting for the stream */
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
    hipEvent_t batch_chunk[2][4] = {};         /* a step's download in four pieces: the scatter into the caller's files starts when the first has landed */
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
    CopyPool *copy_pool_up = nullptr;          /* ... a second set for the batch run's upload side: the next step's bytes are gathered while this step's are scattered */
    /* metronome (metronome/metronome.go): sounds in HBM, the two counters on the host */
    double *d_tick = nullptr, *d_tock = nullptr;
    uint32_t n_tick = 0, n_tock = 0;
    uint32_t met_sample_counter = 0, met_tick_counter = 0, met_beats = 4, met_bpm = 120, met_sr = 96000;
};

inline int fail(const gdg_ctx *ctx, int code, const char *fmt, ...) {       /* one definition, one mutex, for all files */
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    if (ctx) {
        static std::mutex mu;                      /* the batch run's helper thread may fail beside the calling thread (api_batch.cpp) */
        std::lock_guard<std::mutex> lk(mu);
        ctx->err = buf;
    }
    return code;
}

/* Device-resident calls may leave their channel groups running on streams of their own (process_rows, `free_run`); whatever
 * touches the context next -- any entry point -- first makes the context's stream wait for them. */
static inline void join_groups(gdg_ctx *ctx) {
    if (!ctx->groups_pending) return;
    for (size_t g = 0; g < ctx->gstreams.size() && g < ctx->gjoin.size(); g++) {
        hipEventRecord(ctx->gjoin[g], ctx->gstreams[g]);
        hipStreamWaitEvent(ctx->stream, ctx->gjoin[g], 0);
    }
    ctx->groups_pending = false;
}
/* the context's stream behind the premac launch; `keep` = the caller changes no state (synchronize, stream, profiling): the sums stay usable */
static inline void join_premac(gdg_ctx *ctx, bool keep) {
    if (ctx->premac_outstanding) {
        hipStreamWaitEvent(ctx->stream, ctx->ev_premac, 0);
        ctx->premac_outstanding = false;
    }
    if (!keep) ctx->premac_valid = false;
}
static inline void enter(gdg_ctx *ctx, bool read_only = false) {
    hipSetDevice(ctx->device);
    join_groups(ctx);
    join_premac(ctx, read_only);
}

/* largest launch (channels) that takes the split convolution shape: by the number of power amps per channel in the plan */
static inline int fir_split_limit(const gdg_ctx *ctx) {
    return ctx->plan_fir_steps >= 2 ? ctx->fir_split_max : std::min(ctx->fir_split_max, ctx->fir_split_max_single);
}

#define HIP_TRY(ctx, call)                                                                          \
    do {                                                                                            \
        hipError_t e_ = (call);                                                                     \
        if (e_ != hipSuccess) return fail(ctx, GDG_ERR_HIP, "%s failed: %s", #call, hipGetErrorString(e_)); \
    } while (0)

static inline double decibels_to_factor(int32_t decibels) {          /* effects/effects.go:389-394 */
    double e = 0.05 * (double)decibels;
    return pow(10.0, e);
}

static inline double lanczos_kernel(double x, double a) {             /* resample/resample.go:10-31 */
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


/* ---- shared between the source files ------------------------------------------------------------------------------------------- */
/* what a host-buffer entry point does around group g's kernels, on group g's stream (upload before, download after) */
typedef std::function<hipError_t(int g, hipStream_t s)> GroupHook;

hipEvent_t take_event(gdg_ctx *ctx);
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

void numa_bind_thread(const std::vector<int> &cpus);

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

Unit *get_unit(gdg_ctx *ctx, int handle);
bool segf_unit_ok(const Unit &u, int frames, uint32_t sample_rate);
bool reverb_ahead_ok(int frames, uint32_t sample_rate);
hipError_t pinned_alloc(gdg_ctx *ctx, void **p, size_t bytes);
int build_plan(gdg_ctx *ctx, const std::vector<int> &active, const double *d_in, double *d_out, int frames, uint32_t sample_rate,
                      int stride, int stride_out, bool rows_by_channel, int G, const std::vector<size_t> &bounds);
int check_device_error(gdg_ctx *ctx);
int ensure_io(gdg_ctx *ctx, int which, size_t bytes);
int ensure_staging(gdg_ctx *ctx);
int fir_tables(gdg_ctx *ctx, int P, double2 **tw, double2 **tw2);
int fir_transform_size(int frames);
int meter_rows(gdg_ctx *ctx, const double *d_rows, size_t row_stride, int port0, int n_ports, int frames, uint32_t sample_rate);
int numa_rebind(gdg_ctx *ctx, int mode);
int numa_target(const gdg_ctx *ctx, const std::vector<int> **cpus);
int prepare_unit(gdg_ctx *ctx, Unit &u, int frames, uint32_t sample_rate, gdg_seg_unit &d, int chk = GDG_CHK);
int process_rows(gdg_ctx *ctx, const std::vector<int> &active, const double *d_in, double *d_out, int frames, uint32_t sample_rate,
                        int stride = 0, bool rows_by_channel = false, int groups = 1, const GroupHook *before = nullptr, const GroupHook *after = nullptr,
                        int window = 1, int stride_out = 0, const std::vector<size_t> *group_bounds_in = nullptr);
int spatialize_rows(gdg_ctx *ctx, const double *d_in, int in_stride, double *d_left, int out_stride, int frames);
int tuner_enqueue_rows(gdg_ctx *ctx, const double *d_samples, size_t stride, int frames, uint32_t sample_rate);
void copy_rows_parallel(gdg_ctx *ctx, size_t a, size_t b, const std::function<void(size_t)> &copy_row, size_t row_bytes, int which = 0);      /* which: 1 = the upload side's workers */
void destroy_copy_pool(CopyPool *p);
void ensure_copy_pool(gdg_ctx *ctx, int which);     /* makes the pool NOW, on the calling thread (option "numa" = 2 binds the workers to the node of the thread that makes them) */

#pragma GCC visibility pop

#endif
