#!/usr/bin/env python3
"""bench.py -- the hot path's headline metric on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: N ranks under torch.distributed.run -- started by a launcher, or by
                                                             this file itself when it finds none; a launcher with another WORLD_SIZE is an error)

One "step" = one pass of the full effects chain over one 8192-frame block of every channel
(what controller.process() does per BLOCK_SIZE block, controller/controller.go:3076-3107), with
the input block already resident in HBM.  Workload (BASELINE.json metric config): 512 channels
@ 192 kHz, chain = compressor -> overdrive -> tone_stack -> chorus -> power_amp (64k-tap cabinet
IR) -> power_amp (64k-tap reverb IR) -> cabinet (IIR) -> reverb, every channel with its own IR
spectra in HBM (d = 1).  Channels are independent: N GPUs = N shards, no
collective on the data path, no RCCL (the barrier and the max-over-ranks of the elapsed time go
through a gloo group).

The job is FIXED, like the reference's `-channels 512` (controller.go:3262-3269, :3333-3341) -- "scaling": "strong":
  * default: the 512 channels are split over the N GPUs in contiguous blocks, rank r taking shard.channel_shard(512, N, r)
    (BASELINE config 4 at N = 8: 64 channels per GPU; at N = 1 the whole job on the one GPU); IRs and inputs are seeded by GLOBAL
    channel number.  `value` = 512 x frames x K / MAX-over-ranks time.
    With N > 1 the line also carries `one_gpu_alone` (rank 0's shard stepping alone: the one-GPU prediction of this split, next to
    the measurement), the split's batch-mode legs, `weak_scaling` (512 channels on EVERY GPU) and `config5_sharded` (256 tuners and
    the 256 -> 2 mixdown split the same way, partial mixes added on rank 0's host); with N = 1 the per-GPU legs of the split
    (64 / 128 / 256 channels on the one GPU: `strong_split_legs`).
  * --weak ("scaling": "weak"): 512 channels on EVERY GPU as the headline, the strong split as the extra leg `strong_split`.
  * --total-channels T: another fixed job, headline only.
When --steps / --warmup are shorter than the settled default (40 after 25), `settled` carries the same context's rate after 25 more
warm-up steps over 40 steps, so one line holds the caller's figure and the settled one.

Prints ONE JSON line (rank 0) with the metric, the roofline of the dominant kernel measured with
HIP events over the timed region, the PCIe-inclusive host-buffer rates ("end_to_end"), the other
BASELINE configs' rates ("configs") and a CPU baseline (the oracle "port") timed on the host cores.
"""
import argparse
import datetime
import json
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

HBM_PEAK_GBS = 8000.0        # MI355X_MICROARCH.md: 8 TB/s spec
METRIC = "Msamples/s through full chain incl. 64k-tap cab IR, 512ch@192kHz; %HBM roofline"
PROFILE_EVERY = 4             # timed region: HIP events around the dominant kernel on every 4th step

CHAIN = [
    ("compressor", [1, 30, -20]),
    ("overdrive", [0, 20, 100, 0, 1, 0]),
    ("tone_stack", None),
    ("chorus", None),
    ("power_amp", "cab"),
    ("power_amp", "rev"),
    ("cabinet", None),
    ("reverb", [50]),
]


def _synth():
    """go-dsp-guitar_amd/synth.py: SURVEY 8(d)'s inputs and IRs on the reference's LCG (random/random.go), ONE seed table for the HIP
    legs, the parity gate and the CPU baseline (inputs 1337 + channel; cabinet IR 4242 + 2 channel, reverb IR 4243 + 2 channel)."""
    import importlib
    import __graft_entry__ as entry
    entry.load_package()
    return importlib.import_module("go_dsp_guitar_amd.synth")


def synth_ir(n_taps, seed):
    return _synth().synth_ir(n_taps, seed)


def synth_block(n_channels, frames, sample_rate, channel0=0):
    return _synth().synth_block(n_channels, frames, sample_rate, channel0=channel0)


INPUT_BLOCKS = 4              # the headline walks through this many DISTINCT consecutive blocks of the synthetic stream, round robin


def synth_blocks(n_channels, frames, sample_rate, channel0=0, blocks=INPUT_BLOCKS):
    """[blocks][channels][frames]: consecutive blocks of SURVEY 8(d)'s stream (block b = samples b * frames ..)"""
    rows = _synth().synth_rows(n_channels, blocks * frames, sample_rate, channel0=channel0)
    return np.ascontiguousarray(rows.reshape(n_channels, blocks, frames).transpose(1, 0, 2))


def device_identity(index):
    """PCI bus id of HIP device `index` (hipDeviceGetPCIBusId), so the line says WHICH devices the ranks ran on."""
    import ctypes as C
    try:
        hip = C.CDLL("libamdhip64.so")
        buf = C.create_string_buffer(64)
        if hip.hipDeviceGetPCIBusId(buf, 64, int(index)) == 0:
            return buf.value.decode()
    except OSError:
        pass
    return None


def ir_for(kind, index, taps):
    return synth_ir(taps, _synth().ir_seed(kind, index))


def make_context(pkg, nch, frames, device, taps, channel0=0, n_distinct=0, chain=CHAIN, second_amp=True):
    """One shard: `nch` channels carrying GLOBAL channel numbers channel0 .. channel0 + nch - 1 (IR seeds follow the global
    number, so a channel sounds the same whichever GPU it lands on)."""
    ctx = pkg.Context(nch, frames, device)
    cache = {}
    for c in range(nch):
        g = channel0 + c
        for name, p in chain:
            if isinstance(p, str):
                if p == "rev" and not second_amp:
                    continue
                key = (p, g % n_distinct if n_distinct > 0 else g)
                if n_distinct > 0:
                    if key not in cache:
                        cache[key] = ir_for(p, key[1], taps)
                    ir = cache[key]
                else:
                    ir = ir_for(p, key[1], taps)
                ctx.append_unit(c, name, fir=ir)
            else:
                ctx.append_unit(c, name, params=p)
    return ctx


# ---- CPU baseline ---------------------------------------------------------------------------------------------------------

def host_cpu_info():
    """Physical cores, sockets and model name of the host from /proc/cpuinfo (north_star: "core count stated"): distinct (physical id, core id)
    pairs -- logical CPUs that share a pair are SMT siblings."""
    cores, sockets, model = set(), set(), None
    try:
        phys = core = None
        with open("/proc/cpuinfo") as f:
            for line in f:
                key, _, val = line.partition(":")
                key, val = key.strip(), val.strip()
                if key == "physical id":
                    phys = val
                elif key == "core id":
                    core = val
                elif key == "model name" and model is None:
                    model = val
                elif not key and phys is not None and core is not None:
                    cores.add((phys, core))
                    sockets.add(phys)
                    phys = core = None
            if phys is not None and core is not None:
                cores.add((phys, core))
                sockets.add(phys)
    except OSError:
        pass
    return {"physical_cores": len(cores) or None, "sockets": len(sockets) or None, "cpu_model": model}


def cpu_baseline(sample_rate, frames, taps, target_seconds=12.0):
    """The oracle ("port": structure-preserving C restatement of the Go path: unpartitioned 2 * nextpow2(L)-point radix-2 FFT
    pair per block and power amp, 8 exp per sample in the tone stack ...) on the host cores THIS process may use
    (sched_getaffinity), one thread per channel like the reference's goroutine per channel (controller.go:3339-3341).
    Reports the single-thread rate, the all-core rate and the parallel efficiency between them."""
    import __graft_entry__ as entry
    orc = entry.load_oracle()
    orc.build()
    try:
        cores = len(os.sched_getaffinity(0))
    except AttributeError:
        cores = os.cpu_count() or 1
    irs = {"cab": ir_for("cab", 0, taps), "rev": ir_for("rev", 0, taps)}          # channel 0's pair: SURVEY 8(d)'s seeds 4242 / 4243

    def make_chain():
        ch = orc.Chain()
        for name, p in CHAIN:
            if isinstance(p, str):
                ch.append_unit(name, fir=irs[p])
            else:
                ch.append_unit(name, params=p)
        return ch

    x = synth_block(cores, frames, sample_rate)
    probe = make_chain()
    probe.process(x[0], sample_rate)                      # builds H, the FFT tables, the buffers
    n1 = 0
    t0 = time.perf_counter()
    while n1 < 3 or time.perf_counter() - t0 < 1.0:
        probe.process(x[0], sample_rate)
        n1 += 1
    t_block = (time.perf_counter() - t0) / n1
    single = frames / t_block / 1e6

    def run(n_threads, blocks, chains):
        def work(c):
            for _ in range(blocks):
                chains[c].process(x[c], sample_rate)
        threads = [threading.Thread(target=work, args=(c,)) for c in range(n_threads)]
        t0 = time.perf_counter()
        for th in threads:
            th.start()
        for th in threads:
            th.join()
        return time.perf_counter() - t0

    chains = [make_chain() for _ in range(cores)]
    run(cores, 1, chains)                                 # first block of every chain: allocations, H
    dt = run(cores, 2, chains)                            # calibration
    blocks = int(max(2, min(2000, target_seconds / (dt / 2.0))))
    dt = run(cores, blocks, chains)
    all_core = cores * blocks * frames / dt / 1e6
    scaling = {}
    for n in sorted({max(1, cores // 16), max(1, cores // 8), max(1, cores // 4), max(1, cores // 2)}):
        if n >= cores:
            continue
        b = max(2, blocks // 2)
        scaling[str(n)] = n * b * frames / run(n, b, chains) / 1e6
    scaling[str(cores)] = all_core
    best_threads = max(scaling, key=lambda k: scaling[k])
    return {
        # the most favourable thread count for the CPU, not the reference's own (one goroutine per channel on every logical CPU)
        "value": scaling[best_threads],
        "unit": "Msamples/s",
        "cores": int(best_threads),                     # the contract's key: the threads `value` was measured with
        "best_threads": int(best_threads),
        "kind": "port",
        "logical_cpus": cores,
        **host_cpu_info(),
        "single_thread": single,
        "all_core": all_core,
        "parallel_efficiency_all_core": all_core / (cores * single),
        "threads_to_msamples": scaling,
        "sample": "%d channels x %d blocks of %d frames, same chain and IR lengths, one thread per channel on the %d logical CPUs of "
                  "this process's affinity mask (%.1f s wall); single thread: %d blocks, %.1f ms per block.  Per block and channel the "
                  "port streams ~340 MiB (2 power amps x 2 transforms of 131072 points x 17 radix-2 passes over data + twiddle table; "
                  "working set ~10 MB per channel): with few threads each has a whole CCD's L3 and scales, with one thread per logical CPU the working "
                  "sets evict each other and the run is DRAM-bound -- more threads give LESS throughput (threads_to_msamples); `value` is the best "
                  "thread count, `all_core` what the reference's goroutine-per-channel scheduling would get"
                  % (cores, blocks, frames, cores, dt, n1, t_block * 1e3),
    }


# ---- helpers for the extra legs ------------------------------------------------------------------------------------------------

def robust_time(run, sync, units=1, reps=5, settle=6):
    """Seconds per unit of `run()` (which enqueues `units` steps), measured the way every leg of this file is: warm-up calls until
    two consecutive timings agree within 5 % (at most `settle`), then `reps` timed calls, each bracketed by a synchronize.  Returns
    the MEDIAN with min / max and every repetition, so one slow repetition cannot pass for the leg's rate (and cannot hide either)."""
    def once():
        sync()
        t0 = time.perf_counter()
        run()
        sync()
        return (time.perf_counter() - t0) / units

    prev = once()
    settled = 0
    for settled in range(1, settle + 1):
        cur = once()
        ok = abs(cur - prev) <= 0.05 * min(cur, prev)
        prev = cur
        if ok:
            break
    ts = sorted(once() for _ in range(reps))
    return {"median": ts[len(ts) // 2], "min": ts[0], "max": ts[-1], "reps": ts, "warmup_calls": settled + 1}


def us_stats(st):
    return {"us_median": st["median"] * 1e6, "us_min": st["min"] * 1e6, "us_max": st["max"] * 1e6, "repetitions": len(st["reps"]),
            "warmup_calls": st["warmup_calls"]}


def leg_on_one_gpu(pkg, nch, frames, sr, taps, device, steps, channel0=0, chain=CHAIN, second_amp=True, n_distinct=0):
    """Per-frame calls of a fresh `nch`-channel context: `steps` steps per timed call (robust_time)."""
    ctx = make_context(pkg, nch, frames, device, taps, channel0=channel0, chain=chain, second_amp=second_amp, n_distinct=n_distinct)
    d_in, d_out = ctx.alloc(nch, frames), ctx.alloc(nch, frames)
    d_in.upload(synth_block(nch, frames, sr, channel0=channel0))

    def run():
        for _ in range(steps):
            ctx.process_device(d_in, d_out, frames, sr)
    st = robust_time(run, ctx.synchronize, units=steps)
    d_in.free()
    d_out.free()
    ctx.close()
    return st


def window_leg_on_one_gpu(pkg, nch, frames, sr, taps, device, W=16, windows=2, channel0=0, chain=CHAIN, second_amp=True):
    """Seconds per frame with W consecutive frames per call (batch mode, time blocked) over `windows` windows resident in HBM."""
    ctx = make_context(pkg, nch, frames, device, taps, channel0=channel0, chain=chain, second_amp=second_amp)
    ctx.set_window(W)
    n = W * windows * frames
    d_in, d_out = ctx.alloc(nch, n), ctx.alloc(nch, n)
    d_in.upload(np.tile(synth_block(nch, frames, sr, channel0=channel0), (1, W * windows)))

    def run():
        for b in range(0, W * windows, W):
            ctx.process_window_device(d_in.ptr + 8 * b * frames, d_out.ptr + 8 * b * frames, n, W, sr)
    st = robust_time(run, ctx.synchronize, units=W * windows, reps=3)
    d_in.free()
    d_out.free()
    ctx.close()
    return st


def end_to_end(pkg, ctx, nch, frames, sr, steps=6):
    """The boundary's host-buffer entry points on the SAME context (PCIe both ways inside the call; never the headline)."""
    import ctypes as C
    lib = pkg.lib()
    x = synth_block(nch, frames, sr)
    out = np.empty_like(x)
    ins = (C.c_void_p * nch)(*[x[c].ctypes.data for c in range(nch)])
    outs = (C.c_void_p * nch)(*[out[c].ctypes.data for c in range(nch)])
    carr = (C.c_int * nch)(*range(nch))
    ctx.process_staged(list(range(nch)), x, sr)            # fills the pinned slab once (the Go workers write it in parallel)
    res = {}
    for key, fn in (("staged", lambda: ctx._check(lib.gdg_process_staged(ctx._h, carr, nch, frames, sr))),
                    ("pageable", lambda: ctx._check(lib.gdg_process(ctx._h, ins, outs, frames, sr)))):
        st = robust_time(lambda: [fn() for _ in range(steps)], ctx.synchronize, units=steps, reps=3)
        dt = st["median"]
        res[key] = {"value": nch * frames / dt / 1e6, "unit": "Msamples/s", "ms_per_block": dt * 1e3,
                    "ms_per_block_min_max": [st["min"] * 1e3, st["max"] * 1e3],
                    "pcie_gbs_in_plus_out": 2 * nch * frames * 8 / dt / 1e9}
    res["what"] = ("gdg_process_staged: frames already in the pinned slab (what the Go workers fill), H2D + chain + D2H inside the call; "
                   "gdg_process: caller's pageable rows staged through the pinned slab by the library's copy threads")
    return res


def batch_run(pkg, ctx, nch, sr, blocks=128):
    """gdg_batch_run on the SAME context: 16-bit files in, 24-bit files out (N + 3 of them), everything between in HBM;
    per window size W of the block loop (1 = the reference's loop, 16 = time blocked)."""
    frames = 8192
    n = blocks * frames
    files = batch_files(nch, sr, blocks)
    res = {"blocks": blocks, "files_in": "lpcm16 x %d" % nch, "files_out": "lpcm24 x %d" % (nch + 3), "unit": "Msamples/s",
           "what": "controller.processFiles between 'files read' and 'files written' in one call: H2D of the file bytes, decode, the block "
                   "loop in steps of W blocks (N chains + metronome + spatializer + encode, the encoded step going down while the next one "
                   "runs); caller's buffers are pageable, already touched; timed at the C boundary (the 512 input structs and 515 pointers marshalled once)"}
    for W in (1, 16):
        ctx.set_window(W)
        call, outs = ctx.batch_prepared(files, sr, "lpcm24")    # arguments marshalled once: the timed call is the C call alone
        call()                                                  # also touches the output pages once
        ts = []
        for _ in range(3):
            t0 = time.perf_counter()
            call()
            ts.append(time.perf_counter() - t0)
        ts.sort()
        dt = ts[1]
        res["window_%d" % W] = {"value": nch * n / dt / 1e6, "ms": dt * 1e3, "ms_min_max": [ts[0] * 1e3, ts[2] * 1e3]}
        res["host_bytes_in_plus_out"] = sum(f[0].nbytes for f in files) + sum(o.nbytes for o in outs)
    res["value"] = res["window_16"]["value"]
    ctx.set_window(1)
    return res


def batch_files(nch, sr, blocks, channel0=0):
    """16-bit files of `blocks` x 8192 samples: wave.go's 16-bit export of the synthetic block of global channels channel0 .., tiled"""
    blk = np.trunc(32767.5 * synth_block(nch, 8192, sr, channel0=channel0)).astype(np.int16)
    return [(np.tile(blk[c], blocks).view(np.uint8), "lpcm16", sr) for c in range(nch)]


def sharded_batch_one_gpu(pkg, device, total, shards, sr, taps, blocks=32, W=16):
    """The batch job the way N GPUs run it -- one context per contiguous channel block, gdg_batch_run_shard on each (float64 partial
    master mixes), gdg_batch_finish_master once -- with all `shards` contexts on THIS GPU, one after the other: exercises and times
    the path (the shards run concurrently on N GPUs: the job takes the slowest shard + the finish)."""
    from go_dsp_guitar_amd import shard as sh
    res = {"shards": shards, "total_channels": total, "blocks": blocks, "window": W, "files_in": "lpcm16", "files_out": "lpcm24",
           "what": "gdg_batch_run_shard per contiguous channel block + gdg_batch_finish_master (shard partials added in shard order, "
                   "then encoded); shards run one after the other on the one GPU"}
    lefts, rights, t_shard = [], [], []
    n = blocks * 8192
    for g in range(shards):
        c0, cnt = sh.channel_shard(total, shards, g)
        ctx = make_context(pkg, cnt, 8192, device, taps, channel0=c0)
        ctx.set_window(W)
        for c in range(cnt):
            ctx.spatializer_set_position(c, -90.0 + 180.0 * (c0 + c) / max(total - 1, 1), 1.0 + 0.01 * (c0 + c), 0.5)
        files = batch_files(cnt, sr, blocks, channel0=c0)
        call, (outs, l, r, mb, mf) = ctx.batch_shard_prepared(files, sr, "lpcm24", job_samples=n, metronome=(g == 0))
        call()                                                  # touches pages, builds plans
        ts = []
        for _ in range(2):
            t0 = time.perf_counter()
            call()                                              # the C call alone (arguments marshalled, result buffers allocated once)
            ts.append(time.perf_counter() - t0)
        t_shard.append(min(ts))
        lefts.append(l)
        rights.append(r)
        if g == shards - 1:
            t0 = time.perf_counter()
            ml, mr = ctx.batch_finish_master("lpcm24", lefts, rights, aux=None)
            res["finish_master_ms"] = (time.perf_counter() - t0) * 1e3
            res["master_nonzero"] = bool(ml.any() and mr.any())
        ctx.close()
    res["shard_ms"] = [t * 1e3 for t in t_shard]
    slowest = max(t_shard)
    res["predicted_job_ms_on_%d_gpus" % shards] = slowest * 1e3 + res["finish_master_ms"]
    res["predicted_job_value"] = total * n / (slowest + res["finish_master_ms"] * 1e-3) / 1e6
    res["unit"] = "Msamples/s"
    return res


def time_blocked(pkg, ctx, nch, frames, sr, blocks=32):
    """The same chains over `blocks` consecutive frames per channel resident in HBM (what the batch run holds), W frames per call:
    every power amp reads its IR spectra and delay line once per W frames (gdg_process_window_device)."""
    d_in, d_out = ctx.alloc(nch, blocks * frames), ctx.alloc(nch, blocks * frames)
    d_in.upload(np.tile(synth_block(nch, frames, sr), (1, blocks)))
    res = {"unit": "Msamples/s", "frames_per_channel": blocks,
           "what": "device-resident, W consecutive 8192-sample frames per channel and call; W = 1 is the headline's per-frame call"}
    ctx.set_overlap(2 if nch >= 384 else 1)     # two free-running channel groups (W = 16: 323 -> 310 us)
    res["channel_groups"] = 2 if nch >= 384 else 1
    for W in (1, 2, 4, 8, 16):
        ctx.set_window(W)

        def run():
            for b in range(0, blocks, W):
                ctx.process_window_device(d_in.ptr + 8 * b * frames, d_out.ptr + 8 * b * frames, blocks * frames, W, sr)
        st = robust_time(run, ctx.synchronize, units=blocks, reps=3)
        dt = st["median"]
        res["window_%d" % W] = {"value": nch * frames / dt / 1e6, "us_per_frame": dt * 1e6, "us_per_frame_min_max": [st["min"] * 1e6, st["max"] * 1e6],
                                "realtime_factor": frames / sr / dt}
    ctx.set_window(1)
    d_in.free()
    d_out.free()
    return res


def other_configs(pkg, device):
    """BASELINE.json configs 2, 3 and 5 on one GPU, device-resident frames (config 1 is the CPU oracle by definition)."""
    out = {}
    chain3 = [(n, ([0, 20, 100, 0, 1, 2] if n == "overdrive" else p)) for n, p in CHAIN]        # 4x oversampling
    for key, nch, frames, sr, taps, chain, steps in (("config2_1ch_48k_8ktaps_1024frames", 1, 1024, 48000, 8192, CHAIN, 200),
                                                    ("config2_1ch_48k_8ktaps_8192frames", 1, 8192, 48000, 8192, CHAIN, 100),
                                                    ("config3_64ch_96k_4xOS_32ktaps", 64, 8192, 96000, 32768, chain3, 30)):
        st = leg_on_one_gpu(pkg, nch, frames, sr, taps, device, steps, chain=chain, second_amp=False)
        dt = st["median"]
        out[key] = {"value": nch * frames / dt / 1e6, "unit": "Msamples/s", "us_per_block": dt * 1e6, "realtime_factor": frames / sr / dt,
                    "timing": us_stats(st)}
    # config 3 in batch mode (whole files in HBM: windows of 16 frames per call)
    stw = window_leg_on_one_gpu(pkg, 64, 8192, 96000, 32768, device, chain=chain3, second_amp=False)
    out["config3_64ch_96k_4xOS_32ktaps"]["batch_mode_window_16"] = {"us_per_frame": stw["median"] * 1e6, "value": 64 * 8192 / stw["median"] / 1e6,
                                                                    "realtime_factor": 8192 / 96000 / stw["median"], "timing": us_stats(stw)}
    # the ends of the power amp's range (effects/poweramp.go:303-329): the DEFAULT filter order, 1 048 576 taps = 128 partitions at the batch block
    # size, and a 9600-tap filter at the live path's 64-sample hops = 150 partitions; one power amp per chain, 64 channels, 4 distinct IRs
    # (the spectra of 64 private 1M-tap IRs alone would be 2 GiB of synthetic data to make on the host)
    for key, nch, frames, sr, taps, steps in (("long_filter_1048576_taps_64ch_192k", 64, 8192, 192000, 1048576, 10),
                                              ("small_hop_64_frames_9600_taps_64ch_96k", 64, 64, 96000, 9600, 200)):
        st = leg_on_one_gpu(pkg, nch, frames, sr, taps, device, steps, chain=CHAIN, second_amp=False, n_distinct=4)
        dt = st["median"]
        out[key] = {"value": nch * frames / dt / 1e6, "unit": "Msamples/s", "us_per_block": dt * 1e6, "realtime_factor": frames / sr / dt,
                    "partitions": -(-taps // frames), "timing": us_stats(st)}
    # config 5: 256 tuners (96000-sample windows, 262144-point autocorrelation each) + spatializer 256 -> 2 at 192 kHz
    nch, frames, sr = 256, 8192, 192000
    ctx = pkg.Context(nch, frames, device)
    d_x = ctx.alloc(nch, frames)
    d_x.upload(synth_block(nch, frames, sr))
    d_lr = ctx.alloc(2, frames)
    ctx.spatializer_set_sample_rate(sr)
    for c in range(nch):
        ctx.spatializer_set_position(c, -90.0 + 180.0 * c / (nch - 1), 0.5 + 0.01 * c, 0.5)
    for _ in range(13):
        ctx.tuner_enqueue_device(d_x, frames, sr)
    ctx.spatialize_device(d_x, d_lr, frames)
    ctx.tuner_analyze()
    st_sp = robust_time(lambda: [ctx.spatialize_device(d_x, d_lr, frames) for _ in range(20)], ctx.synchronize, units=20)
    st_an = robust_time(lambda: [ctx.tuner_analyze(raw=True) for _ in range(20)], ctx.synchronize, units=20)
    t_sp, t_an = st_sp["median"], st_an["median"]
    # the two kernels' own durations (HIP events on the launches) against SURVEY 8(d)'s algorithmic bytes: 768 kB per analysis (the ring,
    # read once), 8 B per channel-sample for the mix
    ctx.profile_enable(kinds=[pkg.K_TUNER, pkg.K_SPATIALIZER])
    for _ in range(10):
        ctx.spatialize_device(d_x, d_lr, frames)
        ctx.tuner_analyze(raw=True)
    ctx.synchronize()
    ctx.profile_enable(False)
    roof = {}
    for kind, name, nbytes in ((pkg.K_TUNER, "tuner", nch * 96000 * 8.0), (pkg.K_SPATIALIZER, "spatializer", nch * frames * 8.0 + 2 * frames * 8.0)):
        ms, n = ctx.profile_read(kind)
        avg = (ms / n) if n else None
        gbs = (nbytes / (avg * 1e-3) / 1e9) if avg else None
        roof[name] = {"bound": "hbm", "algorithmic_bytes_per_launch": nbytes, "avg_launch_ms": avg, "launches": n, "achieved": gbs, "peak": HBM_PEAK_GBS,
                      "unit": "GB/s", "frac": (gbs / HBM_PEAK_GBS) if gbs else None}
    if roof["tuner"]["avg_launch_ms"]:
        # The short-lag analysis is FP64-issue bound, not HBM bound (profiles/experiments/README.md, r05): ~460 000 vector lane-operations per
        # 4096-sample block (4096-point transform with its twiddle powers, un-packing, accumulation), 24 blocks + the inverse per analysis;
        # a CU issues 64 FP64 lanes per clock (16 per SIMD) at 2.4 GHz.
        lane_ops = nch * 25 * 460e3
        peak = 256 * 64 * 2.4e9
        roof["tuner"]["fp64_issue"] = {"lane_operations_per_launch": lane_ops, "peak_lane_operations_per_s": peak,
                                       "achieved": lane_ops / (roof["tuner"]["avg_launch_ms"] * 1e-3),
                                       "frac": lane_ops / (roof["tuner"]["avg_launch_ms"] * 1e-3) / peak,
                                       "note": "estimate from the instruction mix; the bound that applies to this kernel (`frac` above is against HBM)"}
    d_x.free()
    d_lr.free()
    ctx.close()
    out["config5_256_tuners"] = {"value": nch / t_an, "unit": "analyses/s", "ms_per_256_analyses": t_an * 1e3, "timing": us_stats(st_an),
                                 "roofline": roof["tuner"]}
    # config 5's per-GPU shape on 8 GPUs: 32 tuners (a channel's blocks then go over 8 workgroups: gdg_tuner_short_parts)
    ctx32 = pkg.Context(32, frames, device)
    d32 = ctx32.alloc(32, frames)
    d32.upload(synth_block(32, frames, sr))
    for _ in range(13):
        ctx32.tuner_enqueue_device(d32, frames, sr)
    ctx32.tuner_analyze()
    # (50 calls per repetition: the synchronize that brackets a repetition -- an error-word read-back + a stream wait, ~40 us -- is not part of an analysis)
    st32 = robust_time(lambda: [ctx32.tuner_analyze(raw=True) for _ in range(50)], ctx32.synchronize, units=50)
    d32.free()
    ctx32.close()
    out["config5_32_tuners_per_gpu"] = {"value": 32 / st32["median"], "unit": "analyses/s", "us_per_32_analyses": st32["median"] * 1e6,
                                        "predicted_256_tuners_on_8_gpus": 256 / st32["median"], "timing": us_stats(st32)}
    out["config5_spatializer_256_to_2"] = {"value": nch * frames / t_sp / 1e6, "unit": "Msamples/s", "us_per_block": t_sp * 1e6,
                                           "timing": us_stats(st_sp), "roofline": roof["spatializer"]}
    return out


def split_job_legs(pkg, torch, dist, shard, sctx, sx, n_loc, c0, T, frames, sr, rank, ssync):
    """The batch-mode legs of the job split over the ranks, on a rank's own shard context `sctx` (n_loc channels from global channel c0;
    `sx`: one resident block): windows of 16 frames per call, and the batch run proper (gdg_batch_run_shard + the gather of the partial
    master mixes on rank 0's host + gdg_batch_finish_master).  Every timing is the MAX over ranks."""
    res = {}
    if frames == 8192:
        # the same split in batch mode: 16 consecutive frames per call, time blocked (every rank walks 2 windows of its shard)
        W, windows = 16, 2
        sctx.set_window(W)
        wx = sx.repeat(1, W * windows).contiguous()
        wy = torch.empty_like(wx)

        def wstep():
            for b in range(0, W * windows, W):
                sctx.process_window_device(wx.data_ptr() + 8 * b * frames, wy.data_ptr() + 8 * b * frames, W * windows * frames, W, sr)
        wstep()
        w_elapsed = shard.timed_steps(wstep, 1, ssync, dist, None) / (W * windows)
        res["batch_mode_window_16"] = {"us_per_frame": w_elapsed * 1e6, "value": T * frames / w_elapsed / 1e6,
                                                         "realtime_factor": frames / sr / w_elapsed}
        del wx, wy
        # ... and as the batch run proper: file bytes in, file bytes out, every rank its shard (gdg_batch_run_shard), the float64
        # partial master mixes gathered on rank 0's HOST over gloo ("the host adds the partials") and finished there
        blocks = 32
        n = blocks * frames
        for c in range(n_loc):
            sctx.spatializer_set_position(c, -90.0 + 180.0 * (c0 + c) / max(T - 1, 1), 1.0 + 0.01 * (c0 + c), 0.5)
        files = batch_files(n_loc, sr, blocks, channel0=c0)
        bcall, bres = sctx.batch_shard_prepared(files, sr, "lpcm24", job_samples=n, metronome=(rank == 0))
        bcall()
        held = {"r": bres}

        def bstep():
            bcall()
        b_elapsed = shard.timed_steps(bstep, 1, ssync, dist, None)
        dist.barrier()
        t0 = time.perf_counter()
        lefts, rights = shard.gather_master_partials(held["r"][1], held["r"][2], dist, dst=0)
        master_ok = None
        if rank == 0:
            ml, mr = sctx.batch_finish_master("lpcm24", lefts, rights, aux=None)
            master_ok = bool(ml.any() and mr.any())
        t_finish = time.perf_counter() - t0
        res["batch_run_sharded"] = {
            "blocks": blocks, "window": W, "files_in": "lpcm16", "files_out": "lpcm24", "shard_ms_max_over_ranks": b_elapsed * 1e3,
            "gather_and_finish_master_ms": t_finish * 1e3, "value": T * n / (b_elapsed + t_finish) / 1e6, "unit": "Msamples/s",
            "realtime_factor": n / sr / (b_elapsed + t_finish), "master_nonzero": master_ok,
            "what": "gdg_batch_run_shard on every rank (file bytes to file bytes, PCIe inside), partial master mixes gathered on "
                    "rank 0's host over gloo, gdg_batch_finish_master there"}
    return res


def config5_sharded(pkg, torch, dist, shard, device, world, rank, total=256, frames=8192, sr=192000):
    """BASELINE config 5 over the ranks: 256 tuners and the 256 -> 2 mixdown, rank r holding the contiguous block channel_shard(256, N, r).
    Tuners are independent (three scalars per channel come back); the spatializer gives one partial L/R pair per rank, added in rank order on
    rank 0's HOST (spatializer/spatializer.go:300-310) -- no device collective.  Timings are the MAX over ranks."""
    c0, n_loc = shard.channel_shard(total, world, rank)
    ctx = pkg.Context(n_loc, frames, device)
    d_x = ctx.alloc(n_loc, frames)
    d_x.upload(synth_block(n_loc, frames, sr, channel0=c0))
    d_lr = ctx.alloc(2, frames)
    ctx.spatializer_set_sample_rate(sr)
    for c in range(n_loc):
        g = c0 + c
        ctx.spatializer_set_position(c, -90.0 + 180.0 * g / (total - 1), 0.5 + 0.01 * g, 0.5)
    for _ in range(13):
        ctx.tuner_enqueue_device(d_x, frames, sr)
    ctx.spatialize_device(d_x, d_lr, frames)
    ctx.tuner_analyze()
    reps_an, reps_sp = 5, 20
    for _ in range(2):
        t_an = shard.timed_steps(lambda: ctx.tuner_analyze(raw=True), reps_an, ctx.synchronize, dist, None) / reps_an
        t_sp = shard.timed_steps(lambda: ctx.spatialize_device(d_x, d_lr, frames), reps_sp, ctx.synchronize, dist, None) / reps_sp
    # the mixdown's one exchange: the partial pair of every rank to rank 0's host, added there in rank order
    ctx.synchronize()
    dist.barrier()
    t0 = time.perf_counter()
    part = d_lr.download()
    lefts, rights = shard.gather_master_partials(part[0], part[1], dist, dst=0)
    mix_ok = None
    if rank == 0:
        left, right = shard.combine_spatializer_partials(list(zip(lefts, rights)))
        mix_ok = bool(np.isfinite(left).all() and np.isfinite(right).all() and left.any() and right.any())
    t_mix = time.perf_counter() - t0
    d_x.free()
    d_lr.free()
    ctx.close()
    return {"tuners_total": total, "tuners_per_gpu": n_loc, "n_gpus": world,
            "tuner": {"value": total / t_an, "unit": "analyses/s", "us_per_round_of_analyses_max_over_ranks": t_an * 1e6},
            "spatializer": {"value": total * frames / t_sp / 1e6, "unit": "Msamples/s", "us_per_block_max_over_ranks": t_sp * 1e6,
                            "gather_partials_and_host_sum_us": t_mix * 1e6, "mix_nonzero_and_finite": mix_ok},
            "what": "config 5 split over the ranks in contiguous channel blocks: tuner analyses at the C boundary, the spatializer's partial L/R per rank "
                    "(device-resident), the partials gathered over gloo and added on rank 0's host in rank order"}


# ---- parity gate (in the same run) ---------------------------------------------------------------------------------

PARITY_TOL_RMS = 1e-9          # north_star: output matches the float64 reference within 1e-9 RMS


def parity_gate(ctx_step, read_output, x_blocks, n_calls, nch, channel0, frames, sr, taps, n_distinct=0, extra_blocks=2):
    """The oracle (CHECKER only, never timed, never on the product path) follows the first, the middle and the last channel of the
    SAME context that was just timed: call k so far processed the block `x_blocks[k % len(x_blocks)]`, so the oracle replays those
    `n_calls` blocks per channel, compares the last one with what the device holds, then follows `extra_blocks` more steps.  Returns
    the worst per-channel RMS and the max-abs difference over the compared blocks."""
    import __graft_entry__ as entry
    orc = entry.load_oracle()
    orc.build()
    channels = sorted({0, nch // 2, nch - 1})
    chains = {}
    for c in channels:
        g = channel0 + c
        ch = orc.Chain()
        for name, p in CHAIN:
            if isinstance(p, str):
                ch.append_unit(name, fir=ir_for(p, (g % n_distinct) if n_distinct > 0 else g, taps))
            else:
                ch.append_unit(name, params=p)
        chains[c] = ch
    want = {c: None for c in channels}
    fed = {c: 0 for c in channels}

    def replay(c, blocks):
        for _ in range(blocks):
            want[c] = chains[c].process(x_blocks[fed[c] % len(x_blocks)][c], sr)
            fed[c] += 1

    def all_channels(blocks):
        ths = [threading.Thread(target=replay, args=(c, blocks)) for c in channels]
        for t in ths:
            t.start()
        for t in ths:
            t.join()

    all_channels(n_calls)
    sq = {c: 0.0 for c in channels}
    max_abs, compared = 0.0, 0
    for b in range(1 + extra_blocks):
        if b > 0:
            ctx_step()
            all_channels(1)
        got = read_output()
        for c in channels:
            d = got[c] - want[c]
            sq[c] += float(np.sum(d * d))
            max_abs = max(max_abs, float(np.max(np.abs(d))))
        compared += 1
    rms_max = max(float(np.sqrt(sq[c] / (compared * frames))) for c in channels)
    return {"channels": [channel0 + c for c in channels], "blocks_replayed_by_the_oracle": n_calls + extra_blocks, "blocks_compared": compared,
            "rms_max": rms_max, "max_abs": max_abs, "tolerance_rms": PARITY_TOL_RMS, "ok": bool(rms_max <= PARITY_TOL_RMS),
            "what": "oracle (C restatement of the Go reference) vs the device output of the timed context, per-channel RMS over the compared blocks"}


# ---- main ----------------------------------------------------------------------------------------------------------------------

def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    # defaults: the chip needs ~10 ms of work before its step time settles (3 warm-up steps: 0.556-0.563 ms per step in the timed region, 25: 0.536,
    # the same as runs that follow; profiles/small_shards_r05.txt section 11) -- a batch job runs for seconds, so the settled figure is the one to quote
    ap.add_argument("--steps", type=int, default=40)
    ap.add_argument("--warmup", type=int, default=25)
    ap.add_argument("--channels", type=int, default=512, help="the job's channels (default run: BASELINE's 512, split over the N GPUs); with --weak: channels per GPU")
    ap.add_argument("--total-channels", type=int, default=-1,
                    help="strong scaling: this many channels in total, split over the GPUs in contiguous blocks (default -1 = --channels; 0 = --weak)")
    ap.add_argument("--weak", action="store_true", help="weak scaling: --channels on EVERY GPU (a default run with N > 1 reports it as the extra leg `weak_scaling`)")
    ap.add_argument("--plan-only", action="store_true", help="print the job's shape (ranks, scaling, channels per GPU) and stop: needs no GPU")
    ap.add_argument("--sample-rate", type=int, default=192000)
    ap.add_argument("--frames", type=int, default=8192)
    ap.add_argument("--taps", type=int, default=65536)
    ap.add_argument("--channel-groups", type=int, default=0,
                    help="free-running channel groups on the GPU (gdg_ctx_set_overlap); 0 = 2 from 384 channels on, else 1")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-parity", action="store_true", help="skip the in-run parity gate (oracle follows 3 channels of the timed context)")
    ap.add_argument("--no-extras", action="store_true", help="only the headline measurement (no end_to_end / configs / split legs)")
    ap.add_argument("--distinct-irs", type=int, default=0,
                    help="number of distinct IR tap sets (0 = one per channel = the metric's d = 1; fewer lets power amps share spectra)")
    args = ap.parse_args()

    import __graft_entry__ as entry
    pkg = entry.load_package()
    from go_dsp_guitar_amd import shard

    # --gpus N means N ranks, whoever starts this file: a plain `python bench.py --gpus 8` re-executes itself under torch.distributed.run
    # (one rank per device), a launcher whose WORLD_SIZE is not N is an error -- never a line that says "n_gpus": 1 for a job asked with N = 8
    try:
        plan = shard.launch_plan(args.gpus, os.environ)
    except shard.LaunchError as e:
        sys.stderr.write("bench.py: %s\n" % e.msg)
        raise
    if plan == "spawn":
        cmd = shard.spawn_command(sys.executable, os.path.abspath(__file__), sys.argv[1:], args.gpus, shard.free_port())
        sys.stderr.write("bench.py: --gpus %d without a launcher: %s\n" % (args.gpus, " ".join(cmd)))
        sys.stderr.flush()
        os.execv(sys.executable, cmd)

    if int(os.environ.get("WORLD_SIZE", "1") or 1) > 1:         # one line per rank before anything can fail (a launcher kills the other ranks when one dies)
        sys.stderr.write("bench.py: rank %s of %s started\n" % (os.environ.get("RANK", "?"), os.environ["WORLD_SIZE"]))
        sys.stderr.flush()

    import torch  # before the library is loaded: libgdg.so then binds to the same HIP runtime (same SONAME)

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus
    distributed = world > 1
    one_device = bool(os.environ.get("GDG_BENCH_ONE_DEVICE"))     # harness self-test on a one-GPU box: every rank shares device 0
    try:
        shape = shard.job_shape(world, rank, args.channels, args.total_channels, args.weak)
    except shard.LaunchError as e:
        sys.stderr.write("bench.py: %s\n" % e.msg)
        raise
    if args.plan_only:
        # the job's shape as the ranks themselves derive it (no device touched): what a CPU test of the N > 1 default can check
        shapes = [shape]
        if distributed:
            import torch.distributed as dist
            os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
            dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=120))
            shapes = [None] * world
            dist.all_gather_object(shapes, shape)
            dist.barrier()
            dist.destroy_process_group()
        if rank == 0:
            print(json.dumps({"plan_only": True, "metric": METRIC, "unit": "Msamples/s", "n_gpus": world, "n_gpus_requested": args.gpus,
                              "scaling": shape["scaling"], "steps": args.steps, "warmup": args.warmup,
                              "config": {"total_channels": shape["total_channels"], "channels_per_gpu": shape["channels_per_gpu"],
                                         "channel_blocks": [[sh["channel0"], sh["channels_per_gpu"]] for sh in shapes],
                                         "sample_rate": args.sample_rate, "frames": args.frames, "ir_taps": args.taps}}))
        return
    try:
        local_rank = shard.pick_device(args.gpus, local_rank, torch.cuda.device_count(), one_device)
    except shard.LaunchError as e:
        sys.stderr.write("bench.py: %s\n" % e.msg)
        raise
    dist = None
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # control plane only (barrier + max of one scalar): gloo, so that NO RCCL / xGMI traffic exists anywhere in this job
        # a rank that dies inside an extra leg must not hang the others for gloo's default 30 minutes: collectives give up after 5
        dist.init_process_group("gloo", timeout=datetime.timedelta(seconds=int(os.environ.get("GDG_BENCH_DIST_TIMEOUT", "300"))))
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    devices = [{"rank": rank, "local_device": local_rank, "pci_bus_id": device_identity(local_rank), "name": torch.cuda.get_device_name(local_rank)}]
    if distributed:
        gathered = [None] * world
        dist.all_gather_object(gathered, devices[0])
        devices = gathered
        try:
            shard.check_distinct([d["pci_bus_id"] for d in devices], one_device)      # the same verdict on every rank
        except shard.LaunchError as e:
            if rank == 0:
                sys.stderr.write("bench.py: %s\n" % e.msg)
            dist.destroy_process_group()
            raise

    frames, sr, taps = args.frames, args.sample_rate, args.taps
    # the job: BASELINE's 512 channels, split over the N GPUs in contiguous blocks (config 4 at N = 8; the headline configuration itself at
    # N = 1) -- shard.job_shape; --weak puts 512 channels on every GPU instead
    strong = shape["scaling"] == "strong"
    channel0, nch, total_channels = shape["channel0"], shape["channels_per_gpu"], shape["total_channels"]
    default_job = strong and args.total_channels < 0          # the driver's own command line: every extra leg belongs to it
    # every channel has its OWN impulse responses (d = 1): identical filters would share one copy of the spectra
    n_distinct = args.distinct_irs
    ctx = make_context(pkg, nch, frames, local_rank, taps, channel0=channel0, n_distinct=n_distinct)
    # channel groups on one GPU are an explicit choice of the caller (gdg_ctx_set_overlap; the library's default is one group, whose
    # calls are ordered on the context's stream): two free-running groups from 384 channels on (0.625 -> 0.565 ms per step on the
    # same box, profiles/probes/bisect_g1.py), as a batch caller that only touches the results through the library would
    headline_groups = args.channel_groups if args.channel_groups > 0 else (2 if nch >= 384 else 1)
    ctx.set_overlap(headline_groups)
    # the launch-shape options in force (gdg_ctx_set_option: defaults unless an environment variable overrode one at context creation)
    library_options = {k: ctx.get_option(k) for k in pkg.option_names()}
    # the steps walk round robin through INPUT_BLOCKS distinct consecutive blocks of the stream (delay lines of identical spectra would be
    # a special case; the traffic is the same either way)
    x_host = synth_blocks(nch, frames, sr, channel0=channel0)
    x = torch.from_numpy(x_host).to(dev)
    y = torch.empty_like(x[0])
    x_ptrs = [x[b].data_ptr() for b in range(INPUT_BLOCKS)]

    calls = [0]

    def step():
        ctx.process_device(x_ptrs[calls[0] % INPUT_BLOCKS], y.data_ptr(), frames, sr)
        calls[0] += 1

    for _ in range(max(args.warmup, 1)):      # the first step also builds the plan and the IR spectra
        step()
    ctx.synchronize()
    # timed region: HIP events on the DOMINANT kernel only (the roofline's kernel).  The library hands the two events to the launch itself
    # (hipExtLaunchKernelGGL: the kernel's own begin / end timestamps, the duration rocprofv3 reports; GDG_PROFILE_ATTACH=0: recorded around it).
    # Bracketing all eight launches of a step costs ~7% of the step, so the other kernels are timed in an untimed pass below.
    # Every PROFILE_EVERY-th step's launches of that kernel are bracketed (gdg_profile_sample): an event pair also keeps the bracketed kernel
    # from overlapping its neighbours' ramp-up and tail -- with every step bracketed the timed region is 5 % slower than unobserved.
    ctx.profile_sample(PROFILE_EVERY)
    MAC_KINDS = [pkg.K_FIR_MAC, pkg.K_FIR_MAC_CHAIN]     # the dominant kernel and its chained variant (adjacent power amps)
    if os.environ.get("GDG_BENCH_TIMED_PROFILE", "1") != "0":       # experiment knob: what the events in the timed region cost
        ctx.profile_enable(kinds=[pkg.K_FIR_MAC])                  # the roofline kernel only (its chained variant: in the passes below)

    def synchronize():
        ctx.synchronize()
        torch.cuda.synchronize()

    # barrier + synchronize | exactly K steps | synchronize; MAX over ranks (tested on CPU with gloo)
    elapsed = shard.timed_steps(step, args.steps, synchronize, dist if distributed else None, None)
    ctx.profile_enable(False)
    ctx.profile_sample(1)
    # the timed region twice more (same K steps, same bracketing, rank-local): how far one run of it can be off
    repeats = []
    for _ in range(2):
        synchronize()
        t0 = time.perf_counter()
        for _ in range(args.steps):
            step()
        synchronize()
        repeats.append((time.perf_counter() - t0) / args.steps * 1e3)
    kernels = {}
    ms, n = ctx.profile_read(pkg.K_FIR_MAC)
    timed_mac = {"ms_total": ms, "launches": n, "avg_ms": (ms / n) if n else None}
    ms_c, n_c = ctx.profile_read(pkg.K_FIR_MAC_CHAIN)
    timed_chain = {"ms_total": ms_c, "launches": n_c, "avg_ms": (ms_c / n_c) if n_c else None}

    # The chip needs ~10 ms of work before its step time settles; a batch job runs for seconds.  When the caller's K / W are shorter than the
    # settled default (40 steps after 25 warm-ups), the same context runs 25 more warm-up steps and 40 timed ones (same bracketing as the timed
    # region, max over ranks), so ONE line carries the driver's figure and the settled one
    settled = None
    if args.steps < 40 or args.warmup < 25:
        for _ in range(25):
            step()
        ctx.profile_sample(PROFILE_EVERY)
        if os.environ.get("GDG_BENCH_TIMED_PROFILE", "1") != "0":
            ctx.profile_enable(kinds=[pkg.K_FIR_MAC])
        st_elapsed = shard.timed_steps(step, 40, synchronize, dist if distributed else None, None)
        settled = {"steps": 40, "warmup_before": args.warmup + 3 * args.steps + 25, "ms_per_step": st_elapsed / 40 * 1e3,
                   "value": total_channels * frames * 40 / st_elapsed / 1e6, "unit": "Msamples/s",
                   "what": "the same context after 25 more warm-up steps, 40 timed steps with the timed region's bracketing (max over ranks): the rate a job "
                           "that runs for seconds sees; `value` at the top is the caller's own --steps / --warmup"}
        ctx.profile_enable(False)
        ctx.profile_sample(1)
        ctx.profile_read(pkg.K_FIR_MAC)                 # (discarded: the roofline's launches are the timed region's)
    groups = headline_groups
    # Second pass: the SAME steps with one channel group and the dominant kernel (both variants) bracketed on every step, then once more
    # with every launch bracketed.  With --channel-groups > 1 a launch of the timed region shares the chip with the other group's kernels
    # and its HIP-event duration says little about the kernel: the roofline then comes from this pass.
    ctx.set_overlap(1)
    step()
    synchronize()
    ctx.profile_enable(kinds=MAC_KINDS)
    t_alone = time.perf_counter()
    for _ in range(args.steps):
        step()
    synchronize()
    t_alone = time.perf_counter() - t_alone
    ctx.profile_enable(False)
    ms, n = ctx.profile_read(pkg.K_FIR_MAC)
    kernels["fir_mac"] = {"ms_total": ms, "launches": n, "avg_ms": (ms / n) if n else None,
                          "pass": "timed region" if groups == 1 else "roofline pass: the timed region's steps again with the channel groups off (the kernel runs alone)"}
    ms_c, n_c = ctx.profile_read(pkg.K_FIR_MAC_CHAIN)
    kernels["fir_mac_chain"] = {"ms_total": ms_c, "launches": n_c, "avg_ms": (ms_c / n_c) if n_c else None,
                                "what": "the same kernel when another power amp follows: it also makes that amp's forward transform (history + delay-line slot)"}
    if groups == 1 and timed_mac["avg_ms"]:
        # one channel group: the timed region's own events time the kernel alone -- that average is the roofline's; the pass above (every
        # step bracketed) stays as a cross-check and for the per-step totals
        kernels["fir_mac"].update({"avg_ms_all_steps_bracketed": kernels["fir_mac"]["avg_ms"], "avg_ms": timed_mac["avg_ms"],
                                   "launches_timed_region": timed_mac["launches"],
                                   "pass": "timed region (HIP events on every %d-th step)" % PROFILE_EVERY})
    ctx.profile_enable(True)                      # untimed pass: the same steps again with every launch bracketed
    for _ in range(args.steps):
        step()
    synchronize()
    ctx.profile_enable(False)
    ctx.profile_read(pkg.K_FIR_MAC_CHAIN)
    for kind, name in enumerate(pkg.KERNEL_KINDS[:4]):
        ms, n = ctx.profile_read(kind)
        if name == "fir_mac":
            kernels["fir_mac"]["avg_ms_untimed_pass"] = (ms / n) if n else None
            continue
        kernels[name] = {"ms_total": ms, "launches": n, "avg_ms": (ms / n) if n else None, "pass": "untimed, all launches bracketed, channel groups off"}
    ctx.set_overlap(headline_groups)
    finite = bool(torch.isfinite(y).all().item())
    # parity gate on the context that was just timed (after the timed region; the oracle is the checker, nothing else)
    parity = None
    if not args.no_parity:
        def read_output():
            synchronize()
            return y.cpu().numpy()
        parity = parity_gate(step, read_output, x_host, calls[0], nch, channel0, frames, sr, taps, n_distinct=n_distinct)
        if distributed:
            t = torch.tensor([parity["rms_max"], parity["max_abs"]], dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            parity["rms_max_all_ranks"], parity["max_abs_all_ranks"] = float(t[0]), float(t[1])
            parity["ok"] = bool(float(t[0]) <= PARITY_TOL_RMS)

    extras = {}
    # the extra legs may not take the headline with them: whatever goes wrong in one is reported in the line, not instead of it
    run_extras = not args.no_extras and (default_job or not strong)
    try:
        if run_extras:
            if world == 1:
                extras["end_to_end"] = end_to_end(pkg, ctx, nch, frames, sr)
                if frames == 8192:
                    extras["end_to_end"]["batch_run"] = batch_run(pkg, ctx, nch, sr)
                    extras["time_blocked"] = time_blocked(pkg, ctx, nch, frames, sr)
            elif strong:
                # rank 0's shard ALONE on the node (the other ranks wait at the barrier): the step time one GPU predicts for this split,
                # next to the measured max over N ranks running together (host threads, PCIe and power are shared; the GPUs are not)
                dist.barrier()
                if rank == 0:
                    t_al = shard.timed_steps(step, args.steps, synchronize, None, None)
                    extras["one_gpu_alone"] = {"channels": nch, "ms_per_step": t_al / args.steps * 1e3,
                                               "predicted_job_value": total_channels * frames * args.steps / t_al / 1e6, "unit": "Msamples/s",
                                               "measured_over_predicted": (t_al / elapsed) if elapsed else None,
                                               "what": "rank 0's shard stepping alone, the other ranks idle at a barrier: the one-GPU prediction "
                                                       "of this split's step time; the headline `value` is the measurement (max over ranks, all running)"}
                dist.barrier()
                extras.update(split_job_legs(pkg, torch, dist, shard, ctx, x[0], nch, channel0, total_channels, frames, sr, rank, synchronize))
        ctx.close()
        del x, y
        if run_extras:
            if world > 1 and strong:
                # the weak-scaling leg: --channels on EVERY GPU (what the default measured up to round 5)
                wctx = make_context(pkg, args.channels, frames, local_rank, taps, channel0=rank * args.channels)
                wctx.set_overlap(2 if args.channels >= 384 else 1)
                wx = torch.from_numpy(synth_block(args.channels, frames, sr, channel0=rank * args.channels)).to(dev)
                wy = torch.empty_like(wx)

                def wwstep():
                    wctx.process_device(wx.data_ptr(), wy.data_ptr(), frames, sr)

                def wwsync():
                    wctx.synchronize()
                    torch.cuda.synchronize()

                for _ in range(max(args.warmup, 1)):
                    wwstep()
                ww_elapsed = shard.timed_steps(wwstep, args.steps, wwsync, dist, None)
                extras["weak_scaling"] = {"scaling": "weak", "channels_per_gpu": args.channels, "total_channels": world * args.channels, "n_gpus": world,
                                          "value": world * args.channels * frames * args.steps / ww_elapsed / 1e6, "unit": "Msamples/s",
                                          "ms_per_step": ww_elapsed / args.steps * 1e3,
                                          "realtime_factor": frames * args.steps / ww_elapsed / sr}
                wctx.close()
                del wx, wy
                if frames == 8192:
                    extras["config5_sharded"] = config5_sharded(pkg, torch, dist, shard, local_rank, world, rank)
            elif world > 1:
                # --weak: the strong split of the same 512-channel job over these N GPUs as the extra leg
                T = args.channels
                c0, n_loc = shard.channel_shard(T, world, rank)
                sctx = make_context(pkg, n_loc, frames, local_rank, taps, channel0=c0)
                sx = torch.from_numpy(synth_block(n_loc, frames, sr, channel0=c0)).to(dev)
                sy = torch.empty_like(sx)

                def sstep():
                    sctx.process_device(sx.data_ptr(), sy.data_ptr(), frames, sr)

                def ssync():
                    sctx.synchronize()
                    torch.cuda.synchronize()

                for _ in range(max(args.warmup, 1)):
                    sstep()
                s_elapsed = shard.timed_steps(sstep, args.steps, ssync, dist, None)
                extras["strong_split"] = {
                    "scaling": "strong", "total_channels": T, "channels_per_gpu": n_loc, "n_gpus": world,
                    "value": T * frames * args.steps / s_elapsed / 1e6, "unit": "Msamples/s",
                    "ms_per_step": s_elapsed / args.steps * 1e3, "realtime_factor": frames * args.steps / s_elapsed / sr,
                }
                extras["strong_split"].update(split_job_legs(pkg, torch, dist, shard, sctx, sx, n_loc, c0, T, frames, sr, rank, ssync))
                sctx.close()
            elif rank == 0:
                legs = {}
                for n_loc in (64, 128, 256):
                    st = leg_on_one_gpu(pkg, n_loc, frames, sr, taps, local_rank, 30)
                    dt = st["median"]
                    legs[str(n_loc)] = {"n_gpus_of_the_split": args.channels // n_loc, "us_per_step": dt * 1e6, "timing": us_stats(st),
                                        "value_this_gpu": n_loc * frames / dt / 1e6,
                                        "predicted_job_value": args.channels * frames / dt / 1e6, "unit": "Msamples/s",
                                        "predicted_realtime_factor": frames / sr / dt}
                    if frames == 8192:
                        stw = window_leg_on_one_gpu(pkg, n_loc, frames, sr, taps, local_rank)
                        dtw = stw["median"]
                        legs[str(n_loc)]["batch_mode_window_16"] = {"us_per_frame": dtw * 1e6, "timing": us_stats(stw),
                                                                   "predicted_job_value": args.channels * frames / dtw / 1e6,
                                                                   "predicted_realtime_factor": frames / sr / dtw}
                extras["strong_split_legs"] = {"what": "one GPU running its share of the 512-channel job split over 8 / 4 / 2 GPUs "
                                                       "(channels are independent: the job's step time is the slowest shard's step time); launch shapes "
                                                       "are the library's own by channel count: per-frame calls of <= 192 channels (option fir_split_max_channels) run the split multiply-accumulate with "
                                                       "the sums over the partitions already in the delay line made ahead of the frame (option fir_premac), windows of "
                                                       "<= 448 channels run a workgroup per frame and channel (option seg_wave_max_channels), oversampled shapers of calls of <= 192 channels "
                                                       "run as launches of their own, a workgroup per tile (option seg_os_tiles_max_channels)",
                                               "legs": legs}
                if frames == 8192:
                    extras["sharded_batch"] = sharded_batch_one_gpu(pkg, local_rank, args.channels, 2, sr, taps)
                extras["configs"] = other_configs(pkg, local_rank)
    except Exception as e:          # noqa: BLE001
        import traceback
        extras["error"] = {"rank": rank, "exception": repr(e), "traceback": traceback.format_exc().splitlines()[-6:]}
        print("bench.py: an extra leg failed on rank %d: %r" % (rank, e), file=sys.stderr, flush=True)

    if rank == 0:
        K = (taps + frames - 1) // frames
        spec_bytes = 16.0 * frames                       # one packed half spectrum (P complex128)
        samples_per_step = nch * frames
        mac = kernels["fir_mac"]
        # algorithmic bytes of the MAC launch: K delay-line spectra + K IR spectra per channel (d = 1);
        # the split variant's write of Y is NOT counted (it vanishes in the fused kernel).  With channel groups a step issues
        # several smaller launches per FIR unit: bytes per launch = bytes per step / launches per step.
        fir_per_chain = sum(1 for _, p in CHAIN if isinstance(p, str))
        d_share = (min(n_distinct, nch) / float(nch)) if n_distinct > 0 else 1.0       # d = 1 with per-channel IRs
        fused = not kernels["fir_inv"]["launches"]        # >= 128 channels per launch: MAC fused into the inverse transform's kernel
        out_bytes = 8.0 * frames if fused else 0.0         # the fused kernel also emits the output frame (8 B y out)
        # `mac` is the pass with the channel groups off: one launch per FIR unit and step covers all nch channels
        mac_bytes = nch * ((1.0 + d_share) * K * spec_bytes + out_bytes)
        mac_gbs = mac_bytes / (mac["avg_ms"] * 1e-3) / 1e9 if mac["avg_ms"] else None
        fir_units = fir_per_chain * args.steps
        chain = kernels["fir_mac_chain"]
        # the chained variant: 2 K spectra in, the NEXT amp's new spectrum out (16 B per sample); the frame itself never leaves the chip
        chain_bytes = nch * ((1.0 + d_share) * K * spec_bytes + spec_bytes)
        chain_gbs = chain_bytes / (chain["avg_ms"] * 1e-3) / 1e9 if chain["avg_ms"] else None
        fir_ms = sum((kernels[k]["ms_total"] or 0.0) for k in ("fir_fwd", "fir_mac", "fir_mac_chain", "fir_inv"))
        fir_bytes_per_sample = 16.0 + 16.0 * (1 + 2 * K)              # SURVEY 8d B_conv with (P+1)/P -> 1 (packed bin 0)
        fir_gbs = fir_units * samples_per_step * fir_bytes_per_sample / (fir_ms * 1e-3) / 1e9 if fir_ms else None
        seg = kernels["segment"]
        # HBM traffic from the PMC counters cannot be sampled from inside this process: it is taken from the committed
        # rocprofv3 --pmc passes of the SAME workload (profiles/pmc_fir_mac.json), else null
        traffic, seg_traffic = None, None
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_fir_mac.json")) as f:
                pmc = json.load(f)
            if pmc.get("workload_key") == "%dx%dx%d" % (nch, frames, taps) and n_distinct == 0 and bool(pmc.get("fused")) == fused:
                traffic = pmc["traffic_bytes_per_launch"]
                seg_traffic = pmc.get("segment_kernel", {}).get("traffic_bytes_per_launch")
        except (OSError, ValueError, KeyError):
            pass
        value = total_channels * frames * args.steps / elapsed / 1e6
        out = {
            "metric": METRIC,
            "value": value,
            "unit": "Msamples/s",
            "n_gpus": world,
            "n_gpus_requested": args.gpus,
            "devices": devices,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "ms_per_step_repeats": repeats,
            "higher_is_better": True,
            "scaling": "strong" if strong else "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "%s @ %d Hz, %d-frame blocks, full chain: compressor>overdrive>tone_stack>chorus>power_amp(%d-tap cab IR)>power_amp(%d-tap reverb IR)>cabinet>reverb; %s; input resident in HBM"
                            % (("%d channels in total, split over %d GPU(s) in contiguous blocks" % (total_channels, world)) if strong
                               else ("%d channels/GPU" % nch), sr, frames, taps, taps,
                               "per-channel IRs (d = 1)" if n_distinct == 0 else "%d distinct IR sets shared by the channels" % n_distinct),
                "channels_per_gpu": nch, "total_channels": total_channels, "sample_rate": sr, "frames": frames, "ir_taps": taps, "partitions": K,
                "realtime_factor": value * 1e6 / (total_channels * sr),
                "output_finite": finite,
                "inputs": "SURVEY 8(d): two sines + 0.05 x the reference LCG (random/random.go) seeded 1337 + channel; IRs (1 - 2 r) exp(-6.9 k / L) on the "
                          "LCG seeded 4242 + 2 channel (cabinet) / 4243 + 2 channel (reverb), unit energy",
                "channel_groups": headline_groups,
                "library_options": library_options,
                "channel_groups_note": "opt-in through gdg_ctx_set_overlap (free-running groups; the library's default is 1, ordered on the context's stream)",
                "control_plane": "gloo barrier + max of one scalar; no RCCL, no data-path collective",
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "fir_inv_kernel<13, 1> (spectrum multiply-accumulate fused into the inverse FFT)" if fused else "fir_mac_kernel",
                "achieved": mac_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": (mac_gbs / HBM_PEAK_GBS) if mac_gbs else None,
                "traffic": traffic,
                "algorithmic_bytes_per_launch": mac_bytes, "avg_launch_ms": mac["avg_ms"], "launches": mac.get("launches_timed_region", mac["launches"]),
                "chained_variant": {"kernel": "fir_inv_kernel<13, 1, CHAIN> (amp 1 of two adjacent power amps: + the forward transform of amp 2)",
                                    "algorithmic_bytes_per_launch": chain_bytes, "avg_launch_ms": chain["avg_ms"], "launches": chain["launches"],
                                    "achieved": chain_gbs, "frac": (chain_gbs / HBM_PEAK_GBS) if chain_gbs else None},
                "fir_unit_all_three_kernels": {"bytes_per_channel_sample": fir_bytes_per_sample, "achieved": fir_gbs,
                                               "frac": (fir_gbs / HBM_PEAK_GBS) if fir_gbs else None},
                "kernels_ms": kernels,
                "roofline_pass": {"channel_groups": 1, "ms_per_step": t_alone / args.steps * 1e3,
                                  "value_this_rank": nch * frames * args.steps / t_alone / 1e6,
                                  "note": "the same steps with every kernel alone on the chip (the pass `achieved` / `frac` come from)"},
                "timed_region": {"channel_groups": groups, "mac_launches": timed_mac["launches"], "mac_avg_launch_ms": timed_mac["avg_ms"],
                                 "bracketed_steps": "every %d-th of the %d timed steps" % (PROFILE_EVERY, args.steps),
                                 "mac_achieved_while_sharing_the_chip": (mac_bytes / groups / (timed_mac["avg_ms"] * 1e-3) / 1e9) if timed_mac["avg_ms"] else None,
                                 "note": "with channel groups > 1 a launch covers 1/groups of the channels and overlaps the other group's kernels"},
                "segment_kernel": {"bytes_per_channel_sample_frame_only": 16.0,
                                   "achieved": (seg["launches"] * samples_per_step * 16.0 / (seg["ms_total"] * 1e-3) / 1e9) if seg["ms_total"] else None,
                                   # counter traffic incl. the state of the delay-type units (rings that cannot stay on chip)
                                   "traffic": seg_traffic,
                                   "achieved_incl_state": (seg_traffic / (seg["avg_ms"] * 1e-3) / 1e9) if (seg_traffic and seg["avg_ms"]) else None},
            },
        }
        if parity is not None:
            out["parity"] = parity
        out.update(extras)
        ss = extras.get("strong_split")
        if ss:
            # the strong split of the fixed 512-channel job (BASELINE config 4's shape at N = 8) in TOP-LEVEL keys, next to the weak headline
            out["strong_value"], out["strong_unit"], out["strong_ms_per_step"] = ss["value"], ss["unit"], ss["ms_per_step"]
            out["strong_total_channels"], out["strong_channels_per_gpu"] = ss["total_channels"], ss["channels_per_gpu"]
            out["strong_realtime_factor"] = ss["realtime_factor"]
            if "batch_mode_window_16" in ss:
                out["strong_batch_mode_value"] = ss["batch_mode_window_16"]["value"]
                out["strong_batch_mode_realtime_factor"] = ss["batch_mode_window_16"]["realtime_factor"]
        ws = extras.get("weak_scaling")
        if ws:
            # the weak leg (512 channels on EVERY GPU) in top-level keys next to the strong headline
            out["weak_value"], out["weak_unit"], out["weak_ms_per_step"] = ws["value"], ws["unit"], ws["ms_per_step"]
            out["weak_total_channels"], out["weak_channels_per_gpu"] = ws["total_channels"], ws["channels_per_gpu"]
        if settled is not None:
            out["settled"] = settled
        cfg = extras.get("configs") or {}
        for key, name in (("config5_256_tuners", "tuner"), ("config5_spatializer_256_to_2", "spatializer")):
            if key in cfg and "roofline" in cfg[key]:
                out["roofline"][name + "_kernel"] = cfg[key]["roofline"]
        if not args.no_cpu_baseline and world == 1:       # rank 0 at N = 1 only: it is a property of the host, not of the GPU count
            out["cpu_baseline"] = cpu_baseline(sr, frames, taps)
        print(json.dumps(out))
    if distributed:
        dist.barrier()
        dist.destroy_process_group()
    if parity is not None and not parity["ok"]:
        sys.stderr.write("bench.py: PARITY GATE FAILED: per-channel RMS %.3e > %.1e\n" % (parity["rms_max"], PARITY_TOL_RMS))
        sys.exit(1)


if __name__ == "__main__":
    main()
