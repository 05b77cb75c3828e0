#!/usr/bin/env python3
"""bench.py -- the hot path's headline metric on MI355X.

    python bench.py --gpus N --steps K --warmup W          (N > 1: launched by torch.distributed.run)

One "step" = one pass of the full effects chain over one 8192-frame block of every channel
(what controller.process() does per BLOCK_SIZE block, controller/controller.go:3076-3107), with
the input block already resident in HBM.  Workload (BASELINE.json metric config): 512 channels
@ 192 kHz per GPU, chain = compressor -> overdrive -> tone_stack -> chorus -> power_amp (64k-tap
cabinet IR) -> power_amp (64k-tap reverb IR) -> cabinet (IIR) -> reverb, every channel with its
own IR spectra in HBM.  Channels are independent: N GPUs = N shards, no
collective on the data path; per-GPU work is fixed (weak scaling).

Prints ONE JSON line (rank 0) with the metric, the roofline of the dominant kernel measured with
HIP events over the timed region, and a CPU baseline (the oracle "port") timed on the host cores.
"""
import argparse
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


def synth_ir(n_taps, seed):
    rng = np.random.default_rng(seed)
    k = np.arange(n_taps)
    h = (1.0 - 2.0 * rng.random(n_taps)) * np.exp(-6.9 * k / float(n_taps))
    return h / np.sqrt(np.sum(h * h))


def synth_block(n_channels, frames, sample_rate, channel0=0):
    t = np.arange(frames) / float(sample_rate)
    x = np.empty((n_channels, frames))
    for c in range(n_channels):
        f = 82.4069 * 2.0 ** (((channel0 + c) % 48) / 12.0)
        rng = np.random.default_rng(1337 + channel0 + c)
        x[c] = 0.5 * np.sin(2 * np.pi * f * t) + 0.25 * np.sin(2 * np.pi * 3 * f * t) + 0.05 * (1.0 - 2.0 * rng.random(frames))
    return x


def cpu_baseline(sample_rate, frames, taps, target_seconds=20.0):
    """The oracle ("port": structure-preserving C restatement of the Go path) on the host cores,
    one thread per channel like the reference's goroutine-per-channel (controller.go:3339-3341)."""
    import __graft_entry__ as entry
    orc = entry.load_oracle()
    orc.build()
    cores = os.cpu_count() or 1
    irs = {"cab": synth_ir(taps, 4242), "rev": synth_ir(taps, 4243)}

    def make_chain():
        ch = orc.Chain()
        for name, p in CHAIN:
            if isinstance(p, str):
                ch.append_unit(name, fir=irs[p])
            else:
                ch.append_unit(name, params=p)
        return ch

    probe = make_chain()
    x = synth_block(cores, frames, sample_rate)
    probe.process(x[0], sample_rate)
    t0 = time.perf_counter()
    probe.process(x[0], sample_rate)
    t_block = max(time.perf_counter() - t0, 1e-4)
    blocks = int(max(4, min(400, target_seconds / (cores * t_block))))
    chains = [make_chain() for _ in range(cores)]

    def work(c):
        for _ in range(blocks):
            chains[c].process(x[c], sample_rate)

    threads = [threading.Thread(target=work, args=(c,)) for c in range(cores)]
    t0 = time.perf_counter()
    for th in threads:
        th.start()
    for th in threads:
        th.join()
    dt = time.perf_counter() - t0
    return {
        "value": cores * blocks * frames / dt / 1e6,
        "unit": "Msamples/s",
        "cores": cores,
        "kind": "port",
        "sample": "%d channels x %d blocks of %d frames, same chain and IR lengths, one thread per channel; single-thread block time %.1f ms"
                  % (cores, blocks, frames, t_block * 1e3),
    }


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--channels", type=int, default=512, help="channels per GPU")
    ap.add_argument("--sample-rate", type=int, default=192000)
    ap.add_argument("--frames", type=int, default=8192)
    ap.add_argument("--taps", type=int, default=65536)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--distinct-irs", type=int, default=0,
                    help="number of distinct IR tap sets (0 = one per channel = the metric's d = 1; fewer lets power amps share spectra)")
    args = ap.parse_args()

    import torch  # first: libgdg.so then binds to the same HIP runtime (same SONAME)
    import __graft_entry__ as entry
    pkg = entry.load_package()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    distributed = world > 1
    if distributed:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        torch.cuda.set_device(local_rank)
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    else:
        torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    nch, frames, sr, taps = args.channels, args.frames, args.sample_rate, args.taps
    ctx = pkg.Context(nch, frames, local_rank)
    # every channel has its OWN impulse responses (d = 1): identical filters would share one copy of the spectra
    n_distinct = args.distinct_irs if args.distinct_irs > 0 else nch
    irs = {"cab": [synth_ir(taps, 4242 + i) for i in range(n_distinct)], "rev": [synth_ir(taps, 5242 + i) for i in range(n_distinct)]}
    for c in range(nch):
        for name, p in CHAIN:
            if isinstance(p, str):
                ctx.append_unit(c, name, fir=irs[p][c % n_distinct])
            else:
                ctx.append_unit(c, name, params=p)
    x = torch.from_numpy(synth_block(nch, frames, sr, channel0=rank * nch)).to(dev)
    y = torch.empty_like(x)

    def step():
        ctx.process_device(x.data_ptr(), y.data_ptr(), frames, sr)

    for _ in range(max(args.warmup, 1)):      # the first step also builds the plan and the IR spectra
        step()
    ctx.synchronize()
    # timed region: HIP events around the DOMINANT kernel only (the roofline's kernel; ~0.5 us per event record).
    # Bracketing all eight launches of a step costs ~7% of the step, so the other kernels are timed in an untimed pass below.
    ctx.profile_enable(kinds=[pkg.K_FIR_MAC])
    from go_dsp_guitar_amd import shard

    def synchronize():
        ctx.synchronize()
        torch.cuda.synchronize()

    # barrier + synchronize | exactly K steps | synchronize; MAX over ranks (tested on CPU with gloo)
    elapsed = shard.timed_steps(step, args.steps, synchronize, dist if distributed else None, dev)
    ctx.profile_enable(False)
    kernels = {}
    ms, n = ctx.profile_read(pkg.K_FIR_MAC)
    kernels["fir_mac"] = {"ms_total": ms, "launches": n, "avg_ms": (ms / n) if n else None, "pass": "timed region"}
    ctx.profile_enable(True)                      # untimed pass: the same steps again with every launch bracketed
    for _ in range(args.steps):
        step()
    synchronize()
    ctx.profile_enable(False)
    for kind, name in enumerate(pkg.KERNEL_KINDS[:4]):
        ms, n = ctx.profile_read(kind)
        if name == "fir_mac":
            kernels["fir_mac"]["avg_ms_untimed_pass"] = (ms / n) if n else None
            continue
        kernels[name] = {"ms_total": ms, "launches": n, "avg_ms": (ms / n) if n else None, "pass": "untimed, all launches bracketed"}
    finite = bool(torch.isfinite(y).all().item())

    if rank == 0:
        K = (taps + frames - 1) // frames
        spec_bytes = 16.0 * frames                       # one packed half spectrum (P complex128)
        samples_per_step = nch * frames
        mac = kernels["fir_mac"]
        # algorithmic bytes of the MAC launch: K delay-line spectra + K IR spectra per channel (d = 1);
        # the split variant's write of Y is NOT counted (it vanishes in the fused kernel).  With channel groups a step issues
        # several smaller launches per FIR unit: bytes per launch = bytes per step / launches per step.
        fir_per_chain = sum(1 for _, p in CHAIN if isinstance(p, str))
        d_share = min(n_distinct, nch) / float(nch)                       # d = 1 with per-channel IRs
        fused = not kernels["fir_inv"]["launches"]        # the library's default: MAC fused into the inverse transform's kernel
        out_bytes = 8.0 * frames if fused else 0.0         # the fused kernel also emits the output frame (8 B y out)
        mac_bytes = nch * ((1.0 + d_share) * K * spec_bytes + out_bytes) * fir_per_chain * args.steps / max(mac["launches"], 1)
        mac_gbs = mac_bytes / (mac["avg_ms"] * 1e-3) / 1e9 if mac["avg_ms"] else None
        fir_units = mac["launches"]
        fir_ms = sum((kernels[k]["avg_ms"] or 0.0) * fir_units for k in ("fir_fwd", "fir_mac", "fir_inv"))
        fir_bytes_per_sample = 16.0 + 16.0 * (1 + 2 * K)              # SURVEY 8d B_conv with (P+1)/P -> 1 (packed bin 0)
        fir_gbs = fir_units * samples_per_step * fir_bytes_per_sample / (fir_ms * 1e-3) / 1e9 if fir_ms else None
        seg = kernels["segment"]
        # HBM traffic from the PMC counters cannot be sampled from inside this process: it is taken from the committed
        # rocprofv3 --pmc passes of the SAME workload (profiles/pmc_fir_mac.json), else null
        traffic = None
        try:
            with open(os.path.join(ROOT, "profiles", "pmc_fir_mac.json")) as f:
                pmc = json.load(f)
            if pmc.get("workload_key") == "%dx%dx%d" % (nch, frames, taps) and n_distinct >= nch and bool(pmc.get("fused")) == fused:
                traffic = pmc["traffic_bytes_per_launch"]
        except (OSError, ValueError, KeyError):
            pass
        out = {
            "metric": "Msamples/s through full chain incl. 64k-tap cab IR, 512ch@192kHz; %HBM roofline",
            "value": world * samples_per_step * args.steps / elapsed / 1e6,
            "unit": "Msamples/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "f64",
            "data": "synthetic",
            "config": {
                "workload": "%d channels/GPU @ %d Hz, %d-frame blocks, full chain: compressor>overdrive>tone_stack>chorus>power_amp(%d-tap cab IR)>power_amp(%d-tap reverb IR)>cabinet>reverb; %s; input resident in HBM"
                            % (nch, sr, frames, taps, taps, "per-channel IRs (d = 1)" if n_distinct >= nch else "%d distinct IR sets shared by the channels" % n_distinct),
                "channels_per_gpu": nch, "sample_rate": sr, "frames": frames, "ir_taps": taps, "partitions": K,
                "realtime_factor": world * samples_per_step * args.steps / elapsed / (world * nch * sr),
                "output_finite": finite,
            },
            "roofline": {
                "bound": "hbm",
                "kernel": "fir_inv_kernel<13, 1> (spectrum multiply-accumulate fused into the inverse FFT)" if fused else "fir_mac_kernel",
                "achieved": mac_gbs, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": (mac_gbs / HBM_PEAK_GBS) if mac_gbs else None,
                "traffic": traffic,
                "algorithmic_bytes_per_launch": mac_bytes, "avg_launch_ms": mac["avg_ms"], "launches": mac["launches"],
                "fir_unit_all_three_kernels": {"bytes_per_channel_sample": fir_bytes_per_sample, "achieved": fir_gbs,
                                               "frac": (fir_gbs / HBM_PEAK_GBS) if fir_gbs else None},
                "kernels_ms": kernels,
                "segment_kernel": {"bytes_per_channel_sample_frame_only": 16.0,
                                   "achieved": (seg["launches"] * samples_per_step * 16.0 / (seg["ms_total"] * 1e-3) / 1e9) if seg["ms_total"] else None},
            },
        }
        if not args.no_cpu_baseline and world == 1:       # rank 0 at N = 1 only: it is a property of the host, not of the GPU count
            out["cpu_baseline"] = cpu_baseline(sr, frames, taps)
        print(json.dumps(out))
    ctx.close()
    if distributed:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
