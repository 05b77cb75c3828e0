#!/usr/bin/env python3
"""Small shards (a GPU's share of the 512-channel job on 8 / 4 GPUs): per-frame device calls and windows of 16 frames by number of
free-running channel groups, with the per-kernel-kind HIP-event times of the one-group run.

    NCH=64 MODE=frame|window GROUPS=1,2,4,8 [CHAIN=config3] [TAPS=65536 AMPS=2 DISTINCT=0] [OPTIONS=key=value,...] python profiles/probes/small_ctx.py
"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import bench
import __graft_entry__ as entry
pkg = entry.load_package()
nch, frames = int(os.environ.get("NCH", "64")), 8192
mode = os.environ.get("MODE", "frame")
groups = [int(g) for g in os.environ.get("NGROUPS_LIST", "1,2,4,8").split(",")]
W = int(os.environ.get("W", "16"))
if os.environ.get("CHAIN", "") == "config3":
    sr, taps, second = 96000, 32768, False
    chain = [(n, ([0, 20, 100, 0, 1, 2] if n == "overdrive" else p)) for n, p in bench.CHAIN]
else:
    sr, taps, second, chain = 192000, int(os.environ.get("TAPS", "65536")), os.environ.get("AMPS", "2") == "2", bench.CHAIN
ctx = bench.make_context(pkg, nch, frames, 0, taps, chain=chain, second_amp=second, n_distinct=int(os.environ.get("DISTINCT", "0")))
opts = os.environ.get("OPTIONS", "")                 # "key=value,key=value": gdg_ctx_set_option before the first call
for kv in [o for o in opts.split(",") if o]:
    k, v = kv.split("=")
    ctx.set_option(k, int(v))
tag = "%s %d ch %s hwq=%s%s" % (os.environ.get("CHAIN", "bench"), nch, mode, os.environ.get("GPU_MAX_HW_QUEUES", "default"), (" [" + opts + "]") if opts else "")
if mode == "frame":
    d_in, d_out = ctx.alloc(nch, frames), ctx.alloc(nch, frames)
    d_in.upload(bench.synth_block(nch, frames, sr))
    units = 30
    def run():
        for _ in range(units):
            ctx.process_device(d_in, d_out, frames, sr)
else:
    blocks = 2 * W
    units = blocks
    ctx.set_window(W)
    d_in, d_out = ctx.alloc(nch, blocks * frames), ctx.alloc(nch, blocks * frames)
    d_in.upload(np.tile(bench.synth_block(nch, frames, sr), (1, blocks)))
    def run():
        for b in range(0, blocks, W):
            ctx.process_window_device(d_in.ptr + 8 * b * frames, d_out.ptr + 8 * b * frames, blocks * frames, W, sr)
for G in groups:
    ctx.set_overlap(G)
    st = bench.robust_time(run, ctx.synchronize, units=units, reps=5 if mode == "frame" else 3)
    print("%s, %d groups: %.1f us per frame (min %.1f max %.1f)" % (tag, G, st["median"] * 1e6, st["min"] * 1e6, st["max"] * 1e6), flush=True)
if os.environ.get("KINDS", "1") != "0":
    ctx.set_overlap(1)
    run(); ctx.synchronize()
    ctx.profile_enable(True)
    run(); ctx.synchronize()
    ctx.profile_enable(False)
    line = "%s, 1 group, kernels bracketed:" % tag
    for kind in (pkg.K_FIR_FWD, pkg.K_FIR_MAC, pkg.K_FIR_INV, pkg.K_SEGMENT, pkg.K_FIR_MAC_CHAIN):
        ms, n = ctx.profile_read(kind)
        if n:
            line += " %s %.1f us/frame (x%d)" % (pkg.KERNEL_KINDS[kind] if kind < len(pkg.KERNEL_KINDS) else "kind%d" % kind, 1e3 * ms / units, n)
    print(line, flush=True)
ctx.close()
