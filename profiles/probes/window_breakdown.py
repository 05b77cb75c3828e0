#!/usr/bin/env python3
"""Per-kernel-kind HIP-event times of the time-blocked path (W frames per call), channel groups off, bench workload."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import bench
import __graft_entry__ as entry
pkg = entry.load_package()
nch, frames, sr = int(os.environ.get("NCH", "512")), 8192, 192000
W = int(os.environ.get("W", "16"))
blocks = 2 * W
ctx = bench.make_context(pkg, nch, frames, 0, 65536)
ctx.set_window(W)
ctx.set_overlap(1)
d_in, d_out = ctx.alloc(nch, blocks * frames), ctx.alloc(nch, blocks * frames)
d_in.upload(np.tile(bench.synth_block(nch, frames, sr), (1, blocks)))
def run():
    for b in range(0, blocks, W):
        ctx.process_window_device(d_in.ptr + 8 * b * frames, d_out.ptr + 8 * b * frames, blocks * frames, W, sr)
run(); ctx.synchronize()
st = bench.robust_time(run, ctx.synchronize, units=blocks, reps=3)
ctx.profile_enable(True)
run(); ctx.synchronize()
ctx.profile_enable(False)
line = "W=%d %d ch chain=%s: %.1f us/frame |" % (W, nch, os.environ.get("GDG_FIR_CHAIN", "1"), st["median"] * 1e6)
tot = 0.0
for kind, name in enumerate(pkg.KERNEL_KINDS[:4]):
    ms, n = ctx.profile_read(kind)
    line += " %s %.1f us/frame (x%d)" % (name, 1e3 * ms / blocks, n)
    tot += 1e3 * ms / blocks
print(line + " | sum %.1f" % tot, flush=True)
ctx.close()
