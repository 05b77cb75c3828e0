#!/usr/bin/env python3
"""Time-blocked path (W = 16) by number of free-running channel groups: us per frame, 512 channels, bench chain."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import bench
import __graft_entry__ as entry
pkg = entry.load_package()
nch, frames, sr = int(os.environ.get("NCH", "512")), 8192, 192000
W = int(os.environ.get("W", "16"))
blocks = 4 * W
ctx = bench.make_context(pkg, nch, frames, 0, 65536)
ctx.set_window(W)
d_in, d_out = ctx.alloc(nch, blocks * frames), ctx.alloc(nch, blocks * frames)
d_in.upload(np.tile(bench.synth_block(nch, frames, sr), (1, blocks)))
def run():
    for b in range(0, blocks, W):
        ctx.process_window_device(d_in.ptr + 8 * b * frames, d_out.ptr + 8 * b * frames, blocks * frames, W, sr)
for G in (1, 2, 3, 4, 6, 8):
    ctx.set_overlap(G)
    st = bench.robust_time(run, ctx.synchronize, units=blocks, reps=3)
    print("W=%d %d ch, %d groups: %.1f us/frame (min %.1f max %.1f)" % (W, nch, G, st["median"] * 1e6, st["min"] * 1e6, st["max"] * 1e6), flush=True)
ctx.close()
