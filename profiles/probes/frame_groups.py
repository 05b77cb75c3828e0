#!/usr/bin/env python3
"""Per-frame device calls by number of free-running channel groups: us per step, 512 channels, bench chain (warmed up)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import bench
import __graft_entry__ as entry
pkg = entry.load_package()
nch, frames, sr = int(os.environ.get("NCH", "512")), 8192, 192000
ctx = bench.make_context(pkg, nch, frames, 0, 65536)
d_in, d_out = ctx.alloc(nch, frames), ctx.alloc(nch, frames)
d_in.upload(bench.synth_block(nch, frames, sr))
def run():
    for _ in range(20):
        ctx.process_device(d_in, d_out, frames, sr)
for G in (1, 2, 3, 4, 2):
    ctx.set_overlap(G)
    st = bench.robust_time(run, ctx.synchronize, units=20, reps=5)
    print("%d ch, %d groups: %.1f us/step (min %.1f max %.1f)" % (nch, G, st["median"] * 1e6, st["min"] * 1e6, st["max"] * 1e6), flush=True)
ctx.close()
