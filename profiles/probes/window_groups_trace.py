#!/usr/bin/env python3
"""W = 16 with G channel groups (argv[1]), a few windows: for a rocprofv3 --kernel-trace run (what runs beside what)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import bench
import __graft_entry__ as entry
pkg = entry.load_package()
G = int(sys.argv[1]) if len(sys.argv) > 1 else 2
nch, frames, sr, W = 512, 8192, 192000, 16
blocks = 4 * W
ctx = bench.make_context(pkg, nch, frames, 0, 65536)
ctx.set_window(W)
ctx.set_overlap(G)
d_in, d_out = ctx.alloc(nch, blocks * frames), ctx.alloc(nch, blocks * frames)
d_in.upload(np.tile(bench.synth_block(nch, frames, sr), (1, blocks)))
for rep in range(3):
    for b in range(0, blocks, W):
        ctx.process_window_device(d_in.ptr + 8 * b * frames, d_out.ptr + 8 * b * frames, blocks * frames, W, sr)
ctx.synchronize()
ctx.close()
