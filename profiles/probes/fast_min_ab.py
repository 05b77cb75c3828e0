#!/usr/bin/env python3
"""64 / 128 / 256 channels of the bench chain (the legs of the strong split): us per per-frame step and per frame of a W = 16 window.
Run under GDG_SEG_FAST_MIN=257 (general segment kernel up to a chip's worth of channels) and =1 (two-per-CU kernel for all)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import bench
import __graft_entry__ as entry
pkg = entry.load_package()
frames, sr = 8192, 192000
for nch in (64, 128, 256):
    st = bench.leg_on_one_gpu(pkg, nch, frames, sr, 65536, 0, 30)
    ctx = bench.make_context(pkg, nch, frames, 0, 65536)
    W, blocks = 16, 32
    ctx.set_window(W)
    d_in, d_out = ctx.alloc(nch, blocks * frames), ctx.alloc(nch, blocks * frames)
    d_in.upload(np.tile(bench.synth_block(nch, frames, sr), (1, blocks)))
    def run():
        for b in range(0, blocks, W):
            ctx.process_window_device(d_in.ptr + 8 * b * frames, d_out.ptr + 8 * b * frames, blocks * frames, W, sr)
    w = bench.robust_time(run, ctx.synchronize, units=blocks, reps=3)
    print("%d channels: per-frame step %.1f us, W=16 %.1f us per frame" % (nch, st["median"] * 1e6, w["median"] * 1e6), flush=True)
    ctx.close()
