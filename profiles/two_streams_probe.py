"""Do kernels of two channel groups overlap usefully?  512 channels as one context vs 2 x 256 / 4 x 128 contexts (own streams) on one GPU,
driven round-robin from one host thread; W = 8 windows over 32 frames."""
import sys, time
import numpy as np
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from importlib import import_module
pkg = import_module("go-dsp-guitar_amd")
sr, frames, taps, blocks, W = 192000, 8192, 65536, 32, int(sys.argv[1]) if len(sys.argv) > 1 else 8
total = 512
for groups in (1, 2, 4):
    n = total // groups
    ctxs, bufs = [], []
    for g in range(groups):
        ctx = bench.make_context(pkg, n, frames, 0, taps, channel0=g * n)
        ctx.set_window(W)
        d_in, d_out = ctx.alloc(n, blocks * frames), ctx.alloc(n, blocks * frames)
        d_in.upload(np.tile(bench.synth_block(n, frames, sr, channel0=g * n), (1, blocks)))
        ctxs.append(ctx); bufs.append((d_in, d_out))
    def run():
        for b in range(0, blocks, W):
            for ctx, (d_in, d_out) in zip(ctxs, bufs):
                ctx.process_window_device(d_in.ptr + 8 * b * frames, d_out.ptr + 8 * b * frames, blocks * frames, W, sr)
    def sync():
        for c in ctxs: c.synchronize()
    run(); sync()
    t0 = time.perf_counter(); run(); sync(); dt = (time.perf_counter() - t0) / blocks
    print("W=%d groups=%d: %.1f us per frame, %.1f Msamples/s" % (W, groups, dt * 1e6, total * frames / dt / 1e6))
    for c in ctxs: c.close()
