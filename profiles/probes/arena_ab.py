#!/usr/bin/env python3
"""A/B of the per-unit state's placement (DevArena in ctx.h): one hipMalloc per block vs the arena at several alignments.
Runs itself once per setting in a child process (the knobs are read when a context is created)."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

def child():
    import numpy as np
    import bench
    import __graft_entry__ as entry
    pkg = entry.load_package()
    out = []
    for nch in (64, 128, 512):
        t0 = time.perf_counter()
        ctx = bench.make_context(pkg, nch, 8192, 0, 65536)
        if nch >= 384:
            ctx.set_overlap(2)
        d_in, d_out = ctx.alloc(nch, 8192), ctx.alloc(nch, 8192)
        d_in.upload(bench.synth_block(nch, 8192, 192000))
        ctx.process_device(d_in, d_out, 8192, 192000)
        ctx.synchronize()
        t_setup = time.perf_counter() - t0
        st = bench.robust_time(lambda: [ctx.process_device(d_in, d_out, 8192, 192000) for _ in range(30)], ctx.synchronize, units=30)
        W = 16
        ctx.set_window(W)
        n = 2 * W * 8192
        w_in, w_out = ctx.alloc(nch, n), ctx.alloc(nch, n)
        w_in.upload(np.tile(bench.synth_block(nch, 8192, 192000), (1, 2 * W)))
        def run():
            for b in range(0, 2 * W, W):
                ctx.process_window_device(w_in.ptr + 8 * b * 8192, w_out.ptr + 8 * b * 8192, n, W, 192000)
        sw = bench.robust_time(run, ctx.synchronize, units=2 * W, reps=3)
        t0 = time.perf_counter()
        ctx.close()
        t_close = time.perf_counter() - t0
        out.append("%d ch: %.1f us/step (min %.1f max %.1f) | W=16 %.1f us/frame | setup %.0f ms close %.0f ms"
                   % (nch, st["median"] * 1e6, st["min"] * 1e6, st["max"] * 1e6, sw["median"] * 1e6, t_setup * 1e3, t_close * 1e3))
    print("   " + "\n   ".join(out), flush=True)

if len(sys.argv) > 1 and sys.argv[1] == "child":
    child()
else:
    for env in ({"GDG_ARENA": "0"}, {"GDG_ARENA_ALIGN": "256"}, {"GDG_ARENA_ALIGN": "4096"}, {"GDG_ARENA_ALIGN": "65536"}, {"GDG_ARENA_ALIGN": "2097152"}):
        print(env, flush=True)
        e = dict(os.environ); e.update(env)
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=e, check=False)
