#!/usr/bin/env python3
"""Round 3 probe: the 64-channel leg that ran 26-36x slow right after the 512-channel context was closed.
Reproduces bench.py's sequence -- large context, steps, window, close -> fresh 64-channel context -> per-frame steps -- and
prints every batch of 10 steps with the per-kernel-kind HIP-event times, plus free VRAM before / after."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import bench
import __graft_entry__ as entry
pkg = entry.load_package()

def mem(tag):
    free, total = torch.cuda.mem_get_info(0)
    print("  [%s] free VRAM %.2f GB of %.2f" % (tag, free / 1e9, total / 1e9), flush=True)

def small_leg(nch, tag, batches=8, per=10, sleep=0.0):
    t0 = time.perf_counter()
    ctx = bench.make_context(pkg, nch, 8192, 0, 65536)
    d_in, d_out = ctx.alloc(nch, 8192), ctx.alloc(nch, 8192)
    d_in.upload(bench.synth_block(nch, 8192, 192000))
    t1 = time.perf_counter()
    ctx.process_device(d_in, d_out, 8192, 192000)
    ctx.synchronize()
    t2 = time.perf_counter()
    print("  %s: create %.1f ms, first step (plan + spectra) %.1f ms" % (tag, (t1 - t0) * 1e3, (t2 - t1) * 1e3), flush=True)
    if sleep:
        time.sleep(sleep)
    for b in range(batches):
        prof = b in (0, batches - 1)
        if prof:
            ctx.profile_enable(True)
        t0 = time.perf_counter()
        for _ in range(per):
            ctx.process_device(d_in, d_out, 8192, 192000)
        ctx.synchronize()
        dt = (time.perf_counter() - t0) / per
        line = "  %s batch %d: %.1f us/step" % (tag, b, dt * 1e6)
        if prof:
            ctx.profile_enable(False)
            for kind, name in enumerate(pkg.KERNEL_KINDS[:4]):
                ms, n = ctx.profile_read(kind)
                line += "  %s %.1f us x%d" % (name, 1e3 * ms / max(n, 1), n)
        print(line, flush=True)
    ctx.close()

for rep in range(int(os.environ.get("REPS", "3"))):
    print("rep %d" % rep, flush=True)
    mem("start")
    big = bench.make_context(pkg, 512, 8192, 0, 65536)
    x = torch.from_numpy(bench.synth_block(512, 8192, 192000)).cuda()
    y = torch.empty_like(x)
    for _ in range(5):
        big.process_device(x.data_ptr(), y.data_ptr(), 8192, 192000)
    big.synchronize()
    if os.environ.get("BIG_EXTRAS", "1") == "1":
        bench.time_blocked(pkg, big, 512, 8192, 192000, blocks=16)
        bench.batch_run(pkg, big, 512, 192000, blocks=32)
    mem("big context live")
    t0 = time.perf_counter()
    big.close()
    del x, y
    print("  close of the big context: %.1f ms" % ((time.perf_counter() - t0) * 1e3), flush=True)
    mem("after close")
    small_leg(64, "64ch", sleep=(1.0 if rep == 2 else 0.0))
    mem("after 64ch leg")
    small_leg(128, "128ch", batches=3)
