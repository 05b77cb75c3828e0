#!/usr/bin/env python3
"""Do a 64-channel segment launch (64 workgroups of 1024 threads) and a 64-channel bin-tiled multiply-accumulate share the chip?  Two contexts on
one GPU, each on its own stream, no dependencies: A runs the segments of the bench chain only, B two power amps only."""
import os, sys, time, threading
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import bench
import __graft_entry__ as entry
pkg = entry.load_package()
nch, frames, sr, taps = 64, 8192, 192000, 65536
seg_chain = [(n, p) for n, p in bench.CHAIN if not isinstance(p, str)]
fir_chain = [(n, p) for n, p in bench.CHAIN if isinstance(p, str)]
A = bench.make_context(pkg, nch, frames, 0, taps, chain=seg_chain)
B = bench.make_context(pkg, nch, frames, 0, taps, chain=fir_chain)
bufs = {}
for name, ctx in (("A", A), ("B", B)):
    d_in, d_out = ctx.alloc(nch, frames), ctx.alloc(nch, frames)
    d_in.upload(bench.synth_block(nch, frames, sr))
    bufs[name] = (d_in, d_out)
def run(ctx, name, n):
    d_in, d_out = bufs[name]
    for _ in range(n):
        ctx.process_device(d_in, d_out, frames, sr)
def timed(fn, sync, n):
    fn(); sync()
    t0 = time.perf_counter(); fn(); sync()
    return (time.perf_counter() - t0) / n * 1e6
N = 40
ta = timed(lambda: run(A, "A", N), A.synchronize, N)
tb = timed(lambda: run(B, "B", N), B.synchronize, N)
def both():
    for _ in range(N):
        A.process_device(*bufs["A"], frames, sr)
        B.process_device(*bufs["B"], frames, sr)
tab = timed(both, lambda: (A.synchronize(), B.synchronize()), N)
print("segments alone %.1f us/step, power amps alone %.1f us/step, both interleaved on two streams %.1f us/step (sum %.1f)" % (ta, tb, tab, ta + tb))
A.close(); B.close()
