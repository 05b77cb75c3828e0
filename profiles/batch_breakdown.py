"""Where gdg_batch_run's wall time goes (512 channels, 16 blocks, lpcm16 in / lpcm24 out): wall time vs the chain's own share."""
import sys, time
import numpy as np
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from importlib import import_module
pkg = import_module("go-dsp-guitar_amd")
nch, sr, frames, taps = 512, 192000, 8192, 65536
ctx = bench.make_context(pkg, nch, frames, 0, taps)
for blocks, W in ((4, 1), (16, 1), (64, 1), (16, 8), (64, 8), (32, 16), (64, 16), (128, 16)):
    ctx.set_window(W)
    n = blocks * frames
    rng = np.random.default_rng(5)
    files = [(rng.integers(-20000, 20000, n, dtype=np.int16).view(np.uint8), "lpcm16", sr) for _ in range(nch)]
    for fmt in ("lpcm24", "ieee64"):
        outs = ctx.batch_run(files, sr, fmt)
        t0 = time.perf_counter(); ctx.batch_run(files, sr, fmt, outs=outs); dt = time.perf_counter() - t0
        hb = sum(f[0].nbytes for f in files) + sum(o.nbytes for o in outs)
        print("W %d blocks %3d out %-6s: %8.2f ms  %7.1f Msamples/s  host bytes %6.1f MB  -> %5.1f GB/s if all PCIe" % (W, blocks, fmt, dt * 1e3, nch * n / dt / 1e6, hb / 1e6, hb / dt / 1e9))
ctx.set_window(1)
d_x = ctx.alloc(nch, frames); d_y = ctx.alloc(nch, frames)
d_x.upload(bench.synth_block(nch, frames, sr))
for _ in range(3): ctx.process_device(d_x, d_y, frames, sr)
ctx.synchronize(); t0 = time.perf_counter()
for _ in range(16): ctx.process_device(d_x, d_y, frames, sr)
ctx.synchronize(); print("16 x chain only: %.2f ms" % ((time.perf_counter() - t0) * 1e3))
