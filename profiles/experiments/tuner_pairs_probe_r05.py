#!/usr/bin/env python3
"""gdg_tuner_analyze per call at the C boundary for 32 .. 256 channels at 192 kHz, full rings: one block at a time (256 threads) against two
blocks at a time (512 threads), by parts per channel."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, __graft_entry__ as entry
from helpers import synth_signal
pkg = entry.load_package()
sr, frames = 192000, 8192
for nch in (32, 64, 128, 256, 512):
    ctx = pkg.Context(nch, frames)
    x = np.stack([synth_signal(c, 13 * frames, sr) for c in range(nch)])
    for b in range(13): ctx.tuner_enqueue(x[:, b * frames:(b + 1) * frames], sr)
    line = "%4d channels:" % nch
    for pairs in (0, 1):
        ctx.set_option("tuner_pairs", pairs)
        for parts in ((0, 1, 2, 3, 4, 6, 8) if nch < 256 else (0, 1, 2)):
            ctx.set_option("tuner_parts", parts)
            for _ in range(3): ctx.tuner_analyze(raw=True)
            ts = []
            for _ in range(5):
                t0 = time.perf_counter()
                for _ in range(10): ctx.tuner_analyze(raw=True)
                ts.append((time.perf_counter() - t0) / 10)
            line += "  %s/%d %6.1f" % ("pairs" if pairs else "one", parts, sorted(ts)[2] * 1e6)
    ctx.set_option("tuner_parts", 0); ctx.set_option("tuner_pairs", 1)
    print(line + "   (us per analysis call)", flush=True)
    ctx.close()
