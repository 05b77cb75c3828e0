#!/usr/bin/env python3
"""gdg_tuner_analyze per call at the C boundary (raw result structs, no Python post-processing) for 32 .. 256 channels at 192 kHz, full rings."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, __graft_entry__ as entry
from helpers import synth_signal
pkg = entry.load_package()
sr, frames = 192000, 8192
for nch in (32, 64, 128, 256):
    ctx = pkg.Context(nch, frames)
    x = np.stack([synth_signal(c, 13 * frames, sr) for c in range(nch)])
    for b in range(13): ctx.tuner_enqueue(x[:, b * frames:(b + 1) * frames], sr)
    for _ in range(3): ctx.tuner_analyze(raw=True)
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        for _ in range(10): ctx.tuner_analyze(raw=True)
        ts.append((time.perf_counter() - t0) / 10)
    t = sorted(ts)[2]
    print("%4d channels: %6.1f us per analysis call = %8.0f analyses/s on this GPU (parts per channel: %d)" % (nch, t * 1e6, nch / t, pkg.lib().gdg_tuner_short_parts(nch) if hasattr(pkg.lib(), "gdg_tuner_short_parts") else -1), flush=True)
    ctx.close()
