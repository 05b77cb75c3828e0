#!/usr/bin/env python3
"""BASELINE config 5's per-GPU shape (32 tuners at 192 kHz, full rings): gdg_tuner_analyze per call at the C boundary by parts per channel
(option tuner_parts; 0 = the library's choice) and by how the caller learns that the results are there (option tuner_poll_results)."""
import os, sys, time
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, __graft_entry__ as entry
from helpers import synth_signal
pkg = entry.load_package()
sr, frames = 192000, 8192
for nch in [int(v) for v in os.environ.get("NCH_LIST", "32").split(",")]:
    ctx = pkg.Context(nch, frames)
    x = np.stack([synth_signal(c, 13 * frames, sr) for c in range(nch)])
    for b in range(13): ctx.tuner_enqueue(x[:, b * frames:(b + 1) * frames], sr)
    ref = None
    for parts in [int(v) for v in os.environ.get("PARTS_LIST", "0,4,6,8,12,24").split(",")]:
        for poll in (0, 1):
            ctx.set_option("tuner_parts", parts)
            ctx.set_option("tuner_poll_results", poll)
            for _ in range(3): res = ctx.tuner_analyze(raw=True)
            ts = []
            for _ in range(7):
                t0 = time.perf_counter()
                for _ in range(20): ctx.tuner_analyze(raw=True)
                ts.append((time.perf_counter() - t0) / 20)
            t = sorted(ts)[3]
            got = [(r.note_index, r.cents, r.frequency) for r in res] if hasattr(res[0], "note_index") else None
            print("%4d channels, parts %2d, poll %d: %6.1f us per analysis call (min %5.1f) = %8.0f analyses/s" % (nch, parts, poll, t * 1e6, min(ts) * 1e6, nch / t), flush=True)
    ctx.set_option("tuner_parts", 0)
    ctx.close()
