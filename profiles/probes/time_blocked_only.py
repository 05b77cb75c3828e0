#!/usr/bin/env python3
"""bench.py's time_blocked leg alone (512 channels, the bench chain, W = 1 .. 16, two channel groups): us per frame per W."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
import __graft_entry__ as entry
pkg = entry.load_package()
nch, frames, sr = 512, 8192, 192000
ctx = bench.make_context(pkg, nch, frames, 0, 65536)
tb = bench.time_blocked(pkg, ctx, nch, frames, sr)
print(" ".join("W=%s %.1f" % (k.split("_")[1], v["us_per_frame"]) for k, v in tb.items() if k.startswith("window_")), flush=True)
ctx.close()
