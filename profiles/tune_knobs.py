import os, sys, time, ctypes as C
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import __graft_entry__ as ge, bench
pkg = ge.load_package()
lib = pkg.lib()
what = sys.argv[1]
frames, sr, taps = 8192, 192000, 65536
if what == "split":
    for nch in (96, 112, 128, 160, 192):
        dt = bench.leg_on_one_gpu(pkg, nch, frames, sr, taps, 0, 30)
        print("channels %d: %.1f us per step" % (nch, dt * 1e6))
else:
    nch = 512
    ctx = bench.make_context(pkg, nch, frames, 0, taps, n_distinct=8)
    r = bench.end_to_end(pkg, ctx, nch, frames, sr, steps=8)
    print({k: (round(v["ms_per_block"], 3) if isinstance(v, dict) else None) for k, v in r.items()})
