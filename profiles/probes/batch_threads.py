#!/usr/bin/env python3
"""gdg_batch_run (512 x 16-bit files of 128 blocks -> 515 x 24-bit files, W = 16) by number of copy threads: one child per setting."""
import os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
def child():
    import bench
    import __graft_entry__ as entry
    pkg = entry.load_package()
    ctx = bench.make_context(pkg, 512, 8192, 0, 65536)
    r = bench.batch_run(pkg, ctx, 512, 192000)
    print("   W=1 %.1f ms (%.0f Msamples/s) | W=16 %.1f ms (%.0f Msamples/s, min %.1f max %.1f)" % (
        r["window_1"]["ms"], r["window_1"]["value"], r["window_16"]["ms"], r["window_16"]["value"], *r["window_16"]["ms_min_max"]), flush=True)
    ctx.close()
if len(sys.argv) > 1 and sys.argv[1] == "child":
    child()
else:
    for t in ("4", "8", "12", "16", "24", "32"):
        print({"GDG_COPY_THREADS": t}, flush=True)
        e = dict(os.environ); e["GDG_COPY_THREADS"] = t
        subprocess.run([sys.executable, os.path.abspath(__file__), "child"], env=e, check=False)
