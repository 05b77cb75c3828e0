#!/usr/bin/env python3
"""Where a gdg_batch_run goes (512 x 16-bit files of 128 blocks -> 515 x 24-bit files, W = 16): wall time vs the HIP-event time per kernel kind."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench
import __graft_entry__ as entry
pkg = entry.load_package()
nch, sr, blocks = 512, 192000, 128
ctx = bench.make_context(pkg, nch, 8192, 0, 65536)
ctx.set_window(16)
if os.environ.get("NGROUPS"): ctx.set_overlap(int(os.environ["NGROUPS"]))
files = bench.batch_files(nch, sr, blocks)
outs = ctx.batch_run(files, sr, "lpcm24")
for rep in range(2):
    t0 = time.perf_counter(); ctx.batch_run(files, sr, "lpcm24", outs=outs); t = time.perf_counter() - t0
    print("plain run: %.1f ms" % (t * 1e3), flush=True)
ctx.profile_enable(True)
t0 = time.perf_counter(); ctx.batch_run(files, sr, "lpcm24", outs=outs); t = time.perf_counter() - t0
ctx.profile_enable(False)
names = ["fir_fwd", "fir_mac", "fir_inv", "segment", "tuner", "spatializer", "wave", "resample", "meter", "fir_mac_chain"]
tot = 0.0
line = "profiled run: %.1f ms |" % (t * 1e3)
for k, name in enumerate(names):
    ms, n = ctx.profile_read(k)
    if n:
        line += " %s %.2f ms (x%d)" % (name, ms, n); tot += ms
print(line + " | sum of kinds on the compute stream %.1f ms" % tot, flush=True)
ctx.close()
