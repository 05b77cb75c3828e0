#!/usr/bin/env python3
"""Why does bench.py's one-group headline run 10 % slower than the same steps in ramp_probe.py?  The bench sequence, features toggled by env."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
import __graft_entry__ as entry
pkg = entry.load_package()
nch, frames, sr = 512, 8192, 192000
ctx = bench.make_context(pkg, nch, frames, 0, 65536)
ctx.set_overlap(int(os.environ.get("NGROUPS", "1")))
dev = torch.device("cuda", 0)
x = torch.from_numpy(bench.synth_block(nch, frames, sr)).to(dev); y = torch.empty_like(x)
def step(): ctx.process_device(x.data_ptr(), y.data_ptr(), frames, sr)
def sync():
    ctx.synchronize()
    if os.environ.get("TORCH_SYNC", "1") == "1": torch.cuda.synchronize()
for _ in range(5): step()
ctx.synchronize()
mode = os.environ.get("MODE", "none")
if mode in ("sample", "sample_enable"): ctx.profile_sample(4)
if mode in ("enable", "sample_enable"): ctx.profile_enable(kinds=[pkg.K_FIR_MAC])
def timed():
    sync(); t0 = time.perf_counter()
    for _ in range(20): step()
    sync(); return (time.perf_counter() - t0) / 20 * 1e6
r = [timed()]
ctx.profile_enable(False); ctx.profile_sample(1)
r += [timed() for _ in range(4)]
print("NGROUPS=%s MODE=%s TORCH_SYNC=%s:" % (os.environ.get("NGROUPS", "1"), mode, os.environ.get("TORCH_SYNC", "1")), " ".join("%.0f" % v for v in r), flush=True)
