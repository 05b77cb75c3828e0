#!/usr/bin/env python3
"""(Env NGROUPS, not GROUPS: bash treats GROUPS as a special variable and drops the assignment.)
How the per-frame step time settles after set-up: the bench context, steps timed in batches of 5 (synchronised per batch)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np, torch
import bench
import __graft_entry__ as entry
pkg = entry.load_package()
nch, frames, sr = 512, 8192, 192000
ctx = bench.make_context(pkg, nch, frames, 0, 65536)
ctx.set_overlap(int(os.environ.get("NGROUPS", "2")))
x = torch.from_numpy(bench.synth_block(nch, frames, sr)).cuda(); y = torch.empty_like(x)
ctx.process_device(x.data_ptr(), y.data_ptr(), frames, sr); ctx.synchronize()       # plan + spectra
if os.environ.get("IDLE"): time.sleep(float(os.environ["IDLE"]))
if os.environ.get("EVENTS"):          # record HIP events on a few steps first, then switch them off: does the stream stay slower?
    ctx.profile_enable(kinds=[pkg.K_FIR_MAC])
    for _ in range(int(os.environ["EVENTS"])):
        ctx.process_device(x.data_ptr(), y.data_ptr(), frames, sr)
    ctx.synchronize()
    ctx.profile_enable(False)
out = []
PER = int(os.environ.get("PER", "5"))
for b in range(80 // PER):
    t0 = time.perf_counter()
    for _ in range(PER):
        ctx.process_device(x.data_ptr(), y.data_ptr(), frames, sr)
    t1 = time.perf_counter()
    ctx.synchronize()
    if os.environ.get("TORCH_SYNC"): torch.cuda.synchronize()
    out.append(((time.perf_counter() - t0) / PER * 1e6, (t1 - t0) / PER * 1e6))
print("us/step per batch of %d (wall | host enqueue):" % PER, " ".join("%.0f|%.0f" % v for v in out), flush=True)
ctx.close()
