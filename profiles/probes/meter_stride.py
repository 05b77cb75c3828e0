import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np, __graft_entry__ as entry
pkg = entry.load_package()
nch, frames, sr = 512, 8192, 192000
ports = 2 * nch + 3
for stride in (8192, 8192 + 32, 8192 + 264, 16 * 8192):
  for nports in (ports, 256, 512, 1024):
    ctx = pkg.Context(nch, frames)
    ctx.meter_configure(nports); ctx.meter_set_enabled(True)
    rows = ctx.alloc(nports, stride)
    rows.upload(np.random.default_rng(0).uniform(-0.9, 0.9, (nports, stride)))
    for _ in range(3): ctx.meter_process_device(rows.ptr, stride, frames, sr)
    ctx.synchronize(); ctx.profile_enable(True)
    for _ in range(20): ctx.meter_process_device(rows.ptr, stride, frames, sr)
    ctx.synchronize(); ms, n = ctx.profile_read(pkg.K_METER)
    print("stride %6d ports %5d: %.1f us" % (stride, nports, ms / n * 1e3), flush=True)
    ctx.close()
