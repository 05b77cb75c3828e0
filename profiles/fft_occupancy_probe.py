import os, sys
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
import __graft_entry__ as ge
from helpers import synth_ir
pkg = ge.load_package()
for nch, frames in ((512, 8192), (1024, 4096), (2048, 2048)):
    ctx = pkg.Context(nch, frames)
    ir = synth_ir(frames * 2)
    for c in range(nch):
        ctx.append_unit(c, "power_amp", fir=ir)
    d_in, d_out = ctx.alloc(nch, frames), ctx.alloc(nch, frames)
    d_in.upload(np.random.default_rng(0).uniform(-0.5, 0.5, (nch, frames)))
    for _ in range(3): ctx.process_device(d_in, d_out, frames, 96000)
    ctx.synchronize(); ctx.profile_enable(True)
    for _ in range(10): ctx.process_device(d_in, d_out, frames, 96000)
    ctx.synchronize()
    r = {k: ctx.profile_read(getattr(pkg, "K_FIR_" + k)) for k in ("FWD", "MAC", "INV")}
    print(nch, frames, {k: round(v[0] / v[1] * 1e3, 1) for k, v in r.items()})
    ctx.close()
