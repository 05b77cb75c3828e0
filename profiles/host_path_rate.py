"""PCIe-inclusive rate of the boundary's host-buffer entry points on the bench workload (never the headline value):
gdg_process (caller's pageable rows -> pinned staging -> HBM -> chain -> back) and gdg_process_staged (caller already
wrote the pinned slab).   python profiles/host_path_rate.py > gpurun_out/host_path_rate_r01.txt"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
nch, frames, sr, taps, steps = 512, 8192, 192000, 65536, 10
ctx = pkg.Context(nch, frames)
irs = {"cab": [bench.synth_ir(taps, 4242 + i) for i in range(8)], "rev": [bench.synth_ir(taps, 5242 + i) for i in range(8)]}
for c in range(nch):
    for name, p in bench.CHAIN:
        if isinstance(p, str):
            ctx.append_unit(c, name, fir=irs[p][c % 8])
        else:
            ctx.append_unit(c, name, params=p)
x = bench.synth_block(nch, frames, sr)
chans = list(range(nch))
print("entry point,ms_per_block,Msamples/s,GB/s over PCIe (in+out)")
import ctypes as C  # noqa: E402

lib = pkg.lib()
out = np.empty_like(x)
ins = (C.c_void_p * nch)(*[x[c].ctypes.data for c in range(nch)])
outs = (C.c_void_p * nch)(*[out[c].ctypes.data for c in range(nch)])
carr = (C.c_int * nch)(*chans)
ctx.process_staged(chans, x, sr)                      # fills the pinned input slab once (the Go workers do that in parallel)
for name, fn in (("gdg_process (pageable rows; C call only)", lambda: ctx._check(lib.gdg_process(ctx._h, ins, outs, frames, sr))),
                 ("gdg_process_staged (pinned slab already written; C call only)", lambda: ctx._check(lib.gdg_process_staged(ctx._h, carr, nch, frames, sr)))):
    for _ in range(2):
        fn()
    t0 = time.perf_counter()
    for _ in range(steps):
        fn()
    dt = (time.perf_counter() - t0) / steps
    print("%s,%.3f,%.0f,%.1f" % (name, dt * 1e3, nch * frames / dt / 1e6, 2 * nch * frames * 8 / dt / 1e9))
