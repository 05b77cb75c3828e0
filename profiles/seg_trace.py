import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import __graft_entry__ as ge, bench
pkg = ge.load_package()
nch, frames, sr, taps = 512, 8192, 192000, 65536
ctx = bench.make_context(pkg, nch, frames, 0, taps, n_distinct=8)
d_in, d_out = ctx.alloc(nch, frames), ctx.alloc(nch, frames)
d_in.upload(bench.synth_block(nch, frames, sr))
for _ in range(6):
    ctx.process_device(d_in, d_out, frames, sr)
ctx.synchronize()
buf = np.zeros(1024 * 16, dtype=np.uint64)
pkg.lib().gdg_debug_seg_trace(buf.ctypes.data_as(C.c_void_p))
t = buf.reshape(1024, 16)[:nch].astype(np.int64)
n = int(t[0, 15])
print("stamps per WG:", n, "(the LAST segment launch of the step = seg1: cabinet, reverb)")
d = np.diff(t[:, :n], axis=1)
names = ["load frame"] + ["unit %d" % i for i in range(n - 3)] + ["store frame"]
start = t[:, 0] - t[:, 0].min()
print("WG start offsets (cycles): median %.0f  p90 %.0f  max %.0f" % (np.median(start), np.percentile(start, 90), start.max()))
for i, nm in enumerate(names):
    print("%-12s mean %8.0f  median %8.0f  p90 %8.0f cycles" % (nm, d[:, i].mean(), np.median(d[:, i]), np.percentile(d[:, i], 90)))
print("total per WG: mean %.0f cycles; kernel span %.0f cycles" % ((t[:, n - 1] - t[:, 0]).mean(), t[:, n - 1].max() - t[:, 0].min()))
r = buf.reshape(1024, 16)[:nch, 8:15].astype(np.int64)
dr = np.diff(r, axis=1)
for i, nm in enumerate(["fetch heads + taps (to own loads home)", "barrier wait", "all-pass 1", "all-pass 2 + 3", "mix", "ring append"]):
    print("reverb: %-40s mean %8.0f  median %8.0f  p90 %8.0f" % (nm, dr[:, i].mean(), np.median(dr[:, i]), np.percentile(dr[:, i], 90)))
