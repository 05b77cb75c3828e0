#!/usr/bin/env python3
"""Instrumented build only (STAMP macros in seg.hip, gdg_debug_stamps): cycle stamps of one unit, waves 0 / 7 / 15 of workgroups 0 and 1."""
import sys, ctypes, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, __graft_entry__ as e
pkg = e.load_package()
nch = int(os.environ.get("NCH", "512"))
chain = sys.argv[1:] or ["compressor"]
ctx = pkg.Context(nch, 8192)
for c in range(nch):
    for u in chain: ctx.append_unit(c, u)
a, b = ctx.alloc(nch, 8192), ctx.alloc(nch, 8192)
a.upload(np.random.default_rng(0).uniform(-0.5, 0.5, (nch, 8192)))
for _ in range(5): ctx.process_device(a, b, 8192, 192000)
ctx.synchronize()
buf = (ctypes.c_ulonglong * (64 * 64))()
pkg.lib().gdg_debug_stamps(buf)
st = np.array(buf[:], dtype=np.int64)[:1024].reshape(4, 16, 16)
for blk in range(2):
    for w in (0, 7, 15):
        r = st[blk, w]
        print("wg %d wave %2d:" % (blk, w), " ".join("%6d" % (r[i] - r[0]) for i in range(1, 8)))
