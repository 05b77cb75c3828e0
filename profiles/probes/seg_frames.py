#!/usr/bin/env python3
"""seg_kernel time per launch for a few one-unit chains at a given frame size (FRAMES, default 8192), 512 channels @ 192 kHz."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as entry
pkg = entry.load_package()
frames, nch, sr = int(os.environ.get("FRAMES", "8192")), int(os.environ.get("NCH", "512")), 192000
CASES = [("(copy)", []), ("compressor", [("compressor", None)]), ("tone_stack", [("tone_stack", None)]), ("chorus", [("chorus", None)]),
         ("cabinet", [("cabinet", None)]), ("reverb", [("reverb", None)]),
         ("seg0", [("compressor", None), ("overdrive", [0, 20, 100, 0, 1, 0]), ("tone_stack", None), ("chorus", None)]),
         ("seg1", [("cabinet", None), ("reverb", None)])]
x = np.random.default_rng(0).uniform(-0.5, 0.5, (nch, frames))
for name, chain in CASES:
    ctx = pkg.Context(nch, frames)
    for c in range(nch):
        for u, p in chain: ctx.append_unit(c, u, params=p)
    a, b = ctx.alloc(nch, frames), ctx.alloc(nch, frames)
    a.upload(x)
    for _ in range(3): ctx.process_device(a, b, frames, sr)
    ctx.synchronize(); ctx.profile_enable(True)
    for _ in range(10): ctx.process_device(a, b, frames, sr)
    ctx.synchronize(); ms, n = ctx.profile_read(pkg.K_SEGMENT)
    print("%-12s frames %5d: %7.1f us per launch, %7.2f ns per channel-sample x 1e3" % (name, frames, ms / n * 1e3, ms / n * 1e6 / (nch * frames) * 1e3), flush=True)
    ctx.close()
