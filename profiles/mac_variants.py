#!/usr/bin/env python3
"""A/B of the fir_mac_kernel variants (GDG_MAC_VARIANT) on the bench's FIR geometry:
512 channels, P = 8192, L = 65536 (K = 8).  One process per variant (the knob is read once).

    python profiles/mac_variants.py            # driver: spawns one child per variant
"""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def child():
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import numpy as np
    import __graft_entry__ as entry
    from helpers import synth_ir, synth_signal
    pkg = entry.load_package()
    nch, frames, sr, taps = 512, 8192, 192000, 65536
    ctx = pkg.Context(nch, frames)
    h = synth_ir(taps)
    for c in range(nch):
        ctx.append_unit(c, "power_amp", fir=h)
    d_in, d_out = ctx.alloc(nch, frames), ctx.alloc(nch, frames)
    d_in.upload(np.stack([synth_signal(c % 8, frames, sr) for c in range(nch)]))
    for _ in range(10):
        ctx.process_device(d_in, d_out, frames, sr)
    ctx.synchronize()
    ctx.profile_enable(True)
    for _ in range(30):
        ctx.process_device(d_in, d_out, frames, sr)
    ctx.synchronize()
    ms, n = ctx.profile_read(pkg.K_FIR_MAC)
    us = ms / n * 1e3
    gbs = nch * 2 * 8 * frames * 16 / us / 1e3
    print("variant %s: fir_mac %.1f us  %.0f GB/s  %.1f%% of 8 TB/s" % (os.environ.get("GDG_MAC_VARIANT", "0"), us, gbs, gbs / 80.0))


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "child":
        child()
    else:
        for v in ([int(a) for a in sys.argv[1:]] or range(9)):
            env = dict(os.environ, GDG_MAC_VARIANT=str(v))
            subprocess.call([sys.executable, os.path.abspath(__file__), "child"], env=env)
