"""Per-block latency of the device-resident path on the small configurations (launch-bound):
python profiles/small_config_latency.py > gpurun_out/small_config_latency_r01.txt"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge  # noqa: E402
from helpers import synth_ir, synth_signal  # noqa: E402

pkg = ge.load_package()
print("config,channels,frames,sample_rate,taps,us_per_block,realtime_factor")
for name, nch, frames, sr, taps, two in (("config2", 1, 1024, 48000, 8192, False), ("config2b", 1, 8192, 48000, 8192, False),
                                         ("8ch", 8, 1024, 48000, 8192, False), ("config3", 64, 8192, 96000, 32768, False),
                                         ("config4", 512, 8192, 192000, 65536, True)):
    ctx = pkg.Context(nch, frames)
    for c in range(nch):
        ctx.append_unit(c, "compressor", params=[1, 30, -20])
        ctx.append_unit(c, "overdrive", params=[0, 20, 100, 0, 1, 2 if name == "config3" else 0])
        ctx.append_unit(c, "tone_stack")
        ctx.append_unit(c, "chorus")
        ctx.append_unit(c, "power_amp", fir=synth_ir(taps, seed=c % 8))
        if two:
            ctx.append_unit(c, "power_amp", fir=synth_ir(taps, seed=100 + c % 8))
        ctx.append_unit(c, "cabinet")
        ctx.append_unit(c, "reverb", params=[50])
    d_in, d_out = ctx.alloc(nch, frames), ctx.alloc(nch, frames)
    d_in.upload(np.stack([synth_signal(c, frames, sr) for c in range(nch)]))
    for _ in range(5):
        ctx.process_device(d_in, d_out, frames, sr)
    ctx.synchronize()
    n = 200 if nch < 64 else 30
    t0 = time.perf_counter()
    for _ in range(n):
        ctx.process_device(d_in, d_out, frames, sr)
    ctx.synchronize()
    us = (time.perf_counter() - t0) / n * 1e6
    print("%s,%d,%d,%d,%d,%.1f,%.1f" % (name, nch, frames, sr, taps, us, frames / sr / (us * 1e-6)))
    ctx.close()
