"""BASELINE config 5 on one GPU: 256 tuners (96000-sample windows, 262144-point autocorrelation each) and the spatializer's
256 -> 2 mixdown at 192 kHz, device-resident frames.   python profiles/config5_rates.py > gpurun_out/config5_rates_r01.txt"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge  # noqa: E402
from helpers import synth_signal  # noqa: E402

pkg = ge.load_package()
nch, frames, sr = 256, 8192, 192000
ctx = pkg.Context(nch, frames)
x = np.stack([synth_signal(c, frames, sr) for c in range(nch)])
d_x = ctx.alloc(nch, frames)
d_x.upload(x)
d_lr = ctx.alloc(2, frames)
ctx.spatializer_set_sample_rate(sr)
for c in range(nch):
    ctx.spatializer_set_position(c, -90.0 + 180.0 * c / (nch - 1), 0.5 + 0.01 * c, 0.5)
for _ in range(13):
    ctx.tuner_enqueue_device(d_x, frames, sr)
ctx.spatialize_device(d_x, d_lr, frames)
ctx.tuner_analyze()
ctx.synchronize()
ctx.profile_enable(True)
n = 20
t0 = time.perf_counter()
for _ in range(n):
    ctx.spatialize_device(d_x, d_lr, frames)
ctx.synchronize()
t_sp = (time.perf_counter() - t0) / n
t0 = time.perf_counter()
for _ in range(n):
    ctx.tuner_enqueue_device(d_x, frames, sr)
ctx.synchronize()
t_enq = (time.perf_counter() - t0) / n
t0 = time.perf_counter()
for _ in range(5):
    ctx.tuner_analyze(raw=True)
t_an = (time.perf_counter() - t0) / 5
ms_sp, n_sp = ctx.profile_read(pkg.K_SPATIALIZER)
ms_tu, n_tu = ctx.profile_read(pkg.K_TUNER)
print("spatializer 256 -> 2, 8192 frames: %.1f us per block wall, %.1f us kernel  (%.0f Msamples/s in, %.0f GB/s of input)" %
      (t_sp * 1e6, ms_sp / max(n_sp, 1) * 1e3, nch * frames / t_sp / 1e6, nch * frames * 8 / t_sp / 1e9))
print("tuner enqueue 256 x 8192: %.1f us per block" % (t_enq * 1e6))
print("tuner analyze 256 channels (262144-point autocorrelation each): %.2f ms wall, kernels %.2f ms per analysis  (%.0f analyses/s)" %
      (t_an * 1e3, ms_tu / 5, nch / t_an))
