"""One GPU, the bench chain (2 x 64k-tap per-channel IRs, d = 1) at 8 .. 512 channels: the predicted per-GPU leg of the
strong split of BASELINE config 4 (512 channels over N GPUs = 512/N channels per GPU, controller.go:3262-3269).

    python profiles/channels_sweep.py [--channels 8,32,64,...] > gpurun_out/channels_sweep_r02.txt
"""
import argparse
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as ge  # noqa: E402
import bench  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--channels", default="8,32,64,128,256,512")
ap.add_argument("--taps", type=int, default=65536)
ap.add_argument("--steps", type=int, default=30)
args = ap.parse_args()

pkg = ge.load_package()
frames, sr, taps = 8192, 192000, args.taps
print("# bench chain, %d-tap per-channel IRs x 2, %d-frame blocks @ %d Hz, input resident in HBM; env GDG_FIR_SPLIT=%s"
      % (taps, frames, sr, os.environ.get("GDG_FIR_SPLIT", "(default)")))
print("channels,us_per_step,Msamples_s,realtime_x_at_512ch,fir_fwd_us,fir_mac_us,fir_inv_us,segment_us")
for nch in [int(c) for c in args.channels.split(",")]:
    ctx = pkg.Context(nch, frames)
    for c in range(nch):
        for name, p in bench.CHAIN:
            if isinstance(p, str):
                ctx.append_unit(c, name, fir=bench.synth_ir(taps, (4242 if p == "cab" else 5242) + c))
            else:
                ctx.append_unit(c, name, params=p)
    d_in, d_out = ctx.alloc(nch, frames), ctx.alloc(nch, frames)
    d_in.upload(bench.synth_block(nch, frames, sr))
    for _ in range(5):
        ctx.process_device(d_in, d_out, frames, sr)
    ctx.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        ctx.process_device(d_in, d_out, frames, sr)
    ctx.synchronize()
    us = (time.perf_counter() - t0) / args.steps * 1e6
    ctx.profile_enable(True)
    for _ in range(10):
        ctx.process_device(d_in, d_out, frames, sr)
    ctx.synchronize()
    ctx.profile_enable(False)
    per = []
    for kind in range(4):
        ms, n = ctx.profile_read(kind)
        per.append("%.1f" % (ms / n * 1e3) if n else "")
    rate = nch * frames / us          # Msamples/s
    # 512 channels split over 512/nch GPUs, every GPU doing this step: real-time factor of the whole job
    print("%d,%.1f,%.1f,%.1f,%s" % (nch, us, rate, frames / sr / (us * 1e-6), ",".join(per)))
    sys.stdout.flush()
    ctx.close()
