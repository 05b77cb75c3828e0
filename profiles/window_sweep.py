"""Time blocking: the bench chain (512 channels, 2 x 65536 taps, 192 kHz) over whole 'files' of 32 frames resident in HBM, in windows of
W frames per call.  us per 8192-frame block, Msamples/s, and the HIP-event time of the four kernel kinds per block."""
import sys, time
import numpy as np
import os; sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import bench
from importlib import import_module
pkg = import_module("go-dsp-guitar_amd")
nch = int(os.environ.get("SWEEP_CHANNELS", "512"))
sr, frames, taps, blocks = 192000, 8192, 65536, 32
x = np.tile(bench.synth_block(nch, frames, sr), (1, blocks))
print("W,us_per_block,Msamples_s,realtime_x,fir_fwd_us,fir_mac_us,fir_inv_us,segment_us")
for W in ([int(a) for a in sys.argv[1:]] or (1, 2, 4, 8, 16)):
    ctx = bench.make_context(pkg, nch, frames, 0, taps)
    ctx.set_window(W)
    d_in, d_out = ctx.alloc(nch, blocks * frames), ctx.alloc(nch, blocks * frames)
    d_in.upload(x)
    def run():
        for b in range(0, blocks, W):
            ctx.process_window_device(d_in.ptr + 8 * b * frames, d_out.ptr + 8 * b * frames, blocks * frames, W, sr)
    run(); ctx.synchronize()
    t0 = time.perf_counter(); run(); ctx.synchronize(); dt = (time.perf_counter() - t0) / blocks
    ctx.profile_enable(True); run(); ctx.synchronize(); ctx.profile_enable(False)
    k = [ctx.profile_read(i)[0] * 1e3 / blocks for i in range(4)]
    names = pkg.KERNEL_KINDS[:4]
    d = dict(zip(names, k))
    print("%d,%.1f,%.1f,%.1f,%.1f,%.1f,%.1f,%.1f" % (W, dt * 1e6, nch * frames / dt / 1e6, frames / sr / dt, d["fir_fwd"], d["fir_mac"], d["fir_inv"], d["segment"]))
    ctx.close()
