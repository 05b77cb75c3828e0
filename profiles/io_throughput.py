"""Device throughput of the section-8f kernels (wave codecs, resample.Time, level meters), timed with the library's
own HIP-event profiler (gdg_profile_*).  Algorithmic bytes: codecs = width + 8 B per sample; resample = 8 B read
(every source sample once, neighbours hit in cache) + 8 B written per output sample at 1:1, reported per output
sample; meters = 8 B per sample.   python profiles/io_throughput.py > gpurun_out/io_throughput.txt"""
import importlib.util
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
C = __import__("ctypes")
lib = pkg.lib()
ctx = pkg.Context(1, 8192)
REPS = 10


def timed(kind, fn):
    fn()
    ctx.synchronize()
    ctx.profile_enable(True)
    ctx.profile_read(kind)
    for _ in range(REPS):
        fn()
    ms, n = ctx.profile_read(kind)
    ctx.profile_enable(False)
    return ms / n


N = 1 << 27                                        # 128 Mi samples = 1 GiB of float64
d_samples = ctx.alloc(1, N)
d_bytes = ctx.alloc(1, N)                          # 8 B per sample is enough for every format
rng = np.random.default_rng(0)
d_samples.upload(np.tile(rng.uniform(-1.2, 1.2, 1 << 20), N >> 20))
print("kernel,format,samples,us,GB/s(algorithmic),frac_of_8TB/s")
for name, f in pkg.WAVE_FORMATS.items():
    w = lib.gdg_wave_bytes_per_sample(f)
    t = timed(pkg.K_WAVE, lambda: ctx._check(lib.gdg_wave_encode_device(ctx._h, f, d_samples.ptr, N, 1, d_bytes.ptr)))
    gb = N * (8 + w) / t / 1e6
    print("wave_encode,%s,%d,%.1f,%.0f,%.3f" % (name, N, t * 1e3, gb, gb / 8000))
    t = timed(pkg.K_WAVE, lambda: ctx._check(lib.gdg_wave_decode_device(ctx._h, f, d_bytes.ptr, N, 1, d_samples.ptr)))
    gb = N * (8 + w) / t / 1e6
    print("wave_decode,%s,%d,%.1f,%.0f,%.3f" % (name, N, t * 1e3, gb, gb / 8000))
    d_samples.upload(np.tile(rng.uniform(-1.2, 1.2, 1 << 20), N >> 20))

for src, dst in ((44100, 96000), (96000, 44100), (48000, 192000)):
    n_in = 1 << 24
    n_out = lib.gdg_resample_time_length(n_in, src, dst)
    if n_out > N:
        n_in = int(N * src / dst) - 8
        n_out = lib.gdg_resample_time_length(n_in, src, dst)
    t = timed(pkg.K_RESAMPLE, lambda: ctx._check(lib.gdg_resample_time_device(ctx._h, d_samples.ptr, n_in, src, dst, d_bytes.ptr, n_out)))
    gb = (n_in + n_out) * 8 / t / 1e6
    print("resample_time,%d->%d,%d,%.1f,%.0f,%.3f  (%.0f Msamples/s out)" % (src, dst, n_out, t * 1e3, gb, gb / 8000, n_out / t / 1e3))

ports, frames = 2 * 512 + 3, 8192                 # the reference's 2N+3 ports at N = 512
ctx.meter_configure(ports)
ctx.meter_set_enabled(True)
t = timed(pkg.K_METER, lambda: ctx.meter_process_device(d_samples, frames, frames, 192000))
gb = ports * frames * 8 / t / 1e6
print("meter,%d ports x %d,%d,%.1f,%.0f,%.3f" % (ports, frames, ports * frames, t * 1e3, gb, gb / 8000))
