#!/usr/bin/env python3
"""level meters over the 2 N + 3 ports of a 512-channel job and the metronome, per 8192-frame block (HIP events),
with the oracle's recurrences on one host core beside them.   python profiles/meter_rate.py > gpurun_out/meter_rate.txt"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as entry
pkg = entry.load_package()
oracle = entry.load_oracle()
nch, frames, sr = 512, 8192, 192000
ports = 2 * nch + 3
ctx = pkg.Context(nch, frames)
ctx.meter_configure(ports)
ctx.meter_set_enabled(True)
rows = ctx.alloc(ports, frames)
x = np.random.default_rng(0).uniform(-0.9, 0.9, (ports, frames))
rows.upload(x)
for _ in range(3): ctx.meter_process_device(rows.ptr, frames, frames, sr)
ctx.synchronize(); ctx.profile_enable(True)
for _ in range(20): ctx.meter_process_device(rows.ptr, frames, frames, sr)
ctx.synchronize(); ms, n = ctx.profile_read(pkg.K_METER)
us = ms / n * 1e3
m = oracle.ChannelMeter() if hasattr(oracle, "ChannelMeter") else None
cpu = None
if m is not None:
    m.set_enabled(True)
    m.process(x[0], sr); t0 = time.perf_counter()
    for _ in range(8): m.process(x[0], sr)
    cpu = (time.perf_counter() - t0) / 8 * 1e6
print("level meters: %d ports x %d frames: %.1f us per block on the device = %.0f Msamples/s (%.2f TB/s of the 8 B per sample read)%s" % (
    ports, frames, us, ports * frames / us, ports * frames * 8 / us / 1e6, "" if cpu is None else "; oracle, one port on one core: %.0f us" % cpu))
tick, tock = np.random.default_rng(1).uniform(-0.5, 0.5, 2000), np.random.default_rng(2).uniform(-0.5, 0.5, 1500)
ctx.metronome_set_sounds(tick, tock); ctx.metronome_configure(4, 120, sr)
for _ in range(3): ctx.metronome_process(frames)
t0 = time.perf_counter()
for _ in range(50): ctx.metronome_process(frames)
print("metronome: %d frames per call incl. the download of the block: %.1f us" % (frames, (time.perf_counter() - t0) / 50 * 1e6))
ctx.close()
