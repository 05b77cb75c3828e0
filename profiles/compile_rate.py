#!/usr/bin/env python3
"""one power-amp compile -- 8 impulse-response slots (reduce to the target order, normalise, scale, add; effects/poweramp.go:25-127) --
on the device (gdg_unit_compile_fir, taps uploaded in the call, composite left in the unit) against the oracle's filter algebra on one host core.
    python profiles/compile_rate.py > gpurun_out/compile_rate.txt"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import __graft_entry__ as entry
from helpers import synth_ir
from test_gpu_compile import oracle_compile
pkg = entry.load_package()
oracle = entry.load_oracle() if hasattr(entry, "load_oracle") else __import__("oracle")
print("%-44s %12s %12s %8s" % ("job (8 slots)", "device ms", "oracle ms", "ratio"))
for taps, order in ((8192, 8192), (65536, 65536), (65536, 8192), (200000, 65536)):
    filters = [(synth_ir(taps - 37 * k, seed=100 + k), 10.0 ** (0.05 * -(10 + k)), -k) for k in range(8)]
    ctx = pkg.Context(1, 1024)
    h = ctx.append_unit(0, "power_amp")
    ctx.unit_compile_fir(h, filters, order)                       # tables, buffers
    ctx.synchronize()
    ts = []
    for _ in range(5):
        t0 = time.perf_counter()
        ctx.unit_compile_fir(h, filters, order)
        ctx.synchronize()
        ts.append(time.perf_counter() - t0)
    t0 = time.perf_counter()
    oracle_compile(oracle, filters, order)
    t_cpu = time.perf_counter() - t0
    dev = sorted(ts)[len(ts) // 2]
    print("%-44s %12.2f %12.1f %8.0f" % ("%d-tap IRs -> order %d" % (taps, order), dev * 1e3, t_cpu * 1e3, t_cpu / dev), flush=True)
    ctx.close()
