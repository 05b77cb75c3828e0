#!/usr/bin/env python3
"""How far is "within 1e-9 RMS of the Go binary" from the libm each side calls?  

The oracle calls glibc, the reference Go's math package, the HIP path ocml: all within 1-2 ulp of one another for exp / sin / cos / atan /
pow / log10 / log2.  oracle/libgdg_oracle_jitter.so is the oracle with every such result moved by a seeded -2 .. +2 ulp (oracle/libm_jitter.h).
This script streams every unit of tests/test_gpu_parity.UNIT_CASES and SURVEY 8(d)'s full chain through BOTH oracles (CPU only) and, with
--gpu, the HIP path against the perturbed oracle, and prints the worst per-channel RMS per unit over the seeds.

    python profiles/libm_sensitivity.py [--gpu] [--seeds 3] > profiles/libm_sensitivity_r05.txt
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import __graft_entry__ as entry  # noqa: E402
from helpers import synth_signal, synth_ir, rms  # noqa: E402
from test_gpu_parity import UNIT_CASES  # noqa: E402

RATES = [(48000, 1024, 6), (192000, 8192, 3)]


def stream(orc, build, x, frames, sr):
    chains = []
    for c in range(x.shape[0]):
        ch = orc.Chain()
        build(ch, c)
        chains.append(ch)
    out = np.zeros_like(x)
    for b in range(0, x.shape[1], frames):
        for c, ch in enumerate(chains):
            out[c, b:b + frames] = ch.process(x[c, b:b + frames], sr)
    del chains
    return out


def unit_builder(unit, params):
    return lambda ch, c: ch.append_unit(unit, params=params)


def full_chain_builder(taps):
    def build(ch, c):
        ch.append_unit("compressor", params=[1, 30, -20])
        ch.append_unit("overdrive", params=[0, 20, 100, 0, 1, 0])
        ch.append_unit("tone_stack")
        ch.append_unit("chorus")
        ch.append_unit("power_amp", fir=synth_ir(taps, seed=4242 + c))
        ch.append_unit("power_amp", fir=synth_ir(taps, seed=4243 + c))
        ch.append_unit("cabinet")
        ch.append_unit("reverb", params=[50])
    return build


def inputs(sr, frames, blocks, nch=2):
    return np.stack([synth_signal(7 * c + 3, frames * blocks, sr) * (1.0 if c == 0 else 0.2) for c in range(nch)])


def worst_rms(a, b):
    return max(rms(a[c] - b[c]) for c in range(a.shape[0]))


def cases():
    for unit, params in UNIT_CASES:
        yield "%s %s" % (unit, params if params is not None else "defaults"), unit, unit_builder(unit, params)
    yield "full chain (SURVEY 8d), 2 x 8192-tap IRs", "full_chain", full_chain_builder(8192)


def measure_cpu(seeds):
    """{case: (unit, worst RMS plain-vs-jittered over rates and seeds, jittered calls)}"""
    orc = entry.load_oracle()
    res = {}
    for name, unit, build in cases():
        worst, calls = 0.0, 0
        for sr, frames, blocks in RATES:
            x = inputs(sr, frames, blocks)
            plain = stream(orc, build, x, frames, sr)
            for seed in range(1, seeds + 1):
                with orc.jittered(seed) as j:
                    moved = stream(orc, build, x, frames, sr)
                    calls += j.calls()
                worst = max(worst, worst_rms(plain, moved))
        res[name] = (unit, worst, calls)
    return res


def measure_gpu(seeds):
    """{case: worst RMS HIP-vs-jittered-oracle}"""
    pkg = entry.load_package()
    orc = entry.load_oracle()
    res = {}
    for name, unit, build in cases():
        worst = 0.0
        for sr, frames, blocks in RATES:
            x = inputs(sr, frames, blocks)
            ctx = pkg.Context(x.shape[0], frames)

            class Both:                       # the oracle's Chain interface on the HIP context
                def __init__(self, c):
                    self.c = c

                def append_unit(self, unit_type, params=None, fir=None):
                    ctx.append_unit(self.c, unit_type, params=params, fir=fir)
            for c in range(x.shape[0]):
                build(Both(c), c)
            got = np.zeros_like(x)
            for b in range(0, x.shape[1], frames):
                got[:, b:b + frames] = ctx.process(x[:, b:b + frames], sr)
            ctx.close()
            for seed in range(1, seeds + 1):
                with orc.jittered(seed):
                    moved = stream(orc, build, x, frames, sr)
                worst = max(worst, worst_rms(got, moved))
        res[name] = worst
    return res


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpu", action="store_true")
    ap.add_argument("--seeds", type=int, default=3)
    args = ap.parse_args()
    cpu = measure_cpu(args.seeds)
    gpu = measure_gpu(args.seeds) if args.gpu else {}
    print("# worst per-channel RMS over %s and %d seeds of the +-2 ulp perturbation (oracle/libm_jitter.h); bar: 1e-9" % (RATES, args.seeds))
    print("%-58s %14s %12s %14s" % ("case", "oracle vs moved", "moved calls", "HIP vs moved" if gpu else ""))
    for name, (unit, w, calls) in cpu.items():
        print("%-58s %14.3e %12d %14s" % (name, w, calls, ("%.3e" % gpu[name]) if name in gpu else ""))
    by_unit = {}
    for name, (unit, w, calls) in cpu.items():
        g = gpu.get(name, 0.0)
        by_unit[unit] = max(by_unit.get(unit, 0.0), w, g)
    print("\n# per unit (worst of its cases, both comparisons)")
    for unit, w in sorted(by_unit.items(), key=lambda kv: -kv[1]):
        print("%-20s %10.3e  %s" % (unit, w, "ok" if w <= 1e-9 else "DISCRETE DECISION FLIPS"))


if __name__ == "__main__":
    main()
