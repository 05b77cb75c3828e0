#!/usr/bin/env python3
"""One seed of tests/test_gpu_fuzz.py::test_random_chains_follow_the_oracle taken apart: the failing channel's chain unit by unit (prefixes), the
RMS against the oracle after every unit and the first sample where they part.   python profiles/probes/chain_case.py <seed> <channel>"""
import os, sys
R = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import __graft_entry__ as entry
import test_gpu_fuzz as F
from helpers import ChainPair, rms, synth_ir, synth_signal
pkg = entry.load_package(); oracle = entry.load_oracle()
seed, chan = int(sys.argv[1]), int(sys.argv[2])
rng = np.random.default_rng(1000 + seed)
sr = int(rng.choice([22050, 44100, 48000, 96000, 192000]))
frames = int(rng.choice([8192, 8192, 1024, 1000, 480, 4096]))
blocks = 3 if frames >= 4096 else 5
chains = []
for c in range(3):
    units = []
    for _ in range(int(rng.integers(1, 8))):
        t = int(rng.integers(0, 21)); name = pkg.UNIT_NAMES[t]; bypass = bool(rng.random() < 0.15)
        if name == "power_amp":
            taps = int(rng.choice([1, 77, 500, 3000, 9000]))
            while F.reference_panics(frames, taps): taps *= 2
            units.append((name, synth_ir(taps, seed=int(rng.integers(1, 10 ** 6))) * float(rng.choice([0.5, 1.0, 2.5])), bypass))
        else:
            units.append((name, F.random_params(rng, t, allow_oversampling=True), bypass))
    chains.append(units)
x = np.stack([synth_signal(int(rng.integers(0, 48)), frames * blocks, sr) * float(rng.choice([0.05, 0.5, 1.0])) for _ in range(3)])
print("seed", seed, "sr", sr, "frames", frames, "channel", chan)
units = chains[chan]
for k in range(1, len(units) + 1):
    ctx = pkg.Context(1, frames); p = ChainPair(ctx, 0, oracle)
    for name, arg, bypass in units[:k]:
        p.append(name, fir=arg, bypass=bypass) if name == "power_amp" else p.append(name, params=arg, bypass=bypass)
    got, want = np.zeros(frames * blocks), np.zeros(frames * blocks)
    for b in range(blocks):
        sl = slice(b * frames, (b + 1) * frames)
        got[sl] = ctx.process(np.ascontiguousarray(x[chan:chan + 1, sl]), sr)[0]; want[sl] = p.ref.process(x[chan, sl], sr)
    ctx.close()
    d = np.abs(got - want); first = int(np.argmax(d > 1e-9)) if (d > 1e-9).any() else -1
    name, arg, bypass = units[k - 1]
    print(" after %-16s %-40s%s: RMS %.3e, max %.3e, first sample off by > 1e-9: %d%s" % (name, (len(arg) if name == "power_amp" else arg), " (bypassed)" if bypass else "",
          rms(got - want), d.max(), first, "" if first < 0 else "  (device %.12g, oracle %.12g; %d samples differ)" % (got[first], want[first], int((d > 1e-9).sum()))))

# the last unit that parted, fed the ORACLE's signal in front of it (bit-identical input on both sides)
if len(sys.argv) > 3:
    k = int(sys.argv[3])                                  # index of the unit to isolate
    pre = oracle.Chain()
    for name, arg, bypass in units[:k]:
        pre.append_unit(name, bypass=bypass, params=None if name == "power_amp" else arg, fir=arg if name == "power_amp" else None)
    mid = np.concatenate([pre.process(x[chan, b * frames:(b + 1) * frames], sr) for b in range(blocks)])
    ctx = pkg.Context(1, frames); p = ChainPair(ctx, 0, oracle)
    name, arg, bypass = units[k]
    p.append(name, fir=arg, bypass=bypass) if name == "power_amp" else p.append(name, params=arg, bypass=bypass)
    got = np.concatenate([ctx.process(np.ascontiguousarray(mid[None, b * frames:(b + 1) * frames]), sr)[0] for b in range(blocks)])
    want = np.concatenate([p.ref.process(mid[b * frames:(b + 1) * frames], sr) for b in range(blocks)])
    ctx.close()
    d = np.abs(got - want)
    i0 = int(np.argmax(d > 1e-12 * max(np.max(np.abs(want)), 1e-300))) if d.max() > 0 else -1
    print(" %s alone on the oracle's input: RMS %.3e, max %.3e; input |x| range %.3e .. %.3e; first relative difference > 1e-12 at %d" % (name, rms(got - want), d.max(), np.min(np.abs(mid)), np.max(np.abs(mid)), i0))
    if i0 >= 0:
        print("   input around it:", ["%.17g" % v for v in mid[max(i0 - 3, 0):i0 + 2]])
        print("   device:", ["%.10g" % v for v in got[max(i0 - 1, 0):i0 + 2]], "oracle:", ["%.10g" % v for v in want[max(i0 - 1, 0):i0 + 2]])
