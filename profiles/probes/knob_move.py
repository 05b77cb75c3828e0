"""Control-plane latency (round-3 review, item 7): what a parameter change costs on a LIVE 512-channel context.  The reference's setter
is a mutex and one store (effects/effects.go:283-345); here the change has to reach the device descriptors.  Measured: wall time of the
gdg_process_device call that follows the change (+ synchronize) against a steady-state call, for
  - one knob of one unit (overdrive gain, tone-stack band, chorus depth, reverb mix, delay time = history re-made),
  - the same knob on ALL 512 channels (a preset change),
  - a bypass toggle (layout change: full plan rebuild),
  - new taps for ONE power amp (gdg_unit_set_fir: full rebuild + one IR transform).
GDG_PLAN_PATCH=0 gives round 3's behaviour (every change rebuilds the whole plan).
    python profiles/probes/knob_move.py > gpurun_out/knob_move_r04.txt"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
lib = pkg.lib()
nch, frames, sr, taps = 512, 8192, 192000, 65536
ctx = bench.make_context(pkg, nch, frames, 0, taps, n_distinct=8)
d_in, d_out = ctx.alloc(nch, frames), ctx.alloc(nch, frames)
d_in.upload(bench.synth_block(nch, frames, sr))
names = [n for n, _ in bench.CHAIN]
handles = {}                                              # (channel, unit name) -> handles, read back from the wrapper's chain bookkeeping
for c in range(nch):
    for k, (h, _) in enumerate(ctx._chains[c]):
        handles.setdefault((c, names[k]), []).append(h)


def step():
    t0 = time.perf_counter()
    ctx.process_device(d_in, d_out, frames, sr)
    ctx.synchronize()
    return (time.perf_counter() - t0) * 1e6


first = step()                                            # builds the plan, the IR spectra, every unit's state
print("# first call on the fresh 512-channel context (plan + %d IR transforms + state): %.1f ms" % (2 * nch, first / 1e3))
for _ in range(5):
    step()
steady = sorted(step() for _ in range(20))[10]
print("# plan patching %s; steady-state call + synchronize: %.0f us" % ("OFF (GDG_PLAN_PATCH=0)" if os.environ.get("GDG_PLAN_PATCH") == "0" else "on", steady))
print("change,us of the next call,extra us over steady state,us of the call after")


def measure(label, change):
    res = []
    for rep in range(5):
        change(rep)
        a = step()
        b = step()
        res.append((a, b))
    res.sort()
    a, b = res[2]
    print("%s,%.0f,%.0f,%.0f" % (label, a, a - steady, b))


def set_param(h, i, v):
    ctx._check(lib.gdg_unit_set_param(ctx._h, h, i, v))


measure("overdrive gain on channel 7", lambda r: set_param(handles[(7, "overdrive")][0], 1, 10 + r))
measure("tone stack band 2 on channel 7", lambda r: set_param(handles[(7, "tone_stack")][0], 1, -3 - r))
measure("chorus depth on channel 7", lambda r: set_param(handles[(7, "chorus")][0], 0, 60 + r))
measure("reverb mix on channel 7", lambda r: set_param(handles[(7, "reverb")][0], 0, 40 + r))
measure("compressor target on channel 7", lambda r: set_param(handles[(7, "compressor")][0], 2, -25 + r))
measure("overdrive gain on ALL 512 channels", lambda r: [set_param(handles[(c, "overdrive")][0], 1, 11 + r) for c in range(nch)])
measure("tone stack band 2 on ALL 512 channels (one new scan table)", lambda r: [set_param(handles[(c, "tone_stack")][0], 1, -8 - r) for c in range(nch)])
# bypass toggle of channel 7's chorus: gdg_chain_set = layout change
chain7 = [h for h, _ in ctx._chains[7]]
measure("bypass toggle of one unit (layout: full rebuild)", lambda r: ctx.chain_set(7, chain7, [(k == 3 and r % 2 == 0) for k in range(len(names))]))
ctx.chain_set(7, chain7, [False] * len(names))
step()
ir = bench.synth_ir(taps, 999)
measure("new taps for ONE power amp (full rebuild + one IR transform)", lambda r: ctx.unit_set_fir(chain7[4], ir * (1.0 + 0.01 * r)))
ctx.close()
