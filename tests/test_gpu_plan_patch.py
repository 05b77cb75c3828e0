"""Parameter changes on a live context are patched into the plan's descriptors in place (api_process.cpp apply_patches; round-3 review, item 7): the
next call must behave exactly as if the whole plan had been rebuilt (GDG_PLAN_PATCH=0, round 3's behaviour) -- same bits -- and follow the
oracle, whose setters are the reference's (effects/effects.go:283-345: a store under a mutex, effective from the next Process)."""
import json
import os

import numpy as np
import pytest

import __graft_entry__ as entry
from helpers import TOL_RMS, package, rms, synth_ir, synth_signal

pytestmark = pytest.mark.gpu

with open(os.path.join(entry.ROOT, "tests", "golden", "params.json")) as f:
    PARAMS = json.load(f)

# every unit the segment kernel runs (all but the power amp), octaver first in its chain (see tests/test_gpu_fuzz.py on octavers behind other stages)
UNITS = ["signal_generator", "noise_gate", "bandpass", "auto_wah", "auto_yoy", "compressor", "octaver", "excess", "fuzz", "overdrive", "distortion",
         "tone_stack", "chorus", "flanger", "phaser", "tremolo", "ring_modulator", "delay", "reverb", "cabinet"]


def draw(rng, name):
    pkg = package()
    out = []
    for p in PARAMS[str(pkg.UNIT[name])]["params"]:
        if p["Type"] == "PARAMETER_TYPE_DISCRETE":
            out.append(int(rng.integers(0, len(p["DiscreteValues"]))))
        else:
            lo, hi = int(p["Minimum"]), int(p["Maximum"])
            out.append(int(rng.choice([lo, hi, int(rng.integers(lo, hi + 1))])))
    return out


def run(pkg, oracle, monkeypatch, patch, name, sr, frames, blocks, edits, groups=1):
    """one context of 3 channels: [unit] on channel 0, [compressor, unit, power amp, cabinet] on channel 1, [unit, tone stack] on channel 2;
    `edits[b]` = list of (channel, param index, value) applied BEFORE block b.  Returns (device stream, oracle stream)."""
    monkeypatch.setenv("GDG_PLAN_PATCH", "1" if patch else "0")
    ctx = pkg.Context(3, frames)
    fir = synth_ir(700, seed=5)
    lead = "compressor" if name != "compressor" else "distortion"           # neighbours of another type than the unit under test
    tail = "cabinet" if name != "cabinet" else "tone_stack"
    other = "tone_stack" if name != "tone_stack" else "cabinet"
    layout = [[name], [lead, name, "power_amp", tail] if name != "octaver" else [name, "power_amp", tail], [name, other]]
    refs, target = [], []
    for c, units in enumerate(layout):
        ref = oracle.Chain() if oracle is not None else None
        for k, u in enumerate(units):
            h = ctx.append_unit(c, u, fir=fir if u == "power_amp" else None)
            if ref is not None:
                ref.append_unit(u, fir=fir if u == "power_amp" else None)
            if u == name:
                target.append((h, k))
        refs.append(ref)
    assert len(target) == 3
    ctx.set_overlap(groups)
    x = np.stack([0.6 * synth_signal(3 + c, frames * blocks, sr) for c in range(3)])
    d_in, d_out = ctx.alloc(3, frames), ctx.alloc(3, frames)
    got, want = np.zeros_like(x), np.zeros_like(x)
    for b in range(blocks):
        for c, i, v in edits.get(b, []):
            ctx.unit_set_param(target[c][0], i, v)
            if refs[c] is not None:
                refs[c].unit(target[c][1]).set_param(i, v)
        sl = slice(b * frames, (b + 1) * frames)
        d_in.upload(np.ascontiguousarray(x[:, sl]))
        ctx.process_device(d_in, d_out, frames, sr)
        got[:, sl] = d_out.download()
        for c in range(3):
            if refs[c] is not None:
                want[c, sl] = refs[c].process(x[c, sl], sr)
    ctx.close()
    return got, want


@pytest.mark.parametrize("name", UNITS)
def test_parameter_edits_are_patched_in_place_with_the_bits_of_a_rebuild(oracle, monkeypatch, name):
    pkg = package()
    rng = np.random.default_rng(pkg.UNIT[name])
    sr, frames, blocks = 96000, 8192, 6
    n_params = len(PARAMS[str(pkg.UNIT[name])]["params"])
    edits = {}
    for b in (1, 2, 4):                                       # block 3 and 5 run on an unchanged plan
        vals = draw(rng, name)
        chans = [0, 1, 2] if b != 2 else [1]                  # one round touches a single channel
        edits[b] = [(c, i, vals[i]) for c in chans for i in range(n_params)]
    edits[4] = edits[4] + edits[4]                            # the same values twice: second set is a no-op
    patched, want = run(pkg, oracle, monkeypatch, True, name, sr, frames, blocks, edits)
    rebuilt, _ = run(pkg, None, monkeypatch, False, name, sr, frames, blocks, edits)
    for c in range(3):
        assert np.array_equal(patched[c], rebuilt[c]), (name, c, float(np.max(np.abs(patched[c] - rebuilt[c]))))
        assert rms(patched[c] - want[c]) <= TOL_RMS, (name, c, rms(patched[c] - want[c]))


@pytest.mark.parametrize("name,frames,sr", [("delay", 1000, 48000), ("chorus", 480, 44100), ("bandpass", 8192, 192000), ("fuzz", 4096, 96000), ("overdrive", 1024, 48000)])
def test_patches_with_free_running_groups_and_other_frame_sizes(oracle, monkeypatch, name, frames, sr):
    """the units whose parameters re-make state (delay time -> history, band-pass order -> capacitors, oversampling factor -> oversampler
    objects), at frame sizes other than the batch block, with two free-running channel groups (the patch waits for them)"""
    pkg = package()
    rng = np.random.default_rng(77)
    blocks = 8
    n_params = len(PARAMS[str(pkg.UNIT[name])]["params"])
    edits = {b: [(c, i, v) for c in range(3) for i, v in enumerate(draw(rng, name))] for b in (1, 3, 4, 6)}
    patched, want = run(pkg, oracle, monkeypatch, True, name, sr, frames, blocks, edits, groups=2)
    rebuilt, _ = run(pkg, None, monkeypatch, False, name, sr, frames, blocks, edits, groups=1)
    for c in range(3):
        assert np.array_equal(patched[c], rebuilt[c]), (name, c)
        assert rms(patched[c] - want[c]) <= TOL_RMS, (name, c, rms(patched[c] - want[c]))
    assert n_params > 0


def test_a_patch_costs_no_plan_rebuild(monkeypatch, capfd):
    """GDG_PLAN_TRACE prints one line per build_plan: a knob move on a live context must not print one, a bypass toggle must."""
    pkg = package()
    monkeypatch.setenv("GDG_PLAN_TRACE", "1")
    monkeypatch.setenv("GDG_PLAN_PATCH", "1")
    ctx = pkg.Context(2, 1024)
    hs = [[ctx.append_unit(c, "overdrive"), ctx.append_unit(c, "power_amp", fir=synth_ir(300, seed=1)), ctx.append_unit(c, "reverb")] for c in range(2)]
    x = np.stack([synth_signal(c, 1024, 48000) for c in range(2)])
    ctx.process(x, 48000)
    capfd.readouterr()
    ctx.unit_set_param(hs[0][0], 1, 17)
    ctx.unit_set_param(hs[1][2], 0, 80)
    ctx.unit_set_param(hs[0][1], 0, 3)                         # a power amp's parameter: nothing on the device depends on it
    ctx.process(x, 48000)
    assert "[plan]" not in capfd.readouterr().err
    ctx.chain_set(0, hs[0], [True, False, False])
    ctx.process(x, 48000)
    assert "[plan]" in capfd.readouterr().err
    ctx.close()
