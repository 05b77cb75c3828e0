"""The two-per-CU segment kernel (seg.hip compiled with SEG_FAST: 512 threads per channel, ONE LDS frame buffer, every unit in place) takes over
segments of 8192-sample frames from 128 channels on.  Here it is forced on for small contexts (GDG_SEG_FAST_MIN=0) and held against the oracle
(1e-9 RMS), against the general kernel (same arithmetic, another association of the workgroup scans: ~1e-16) and against itself in windows
(bit for bit); streams that move between the two kernels -- frame-size changes, a unit that loses its eligibility -- must stay continuous:
both kernels share one state layout in HBM."""
import numpy as np
import pytest

from helpers import TOL_RMS, package, rms, synth_ir, synth_signal

pytestmark = pytest.mark.gpu

B = 8192
FAST_UNITS = [("compressor", [0, 30, -20]), ("compressor", [1, 12, -6]), ("overdrive", [5, 25, 80, -3, 0, 0]), ("overdrive", [0, 20, 100, 0, 1, 0]),
              ("distortion", [10, 20, -6, 0]), ("excess", [24, -3, 0]), ("tone_stack", [3, -4, -9, -1]), ("cabinet", None), ("chorus", [70, 45]),
              ("chorus", [0, 30]), ("reverb", [65]), ("ring_modulator", [37]), ("tremolo", [55, 30, -12]), ("signal_generator", [60, -6, 4, 440, 50, -10]),
              ("signal_generator", [100, 0, 0, 1000, 30, -20])]
BENCH_CHAIN = [("compressor", [1, 30, -20]), ("overdrive", [0, 20, 100, 0, 1, 0]), ("tone_stack", None), ("chorus", None), ("power_amp", "ir"),
               ("cabinet", None), ("reverb", [50])]


def build(pkg, oracle, chains, frames=B):
    ctx = pkg.Context(len(chains), frames)
    refs = []
    for c, chain in enumerate(chains):
        ref = oracle.Chain() if oracle is not None else None
        for name, p in chain:
            fir = synth_ir(3000, seed=40 + c) if p == "ir" else None
            par = None if p == "ir" else p
            ctx.append_unit(c, name, params=par, fir=fir)
            if ref is not None:
                ref.append_unit(name, params=par, fir=fir)
        refs.append(ref)
    return ctx, refs


def stream(ctx, refs, x, sr, sizes):
    got, want = np.zeros_like(x), np.zeros_like(x)
    at = 0
    for n in sizes:
        blk = np.ascontiguousarray(x[:, at:at + n])
        got[:, at:at + n] = ctx.process(blk, sr)
        for c, r in enumerate(refs):
            if r is not None:
                want[c, at:at + n] = r.process(blk[c], sr)
        at += n
    return got, want


@pytest.mark.parametrize("sr", [48000, 192000])
def test_every_in_place_unit_follows_the_oracle_and_the_general_kernel(oracle, monkeypatch, sr):
    pkg = package()
    chains = [[u] for u in FAST_UNITS] + [BENCH_CHAIN]
    x = np.stack([0.7 * synth_signal(c, 5 * B, sr) for c in range(len(chains))])
    monkeypatch.setenv("GDG_SEG_FAST_MIN", "0")
    ctx, refs = build(pkg, oracle, chains)
    fast, want = stream(ctx, refs, x, sr, [B] * 5)
    ctx.close()
    monkeypatch.setenv("GDG_SEG_FAST", "0")
    ctx, _ = build(pkg, None, chains)
    general, _ = stream(ctx, [None] * len(chains), x, sr, [B] * 5)
    ctx.close()
    for c, chain in enumerate(chains):
        assert rms(fast[c] - want[c]) <= TOL_RMS, (chain, rms(fast[c] - want[c]))
        assert np.max(np.abs(fast[c] - general[c])) <= 1e-12, (chain, float(np.max(np.abs(fast[c] - general[c]))))
    # the forced run really took the other kernel: the scans associate differently, so SOME recurrence differs in its last bits
    assert any(not np.array_equal(fast[c], general[c]) for c in range(len(chains)))


def test_windows_of_the_two_per_cu_kernel_equal_its_single_frames(monkeypatch):
    pkg = package()
    monkeypatch.setenv("GDG_SEG_FAST_MIN", "0")
    sr, W, blocks = 96000, 8, 8 + 3
    chains = [BENCH_CHAIN, [("reverb", [30]), ("chorus", [100, 10])], [("tone_stack", None), ("cabinet", None), ("tremolo", None)]]
    x = np.stack([0.6 * synth_signal(c, blocks * B, sr) for c in range(len(chains))])
    ctx, _ = build(pkg, None, chains)
    d_in, d_out = ctx.alloc(len(chains), B), ctx.alloc(len(chains), B)
    single = np.zeros_like(x)
    for b in range(blocks):
        d_in.upload(np.ascontiguousarray(x[:, b * B:(b + 1) * B]))
        ctx.process_device(d_in, d_out, B, sr)
        single[:, b * B:(b + 1) * B] = d_out.download()
    ctx.close()
    ctx, _ = build(pkg, None, chains)
    ctx.set_window(W)
    n = blocks * B
    w_in, w_out = ctx.alloc(len(chains), n), ctx.alloc(len(chains), n)
    w_in.upload(x)
    done = 0
    while done < blocks:
        w = W
        while w > blocks - done:
            w //= 2
        ctx.process_window_device(w_in.ptr + 8 * done * B, w_out.ptr + 8 * done * B, n, w, sr)
        done += w
    win = w_out.download()
    ctx.close()
    for c in range(len(chains)):
        assert np.array_equal(win[c], single[c]), c


def test_a_stream_moves_between_the_two_kernels_without_a_seam(oracle, monkeypatch):
    """frame sizes 8192 (two-per-CU kernel) and 1024 / 4096 / 1000 (general kernel) alternate; then the overdrive switches its 4x oversampling on
    (its segment leaves the two-per-CU kernel) and off again: one continuous stream against the oracle"""
    pkg = package()
    monkeypatch.setenv("GDG_SEG_FAST_MIN", "0")
    sr = 96000
    sizes = [B, B, 1024, 1024, B, 4096, 1000, B, B, B, B, B]
    x = np.stack([0.6 * synth_signal(c, sum(sizes), sr) for c in range(2)])
    ctx, refs = build(pkg, oracle, [BENCH_CHAIN, [("chorus", None), ("reverb", None), ("cabinet", None)]])
    got, want = np.zeros_like(x), np.zeros_like(x)
    at = 0
    for k, n in enumerate(sizes):
        if k in (8, 10):
            v = 2 if k == 8 else 0
            ctx.unit_set_param(ctx._chains[0][1][0], 5, v)
            refs[0].unit(1).set_param(5, v)
        blk = np.ascontiguousarray(x[:, at:at + n])
        got[:, at:at + n] = ctx.process(blk, sr)
        for c in range(2):
            want[c, at:at + n] = refs[c].process(blk[c], sr)
        at += n
    ctx.close()
    for c in range(2):
        assert rms(got[c] - want[c]) <= TOL_RMS, (c, rms(got[c] - want[c]))


def test_small_contexts_stay_on_the_general_kernel_by_default(monkeypatch):
    """below 128 channels per call the general kernel is the default (one round of 1024-thread workgroups; the two-per-CU build takes over where windows of a workgroup per frame gain from twice the frames in flight): same bits as GDG_SEG_FAST=0"""
    pkg = package()
    sr = 48000
    x = np.stack([0.7 * synth_signal(c, 2 * B, sr) for c in range(2)])
    outs = []
    for env in ({}, {"GDG_SEG_FAST": "0"}):
        for k in ("GDG_SEG_FAST", "GDG_SEG_FAST_MIN"):
            monkeypatch.delenv(k, raising=False)
        for k, v in env.items():
            monkeypatch.setenv(k, v)
        ctx, _ = build(pkg, None, [BENCH_CHAIN, [("reverb", None)]])
        outs.append(stream(ctx, [None, None], x, sr, [B, B])[0])
        ctx.close()
    assert np.array_equal(outs[0], outs[1])


@pytest.mark.parametrize("seed", range(24))
def test_random_in_place_chains_on_the_two_per_cu_kernel(oracle, monkeypatch, seed):
    """Random chains of the units the two-per-CU kernel runs (power amps between them cut the chains into segments), parameters anywhere in the
    reference's ranges (oversampling off), 8192-sample frames at a random rate: against the oracle, and run TWICE -- every unit of this kernel
    shares one LDS buffer with its neighbours' threads, a race would show as a difference between the runs."""
    import json
    import os
    import __graft_entry__ as entry
    with open(os.path.join(entry.ROOT, "tests", "golden", "params.json")) as f:
        PARAMS = json.load(f)
    pkg = package()
    monkeypatch.setenv("GDG_SEG_FAST_MIN", "0")
    rng = np.random.default_rng(7000 + seed)
    sr = int(rng.choice([44100, 48000, 96000, 192000]))
    names = ["compressor", "overdrive", "distortion", "excess", "tone_stack", "cabinet", "chorus", "reverb", "ring_modulator", "tremolo", "signal_generator"]
    chains = []
    for c in range(3):
        chain = []
        for _ in range(int(rng.integers(1, 7))):
            if rng.random() < 0.15:
                chain.append(("power_amp", "ir"))
                continue
            name = names[int(rng.integers(0, len(names)))]
            vals = []
            for p in PARAMS[str(pkg.UNIT[name])]["params"]:
                if p["Type"] == "PARAMETER_TYPE_DISCRETE":
                    vals.append(0 if p["Name"] == "oversampling" else int(rng.integers(0, len(p["DiscreteValues"]))))
                else:
                    lo, hi = int(p["Minimum"]), int(p["Maximum"])
                    vals.append(int(rng.choice([lo, hi, int(rng.integers(lo, hi + 1)), int(rng.integers(lo, hi + 1))])))
            chain.append((name, vals))
        chains.append(chain)
    blocks = 3
    x = np.stack([synth_signal(int(rng.integers(0, 48)), blocks * B, sr) * float(rng.choice([0.05, 0.5, 1.0])) for _ in range(3)])
    runs = []
    want = None
    for k in range(2):
        ctx, refs = build(pkg, oracle if k == 0 else None, chains)
        got, w = stream(ctx, refs if k == 0 else [None] * 3, x, sr, [B] * blocks)
        ctx.close()
        runs.append(got)
        if k == 0:
            want = w
    for c in range(3):
        assert np.isfinite(runs[0][c]).all(), (seed, c, chains[c])
        assert np.array_equal(runs[0][c], runs[1][c]), (seed, c, chains[c])
        assert rms(runs[0][c] - want[c]) <= TOL_RMS, (seed, c, sr, chains[c], rms(runs[0][c] - want[c]))


def test_full_load_equals_light_load_bit_for_bit(monkeypatch):
    """512 channels (two workgroups on every CU, all phases of all units colliding) against the SAME channels alone in a three-channel context on
    the same kernel: channels are independent, so every bit must agree -- a race between co-resident workgroups or between the threads that
    share the one LDS buffer would show here (and the big run is repeated: equal to itself as well)."""
    pkg = package()
    sr, nch, blocks = 192000, 512, 4
    ir = synth_ir(4000, seed=9)
    chain = [("compressor", [1, 30, -20]), ("overdrive", [0, 20, 100, 0, 1, 0]), ("tone_stack", None), ("chorus", None), ("power_amp", ir),
             ("cabinet", None), ("reverb", [50]), ("tremolo", None)]
    pick = [0, 255, 511]

    def run(channels):
        ctx = pkg.Context(len(channels), B)
        for c in range(len(channels)):
            for name, p in chain:
                if name == "power_amp":
                    ctx.append_unit(c, name, fir=p)
                else:
                    ctx.append_unit(c, name, params=p)
        x = np.stack([0.6 * synth_signal(g % 48, blocks * B, sr) * (1.0 + 0.001 * g) for g in channels])
        d_in, d_out = ctx.alloc(len(channels), B), ctx.alloc(len(channels), B)
        out = np.zeros_like(x)
        for b in range(blocks):
            d_in.upload(np.ascontiguousarray(x[:, b * B:(b + 1) * B]))
            ctx.process_device(d_in, d_out, B, sr)
            out[:, b * B:(b + 1) * B] = d_out.download()
        ctx.close()
        return out

    big1 = run(list(range(nch)))
    big2 = run(list(range(nch)))
    assert np.array_equal(big1, big2)
    monkeypatch.setenv("GDG_SEG_FAST_MIN", "0")
    small = run(pick)
    for k, g in enumerate(pick):
        assert np.array_equal(small[k], big1[g]), (g, float(np.max(np.abs(small[k] - big1[g]))))
    assert np.isfinite(big1).all() and np.max(np.abs(big1)) > 0.01
