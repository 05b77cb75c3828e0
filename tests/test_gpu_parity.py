"""Parity tests proper: the HIP path (through the C-ABI, libgdg.so) against the CPU oracle on
the same seeded inputs.  Tolerance: 1e-9 RMS per channel (north_star), max-abs reported.
All tests need a real MI355X: run with `pytest -m gpu`.
"""
import numpy as np
import pytest

from helpers import ChainPair, TOL_RMS, package, rms, run_pairs, synth_ir, synth_signal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    p = package()
    assert p.device_count() > 0, "no HIP device: the gpu tests must run on the GPU box"
    return p


def check(got, want, tol=TOL_RMS):
    for c in range(got.shape[0]):
        err = rms(got[c] - want[c])
        assert err <= tol, "channel %d: RMS %.3e (max abs %.3e)" % (c, err, float(np.max(np.abs(got[c] - want[c]))))


# ---- FIR (power amp): filter.Process semantics y = clip(x * h) ------------------------------------------
@pytest.mark.parametrize("frames,taps", [(64, 1), (64, 200), (256, 77), (1024, 1024), (1024, 1025), (1024, 5000),
                                          (4096, 9600), (8192, 511), (8192, 8192), (8192, 20000),
                                          # frame sizes that are not a power of two (e.g. 480-frame periods): hop < transform half
                                          (1000, 3000), (480, 2000), (37, 100), (100, 77), (1, 5), (8000, 20000), (4097, 4097), (63, 64)])
def test_fir_stream_matches_oracle_and_direct_convolution(pkg, oracle, frames, taps):
    sr, blocks = 48000, 5
    ctx = pkg.Context(2, frames)
    h = [synth_ir(taps, seed=4242 + c) * (3.0 if c else 1.0) for c in range(2)]   # channel 1 is driven into the clip
    pairs = []
    for c in range(2):
        p = ChainPair(ctx, c, oracle)
        p.append("power_amp", fir=h[c])
        pairs.append(p)
    x = np.stack([synth_signal(c, frames * blocks, sr) for c in range(2)])
    got, want = run_pairs(ctx, pairs, x, frames, sr)
    check(got, want)
    for c in range(2):      # second, independent formulation: direct-form convolution + clip
        direct = np.clip(np.convolve(x[c], h[c])[:x.shape[1]], -1.0, 1.0)
        assert rms(got[c] - direct) <= TOL_RMS
    ctx.close()


def test_fir_empty_filter_and_missing_filter_give_zeros(pkg, oracle):
    ctx = pkg.Context(2, 256)
    a = ChainPair(ctx, 0, oracle); a.append("power_amp", fir=[])          # filter.Empty
    b = ChainPair(ctx, 1, oracle); b.append("power_amp")                   # never compiled
    x = np.stack([synth_signal(c, 512, 48000) for c in range(2)])
    got, want = run_pairs(ctx, [a, b], x, 256, 48000)
    assert np.all(got == 0.0) and np.all(want == 0.0)
    ctx.close()


def test_fir_set_fir_resets_state(pkg, oracle):
    frames, sr = 512, 48000
    ctx = pkg.Context(1, frames)
    p = ChainPair(ctx, 0, oracle)
    hnd = p.append("power_amp", fir=synth_ir(3000))
    x = synth_signal(0, frames * 4, sr)[None, :]
    got1, want1 = run_pairs(ctx, [p], x[:, :2 * frames], frames, sr)
    h2 = synth_ir(700, seed=9)
    ctx.unit_set_fir(hnd, h2)
    p.ref.unit(0).set_fir(h2)
    got2, want2 = run_pairs(ctx, [p], x[:, 2 * frames:], frames, sr)
    check(np.hstack([got1, got2]), np.hstack([want1, want2]))
    ctx.close()


def test_fir_odd_frame_sizes_work_in_chains(pkg, oracle):
    """Any frame size up to 8192 runs (the reference's filter.Process takes any block length); a full chain at 1000 frames."""
    sr, frames = 44100, 1000
    ctx = pkg.Context(2, frames)
    pairs = []
    for c in range(2):
        p = ChainPair(ctx, c, oracle)
        full_chain(p, c, synth_ir(5000, seed=c), synth_ir(2500, seed=10 + c))
        pairs.append(p)
    x = np.stack([synth_signal(c, frames * 6, sr) for c in range(2)])
    got, want = run_pairs(ctx, pairs, x, frames, sr)
    check(got, want)
    ctx.close()


# ---- single units ------------------------------------------------------------------------------------------
UNIT_CASES = [
    ("compressor", None), ("compressor", [0, 30, -20]), ("compressor", [1, 12, -6]),
    ("overdrive", None), ("overdrive", [10, 20, 70, -3, 0, 0]), ("overdrive", [0, 20, 100, 0, 1, 1]), ("overdrive", [0, 20, 100, 0, 1, 2]),
    ("overdrive", [5, 10, 50, -6, 0, 2]),
    ("distortion", [0, 20, -3, 0]), ("distortion", [0, 20, -3, 1]), ("distortion", [10, 10, 0, 2]),
    ("excess", [20, -3, 0]), ("excess", [30, 0, 1]), ("excess", [12, -6, 2]),
    ("tone_stack", None), ("tone_stack", [-10, 0, -3, -20]),
    ("cabinet", None),
    ("chorus", None), ("chorus", [35, 77]),
    ("flanger", None), ("flanger", [40, 100]),
    ("phaser", None), ("phaser", [70, 33, -60]),
    ("delay", None), ("delay", [3, -10, 0]), ("delay", [1000, 0, 0]),
    ("ring_modulator", None), ("ring_modulator", [7]),
    ("tremolo", None), ("tremolo", [10, 0, -60]), ("tremolo", [100, 100, -3]),
    ("signal_generator", None), ("signal_generator", [50, -6, 1, 1000, 80, -3]), ("signal_generator", [100, 0, 2, 123, 50, 0]),
    ("signal_generator", [100, 0, 3, 5000, 50, 0]), ("signal_generator", [30, 0, 4, 440, 100, -10]),
    ("reverb", None), ("reverb", [100]), ("reverb", [0]),
    ("fuzz", None), ("fuzz", [0, -30, 10, 20, 60, -3, 0]), ("fuzz", [1, 100, 0, 30, 100, 0, 0]),
    ("fuzz", [1, 50, 0, 20, 100, 0, 1]), ("fuzz", [0, 30, 10, 10, 70, -6, 2]), ("fuzz", [1, -50, 0, 30, 100, 0, 2]),
    ("auto_yoy", None), ("auto_yoy", [0, -10, -50, 40]),
    ("auto_wah", None), ("auto_wah", [0, -5, -45, 200, 9000]),
    ("bandpass", None), ("bandpass", [3, 5000, 100]), ("bandpass", [1, 40, 18000]),
    ("octaver", None), ("octaver", [0, -6, -12, -3, 0, -6, -30]), ("octaver", [1, -60, 0, -60, -3, -3, 0]),
    ("noise_gate", None), ("noise_gate", [-10, -14, 5]), ("noise_gate", [-30, -20, 50]), ("noise_gate", [-6, -40, 0]),
    ("noise_gate", [-3, -8, 1]),
]


@pytest.mark.parametrize("unit,params", UNIT_CASES)
@pytest.mark.parametrize("sr,frames", [(48000, 1024), (192000, 8192), (22050, 1000)])
def test_single_unit_stream(pkg, oracle, unit, params, sr, frames):
    blocks = 3 if frames == 8192 else 6
    nch = 2
    ctx = pkg.Context(nch, frames)
    pairs = []
    for c in range(nch):
        p = ChainPair(ctx, c, oracle)
        p.append(unit, params=params)
        pairs.append(p)
    x = np.stack([synth_signal(7 * c + 3, frames * blocks, sr) * (1.0 if c == 0 else 0.2) for c in range(nch)])
    got, want = run_pairs(ctx, pairs, x, frames, sr)
    check(got, want)
    ctx.close()


# ---- chains -----------------------------------------------------------------------------------------------------
def full_chain(p, os_index, cab_ir, reverb_ir=None):
    """SURVEY.md section 8d canonical ordering."""
    p.append("compressor", params=[1, 30, -20])
    p.append("overdrive", params=[0, 20, 100, 0, 1, os_index])
    p.append("tone_stack")
    p.append("chorus")
    p.append("power_amp", fir=cab_ir)
    if reverb_ir is not None:
        p.append("power_amp", fir=reverb_ir)
    p.append("cabinet")
    p.append("reverb", params=[50])


@pytest.mark.parametrize("sr,frames,taps,os_index,two_irs", [(48000, 1024, 8192, 0, False), (48000, 8192, 8192, 0, False),
                                                            (96000, 8192, 32768, 2, False), (192000, 8192, 65536, 0, True)])
def test_full_chain(pkg, oracle, sr, frames, taps, os_index, two_irs):
    blocks = 3
    nch = 3
    ctx = pkg.Context(nch, frames)
    pairs = []
    for c in range(nch):
        p = ChainPair(ctx, c, oracle)
        full_chain(p, os_index, synth_ir(taps, seed=4242 + c), synth_ir(taps, seed=4243 + c) if two_irs else None)
        pairs.append(p)
    x = np.stack([synth_signal(c, frames * blocks, sr) for c in range(nch)])
    got, want = run_pairs(ctx, pairs, x, frames, sr)
    check(got, want)
    ctx.close()


def test_empty_chain_bypass_and_heterogeneous_channels(pkg, oracle):
    sr, frames = 48000, 512
    ctx = pkg.Context(4, frames)
    pairs = [ChainPair(ctx, c, oracle) for c in range(4)]
    # channel 0: empty chain (copy); channel 1: everything bypassed; 2: FIR first; 3: two FIRs back to back
    pairs[1].append("overdrive", bypass=True)
    pairs[1].append("power_amp", fir=synth_ir(100), bypass=True)
    pairs[2].append("power_amp", fir=synth_ir(900))
    pairs[2].append("tone_stack")
    pairs[3].append("cabinet")
    pairs[3].append("power_amp", fir=synth_ir(300, seed=1))
    pairs[3].append("power_amp", fir=synth_ir(2000, seed=2))
    x = np.stack([synth_signal(c, frames * 4, sr) for c in range(4)])
    got, want = run_pairs(ctx, pairs, x, frames, sr)
    check(got, want)
    np.testing.assert_array_equal(got[0], x[0])
    np.testing.assert_array_equal(got[1], x[1])
    ctx.close()


def test_bypassed_unit_keeps_its_state_and_reorder_keeps_state(pkg, oracle):
    sr, frames = 48000, 1024
    ctx = pkg.Context(1, frames)
    p = ChainPair(ctx, 0, oracle)
    h0 = p.append("tone_stack")
    h1 = p.append("compressor")
    x = synth_signal(0, frames * 6, sr)[None, :]
    g1, w1 = run_pairs(ctx, [p], x[:, :2 * frames], frames, sr)
    ctx.chain_set(0, [h0, h1], [True, False]); p.ref.set_bypass(0, True)          # bypass: state frozen
    g2, w2 = run_pairs(ctx, [p], x[:, 2 * frames:4 * frames], frames, sr)
    ctx.chain_set(0, [h1, h0], [False, False]); p.ref.set_bypass(0, False); p.ref.move_down(0)   # MoveDown: state travels
    g3, w3 = run_pairs(ctx, [p], x[:, 4 * frames:], frames, sr)
    check(np.hstack([g1, g2, g3]), np.hstack([w1, w2, w3]))
    ctx.close()


def test_every_unit_type_runs_on_hip_or_fails_loudly(pkg):
    """No silent fallback: a unit either runs on the HIP path or the call returns GDG_ERR_UNSUPPORTED."""
    x = np.zeros((1, 256))
    for name in pkg.UNIT_NAMES:
        ctx = pkg.Context(1, 256)
        ctx.append_unit(0, name)
        try:
            ctx.process(x, 48000)
        except pkg.GdgError as e:
            assert e.code == pkg.GDG_ERR_UNSUPPORTED, name
        ctx.close()
