"""Per-frame calls of few channels make the wet path of a reverb AHEAD of the frame's own samples (seg.hip REVERB_AHEAD / REVERB_CONSUME,
option seg_reverb_ahead_max_channels): the tapped sums and the all-passes need nothing of the frame when every tap lies at least a frame
back, so extra workgroups of an EARLIER segment launch of the same call make them beside the channels' own workgroups and the unit, behind
the power amps, only mixes.  The same expressions on the same operands: the same BITS as the unit run in one piece -- through every kind of
call in between -- and the oracle within 1e-9."""
import numpy as np
import pytest

import __graft_entry__ as entry
from helpers import synth_ir, synth_signal, rms, TOL_RMS

pytestmark = pytest.mark.gpu

FRAMES = 8192


@pytest.fixture(scope="module")
def pkg():
    return entry.load_package()


@pytest.fixture(scope="module")
def oracle():
    o = entry.load_oracle()
    o.build()
    return o


CHAINS = {
    "cabinet_reverb": [("cabinet", None), ("reverb", [50])],
    "reverb_first_of_two": [("reverb", [30]), ("tone_stack", None), ("reverb", [80])],
    "bench": [("compressor", [1, 30, -20]), ("overdrive", [0, 20, 100, 0, 1, 0]), ("tone_stack", None), ("chorus", None),
              ("power_amp", "a"), ("power_amp", "b"), ("cabinet", None), ("reverb", [50])],
    "reverb_between_amps": [("power_amp", "a"), ("reverb", [65]), ("power_amp", "b")],
    "two_hosted": [("chorus", None), ("power_amp", "a"), ("reverb", [40]), ("power_amp", "b"), ("cabinet", None), ("reverb", [70])],
    "config3_shape": [("compressor", [1, 30, -20]), ("overdrive", [0, 20, 100, 0, 1, 2]), ("tone_stack", None), ("chorus", None),
                      ("power_amp", "a"), ("cabinet", None), ("reverb", [50])],
}


def build(pkg, nch, chain, ahead, frames=FRAMES):
    ctx = pkg.Context(nch, frames)
    ctx.set_option("seg_reverb_ahead_max_channels", 80 if ahead else 0)
    for c in range(nch):
        for name, p in chain:
            if isinstance(p, str):
                ctx.append_unit(c, name, fir=synth_ir(20000, seed=11 + 2 * c + (p == "b")))
            else:
                ctx.append_unit(c, name, params=p)
    return ctx


def stream(ctx, x, sr, blocks, events=None):
    nch = x.shape[0]
    got = np.zeros_like(x)
    d_in, d_out = ctx.alloc(nch, FRAMES), ctx.alloc(nch, FRAMES)
    for b in range(blocks):
        if events and b in events:
            events[b](ctx)
        d_in.upload(x[:, b * FRAMES:(b + 1) * FRAMES])
        ctx.process_device(d_in, d_out, FRAMES, sr)
        got[:, b * FRAMES:(b + 1) * FRAMES] = d_out.download()
    ctx.synchronize()
    return got


@pytest.mark.parametrize("sr", [44100, 48000, 96000, 192000])
@pytest.mark.parametrize("name", sorted(CHAINS))
def test_ahead_gives_the_bits_of_the_unit_in_one_piece_and_follows_the_oracle(pkg, oracle, name, sr):
    nch, blocks = 3, 7
    chain = CHAINS[name]
    x = np.stack([synth_signal(c + 2, FRAMES * blocks, sr) * (1.0 if c else 0.2) for c in range(nch)])
    outs = {}
    for ahead in (False, True):
        ctx = build(pkg, nch, chain, ahead)
        outs[ahead] = stream(ctx, x, sr, blocks)
        ctx.close()
    np.testing.assert_array_equal(outs[True], outs[False])
    for c in range(nch):
        ref = oracle.Chain()
        for uname, p in chain:
            if isinstance(p, str):
                ref.append_unit(uname, fir=synth_ir(20000, seed=11 + 2 * c + (p == "b")))
            else:
                ref.append_unit(uname, params=p)
        want = np.concatenate([ref.process(x[c, b * FRAMES:(b + 1) * FRAMES], sr) for b in range(blocks)])
        assert rms(outs[True][c] - want) <= TOL_RMS, (name, sr, c)


def test_rates_outside_the_shape_fall_back_to_the_whole_unit(pkg):
    """32 kHz: the first tap (6143 samples) reaches into the frame -- reverb_ahead_ok says no; 200 kHz: the first all-pass ring (8407) is
    longer than a frame (the ring keeps a position): results as ever"""
    for sr in (32000, 200000):
        x = np.stack([synth_signal(c, FRAMES * 4, sr) for c in range(2)])
        outs = {}
        for ahead in (False, True):
            ctx = build(pkg, 2, CHAINS["cabinet_reverb"], ahead)
            outs[ahead] = stream(ctx, x, sr, 4)
            ctx.close()
        np.testing.assert_array_equal(outs[True], outs[False])


def test_every_kind_of_call_and_edit_between_frames(pkg):
    """One stream of 14 frames with, between frames: a knob move (mix: applied at the mix, what was made ahead stays good), a reset of the
    reverb, the reverb bypassed for two frames (it keeps its state: signal.go:390-401) and back, a reset of ANOTHER unit, a synchronize, a window
    of two frames in the middle, a frame of another size, another sample rate.  With and without the option: the same bits."""
    nch, sr = 2, 96000
    chain = CHAINS["bench"]
    blocks = 14
    x = np.stack([synth_signal(c + 7, FRAMES * (blocks + 4), sr) for c in range(nch)])
    outs = {}
    for ahead in (False, True):
        ctx = build(pkg, nch, chain, ahead)
        ctx.set_window(2)
        rev = [ctx._chains[c][-1][0] for c in range(nch)]
        cab = [ctx._chains[c][-2][0] for c in range(nch)]

        def bypass(on, ctx=ctx):
            for c in range(nch):
                hs = [h for h, _ in ctx._chains[c]]
                ctx.chain_set(c, hs, [False] * (len(hs) - 1) + [on])

        events = {
            2: lambda ctx: [ctx.unit_set_param(h, 0, 80) for h in rev],
            4: lambda ctx: [ctx.unit_reset(h) for h in rev],
            6: lambda ctx: bypass(True),
            8: lambda ctx: bypass(False),
            9: lambda ctx: ctx.unit_reset(cab[0]),
            10: lambda ctx: ctx.synchronize(),
        }
        got = [stream(ctx, x, sr, 12, events)]
        # a window of two frames, then per-frame calls again
        d_in, d_out = ctx.alloc(nch, 2 * FRAMES), ctx.alloc(nch, 2 * FRAMES)
        d_in.upload(x[:, 12 * FRAMES:14 * FRAMES])
        ctx.process_window_device(d_in.ptr, d_out.ptr, 2 * FRAMES, 2, sr)
        got.append(d_out.download())
        got.append(stream(ctx, x[:, 14 * FRAMES:], sr, 2))
        # a frame of another size (host path), then the batch size again, then another rate
        got.append(ctx.process(x[:, :4096], sr))
        got.append(stream(ctx, x[:, 16 * FRAMES:], sr, 1))
        got.append(stream(ctx, x[:, 17 * FRAMES:], 48000, 1))
        outs[ahead] = np.concatenate(got, axis=1)
        ctx.close()
    np.testing.assert_array_equal(outs[True], outs[False])
    assert np.isfinite(outs[True]).all() and np.abs(outs[True]).max() > 0.01


def test_subsets_and_host_buffer_calls_keep_their_bits(pkg):
    """gdg_process (host buffers) and gdg_process_subset (another set of channels: another plan) between device-resident per-frame calls"""
    nch, sr = 4, 192000
    x = np.stack([synth_signal(c + 1, FRAMES * 8, sr) for c in range(nch)])
    outs = {}
    for ahead in (False, True):
        ctx = build(pkg, nch, CHAINS["cabinet_reverb"], ahead)
        got = [stream(ctx, x, sr, 2)]
        got.append(ctx.process(x[:, 2 * FRAMES:3 * FRAMES], sr))
        got.append(ctx.process(x[:, 3 * FRAMES:4 * FRAMES], sr))
        sub = ctx.process_subset([1, 3], x[[1, 3], 4 * FRAMES:5 * FRAMES], sr)
        got.append(stream(ctx, x[:, 5 * FRAMES:], sr, 3))
        outs[ahead] = (np.concatenate(got, axis=1), sub)
        ctx.close()
    np.testing.assert_array_equal(outs[True][0], outs[False][0])
    np.testing.assert_array_equal(outs[True][1], outs[False][1])
