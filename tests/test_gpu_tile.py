"""Per-frame calls of few channels run a segment of compressor / shapers / tone stack / cabinet / chorus with a channel's 8192-sample frame on TWO
workgroups (seg.hip compiled with -DSEG_TILE, option seg_tile_max_channels).  The scans keep the general kernel's association -- a scan's
sixteen wave totals meet in ONE 16-lane scan, eight of them arriving from the other workgroup through HBM --, the chorus's LFO values are the
general kernel's threads' values, so the results must be the general kernel's BITS: frame by frame, through knob moves, resets, other call
kinds in between, at every rate; and the oracle's within 1e-9."""
import numpy as np
import pytest

import __graft_entry__ as entry
from helpers import synth_ir, synth_signal, rms, TOL_RMS

pytestmark = pytest.mark.gpu
FRAMES = 8192

CHAINS = {
    "bench_seg0": [("compressor", [1, 30, -20]), ("overdrive", [0, 20, 100, 0, 1, 0]), ("tone_stack", None), ("chorus", None)],
    "peak_follower_cabinet": [("compressor", [0, 20, -10]), ("cabinet", None), ("distortion", [0, 10, -3, 0]), ("excess", [12, -6, 0])],
    "two_choruses": [("chorus", [40, 70]), ("tone_stack", [3, -4, 2, -6]), ("chorus", None), ("cabinet", None)],
    "bench": [("compressor", [1, 30, -20]), ("overdrive", [0, 20, 100, 0, 1, 0]), ("tone_stack", None), ("chorus", None),
              ("power_amp", "a"), ("power_amp", "b"), ("cabinet", None), ("reverb", [50])],
    "tile_behind_an_amp": [("reverb", [30]), ("power_amp", "a"), ("cabinet", None), ("compressor", None), ("chorus", None)],
    "lone_shaper": [("overdrive", [0, 20, 100, 0, 0, 0])],
}


@pytest.fixture(scope="module")
def pkg():
    return entry.load_package()


@pytest.fixture(scope="module")
def oracle():
    o = entry.load_oracle()
    o.build()
    return o


def build(pkg, nch, chain, tile):
    ctx = pkg.Context(nch, FRAMES)
    ctx.set_option("seg_tile_max_channels", 112 if tile else 0)
    for c in range(nch):
        for name, p in chain:
            if isinstance(p, str):
                ctx.append_unit(c, name, fir=synth_ir(12000, seed=21 + 2 * c + (p == "b")))
            else:
                ctx.append_unit(c, name, params=p)
    return ctx


def stream(ctx, x, sr, blocks, events=None):
    nch = x.shape[0]
    got = np.zeros_like(x[:, :blocks * FRAMES])
    d_in, d_out = ctx.alloc(nch, FRAMES), ctx.alloc(nch, FRAMES)
    for b in range(blocks):
        if events and b in events:
            events[b](ctx)
        d_in.upload(x[:, b * FRAMES:(b + 1) * FRAMES])
        ctx.process_device(d_in, d_out, FRAMES, sr)
        got[:, b * FRAMES:(b + 1) * FRAMES] = d_out.download()
    ctx.synchronize()
    return got


@pytest.mark.parametrize("sr", [44100, 96000, 192000])
@pytest.mark.parametrize("name", sorted(CHAINS))
def test_two_workgroups_give_the_bits_of_one(pkg, oracle, name, sr):
    nch, blocks = 5, 6
    chain = CHAINS[name]
    x = np.stack([synth_signal(c + 1, FRAMES * blocks, sr) * (0.05 if c == 1 else 1.0) for c in range(nch)])
    x[2, FRAMES:2 * FRAMES] = 0.0                                   # a silent frame in mid-stream
    outs = {}
    for tile in (False, True):
        ctx = build(pkg, nch, chain, tile)
        outs[tile] = stream(ctx, x, sr, blocks)
        ctx.close()
    np.testing.assert_array_equal(outs[True], outs[False])
    for c in (0, nch - 1):
        ref = oracle.Chain()
        for uname, p in chain:
            if isinstance(p, str):
                ref.append_unit(uname, fir=synth_ir(12000, seed=21 + 2 * c + (p == "b")))
            else:
                ref.append_unit(uname, params=p)
        want = np.concatenate([ref.process(x[c, b * FRAMES:(b + 1) * FRAMES], sr) for b in range(blocks)])
        assert rms(outs[True][c] - want) <= TOL_RMS, (name, sr, c)


def test_knob_moves_resets_windows_and_other_frame_sizes_between_tiled_frames(pkg):
    nch, sr = 4, 192000
    chain = CHAINS["bench"]
    x = np.stack([synth_signal(c + 9, FRAMES * 14, sr) for c in range(nch)])
    outs = {}
    for tile in (False, True):
        ctx = build(pkg, nch, chain, tile)
        ctx.set_window(2)
        comp = [ctx._chains[c][0][0] for c in range(nch)]
        ts = [ctx._chains[c][2][0] for c in range(nch)]
        cho = [ctx._chains[c][3][0] for c in range(nch)]
        events = {
            2: lambda ctx: [ctx.unit_set_param(h, 1, -7) for h in ts],                  # a tone-stack band: new scan tables
            3: lambda ctx: [ctx.unit_set_param(h, 0, 0) for h in comp],                 # the follower: level -> peak (max-affine scan)
            5: lambda ctx: [ctx.unit_reset(h) for h in cho],
            6: lambda ctx: [ctx.unit_set_param(h, 0, 30) for h in cho],                 # chorus depth
            8: lambda ctx: ctx.synchronize(),
        }
        got = [stream(ctx, x, sr, 10, events)]
        d_in, d_out = ctx.alloc(nch, 2 * FRAMES), ctx.alloc(nch, 2 * FRAMES)
        d_in.upload(x[:, 10 * FRAMES:12 * FRAMES])
        ctx.process_window_device(d_in.ptr, d_out.ptr, 2 * FRAMES, 2, sr)              # a window (WAVE / walk) in between
        got.append(d_out.download())
        got.append(ctx.process(x[:, 12 * FRAMES:12 * FRAMES + 4096], sr))              # another frame size
        got.append(stream(ctx, x[:, 13 * FRAMES:], sr, 1))
        outs[tile] = np.concatenate(got, axis=1)
        ctx.close()
    np.testing.assert_array_equal(outs[True], outs[False])
    assert np.isfinite(outs[True]).all() and np.abs(outs[True]).max() > 0.01


def test_a_channel_with_another_unit_keeps_the_whole_step_on_the_general_kernel(pkg):
    """one launch per step: a flanger in ONE channel's segment and the step is not tiled -- results as ever"""
    nch, sr = 3, 96000
    x = np.stack([synth_signal(c + 4, FRAMES * 4, sr) for c in range(nch)])
    outs = {}
    for tile in (False, True):
        ctx = pkg.Context(nch, FRAMES)
        ctx.set_option("seg_tile_max_channels", 112 if tile else 0)
        for c in range(nch):
            ctx.append_unit(c, "compressor")
            ctx.append_unit(c, "flanger" if c == 1 else "tone_stack")
            ctx.append_unit(c, "chorus")
        outs[tile] = stream(ctx, x, sr, 4)
        ctx.close()
    np.testing.assert_array_equal(outs[True], outs[False])


def test_many_channels_and_a_long_stream(pkg):
    """64 channels (128 tile workgroups + 64 reverb workgroups in one launch), 12 frames: every channel, every sample"""
    nch, sr, blocks = 64, 192000, 12
    chain = CHAINS["bench"]
    x = np.stack([synth_signal(c, FRAMES * blocks, sr) for c in range(nch)])
    outs = {}
    for tile in (False, True):
        ctx = build(pkg, nch, chain, tile)
        outs[tile] = stream(ctx, x, sr, blocks)
        ctx.close()
    np.testing.assert_array_equal(outs[True], outs[False])
