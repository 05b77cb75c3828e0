"""Calls of few channels run every 2 x / 4 x oversampled shaper as a launch of its own, a workgroup per (channel, frame, tile) -- seg.hip
os_tiles_kernel, option seg_os_tiles_max_channels.  What the in-segment unit keeps as state between frames is recomputed there from the previous
frame's inputs, so the two forms must give the same BITS, frame by frame and in windows, through changes of the factor and of the call shape."""
import numpy as np
import pytest

import __graft_entry__ as entry
from helpers import synth_ir, synth_signal, rms, TOL_RMS

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    return entry.load_package()


@pytest.fixture(scope="module")
def oracle():
    o = entry.load_oracle()
    o.build()
    return o


CHAINS = {
    "overdrive4": [("compressor", None), ("overdrive", [0, 20, 100, 0, 1, 2]), ("tone_stack", None), ("chorus", None)],
    "distortion2_first": [("distortion", [0, 20, -3, 1]), ("cabinet", None)],
    "excess4_last_after_fir": [("tone_stack", None), ("power_amp", "ir"), ("excess", [12, -6, 2])],
    "two_shapers": [("overdrive", [5, 10, 50, -6, 0, 1]), ("distortion", [10, 10, 0, 2]), ("reverb", None)],
    "shaper_only": [("overdrive", [0, 20, 100, 0, 1, 2])],
}


def build(pkg, nch, frames, chain, tiles):
    ctx = pkg.Context(nch, frames)
    ctx.set_option("seg_os_tiles_max_channels", 192 if tiles else 0)
    for c in range(nch):
        for name, p in chain:
            if p == "ir":
                ctx.append_unit(c, name, fir=synth_ir(9000, seed=3 + c))
            else:
                ctx.append_unit(c, name, params=p)
    return ctx


@pytest.mark.parametrize("name", sorted(CHAINS))
def test_tiles_give_the_bits_of_the_in_segment_unit_and_follow_the_oracle(pkg, oracle, name):
    nch, frames, sr, blocks = 3, 8192, 96000, 6
    chain = CHAINS[name]
    x = np.stack([synth_signal(c + 1, frames * blocks, sr) * (1.0 if c else 0.3) for c in range(nch)])
    outs = {}
    for tiles in (False, True):
        ctx = build(pkg, nch, frames, chain, tiles)
        got = np.zeros_like(x)
        d_in, d_out = ctx.alloc(nch, frames), ctx.alloc(nch, frames)
        for b in range(blocks):
            d_in.upload(x[:, b * frames:(b + 1) * frames])
            ctx.process_device(d_in, d_out, frames, sr)
            got[:, b * frames:(b + 1) * frames] = d_out.download()
        outs[tiles] = got
        ctx.close()
    np.testing.assert_array_equal(outs[True], outs[False])
    for c in range(nch):
        ref = oracle.Chain()
        for uname, p in chain:
            if p == "ir":
                ref.append_unit(uname, fir=synth_ir(9000, seed=3 + c))
            else:
                ref.append_unit(uname, params=p)
        want = np.concatenate([ref.process(x[c, b * frames:(b + 1) * frames], sr) for b in range(blocks)])
        assert rms(outs[True][c] - want) <= TOL_RMS, (name, c)


def test_windows_frames_and_a_change_of_factor_are_one_stream(pkg):
    """frames 0-1 per call, 2-5 as a window of 4, 6 per call, 7-10 as a window of 4 with the factor changed from 4 x to 2 x before it, 11 per call:
    the tiles' recomputed tails, the state they leave and the state they find must be the in-segment unit's, bit for bit"""
    nch, frames, sr, blocks = 2, 8192, 192000, 12
    chain = CHAINS["overdrive4"]
    x = np.stack([synth_signal(c + 5, frames * blocks, sr) for c in range(nch)])
    outs = {}
    for tiles in (False, True):
        ctx = build(pkg, nch, frames, chain, tiles)
        ctx.set_window(4)
        got = np.zeros_like(x)
        d_in, d_out = ctx.alloc(nch, 4 * frames), ctx.alloc(nch, 4 * frames)

        def per_frame(b):
            blk = np.zeros((nch, 4 * frames))
            blk[:, :frames] = x[:, b * frames:(b + 1) * frames]
            d_in.upload(blk)
            ctx.process_window_device(d_in.ptr, d_out.ptr, 4 * frames, 1, sr)
            got[:, b * frames:(b + 1) * frames] = d_out.download()[:, :frames]

        def window(b):
            d_in.upload(x[:, b * frames:(b + 4) * frames])
            ctx.process_window_device(d_in.ptr, d_out.ptr, 4 * frames, 4, sr)
            got[:, b * frames:(b + 4) * frames] = d_out.download()

        per_frame(0); per_frame(1); window(2); per_frame(6)
        for c in range(nch):
            ctx.unit_set_param(ctx._chains[c][1][0], 5, 1)       # the channel's overdrive (second unit of its chain): oversampling "2"
        window(7); per_frame(11)
        outs[tiles] = got
        ctx.close()
    np.testing.assert_array_equal(outs[True], outs[False])
    assert np.isfinite(outs[True]).all() and np.abs(outs[True]).max() > 0.01


def test_switching_oversampling_off_and_on_in_window_mode_relays_the_counters(pkg):
    """ADVICE r05 (api_plan.cpp, d_wave): switching oversampling OFF merges [tiles launch | segment] into one segment, so the counters of its
    gated units move onto words the old layout used for tile flags (which keep their launch's epoch) -- every plan must start from zeroed
    counters.  64 channels of [distortion 4 x, tone stack, chorus] in windows, oversampling off, on (2 x), off again: no stall, no error word,
    and the bits of the per-frame walk of the same stream."""
    nch, frames, sr, W = 64, 8192, 192000, 4
    chain = [("distortion", [0, 20, -3, 2]), ("tone_stack", None), ("chorus", None)]
    x = np.stack([synth_signal(c + 1, frames * W * 4, sr) for c in range(nch)])
    outs = {}
    for windowed in (False, True):
        ctx = build(pkg, nch, frames, chain, True)
        ctx.set_window(W)
        got = np.zeros_like(x)
        d_in, d_out = ctx.alloc(nch, W * frames), ctx.alloc(nch, W * frames)
        for w, os_index in enumerate((2, 0, 1, 0)):
            for c in range(nch):
                ctx.unit_set_param(ctx._chains[c][0][0], 3, os_index)
            lo = w * W * frames
            if windowed:
                d_in.upload(x[:, lo:lo + W * frames])
                ctx.process_window_device(d_in.ptr, d_out.ptr, W * frames, W, sr)
                ctx.synchronize()                                    # raises on the device error word
                got[:, lo:lo + W * frames] = d_out.download()
            else:
                for f in range(W):
                    blk = np.zeros((nch, W * frames))
                    blk[:, :frames] = x[:, lo + f * frames:lo + (f + 1) * frames]
                    d_in.upload(blk)
                    ctx.process_window_device(d_in.ptr, d_out.ptr, W * frames, 1, sr)
                    got[:, lo + f * frames:lo + (f + 1) * frames] = d_out.download()[:, :frames]
                ctx.synchronize()
        outs[windowed] = got
        ctx.close()
    np.testing.assert_array_equal(outs[True], outs[False])
    assert np.isfinite(outs[True]).all() and np.abs(outs[True]).max() > 0.01


@pytest.mark.parametrize("factor_index", [1, 2])
@pytest.mark.parametrize("follow", [0, 1])
def test_a_lone_compressor_in_front_runs_inside_the_tiles_launch(pkg, oracle, follow, factor_index):
    """compressor > oversampled overdrive (BASELINE config 3's head): a per-frame call does not launch the compressor's step, every tile's workgroup
    runs the unit itself (option seg_os_tiles_prefix) -- the compressor's own code on the same frame: the same bits, with its state handed on
    from frame to frame, through windows in between and a reverb behind a power amp that wants an earlier launch to carry its wet path"""
    nch, frames, sr, blocks = 3, 8192, 96000, 7
    chain = [("compressor", [follow, 30, -20]), ("overdrive", [0, 20, 100, 0, 1, factor_index]), ("tone_stack", None), ("chorus", None),
             ("power_amp", "ir"), ("cabinet", None), ("reverb", [50])]
    x = np.stack([synth_signal(c + 1, frames * blocks, sr) * (1.0 if c else 0.3) for c in range(nch)])
    outs = {}
    for prefix in (0, 1):
        ctx = build(pkg, nch, frames, chain, True)
        ctx.set_option("seg_os_tiles_prefix", prefix)
        ctx.set_window(2)
        got = np.zeros_like(x)
        d_in, d_out = ctx.alloc(nch, 2 * frames), ctx.alloc(nch, 2 * frames)
        f_in, f_out = ctx.alloc(nch, frames), ctx.alloc(nch, frames)

        def per_frame(b):
            f_in.upload(x[:, b * frames:(b + 1) * frames])
            ctx.process_device(f_in, f_out, frames, sr)
            got[:, b * frames:(b + 1) * frames] = f_out.download()

        for b in (0, 1, 2):
            per_frame(b)
        d_in.upload(x[:, 3 * frames:5 * frames])
        ctx.process_window_device(d_in.ptr, d_out.ptr, 2 * frames, 2, sr)          # a window: the compressor's step is launched as ever
        got[:, 3 * frames:5 * frames] = d_out.download()
        for b in (5, 6):
            per_frame(b)
        ctx.synchronize()
        outs[prefix] = got
        ctx.close()
    bad = [b for b in range(blocks) if not np.array_equal(outs[1][:, b * frames:(b + 1) * frames], outs[0][:, b * frames:(b + 1) * frames])]
    assert bad == [], "frames that differ: %s" % bad
    np.testing.assert_array_equal(outs[1], outs[0])
    ref = oracle.Chain()
    for uname, p in chain:
        if p == "ir":
            ref.append_unit(uname, fir=synth_ir(9000, seed=3 + 2))
        else:
            ref.append_unit(uname, params=p)
    want = np.concatenate([ref.process(x[2, b * frames:(b + 1) * frames], sr) for b in range(blocks)])
    assert rms(outs[1][2] - want) <= TOL_RMS
