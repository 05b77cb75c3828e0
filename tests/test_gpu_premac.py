"""Small shards, per-frame calls: the sums of the NEXT frame's convolution over the partitions already in the delay line are launched when a call
ends (premac, api_process.cpp) and the next call only adds the newest term.  Every multiply-accumulate kernel sums k descending, so the split sum must
have the bits of the whole one -- and anything that touches the context between two calls must drop the speculative part."""
import numpy as np
import pytest

import __graft_entry__ as entry
from helpers import ChainPair, synth_ir, synth_signal, rms, TOL_RMS

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    return entry.load_package()


@pytest.fixture(scope="module")
def oracle():
    o = entry.load_oracle()
    o.build()
    return o


def build(pkg, nch, frames, taps, premac, two_amps=True):
    ctx = pkg.Context(nch, frames)
    ctx.set_option("fir_premac", 1 if premac else 0)
    ctx.set_option("fir_premac_min_partitions", 1)
    for c in range(nch):
        ctx.append_unit(c, "compressor")
        ctx.append_unit(c, "power_amp", fir=synth_ir(taps, seed=100 + c))
        if two_amps:
            ctx.append_unit(c, "power_amp", fir=synth_ir(taps // 2 + 1000, seed=200 + c))
        ctx.append_unit(c, "cabinet")
    return ctx


@pytest.mark.parametrize("two_amps", [True, False])
def test_premac_gives_the_bits_of_the_whole_sum(pkg, two_amps):
    nch, frames, sr, taps, blocks = 3, 8192, 192000, 40000, 7
    x = np.stack([synth_signal(c, frames * blocks, sr) for c in range(nch)])
    outs = {}
    for premac in (False, True):
        ctx = build(pkg, nch, frames, taps, premac, two_amps)
        d_in, d_out = ctx.alloc(nch, frames), ctx.alloc(nch, frames)
        got = np.zeros_like(x)
        for b in range(blocks):
            d_in.upload(x[:, b * frames:(b + 1) * frames])
            ctx.process_device(d_in, d_out, frames, sr)
            got[:, b * frames:(b + 1) * frames] = d_out.download()
        outs[premac] = got
        ctx.close()
    np.testing.assert_array_equal(outs[True], outs[False])


def test_premac_is_dropped_by_whatever_touches_the_context(pkg, oracle):
    """Between two process calls: a parameter change, a reset of the power amp (its delay line is zeroed: sums made from the old one must not be
    used), a new filter, a bypass, a frame-size change, a host-buffer call -- the stream must follow the oracle through all of them."""
    nch, frames, sr = 2, 8192, 96000
    ctx = pkg.Context(nch, frames)
    ctx.set_option("fir_premac_min_partitions", 1)
    pairs = []
    for c in range(nch):
        p = ChainPair(ctx, c, oracle)
        p.append("tone_stack")
        p.append("power_amp", fir=synth_ir(30000, seed=7 + c))            # channel 1's is reset below: the oracle's way to a fresh filter is setting it again
        p.append("reverb")
        pairs.append(p)
    x = np.stack([synth_signal(c, frames * 16, sr) for c in range(nch)])
    d_in, d_out = ctx.alloc(nch, frames), ctx.alloc(nch, frames)
    got, want = np.zeros_like(x), np.zeros_like(x)

    def step(b, n=frames):
        blk = x[:, b * frames:b * frames + n]
        if n == frames:
            d_in.upload(blk)
            ctx.process_device(d_in, d_out, frames, sr)
            got[:, b * frames:(b + 1) * frames] = d_out.download()
        else:
            got[:, b * frames:b * frames + n] = ctx.process(blk, sr)
        for c in range(nch):
            want[c, b * frames:b * frames + n] = pairs[c].ref.process(blk[c], sr)

    step(0); step(1)
    ctx.unit_set_param(pairs[0].handles[0], 1, -7); pairs[0].ref.unit(0).set_params([0, -7, -5, -5])
    step(2); step(3)
    h = pairs[1].handles[1]
    ctx.unit_reset(h); pairs[1].ref.unit(1).set_fir(synth_ir(30000, seed=8))       # poweramp.go:132-181: a set replaces the filter, state and all
    step(4); step(5)
    ir = synth_ir(12000, seed=99)
    ctx.unit_set_fir(pairs[0].handles[1], ir); pairs[0].ref.unit(1).set_fir(ir)
    step(6); step(7)
    ctx.chain_set(1, pairs[1].handles, [False, True, False]); pairs[1].ref.set_bypass(1, True)
    step(8)
    ctx.chain_set(1, pairs[1].handles, [False, False, False]); pairs[1].ref.set_bypass(1, False)
    step(9); step(10)
    step(11, 4096)                                     # another frame size (host-buffer call): the delay line is re-partitioned
    for c in range(nch):
        assert rms(got[c, :11 * frames + 4096] - want[c, :11 * frames + 4096]) <= TOL_RMS, c
    ctx.close()
