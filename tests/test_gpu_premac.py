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


@pytest.mark.parametrize("two_amps", [True, False])
@pytest.mark.parametrize("nch,taps", [(3, 40000), (5, 65536), (2, 200000)])
def test_back_to_back_calls_use_the_sums_made_ahead_and_keep_the_bits(pkg, oracle, two_amps, nch, taps):
    """The test above uploads a frame between two calls -- a library call, which DROPS the sums made ahead: it compares the path that throws them
    away.  Here every frame is resident before the first call and the calls follow each other with nothing in between (what the bench and the
    batch loop do): the inverse launches must CONTINUE the sums (counter stat_premac_launches_used), with the bits of a context that never
    makes any, and follow the oracle."""
    frames, sr, blocks = 8192, 192000, 6
    x = np.stack([synth_signal(c + 5, frames * blocks, sr) for c in range(nch)])
    outs, used = {}, {}
    for premac in (False, True):
        ctx = build(pkg, nch, frames, taps, premac, two_amps)
        d_in = [ctx.alloc(nch, frames) for _ in range(blocks)]
        d_out = [ctx.alloc(nch, frames) for _ in range(blocks)]
        for b in range(blocks):
            d_in[b].upload(x[:, b * frames:(b + 1) * frames])
        for b in range(blocks):
            ctx.process_device(d_in[b], d_out[b], frames, sr)
        ctx.synchronize()
        used[premac] = ctx.get_option("stat_premac_launches_used")
        outs[premac] = np.concatenate([d.download() for d in d_out], axis=1)
        ctx.close()
    assert used[False] == 0
    assert used[True] == (blocks - 1) * (2 if two_amps else 1), used          # every call but the first, every power amp
    np.testing.assert_array_equal(outs[True], outs[False])
    ref = oracle.Chain()
    ref.append_unit("compressor")
    ref.append_unit("power_amp", fir=synth_ir(taps, seed=100))
    if two_amps:
        ref.append_unit("power_amp", fir=synth_ir(taps // 2 + 1000, seed=200))
    ref.append_unit("cabinet")
    want = np.concatenate([ref.process(x[0, b * frames:(b + 1) * frames], sr) for b in range(blocks)])
    assert rms(outs[True][0] - want) <= TOL_RMS


def test_a_64_channel_shard_of_the_bench_job_back_to_back(pkg, oracle):
    """One GPU's share of BASELINE config 4 on eight GPUs, as the bench runs it: the bench's chain (two 65536-tap power amps), 64 channels,
    resident frames, calls back to back -- sums made ahead, a frame on two workgroups and the reverb's wet path beside the first segment all at
    once.  The bits of a context with all three off, and the oracle on the first and the last channel."""
    import bench
    nch, frames, sr, taps, blocks = 64, 8192, 192000, 65536, 5
    xs = [bench.synth_block(nch, frames, sr) * (1.0 - 0.1 * b) for b in range(blocks)]
    outs = {}
    for name, opts in (("default", {}), ("plain", {"fir_premac": 0, "seg_tile_max_channels": 0, "seg_reverb_ahead_max_channels": 0})):
        ctx = bench.make_context(pkg, nch, frames, 0, taps)
        for k, v in opts.items():
            ctx.set_option(k, v)
        d_in = [ctx.alloc(nch, frames) for _ in range(blocks)]
        d_out = [ctx.alloc(nch, frames) for _ in range(blocks)]
        for b in range(blocks):
            d_in[b].upload(xs[b])
        for b in range(blocks):
            ctx.process_device(d_in[b], d_out[b], frames, sr)
        ctx.synchronize()
        assert ctx.get_option("stat_premac_launches_used") == (2 * (blocks - 1) if name == "default" else 0)
        outs[name] = [d.download() for d in d_out]
        ctx.close()
    for b in range(blocks):
        np.testing.assert_array_equal(outs["default"][b], outs["plain"][b])
    for c in (0, nch - 1):
        ref = oracle.Chain()
        for uname, p in bench.CHAIN:
            if isinstance(p, str):
                ref.append_unit(uname, fir=bench.ir_for(p, c, taps))
            else:
                ref.append_unit(uname, params=p)
        want = np.concatenate([ref.process(xs[b][c], sr) for b in range(blocks)])
        got = np.concatenate([outs["default"][b][c] for b in range(blocks)])
        assert rms(got - want) <= TOL_RMS, c


def test_the_split_shape_up_to_192_channels_and_the_premacs_lds_request_keep_the_bits(pkg):
    """160 channels with two power amps take the split multiply-accumulate with sums made ahead by default (fir_split_max_channels 192); the fused
    kernel (fir_fused 1), the split shape without sums ahead (fir_premac 0) and every LDS request of the premac's launch (fir_premac_lds_bytes: LDS
    the kernel never touches, only where its workgroups land) give the same bits."""
    nch, frames, sr, blocks = 160, 8192, 192000, 4
    x = np.stack([synth_signal(c % 7, frames * blocks, sr) * (0.3 + 0.004 * c) for c in range(nch)])
    irs = [synth_ir(30000, seed=40 + c % 5) for c in range(nch)], [synth_ir(20000, seed=90 + c % 3) for c in range(nch)]
    outs = {}
    for name, opts in (("default", {}), ("fused", {"fir_fused": 1}), ("no_premac", {"fir_premac": 0}),
                       ("lds_0", {"fir_premac_lds_bytes": 0}), ("lds_65536", {"fir_premac_lds_bytes": 65536})):
        ctx = pkg.Context(nch, frames)
        ctx.set_option("share_ir_spectra", 0)
        for k, v in opts.items():
            ctx.set_option(k, v)
        for c in range(nch):
            ctx.append_unit(c, "tone_stack")
            ctx.append_unit(c, "power_amp", fir=irs[0][c])
            ctx.append_unit(c, "power_amp", fir=irs[1][c])
        d_in = [ctx.alloc(nch, frames) for _ in range(blocks)]
        d_out = [ctx.alloc(nch, frames) for _ in range(blocks)]
        for b in range(blocks):
            d_in[b].upload(x[:, b * frames:(b + 1) * frames])
        for b in range(blocks):
            ctx.process_device(d_in[b], d_out[b], frames, sr)
        ctx.synchronize()
        used = ctx.get_option("stat_premac_launches_used")
        assert used == (2 * (blocks - 1) if name in ("default", "lds_0", "lds_65536") else 0), (name, used)
        outs[name] = np.concatenate([d.download() for d in d_out], axis=1)
        ctx.close()
    for name in outs:
        np.testing.assert_array_equal(outs[name], outs["default"], err_msg=name)
    assert np.isfinite(outs["default"]).all() and np.abs(outs["default"]).max() > 0.01


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
