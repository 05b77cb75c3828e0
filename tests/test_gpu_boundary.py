"""The other entry points of the boundary: subset, staged (cgo-friendly) and device-resident processing
must all give the same samples as gdg_process and the oracle.  Run with `pytest -m gpu`."""
import numpy as np
import pytest

from helpers import ChainPair, TOL_RMS, package, rms, synth_ir, synth_signal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    p = package()
    assert p.device_count() > 0
    return p


def build(pkg, oracle, nch, frames):
    ctx = pkg.Context(nch, frames)
    pairs = []
    for c in range(nch):
        p = ChainPair(ctx, c, oracle)
        p.append("compressor")
        p.append("power_amp", fir=synth_ir(700 + 100 * c, seed=c))
        p.append("cabinet")
        pairs.append(p)
    return ctx, pairs


def test_subset_leaves_other_channels_untouched(pkg, oracle):
    sr, frames, nch = 48000, 512, 4
    ctx, pairs = build(pkg, oracle, nch, frames)
    x = np.stack([synth_signal(c, frames * 4, sr) for c in range(nch)])
    schedule = [[0, 1, 2, 3], [2, 0], [1], [3, 1, 0, 2]]        # which channels take part in each block (any order)
    for b, chans in enumerate(schedule):
        blk = x[:, b * frames:(b + 1) * frames]
        got = ctx.process_subset(chans, blk[chans], sr)
        for i, c in enumerate(chans):
            want = pairs[c].ref.process(blk[c], sr)               # the oracle chain only advances when its channel ran
            assert rms(got[i] - want) <= TOL_RMS
    ctx.close()


def test_staged_equals_host_pointer_path(pkg, oracle):
    sr, frames, nch = 96000, 1024, 3
    ctx, pairs = build(pkg, oracle, nch, frames)
    x = np.stack([synth_signal(c, frames * 3, sr) for c in range(nch)])
    for b in range(3):
        blk = x[:, b * frames:(b + 1) * frames]
        chans = [0, 1, 2] if b != 1 else [2, 0]
        got = ctx.process_staged(chans, blk[chans], sr)
        for i, c in enumerate(chans):
            assert rms(got[i] - pairs[c].ref.process(blk[c], sr)) <= TOL_RMS
    ctx.close()


def test_device_resident_path(pkg, oracle):
    sr, frames, nch = 48000, 2048, 2
    ctx, pairs = build(pkg, oracle, nch, frames)
    d_in, d_out = ctx.alloc(nch, frames), ctx.alloc(nch, frames)
    x = np.stack([synth_signal(c, frames * 3, sr) for c in range(nch)])
    ctx.profile_enable(True)
    for b in range(3):
        blk = x[:, b * frames:(b + 1) * frames]
        d_in.upload(blk)
        ctx.process_device(d_in, d_out, frames, sr)
        ctx.synchronize()
        got = d_out.download()
        for c in range(nch):
            assert rms(got[c] - pairs[c].ref.process(blk[c], sr)) <= TOL_RMS
    ms, n = ctx.profile_read(pkg.K_FIR_MAC)
    assert n >= 3 and n % 3 == 0 and ms > 0.0                    # HIP-event timing of the MAC kernel is live (x channel groups)
    with pytest.raises(pkg.GdgError):
        ctx.process_device(d_in, d_in, frames, sr)               # in-place is rejected
    d_in.free(); d_out.free()
    ctx.close()


@pytest.mark.parametrize("taps,sizes", [(3000, [1024, 1024, 512, 512, 1024, 64, 64, 2048]),
                                        (65536, [8192, 8192, 1024, 1000, 1000, 8192, 8192]),
                                        (5000, [1000, 480, 480, 8192, 37, 4096, 1000, 1000]),
                                        (77, [256, 1024, 100, 100, 1]),
                                        (20000, [64, 8192, 64, 64, 8192])])
def test_frame_size_sequence_carries_the_convolution_state(pkg, oracle, taps, sizes):
    """filter.Process takes any N from call to call: tail and transform sizes depend on L only (filter/filter.go:370-389,
    :419-428), so a stream cut into blocks of changing size is ONE convolution.  The HIP path re-partitions its delay line."""
    sr = 48000
    ctx = pkg.Context(2, 8192)
    h = [synth_ir(taps, seed=77 + c) * (2.5 if c else 1.0) for c in range(2)]          # channel 1 clips
    refs = []
    for c in range(2):
        ctx.append_unit(c, "compressor")
        ctx.append_unit(c, "power_amp", fir=h[c])
        r = oracle.Chain()
        r.append_unit("compressor")
        r.append_unit("power_amp", fir=h[c])
        refs.append(r)
    x = np.stack([synth_signal(c, sum(sizes), sr) for c in range(2)])
    got, want = np.zeros_like(x), np.zeros_like(x)
    at = 0
    for n in sizes:
        blk = np.ascontiguousarray(x[:, at:at + n])
        got[:, at:at + n] = ctx.process(blk, sr)
        for c in range(2):
            want[c, at:at + n] = refs[c].process(blk[c], sr)
        at += n
    for c in range(2):
        assert rms(got[c] - want[c]) <= TOL_RMS, "channel %d: RMS %.3e" % (c, rms(got[c] - want[c]))
    ctx.close()


def test_frame_size_change_matches_direct_convolution(pkg):
    """Second, oracle-free formulation of the same property: y = clip(x * h) over the whole stream."""
    sr, taps, sizes = 96000, 9000, [2048, 300, 300, 8192, 1024, 5, 4096]
    ctx = pkg.Context(1, 8192)
    h = synth_ir(taps, seed=5)
    ctx.append_unit(0, "power_amp", fir=h)
    x = synth_signal(3, sum(sizes), sr)
    got, at = np.zeros_like(x), 0
    for n in sizes:
        got[at:at + n] = ctx.process(np.ascontiguousarray(x[None, at:at + n]), sr)[0]
        at += n
    direct = np.clip(np.convolve(x, h)[:x.size], -1.0, 1.0)
    assert rms(got - direct) <= TOL_RMS
    ctx.close()


def test_frame_size_the_reference_panics_on_is_rejected_with_a_message(pkg, oracle):
    """N = 600, L = 100: nextpow2(N) = 1024 is cut into 8 blocks of 128, the sixth starts at 640 > N and the reference panics on
    the slice bounds (filter/filter.go:443-453).  The oracle reports it, the HIP path rejects the pair and says why."""
    with pytest.raises(ValueError, match="rc=-3"):
        oracle.Filter(synth_ir(100), 48000).process(np.zeros(600))
    oracle.Filter(synth_ir(100), 48000).process(np.zeros(900))    # the last of the 8 blocks starts at 896 <= N: legal
    ctx = pkg.Context(1, 1024)
    ctx.append_unit(0, "power_amp", fir=synth_ir(100))
    with pytest.raises(pkg.GdgError) as e:
        ctx.process(np.zeros((1, 600)), 48000)
    assert e.value.code == pkg.GDG_ERR_UNSUPPORTED and "reference panics" in str(e.value) and "filter.go:443-453" in str(e.value)
    y = ctx.process(np.zeros((1, 512)), 48000)                   # the context stays usable: a legal size runs
    assert y.shape == (1, 512)
    y = ctx.process(np.zeros((1, 900)), 48000)                   # 7 x 128 = 896 <= 900: every block starts inside the frame
    assert y.shape == (1, 900)
    ctx.close()


def test_identical_filters_share_spectra_without_changing_results(pkg, oracle):
    """Power amps with identical taps share one copy of the IR spectra (gdg_ctx_share_ir_spectra): same bits as private
    copies; replacing or destroying one sharer leaves the others alone."""
    sr, frames, nch = 48000, 1024, 6
    ir_a, ir_b = synth_ir(5000, seed=1), synth_ir(5000, seed=2)
    x = np.stack([synth_signal(c, frames * 4, sr) for c in range(nch)])

    def run(share):
        ctx = pkg.Context(nch, frames)
        ctx.share_ir_spectra(share)
        hs = [ctx.append_unit(c, "power_amp", fir=ir_a if c % 2 == 0 else ir_b) for c in range(nch)]
        outs = [ctx.process(x[:, :frames], sr)]
        ctx.unit_set_fir(hs[2], ir_b)                            # channel 2 leaves the ir_a group (fresh state, like the reference)
        outs.append(ctx.process(x[:, frames:2 * frames], sr))
        ctx.chain_set(4, [])                                     # channel 4: power amp removed and destroyed
        ctx.unit_destroy(hs[4])
        outs.append(ctx.process(x[:, 2 * frames:3 * frames], sr))
        outs.append(ctx.process(x[:, 3 * frames:], sr))
        ctx.close()
        return np.concatenate(outs, axis=1)

    shared, private = run(True), run(False)
    np.testing.assert_array_equal(shared, private)
    # and both follow the oracle on a channel that kept its filter throughout
    ref = oracle.Chain()
    ref.append_unit("power_amp", fir=ir_a)
    want = np.concatenate([ref.process(x[0, b * frames:(b + 1) * frames], sr) for b in range(4)])
    assert rms(shared[0] - want) <= TOL_RMS


@pytest.mark.parametrize("nch,frames,taps", [(1, 8192, 65536), (8, 8192, 20000), (5, 1024, 5000), (3, 1000, 3000)])
def test_fir_launch_shapes_agree(pkg, oracle, nch, frames, taps):
    """Few channels take the split shape (bin-tiled multiply-accumulate over 32 workgroups per channel + inverse), many the fused
    kernel (one workgroup per channel); GDG_FIR_FUSED forces one.  Same results from both, and both match the oracle."""
    import os
    sr, blocks = 96000, 3
    x = np.stack([synth_signal(c, frames * blocks, sr) for c in range(nch)])
    irs = [synth_ir(taps, seed=300 + c) for c in range(nch)]
    outs = {}
    for forced in ("0", "1"):
        os.environ["GDG_FIR_FUSED"] = forced
        try:
            ctx = pkg.Context(nch, frames)                          # the environment is read when the context is created
        finally:
            del os.environ["GDG_FIR_FUSED"]
        for c in range(nch):
            ctx.append_unit(c, "power_amp", fir=irs[c])
        outs[forced] = np.concatenate([ctx.process(x[:, b * frames:(b + 1) * frames], sr) for b in range(blocks)], axis=1)
        ctx.close()
    assert np.array_equal(outs["0"], outs["1"])                  # all multiply-accumulate kernels: same order, no fused multiply-adds
    for c in range(nch):
        ref = oracle.Chain()
        ref.append_unit("power_amp", fir=irs[c])
        want = np.concatenate([ref.process(x[c, b * frames:(b + 1) * frames], sr) for b in range(blocks)])
        assert rms(outs["0"][c] - want) <= TOL_RMS and rms(outs["1"][c] - want) <= TOL_RMS


@pytest.mark.parametrize("nch,fused,share", [(3, "0", True), (3, "1", False), (130, None, True), (130, None, False)],
                         ids=["split_inverse", "fused_private", "auto_fused_shared", "auto_fused_private"])
def test_adjacent_power_amps_chain_their_transforms(pkg, oracle, nch, fused, share):
    """Two (and three) power amps in a row at the batch block size: the inverse transform of one produces the forward transform of the
    next inside the same launch (fir_inv_kernel CHAIN: the clipped frame goes from the inverse's registers into the next amp's history
    and delay line).  GDG_FIR_CHAIN=0 keeps the launches separate: both ways give the same bits, in every launch shape (split inverse,
    fused with private / shared spectra), across a change of the chain in mid-stream, and they follow the oracle."""
    import os
    frames, sr, blocks, taps = 8192, 96000, 6, 20000
    x = np.stack([synth_signal(c, frames * blocks, sr) for c in range(nch)])
    n_irs = 2 if share else nch
    irs = [[synth_ir(taps, seed=900 + 10 * k + j) * 0.9 for j in range(3)] for k in range(n_irs)]
    outs = {}
    for chain in ("1", "0"):
        os.environ["GDG_FIR_CHAIN"] = chain
        if fused is not None:
            os.environ["GDG_FIR_FUSED"] = fused
        try:
            ctx = pkg.Context(nch, frames)
        finally:
            del os.environ["GDG_FIR_CHAIN"]
            os.environ.pop("GDG_FIR_FUSED", None)
        ctx.share_ir_spectra(share)
        third = []
        for c in range(nch):
            ctx.append_unit(c, "compressor", params=[1, 30, -20])
            ctx.append_unit(c, "power_amp", fir=irs[c % n_irs][0])
            ctx.append_unit(c, "power_amp", fir=irs[c % n_irs][1])
            third.append(ctx.append_unit(c, "power_amp", fir=irs[c % n_irs][2], bypass=True))
            ctx.append_unit(c, "cabinet")
        d_in, d_out = ctx.alloc(nch, frames), ctx.alloc(nch, frames)
        got = []
        for b in range(blocks):
            if b == 3:                                           # the third amp joins: amp 2 now chains into amp 3 as well
                for c in range(nch):
                    ctx.chain_set(c, [h for h, _ in ctx._chains[c]], [False] * 5)
            d_in.upload(np.ascontiguousarray(x[:, b * frames:(b + 1) * frames]))
            ctx.process_device(d_in, d_out, frames, sr)
            got.append(d_out.download())
        outs[chain] = np.concatenate(got, axis=1)
        ctx.close()
    assert np.array_equal(outs["1"], outs["0"]), float(np.max(np.abs(outs["1"] - outs["0"])))
    for c in (0, nch - 1):
        ref = oracle.Chain()
        ref.append_unit("compressor", params=[1, 30, -20])
        for j in range(3):
            ref.append_unit("power_amp", fir=irs[c % n_irs][j], bypass=(j == 2))
        ref.append_unit("cabinet")
        want = []
        for b in range(blocks):
            if b == 3:
                ref.set_bypass(3, False)
            want.append(ref.process(x[c, b * frames:(b + 1) * frames], sr))
        assert rms(outs["1"][c] - np.concatenate(want)) <= TOL_RMS, c


def test_scan_table_cache_is_trimmed_and_rebuilt(oracle, monkeypatch):
    """A parameter sweep makes a new set of scan tables per setting (the band pass: one per pair of corner frequencies).  Past
    GDG_SCAN_TABLES_MAX entries the cache is dropped at the next plan and the plan re-makes what it needs: the stream goes on unharmed."""
    pkg = package()
    monkeypatch.setenv("GDG_SCAN_TABLES_MAX", "3")
    sr, frames, nch = 96000, 8192, 3
    ctx = pkg.Context(nch, frames)
    pairs = []
    for c in range(nch):
        p = ChainPair(ctx, c, oracle)
        p.append("bandpass", params=[1, 200 + 10 * c, 3000])
        p.append("compressor")
        p.append("cabinet")
        pairs.append(p)
    x = np.stack([synth_signal(c, 14 * frames, sr) for c in range(nch)])
    for b in range(14):
        blk = x[:, b * frames:(b + 1) * frames]
        got = ctx.process(blk, sr)
        for c in range(nch):
            want = pairs[c].ref.process(blk[c], sr)
            assert rms(got[c] - want) <= TOL_RMS, (b, c)
        # other corner frequencies on every channel (different ones per channel): new tables each time
        for c in range(nch):
            for idx, v in ((1, 100 + 37 * b + c), (2, 2000 + 101 * b + 7 * c)):
                ctx.unit_set_param(pairs[c].handles[0], idx, v)
                pairs[c].ref.unit(0).set_param(idx, v)
    ctx.close()
