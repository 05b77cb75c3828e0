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
    assert n == 3 and ms > 0.0                                   # HIP-event timing of the MAC kernel is live
    with pytest.raises(pkg.GdgError):
        ctx.process_device(d_in, d_in, frames, sr)               # in-place is rejected
    d_in.free(); d_out.free()
    ctx.close()


def test_frame_size_change_with_live_fir_state_fails_loudly(pkg):
    ctx = pkg.Context(1, 1024)
    h = ctx.append_unit(0, "power_amp", fir=synth_ir(3000))
    ctx.process(np.zeros((1, 1024)), 48000)
    with pytest.raises(pkg.GdgError) as e:
        ctx.process(np.zeros((1, 512)), 48000)
    assert e.value.code == pkg.GDG_ERR_UNSUPPORTED
    ctx.unit_reset(h)
    ctx.process(np.zeros((1, 512)), 48000)                      # after a reset the new partition size is accepted
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
