"""Power-amp filter compilation on the device (effects/poweramp.go:25-127) against the oracle's
filter algebra (filter.Reduce / Normalize / Multiply / Add).  Run with `pytest -m gpu`."""
import numpy as np
import pytest

from helpers import TOL_RMS, package, rms, synth_ir, synth_signal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    p = package()
    assert p.device_count() > 0
    return p


def oracle_compile(oracle, filters, order, sr=48000):
    comp = oracle.Filter([], sr)
    for taps, gc, level in filters:
        if taps is None or len(taps) == 0:
            comp = comp.add(None)
            continue
        f = oracle.Filter(taps, sr, gc)
        if order > 0:
            f = f.reduce(order)
        comp = comp.add(f.normalize().multiply(10.0 ** (0.05 * level)))
    return comp.coefficients()


@pytest.mark.parametrize("order", [0, 1, 2, 64, 1000, 1024, 4096, 65536])
def test_compile_matches_oracle(pkg, oracle, order):
    ctx = pkg.Context(1, 1024)
    h = ctx.append_unit(0, "power_amp")
    filters = [(synth_ir(3000, seed=1), 10.0 ** (0.05 * -20), -3), (None, 1.0, 0), (synth_ir(70001, seed=2), 10.0 ** (0.05 * -10), 0),
               (synth_ir(500, seed=3), 1.0, -12), (np.zeros(0), 1.0, 0), (synth_ir(4096, seed=4), 0.5, 6)]
    ctx.unit_compile_fir(h, filters, order)
    got = ctx.unit_get_fir(h)
    want = oracle_compile(oracle, filters, order)
    assert len(got) == len(want)
    assert rms(got - want) <= TOL_RMS * max(rms(want), 1e-300)
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-12)
    ctx.close()


def test_compile_long_ir_and_process(pkg, oracle):
    """A 300 000-tap IR reduced to 65536 taps (2^19-point transform), then used: the compiled unit convolves like the oracle."""
    sr, frames = 192000, 8192
    ctx = pkg.Context(2, frames)
    hs = [ctx.append_unit(c, "power_amp") for c in range(2)]
    filters = [(synth_ir(300000, seed=9), 10.0 ** (0.05 * -18), 0), (synth_ir(20000, seed=10), 10.0 ** (0.05 * -25), -6)]
    ctx.unit_compile_fir(hs[0], filters, 65536)
    ctx.unit_compile_fir(hs[1], filters[1:], 0)
    want0 = oracle_compile(oracle, filters, 65536, sr)
    want1 = oracle_compile(oracle, filters[1:], 0, sr)
    got0, got1 = ctx.unit_get_fir(hs[0]), ctx.unit_get_fir(hs[1])
    assert len(got0) == 65536 and len(got1) == 20000
    assert rms(got0 - want0) <= TOL_RMS * rms(want0) and rms(got1 - want1) <= TOL_RMS * rms(want1)
    refs = []
    for w in (want0, want1):
        r = oracle.Chain()
        r.append_unit("power_amp", fir=w)
        refs.append(r)
    x = np.stack([synth_signal(c, frames * 3, sr) for c in range(2)])
    for b in range(3):
        blk = x[:, b * frames:(b + 1) * frames]
        got = ctx.process(blk, sr)
        for c in range(2):
            assert rms(got[c] - refs[c].process(blk[c], sr)) <= TOL_RMS
    ctx.close()


def test_compile_empty_and_errors(pkg):
    ctx = pkg.Context(1, 256)
    h = ctx.append_unit(0, "power_amp")
    ctx.unit_compile_fir(h, [(None, 1.0, 0)] * 8, 1024)              # all slots "- NONE -": the Empty filter
    assert ctx.unit_get_fir(h).size == 0
    np.testing.assert_array_equal(ctx.process(np.ones((1, 256)), 48000), np.zeros((1, 256)))     # poweramp.go:210-214
    h2 = ctx.append_unit(0, "overdrive")
    with pytest.raises(pkg.GdgError):
        ctx.unit_compile_fir(h2, [(np.ones(4), 1.0, 0)], 0)
    ctx.close()
