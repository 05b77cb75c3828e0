"""Every parameter of every unit at its minimum and at its maximum (one at a time, the others at their defaults; ranges are
the reference's own tables, tests/golden/params.json): the HIP path follows the oracle at the 1e-9 RMS bar and stays finite.
Discrete parameters run through every value.  The power amp is covered by the FIR tests.  Run with `pytest -m gpu`."""
import numpy as np
import pytest

from helpers import TOL_RMS, package, rms, synth_signal

pytestmark = pytest.mark.gpu

BLOCKS = 4


def _cases(golden_params):
    cases = []
    for utype in sorted(golden_params, key=int):
        if int(utype) == 19:                                   # power amp: filter_N / level_N live in the host mirror
            continue
        ps = golden_params[utype]["params"]
        defaults = [p["DiscreteValueIndex"] if p["Type"] == "PARAMETER_TYPE_DISCRETE" else p["NumericValue"] for p in ps]
        for i, p in enumerate(ps):
            if p["Type"] == "PARAMETER_TYPE_DISCRETE":
                values = range(len(p["DiscreteValues"]))
            else:
                values = sorted({p["Minimum"], p["Maximum"]})
            for v in values:
                if v == defaults[i]:
                    continue
                q = list(defaults)
                q[i] = v
                cases.append((int(utype), p["Name"], q))
    return cases


@pytest.fixture(scope="module")
def pkg():
    p = package()
    assert p.device_count() > 0
    return p


@pytest.mark.parametrize("SR,FRAMES", [(96000, 2048), (44100, 1000), (192000, 8192)])
def test_every_parameter_extreme_matches_oracle(pkg, oracle, golden, SR, FRAMES):
    cases = _cases(golden("params"))
    assert len(cases) > 100
    nch = len(cases)
    ctx = pkg.Context(nch, FRAMES)
    refs = []
    for c, (utype, name, params) in enumerate(cases):
        ctx.append_unit(c, utype, params=params)
        r = oracle.Chain()
        r.append_unit(utype, params=params)
        refs.append(r)
    x = np.stack([synth_signal(c % 48, FRAMES * BLOCKS, SR) * (1.0 if c % 3 else 0.3) for c in range(nch)])
    worst = (0.0, None)
    for b in range(BLOCKS):
        blk = np.ascontiguousarray(x[:, b * FRAMES:(b + 1) * FRAMES])
        got = ctx.process(blk, SR)
        assert np.isfinite(got).all()
        for c, r in enumerate(refs):
            e = rms(got[c] - r.process(blk[c], SR))
            if e > worst[0]:
                worst = (e, cases[c][:2])
            assert e <= TOL_RMS, (cases[c], b, e)
    print("%d parameter extremes, worst RMS error %.3e at %s" % (nch, worst[0], worst[1]))
    ctx.close()
