"""The 1e-9 claim against another libm.

The oracle calls glibc's exp / sin / cos / atan / pow / log10 / log2; the reference calls Go's math package, the HIP path ocml -- each within
1-2 ulp of the others, none correctly rounded.  oracle/libm_jitter.h builds the oracle a second time with every such result moved by a seeded
-2 .. +2 ulp.  If the output moves by more than the parity bar under that, "matches the oracle" says nothing about "matches the Go binary".
It does not: every unit stays below 2e-11 (profiles/libm_sensitivity_r05.txt, generator profiles/libm_sensitivity.py).

Discrete decisions (octaver polarity and hysteresis, noise-gate thresholds, tremolo counters, the encoders' code boundaries) can flip only
where an input sits within an ulp of the threshold; the synthetic inputs here do not, and the tests say so by passing -- the places where such
inputs were FOUND (digital silence behind an FFT, samples on a code boundary) are documented in DESIGN.md section 3 and compared per section
in tests/test_gpu_fuzz.py."""
import os
import sys

import numpy as np
import pytest

import __graft_entry__ as entry

sys.path.insert(0, os.path.join(entry.ROOT, "profiles"))
from helpers import TOL_RMS  # noqa: E402
from test_gpu_parity import UNIT_CASES  # noqa: E402
import libm_sensitivity as ls  # noqa: E402


@pytest.fixture(scope="module")
def orc():
    o = entry.load_oracle()
    o.build()
    o.build(jitter=True)
    return o


def test_the_perturbed_build_perturbs(orc):
    x = ls.inputs(48000, 1024, 2)
    build = ls.unit_builder("overdrive", None)
    plain = ls.stream(orc, build, x, 1024, 48000)
    with orc.jittered(5) as j:
        moved = ls.stream(orc, build, x, 1024, 48000)
        calls = j.calls()
    assert calls >= x.size - 8                   # one exp per sample (exp(0) = 1 is left alone)
    assert 0.0 < ls.worst_rms(plain, moved) <= 1e-15
    with orc.jittered(5):
        again = ls.stream(orc, build, x, 1024, 48000)
    np.testing.assert_array_equal(moved, again)  # seeded: reproducible
    np.testing.assert_array_equal(plain, ls.stream(orc, build, x, 1024, 48000))      # and the plain library is back afterwards


@pytest.mark.parametrize("unit,params", UNIT_CASES)
def test_unit_is_insensitive_to_two_ulp_of_libm(orc, unit, params):
    sr, frames, blocks = 48000, 1024, 6
    x = ls.inputs(sr, frames, blocks)
    build = ls.unit_builder(unit, params)
    plain = ls.stream(orc, build, x, frames, sr)
    for seed in (1, 2):
        with orc.jittered(seed):
            moved = ls.stream(orc, build, x, frames, sr)
        assert ls.worst_rms(plain, moved) <= TOL_RMS / 10, (unit, params, seed)


def test_full_chain_is_insensitive_to_two_ulp_of_libm(orc):
    sr, frames, blocks = 192000, 8192, 3
    x = ls.inputs(sr, frames, blocks)
    build = ls.full_chain_builder(8192)
    plain = ls.stream(orc, build, x, frames, sr)
    with orc.jittered(7):
        moved = ls.stream(orc, build, x, frames, sr)
    assert ls.worst_rms(plain, moved) <= TOL_RMS / 10


@pytest.mark.gpu
@pytest.mark.parametrize("unit,params", UNIT_CASES)
def test_hip_unit_matches_the_perturbed_oracle(orc, unit, params):
    """ocml against glibc-moved-by-2-ulp: the HIP path is as close to a libm it has never seen as to the one the oracle uses."""
    pkg = entry.load_package()
    sr, frames, blocks = 48000, 1024, 6
    x = ls.inputs(sr, frames, blocks)
    build = ls.unit_builder(unit, params)
    ctx = pkg.Context(x.shape[0], frames)
    for c in range(x.shape[0]):
        ctx.append_unit(c, unit, params=params)
    got = np.zeros_like(x)
    for b in range(0, x.shape[1], frames):
        got[:, b:b + frames] = ctx.process(x[:, b:b + frames], sr)
    ctx.close()
    with orc.jittered(11):
        moved = ls.stream(orc, build, x, frames, sr)
    assert ls.worst_rms(got, moved) <= TOL_RMS, (unit, params)


@pytest.mark.gpu
def test_hip_full_chain_matches_the_perturbed_oracle(orc):
    pkg = entry.load_package()
    sr, frames, blocks, taps = 192000, 8192, 3, 8192
    x = ls.inputs(sr, frames, blocks)
    build = ls.full_chain_builder(taps)
    ctx = pkg.Context(x.shape[0], frames)

    class OnHip:
        def __init__(self, c):
            self.c = c

        def append_unit(self, unit_type, params=None, fir=None):
            ctx.append_unit(self.c, unit_type, params=params, fir=fir)
    for c in range(x.shape[0]):
        build(OnHip(c), c)
    got = np.zeros_like(x)
    for b in range(0, x.shape[1], frames):
        got[:, b:b + frames] = ctx.process(x[:, b:b + frames], sr)
    ctx.close()
    with orc.jittered(13):
        moved = ls.stream(orc, build, x, frames, sr)
    assert ls.worst_rms(got, moved) <= TOL_RMS
