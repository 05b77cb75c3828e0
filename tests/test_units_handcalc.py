"""Third-formulation pins for the rows the reference itself has no vectors for: literal first samples from zero
state, derived from the math of SURVEY.md Appendix B (tests/golden/make_units_handcalc.py), against the oracle (CPU) AND the HIP
path (GPU).  Plus known-answer tests of the HIP FFT itself against numpy.fft (row a18)."""
import json
import os

import numpy as np
import pytest

import __graft_entry__ as entry
from helpers import package, rms

TOL = 1e-12

with open(os.path.join(entry.ROOT, "tests", "golden", "units_handcalc.json")) as f:
    GOLD = json.load(f)
IDS = ["%s-%d" % (c["unit"], i) for i, c in enumerate(GOLD["units"])]


def test_every_unit_type_has_a_handcalc_case():
    units = {c["unit"] for c in GOLD["units"]}
    assert len(units) == 21, sorted(units)                 # all 21 effects units incl. the power amp


@pytest.mark.parametrize("case", GOLD["units"], ids=IDS)
def test_oracle_reproduces_handcalc(oracle, case):
    ch = oracle.Chain()
    ch.append_unit(case["unit"], params=case["params"], fir=case.get("fir"))
    got = ch.process(np.array(case["x"]), case["sample_rate"])
    assert np.max(np.abs(got - np.array(case["y"]))) <= TOL, (case["source"], case["note"], got.tolist(), case["y"])


def test_oracle_spatializer_reproduces_handcalc(oracle):
    s = GOLD["spatializer"]
    sp = oracle.Spatializer(1)
    sp.set_sample_rate(s["sample_rate"])
    sp.set_azimuth(0, s["azimuth"]); sp.set_distance(0, s["distance"]); sp.set_level(0, s["level"])
    left, right = sp.process(np.array([s["x"]]))
    assert np.max(np.abs(left - np.array(s["left"]))) <= TOL and np.max(np.abs(right - np.array(s["right"]))) <= TOL


@pytest.mark.gpu
@pytest.mark.parametrize("case", GOLD["units"], ids=IDS)
def test_hip_reproduces_handcalc(case):
    pkg = package()
    x = np.array(case["x"])
    ctx = pkg.Context(1, max(64, x.size))
    ctx.append_unit(0, case["unit"], params=case["params"], fir=case.get("fir"))
    got = ctx.process(x[None, :], case["sample_rate"])[0]
    ctx.close()
    assert np.max(np.abs(got - np.array(case["y"]))) <= 1e-11, (case["source"], case["note"], got.tolist(), case["y"])


@pytest.mark.gpu
def test_hip_spatializer_reproduces_handcalc():
    pkg = package()
    s = GOLD["spatializer"]
    ctx = pkg.Context(1, 64)
    ctx.spatializer_set_sample_rate(s["sample_rate"])
    ctx.spatializer_set_position(0, s["azimuth"], s["distance"], s["level"])
    left, right = ctx.spatialize(np.array([s["x"]]))
    ctx.close()
    assert np.max(np.abs(left - np.array(s["left"]))) <= TOL and np.max(np.abs(right - np.array(s["right"]))) <= TOL


# ---- the HIP FFT by itself (fft.RealFourier / RealInverseFourier conventions: e^{-2 pi i}, unscaled forward, 1/n inverse) ----------
@pytest.mark.gpu
@pytest.mark.parametrize("n", [128, 256, 512, 1024, 2048, 4096, 8192, 16384])
def test_hip_fft_known_answers(n):
    pkg = package()
    ctx = pkg.Context(1, 64)
    k = np.arange(n)
    # 1. closed forms: impulse -> all ones; shifted impulse -> a phase ramp; a cosine on a bin -> n/2 in that bin; DC -> n in bin 0
    imp = np.zeros(n); imp[0] = 1.0
    np.testing.assert_allclose(ctx.fft_real(imp), np.ones(n // 2 + 1), rtol=0, atol=1e-15)
    sh = np.zeros(n); sh[3] = 1.0
    np.testing.assert_allclose(ctx.fft_real(sh), np.exp(-2j * np.pi * 3 * np.arange(n // 2 + 1) / n), rtol=0, atol=1e-14)
    b = n // 8 + 1
    spec = ctx.fft_real(np.cos(2 * np.pi * b * k / n))
    want = np.zeros(n // 2 + 1, dtype=complex); want[b] = n / 2
    np.testing.assert_allclose(spec, want, rtol=0, atol=1e-11 * n)
    np.testing.assert_allclose(ctx.fft_real(np.ones(n))[0], n, rtol=1e-15)
    nyq = ctx.fft_real(np.cos(np.pi * k))                           # alternating signs: everything in the Nyquist bin, real
    assert abs(nyq[n // 2] - n) <= 1e-12 * n and np.max(np.abs(nyq[:n // 2])) <= 1e-11 * n
    # 2. random data against numpy.fft (second implementation), relative to the spectrum's scale
    rng = np.random.default_rng(n)
    x = rng.standard_normal(n)
    got, ref = ctx.fft_real(x), np.fft.rfft(x)
    assert np.max(np.abs(got - ref)) <= 2e-15 * np.sqrt(n) * np.max(np.abs(ref)) + 1e-13
    # 3. the inverse: 1/n scaling, only Re of bins 0 and n/2 used (fft.go:899-906), round trip
    back = ctx.fft_real_inverse(ref, n)
    assert np.max(np.abs(back - x)) <= 1e-13
    dirty = ref.copy(); dirty[0] += 5j; dirty[-1] -= 7j               # imaginary parts of DC / Nyquist are ignored
    assert np.max(np.abs(ctx.fft_real_inverse(dirty, n) - x)) <= 1e-13
    assert rms(ctx.fft_real_inverse(ctx.fft_real(x), n) - x) <= 1e-14
    ctx.close()


@pytest.mark.gpu
def test_hip_fft_rejects_sizes_outside_the_lds_resident_range():
    pkg = package()
    ctx = pkg.Context(1, 64)
    for n in (0, 3, 100, 32768):
        with pytest.raises(pkg.GdgError):
            ctx.fft_real(np.zeros(n))
    ctx.close()


@pytest.mark.gpu
def test_hip_fft_reproduces_the_reference_tests_own_vectors(golden):
    """fft/fft_test.go on the HIP transforms: TestRealFFT's seven vectors (:237-271, tolerance 1e-8 as there), the orthonormal
    scaling vector (:547-570; the scale 1/sqrt(n) is applied by the test, the library's forward transform is unscaled like
    SCALING_DEFAULT), the single-element transform (:639-746) and the all-zero inputs (:194).  The reference stores the full
    spectrum (n bins, conjugate mirror); the library returns bins 0 .. n/2, which determine the rest."""
    pkg = package()
    ctx = pkg.Context(1, 64)
    t = golden("fft")["tests"]
    real = t["TestRealFFT"]
    for x, re, im in zip(real["in"]["value"], real["outRealExpected"]["value"], real["outImagExpected"]["value"]):
        n = len(x)
        got = ctx.fft_real(np.array(x))
        want = np.array(re) + 1j * np.array(im)
        assert got.size == n // 2 + 1
        assert np.max(np.abs(got - want[:n // 2 + 1])) <= 1e-8, (x, got)
        # the mirror the reference stores: X[n - k] = conj(X[k])
        assert np.max(np.abs(np.conj(got[1:n // 2][::-1]) - want[n // 2 + 1:])) <= 1e-8
        back = ctx.fft_real_inverse(got, n)
        assert np.max(np.abs(back - np.array(x))) <= 1e-14
    o = t["TestOrthonormalScaling"]
    x = np.array(o["in"]["value"])
    got = ctx.fft_real(x) / np.sqrt(x.size)
    want = np.array(o["expectedReal"]["value"]) + 1j * np.array(o["expectedImag"]["value"])
    assert np.max(np.abs(got - want[:x.size // 2 + 1])) <= 1e-8
    one = t["TestSingleElementFFT"]["inReal"]["value"]
    assert ctx.fft_real(np.array(one))[0] == one[0] and ctx.fft_real_inverse(np.array([one[0] + 0j]), 1)[0] == one[0]
    for n in t["TestZeroFloat"]["sizes"]["value"]:
        if n & (n - 1) == 0:
            assert not ctx.fft_real(np.zeros(n)).any()
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("n", [2, 4, 8, 16, 32, 64])
def test_hip_fft_small_sizes_against_numpy(n):
    pkg = package()
    ctx = pkg.Context(1, 64)
    rng = np.random.default_rng(100 + n)
    for _ in range(4):
        x = rng.standard_normal(n)
        got, ref = ctx.fft_real(x), np.fft.rfft(x)
        assert np.max(np.abs(got - ref)) <= 1e-14 * max(1.0, np.max(np.abs(ref)))
        assert np.max(np.abs(ctx.fft_real_inverse(ref, n) - x)) <= 1e-14
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("factor,name", [(2, "TestTwoTimesOversampling"), (4, "TestFourTimesOversampling")])
def test_hip_oversampler_tiles_reproduce_the_reference_tests_own_vectors(golden, factor, name):
    """oversampling/oversampling_test.go:48-131 / :136-219 on the HIP tiles (gdg_debug_oversample_decimate): four sequential frames through ONE
    object -- Oversample, compare, Decimate the oversampled frame, compare -- at the reference's tolerance 1e-7 (:33).  Rows a19 / a20 met these
    vectors only through the oracle before."""
    pkg = package()
    ctx = pkg.Context(1, 64)
    t = golden("oversampling")["tests"][name]
    state = np.zeros(8 + (77 if factor == 2 else 155) - 1)
    for x, up_want, down_want in zip(t["in"]["value"], t["oversampledExpected"]["value"], t["decimatedExpected"]["value"]):
        up, down = ctx.debug_oversample_decimate(factor, np.array(x), state)
        np.testing.assert_allclose(up, up_want, atol=1e-7, rtol=0)
        np.testing.assert_allclose(down, down_want, atol=1e-7, rtol=0)
    ctx.close()


@pytest.mark.gpu
@pytest.mark.parametrize("factor", [2, 4])
@pytest.mark.parametrize("sizes", [[8192, 8192, 8192], [1000, 37, 4096, 5, 8192, 1], [64] * 12])
def test_hip_oversampler_tiles_follow_the_oracle_object(oracle, factor, sizes):
    """The same entry against the oracle's OversamplerDecimator over frames that exercise full tiles, partial tiles, frames shorter than the
    8-sample history and the carried state (the oracle object is pinned by the vectors above: tests/test_oracle_golden.py)."""
    pkg = package()
    ctx = pkg.Context(1, 8192)
    osd = oracle.OversamplerDecimator(factor)
    state = np.zeros(8 + (77 if factor == 2 else 155) - 1)
    rng = np.random.default_rng(factor)
    for k, n in enumerate(sizes):
        x = rng.uniform(-1.2, 1.2, n)                        # beyond full scale: the decimator's clip takes part
        if k > 0 and n != sizes[k - 1]:
            state[:8] = 0.0                                   # bufferPreUpsampling is re-made when the frame size changes (oversampling.go:86-89; api_plan.cpp does the same)
        up, down = ctx.debug_oversample_decimate(factor, x, state)
        up_want = osd.oversample(x)
        down_want = osd.decimate(up_want)
        assert np.max(np.abs(up - up_want)) <= 1e-13, (k, n)
        assert np.max(np.abs(down - down_want)) <= 1e-12, (k, n)
    ctx.close()
