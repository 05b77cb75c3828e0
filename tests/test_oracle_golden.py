"""Pin the CPU oracle against every golden vector the reference's own unit tests hold for
the hot path.  The fixtures under tests/golden/ are the literal tables
of fft/fft_test.go, oversampling/oversampling_test.go, resample/resample_test.go,
random/random_test.go and circular/circular_test.go (extracted by tests/golden/make_golden.py).
Tolerances are the reference's own (1e-8, 1e-7 for oversampling).
"""
import numpy as np
import pytest


def _c(re, im):
    return np.array(re, dtype=np.float64) + 1j * np.array(im, dtype=np.float64)


# ---- fft/fft_test.go --------------------------------------------------------------------------
def test_next_power_of_two(oracle, golden):
    t = golden("fft")["tests"]["TestNextPowerOfTwo"]         # fft_test.go:73-150
    for v, p, e in zip(t["in"]["value"], t["powers"]["value"], t["exponents"]["value"]):
        assert oracle.next_power_of_two(v) == (p, e)


def test_real_fft(oracle, golden):
    t = golden("fft")["tests"]["TestRealFFT"]                # fft_test.go:232-359, tolerance :33
    for x, re, im in zip(t["in"]["value"], t["outRealExpected"]["value"], t["outImagExpected"]["value"]):
        rc, out = oracle.real_fourier(x)
        assert rc == 0
        np.testing.assert_allclose(out, _c(re, im), atol=1e-8, rtol=0)
        np.testing.assert_allclose(out, np.fft.fft(x), atol=1e-13, rtol=0)
        rc, back = oracle.real_inverse_fourier(out)
        assert rc == 0
        np.testing.assert_allclose(back, x, atol=1e-8, rtol=0)


@pytest.mark.parametrize("mode", ["standard", "inplace"])
def test_complex_fft(oracle, golden, mode):
    t = golden("fft")["tests"]["TestComplexFFT"]             # fft_test.go:364-545
    m = oracle.MODE_STANDARD if mode == "standard" else oracle.MODE_INPLACE
    for xr, xi, re, im in zip(t["inReal"]["value"], t["inImag"]["value"],
                              t["outRealExpected"]["value"], t["outImagExpected"]["value"]):
        z = _c(xr, xi)
        out = oracle.fourier(z, mode=m)
        np.testing.assert_allclose(out, _c(re, im), atol=1e-8, rtol=0)
        back = oracle.inverse_fourier(out, mode=m)
        np.testing.assert_allclose(back, z, atol=1e-8, rtol=0)


def test_orthonormal_scaling(oracle, golden):
    t = golden("fft")["tests"]["TestOrthonormalScaling"]     # fft_test.go:547-637
    x = t["in"]["value"]
    rc, out = oracle.real_fourier(x, scaling=oracle.SCALING_ORTHONORMAL)
    assert rc == 0
    np.testing.assert_allclose(out, _c(t["expectedReal"]["value"], t["expectedImag"]["value"]), atol=1e-8, rtol=0)
    rc, back = oracle.real_inverse_fourier(out, scaling=oracle.SCALING_ORTHONORMAL)
    assert rc == 0
    np.testing.assert_allclose(back, x, atol=1e-8, rtol=0)


def test_single_element_and_failures(oracle):
    # fft_test.go:639-699 (n = 1 is the identity) and :701-746 (odd n, length mismatch must fail)
    rc, out = oracle.real_fourier([0.5])
    assert rc == 0 and out[0] == 0.5
    rc, back = oracle.real_inverse_fourier(np.array([0.5 + 0j]))
    assert rc == 0 and back[0] == 0.5
    np.testing.assert_array_equal(oracle.fourier(np.array([0.25 + 0.5j])), [0.25 + 0.5j])
    assert oracle.real_fourier([1.0, 2.0, 3.0])[0] != 0
    assert oracle.real_inverse_fourier(np.array([1.0, 2.0, 3.0], dtype=complex))[0] != 0
    assert oracle.real_fourier([1.0, 2.0, 3.0, 4.0], n_out=2)[0] != 0
    assert oracle.real_inverse_fourier(np.zeros(4, dtype=complex), n_out=2)[0] != 0


def test_shift(oracle, golden):
    t = golden("fft")["tests"]["TestShift"]                  # fft_test.go:748-838
    for key_in, key_out in (("inEven", "outEven"), ("inOdd", "outOdd")):
        z = np.array([complex(a, b) for a, b in t[key_in]["value"]])
        want = np.array([complex(a, b) for a, b in t[key_out]["value"]])
        got = oracle.shift(z)
        np.testing.assert_array_equal(got, want)
        np.testing.assert_array_equal(oracle.shift(got, inverse=True), z)


# ---- oversampling/oversampling_test.go ------------------------------------------------------------
@pytest.mark.parametrize("factor,name", [(2, "TestTwoTimesOversampling"), (4, "TestFourTimesOversampling")])
def test_oversampler_decimator_stateful(oracle, golden, factor, name):
    t = golden("oversampling")["tests"][name]                # oversampling_test.go:48-131 / :136-219, tolerance :33
    osd = oracle.OversamplerDecimator(factor)
    for x, up_want, down_want in zip(t["in"]["value"], t["oversampledExpected"]["value"], t["decimatedExpected"]["value"]):
        up = osd.oversample(x)
        np.testing.assert_allclose(up, up_want, atol=1e-7, rtol=0)
        down = osd.decimate(up)
        np.testing.assert_allclose(down, down_want, atol=1e-7, rtol=0)


def test_aa_taps_symmetric(oracle):
    for f, n in ((2, 77), (4, 155)):
        taps = oracle.aa_taps(f)
        assert len(taps) == n
        np.testing.assert_array_equal(taps, taps[::-1])


# ---- resample/resample_test.go ------------------------------------------------------------------
def test_resample_time(oracle, golden):
    t = golden("resample")["tests"]["TestTimeSeries"]        # resample_test.go:48-99, tolerance 1e-8
    for x, up, down in zip(t["in"]["value"], t["outExpectedUp"]["value"], t["outExpectedDown"]["value"]):
        got_up = oracle.resample_time(x, 96000, 192000)
        assert len(got_up) == len(up)
        np.testing.assert_allclose(got_up, up, atol=1e-8, rtol=0)
        got_down = oracle.resample_time(x, 96000, 44100)
        assert len(got_down) == len(down)
        np.testing.assert_allclose(got_down, down, atol=1e-8, rtol=0)


def test_resample_frequency(oracle, golden):
    t = golden("resample")["tests"]["TestFrequencySeries"]   # resample_test.go:104-175
    for z, re, im in zip(t["in"]["value"], t["outExpectedReal"]["value"], t["outExpectedImag"]["value"]):
        zz = np.array([complex(a, b) for a, b in z])
        got = oracle.resample_frequency(zz, len(re))
        np.testing.assert_allclose(got, _c(re, im), atol=1e-8, rtol=0)


def test_resample_oversample(oracle, golden):
    t = golden("resample")["tests"]["TestOversample"]        # resample_test.go:180-215
    for x, want in zip(t["in"]["value"], t["outExpected"]["value"]):
        got = oracle.resample_oversample(x, len(want), 2)
        np.testing.assert_allclose(got, want, atol=1e-8, rtol=0)


# ---- random/random_test.go ---------------------------------------------------------------------------
def test_prng(oracle, golden):
    t = golden("random")["tests"]["TestRNG"]                 # random_test.go:48-112
    for seed, want in zip(t["seeds"]["value"], t["expectedOutputs"]["value"]):
        g = oracle.Prng(seed)
        np.testing.assert_allclose(g.floats(len(want)), want, atol=1e-8, rtol=0)
        more = g.floats(10000)
        assert more.min() >= 0.0 and more.max() <= 1.0


# ---- circular/circular_test.go ------------------------------------------------------------------------
def test_ring(oracle, golden):
    t = golden("circular")["tests"]["TestBuffer"]            # circular_test.go:42-166
    vin, want = t["in"]["value"], t["expected"]["value"]
    r = oracle.Ring(5)
    got = []
    r.enqueue(vin[0]); got.append(r.retrieve())
    r.enqueue(vin[1]); got.append(r.retrieve())
    r.enqueue(vin[2]); got.append(r.retrieve())
    r.enqueue(vin[3]); r.enqueue(vin[4]); got.append(r.retrieve())
    r.enqueue(vin[5]); got.append(r.retrieve())
    r.enqueue(vin[6]); got.append(r.retrieve())
    for k in (7, 8, 9):
        r.enqueue(vin[k])
    r.enqueue(vin[10][:1]); r.enqueue(vin[11][:1]); got.append(r.retrieve())
    for (rc, g), w in zip(got, want):
        assert rc == 0
        np.testing.assert_array_equal(g, w)
    assert r.retrieve(4)[0] != 0


# ---- wave/wave_test.go: byte-exact sample codecs ------------------------------------------------
WAVE_CASES = [("lpcm8", "PCM8", 1), ("lpcm16", "PCM16", 2), ("lpcm24", "PCM24", 3), ("lpcm32", "PCM32", 4), ("ieee32", "IEEE32", 4), ("ieee64", "IEEE64", 8)]


@pytest.mark.parametrize("fmt,tag,width", WAVE_CASES)
def test_wave_export_data_section_is_byte_exact(oracle, golden, fmt, tag, width):
    t = golden("wave")["tests"]["TestExport%sMono" % tag]       # wave_test.go: samples -> file bytes; the data chunk is the tail
    samples = t["samples"]["value"]
    want = np.array(t["expectedOutput"]["value"], dtype=np.uint8)[-len(samples) * width:]
    np.testing.assert_array_equal(oracle.wave_encode(fmt, samples), want)


@pytest.mark.parametrize("fmt,tag,width", WAVE_CASES)
def test_wave_import_samples(oracle, golden, fmt, tag, width):
    t = golden("wave")["tests"]["TestImport%sMono" % tag]
    want = np.array(t["expectedSamples"]["value"])
    data = np.array(t["buf"]["value"], dtype=np.uint8)[-len(want) * width:]
    # the reference's own per-format tolerances (wave_test.go:319, :567, :819, :1077, :1335, :1613)
    tol = {"lpcm8": 0.078125, "lpcm16": 3.0518e-5, "lpcm24": 1.1921e-7, "lpcm32": 4.6567e-10, "ieee32": 1.1921e-7, "ieee64": 1.0e-16}[fmt]
    got = oracle.wave_decode(fmt, data)
    np.testing.assert_allclose(got, want, rtol=0, atol=tol)
    if fmt != "lpcm8":      # decode(encode(x)) reproduces the quantised value exactly (8-bit: 127 vs 1/127 scaling is lossy by design)
        np.testing.assert_array_equal(oracle.wave_decode(fmt, oracle.wave_encode(fmt, got)), got)


# ---- level/level_test.go: known meter readings ---------------------------------------------------
def test_level_meter_known_readings(oracle):
    sr = 96000                                                    # level_test.go:17-224: one second of a 1 Hz sine
    a = np.sin(2.0 * np.pi * (np.arange(sr) / float(sr)))
    for buf, want in ((a, (-3, 0)), (0.5 * a, (-9, -6))):
        m = oracle.ChannelMeter()
        m.set_enabled(True)
        m.process(buf, sr)
        assert m.analyze() == want
        m.set_enabled(False)
        assert m.analyze() == (-200, -200)
    m = oracle.ChannelMeter()                                     # a disabled meter ignores its input
    m.process(a, sr)
    assert m.analyze() == (-200, -200)
