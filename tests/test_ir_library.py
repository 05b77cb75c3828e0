"""The reference's shipped impulse-response library (24 mono 96 kHz 16-bit files, ir/index.json) through the path a power amp takes:
wave decode -> resample.Time to the session rate (filter.Import, filter/filter.go:704-790) -> poweramp.compile = Reduce / Normalize /
Multiply / Add (effects/poweramp.go:25-127) -> filter.Process.  Fixture: tests/golden/ir_library.npz (the sample words and the
name / compensation table, tests/golden/make_ir_golden.py).

Documented property pinned here (filter.go:127-138, :328-336): after Normalize an impulse response has energy sqrt(sum h^2) equal to
its gain compensation factor 10^(compensation_dB / 20) -- for every shipped response, on the oracle and on the device."""
import os

import numpy as np
import pytest

import __graft_entry__ as entry
from helpers import TOL_RMS, package, rms, synth_signal

LIB = np.load(os.path.join(entry.ROOT, "tests", "golden", "ir_library.npz"))
NAMES = [str(n) for n in LIB["names"]]
COMP = [int(c) for c in LIB["compensation"]]
WORDS = [LIB["ir%02d" % i] for i in range(len(NAMES))]
IDS = ["%02d-%s" % (i, n.split(": ")[-1].replace(" ", "_")) for i, n in enumerate(NAMES)]


def decode(words):
    return words.astype(np.float64) * (2.0 / 65535.0)            # wave/wave.go:347-375: 16-bit LPCM


def test_library_fixture_is_the_reference_table():
    assert len(NAMES) == 24 and len(set(NAMES)) == 24
    assert sorted(set(COMP)) == [-25, -20, -15, -10] or all(-40 <= c <= 0 for c in COMP)
    assert min(w.size for w in WORDS) >= 64 and max(w.size for w in WORDS) <= 9600
    assert int(LIB["sample_rate"]) == 96000


@pytest.mark.parametrize("i", range(24), ids=IDS)
def test_oracle_normalize_gives_the_documented_energy(oracle, i):
    h = decode(WORDS[i])
    fac = 10.0 ** (0.05 * COMP[i])
    n = oracle.Filter(h, 96000, fac).normalize().coefficients()
    assert abs(np.sqrt(np.sum(n * n)) - fac) <= 1e-13 * fac
    # decode as the oracle's wave codec does it
    np.testing.assert_array_equal(oracle.wave_decode("lpcm16", WORDS[i].view(np.uint8)), h)


@pytest.mark.gpu
def test_device_import_and_compile_of_every_shipped_response(oracle):
    """decode + resample.Time to 48 kHz on the device, then gdg_unit_compile_fir slot by slot: energy = compensation factor,
    taps = the oracle's; then all 8 slots of a power amp filled with shipped responses at different levels."""
    pkg = package()
    ctx = pkg.Context(1, 8192)
    h = ctx.append_unit(0, "power_amp")
    taps48 = []
    for i in range(24):
        x96 = ctx.wave_decode("lpcm16", WORDS[i].view(np.uint8))
        np.testing.assert_array_equal(x96, decode(WORDS[i]))
        x48 = ctx.resample_time(x96, 96000, 48000)                                      # filter.go:766-768
        want48 = oracle.resample_time(decode(WORDS[i]), 96000, 48000)
        assert x48.size == want48.size and np.max(np.abs(x48 - want48)) <= 1e-13
        taps48.append(x48)
        fac = 10.0 ** (0.05 * COMP[i])
        for taps, sr in ((x96, 96000), (x48, 48000)):
            ctx.unit_compile_fir(h, [(taps, fac, 0)], 1048576)                          # default filter_order: no Reduce
            got = ctx.unit_get_fir(h)
            assert got.size == taps.size
            assert abs(np.sqrt(np.sum(got * got)) - fac) <= 1e-12 * fac, NAMES[i]
            want = oracle.Filter(taps, sr, fac).normalize().coefficients()
            assert np.max(np.abs(got - want)) <= 1e-13
    # a realistic amp: 8 slots, a Reduce to 2048 taps (poweramp.go:88-96), different levels
    slots = [(taps48[i], 10.0 ** (0.05 * COMP[i]), -3 * k) for k, i in enumerate((0, 3, 8, 11, 15, 16, 19, 23))]
    ctx.unit_compile_fir(h, slots, 2048)
    got = ctx.unit_get_fir(h)
    comp = oracle.Filter([], 48000)
    for taps, fac, level in slots:
        comp = comp.add(oracle.Filter(taps, 48000, fac).reduce(2048).normalize().multiply(10.0 ** (0.05 * level)))
    want = comp.coefficients()
    assert got.size == want.size and np.max(np.abs(got - want)) <= 1e-11
    # ... and the amp sounds like the oracle's
    ref = oracle.Chain()
    ref.append_unit("power_amp", fir=want)
    x = synth_signal(2, 8192 * 3, 48000)
    for b in range(3):
        blk = x[b * 8192:(b + 1) * 8192]
        assert rms(ctx.process(blk[None, :], 48000)[0] - ref.process(blk, 48000)) <= TOL_RMS
    ctx.close()
