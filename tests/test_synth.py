"""The synthetic workload's pseudo random numbers are the reference's (random/random.go:23-55): checked against the known answers
of random/random_test.go:53-70 (tests/golden/random.json) and against the oracle's restatement."""
import json
import os

import numpy as np

from helpers import _synth

HERE = os.path.dirname(os.path.abspath(__file__))


def test_lcg_known_answers():
    synth = _synth()
    with open(os.path.join(HERE, "golden", "random.json")) as f:
        t = json.load(f)["tests"]["TestRNG"]
    seeds = [int(s) for s in t["seeds"]["value"]] if "seeds" in t else [0, 1, 1337, 0xFFFFFFFFFFFFFFFF]
    for seed, want in zip(seeds, t["expectedOutputs"]["value"]):
        got = synth.lcg_floats(seed, len(want))
        assert np.array_equal(got, np.array(want)), (seed, got, want)


def test_lcg_matches_the_oracle_over_a_long_run(oracle):
    synth = _synth()
    orc = oracle
    for seed in (0, 1337 + 511, 4242, 4243 + 2 * 511):
        want = orc.Prng(seed).floats(5000)
        got = synth.lcg_floats(seed, 70000)[:5000]
        assert np.array_equal(got, want)
    # jump-ahead far into the stream
    p = orc.Prng(7)
    for _ in range(65536 - 4):
        p.next_float()
    assert np.array_equal(synth.lcg_floats(7, 65536)[-4:], p.floats(4))


def test_ir_and_rows_are_what_the_survey_specifies():
    synth = _synth()
    h = synth.synth_ir(8192, synth.ir_seed("cab"))
    assert abs(np.sum(h * h) - 1.0) < 1e-12
    r = synth.lcg_floats(4242, 8192)
    k = np.arange(8192)
    raw = (1.0 - 2.0 * r) * np.exp(-6.9 * k / 8192.0)
    assert np.allclose(h, raw / np.sqrt(np.sum(raw * raw)), rtol=0, atol=0)
    assert synth.ir_seed("rev", 0) == 4243 and synth.ir_seed("cab", 3) != synth.ir_seed("rev", 2)
    x = synth.synth_rows(3, 100, 192000, channel0=5, start=50)
    full = synth.synth_rows(8, 150, 192000)
    assert np.array_equal(x, full[5:8, 50:150])
    assert np.max(np.abs(full)) < 1.0
