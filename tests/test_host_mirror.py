"""The host-side mirror of effects.Unit / signal.Chain (go-dsp-guitar_amd/host/): parameter tables,
error strings and chain editing checked on the CPU; Process() checked against the oracle on the GPU.

The expected error texts are the reference's own format strings (effects/effects.go:144-384,
signal/signal.go:52-357); the parameter tables are tests/golden/params.json (extracted from the
create*() functions by tests/golden/make_params_golden.py)."""
import json
import os
import threading

import numpy as np
import pytest

import __graft_entry__ as entry
from helpers import TOL_RMS, rms, synth_ir, synth_signal

TYPE = {"PARAMETER_TYPE_DISCRETE": 1, "PARAMETER_TYPE_NUMERIC": 2}


@pytest.fixture(scope="module")
def host():
    pkg = entry.load_package()
    pkg.build()
    from go_dsp_guitar_amd import host as h
    h.build()
    return h


@pytest.fixture(scope="module")
def params():
    with open(os.path.join(entry.ROOT, "tests", "golden", "params.json")) as f:
        return json.load(f)


def test_parameter_tables_match_reference(host, params):
    eng = host.Engine(1)
    ch = eng.create_chain()
    for t in range(21):
        i = ch.AppendUnit(t)
        assert i == t
        assert ch.UnitType(i) == t
        assert ch.GetBypass(i) is True                       # signal.go:74: new units start bypassed
        got = ch.Parameters(i)
        want = params[str(t)]["params"]
        assert len(got) == len(want), params[str(t)]["unit"]
        for g, w in zip(got, want):
            assert g["Name"] == w["Name"]
            assert g["Type"] == TYPE[w["Type"]]
            assert g["PhysicalUnit"] == w["PhysicalUnit"]
            assert (g["Minimum"], g["Maximum"], g["NumericValue"]) == (w["Minimum"], w["Maximum"], w["NumericValue"])
            assert g["DiscreteValueIndex"] == w["DiscreteValueIndex"]
            assert g["DiscreteValues"] == w["DiscreteValues"]
    assert ch.Length() == 21
    eng.close()


def test_set_get_and_error_strings(host):
    eng = host.Engine(1)
    ch = eng.create_chain()
    i = ch.AppendUnit(9)                                      # overdrive
    ch.SetNumericValue(i, "gain", 20)
    assert ch.GetNumericValue(i, "gain") == 20
    ch.SetDiscreteValue(i, "valve", "ECC82 (12AU7)")
    assert ch.GetDiscreteValue(i, "valve") == "ECC82 (12AU7)"
    cases = [
        (lambda: ch.SetNumericValue(i, "gain", 31), "Failed to set numeric value: Parameter 'gain' must be between '-30' and '30' - got '31'."),
        (lambda: ch.SetNumericValue(i, "nope", 1), "Failed to set numeric value: Could not find parameter with name 'nope'."),
        (lambda: ch.SetNumericValue(i, "valve", 1), "Failed to set numeric value: Parameter 'valve' is not numeric."),
        (lambda: ch.SetDiscreteValue(i, "valve", "EL34"), "Failed to set discrete value: Value 'EL34' is not valid for parameter 'valve'."),
        (lambda: ch.SetDiscreteValue(i, "gain", "x"), "Failed to set discrete value: Parameter 'gain' is not discrete."),
        (lambda: ch.SetDiscreteValue(i, "nope", "x"), "Failed to set discrete value: Could not find parameter with name 'nope'."),
        (lambda: ch.GetNumericValue(i, "valve"), "Failed to get numeric value: Parameter 'valve' is not numeric."),
        (lambda: ch.GetDiscreteValue(i, "gain"), "Failed to get discrete value: Parameter 'gain' is not discrete."),
        (lambda: ch.GetDiscreteValue(i, "nope"), "Failed to get discrete value: Could not find parameter with name 'nope'."),
        (lambda: ch.AppendUnit(21), "Failed to create effects unit."),
        (lambda: ch.RemoveUnit(5), "Cannot remove unit 5."),
        (lambda: ch.MoveUp(0), "Cannot move unit 0 up."),
        (lambda: ch.MoveDown(0), "Cannot move unit 0 down."),
        (lambda: ch.UnitType(3), "Cannot get unit type: No unit 3."),
        (lambda: ch.SetBypass(3, True), "Cannot enable bypass: No unit 3."),
        (lambda: ch.SetBypass(3, False), "Cannot disable bypass: No unit 3."),
        (lambda: ch.GetBypass(-1), "Cannot get bypass value: No unit -1."),
        (lambda: ch.SetDiscreteValue(7, "a", "b"), "Cannot set discrete value: No unit 7."),
        (lambda: ch.GetDiscreteValue(7, "a"), "Cannot get discrete value: No unit 7."),
        (lambda: ch.SetNumericValue(7, "a", 1), "Cannot set numeric value: No unit 7."),
        (lambda: ch.GetNumericValue(7, "a"), "Cannot get numeric value: No unit 7."),
        (lambda: ch.Parameters(7), "Cannot get parameters: No unit 7."),
    ]
    for fn, msg in cases:
        with pytest.raises(host.HostError) as e:
            fn()
        assert str(e.value) == msg
    assert ch.GetNumericValue(i, "gain") == 20               # a rejected value leaves the parameter unchanged
    eng.close()


def test_chain_editing(host):
    eng = host.Engine(1)
    ch = eng.create_chain()
    for t in (5, 9, 11, 20):
        ch.AppendUnit(t)
    ch.MoveUp(2)
    assert [ch.UnitType(i) for i in range(4)] == [5, 11, 9, 20]
    ch.MoveDown(0)
    assert [ch.UnitType(i) for i in range(4)] == [11, 5, 9, 20]
    ch.RemoveUnit(1)
    assert [ch.UnitType(i) for i in range(3)] == [11, 9, 20]
    ch.SetBypass(1, False)
    assert [ch.GetBypass(i) for i in range(3)] == [True, False, True]
    eng.close()


def test_power_amp_parameters_follow_the_ir_library(host):
    irs = host.ImpulseResponses()
    for sr in (48000, 96000):
        irs.add("Guitar: A", sr, -20, synth_ir(300, seed=1))
        irs.add("Guitar: B", sr, -25, synth_ir(500, seed=2))
    eng = host.Engine(1)
    ch = eng.create_chain(irs)
    i = ch.AppendUnit(19)
    p = ch.Parameters(i)
    assert [q["Name"] for q in p] == ["filter_order"] + [n for k in range(1, 9) for n in ("filter_%d" % k, "level_%d" % k)]
    assert p[1]["DiscreteValues"] == ["- NONE -", "Guitar: A", "Guitar: B"]      # poweramp.go:256-281
    assert (p[2]["Minimum"], p[2]["Maximum"], p[2]["NumericValue"]) == (-60, 0, 0)
    ch.SetDiscreteValue(i, "filter_3", "Guitar: B")
    assert ch.GetDiscreteValue(i, "filter_3") == "Guitar: B"
    with pytest.raises(host.HostError):
        ch.SetDiscreteValue(i, "filter_3", "Guitar: C")
    eng.close()


def test_filter_compile_matches_oracle(host, oracle):
    """Reduce -> Normalize -> Multiply (poweramp.go:88-96) against the oracle's filter algebra."""
    taps = synth_ir(3000, seed=5)
    for order, comp, level in ((0, -20, 0), (4096, -25, -6), (1024, -10, -3), (256, 0, 0)):
        got = host.filter_compile(taps, 48000, comp, order, level)
        f = oracle.Filter(taps, 48000, 10.0 ** (0.05 * comp))
        if order:
            f = f.reduce(order)
        want = f.normalize().multiply(10.0 ** (0.05 * level)).coefficients()
        assert len(got) == len(want)
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-12)


# ---- GPU ---------------------------------------------------------------------------------------------------------
def _full_chain(ch, ref, oracle, sr, irs_taps):
    spec = [(5, {"gain_limit": 30, "target_level": -20}), (9, {"gain": 20}), (11, {}), (12, {}), (19, None), (20, {}), (18, {"mix": 50})]
    for t, numeric in spec:
        i = ch.AppendUnit(t)
        if t == 19:
            ch.SetDiscreteValue(i, "filter_1", "Cab")
            ch.SetNumericValue(i, "level_1", -3)
            ch.SetDiscreteValue(i, "filter_2", "Room")
        else:
            for k, v in numeric.items():
                ch.SetNumericValue(i, k, v)
        ch.SetBypass(i, False)
    # the same chain on the oracle; the composite filter is compiled with the oracle's own filter algebra
    a = oracle.Filter(irs_taps["Cab"], sr, 10.0 ** (0.05 * -20)).normalize().multiply(10.0 ** (0.05 * -3))
    b = oracle.Filter(irs_taps["Room"], sr, 10.0 ** (0.05 * -10)).normalize().multiply(1.0)
    composite = oracle.Filter([], sr).add(a).add(b)
    ref.append_unit("compressor", params=[1, 30, -20])
    ref.append_unit("overdrive", params=[0, 20, 100, 0, 1, 0])
    ref.append_unit("tone_stack")
    ref.append_unit("chorus")
    ref.append_unit("power_amp", fir=composite.coefficients())
    ref.append_unit("cabinet")
    ref.append_unit("reverb", params=[50])


@pytest.mark.gpu
def test_chain_process_rendezvous_matches_oracle(host, oracle):
    """N concurrent Chain.Process calls (one thread per channel, like the controller's worker goroutines)."""
    sr, frames, nch, blocks = 48000, 1024, 4, 4
    taps = {"Cab": synth_ir(2000, seed=3), "Room": synth_ir(5000, seed=4)}
    irs = host.ImpulseResponses()
    irs.add("Cab", sr, -20, taps["Cab"])
    irs.add("Room", sr, -10, taps["Room"])
    eng = host.Engine(nch, frames)
    chains, refs = [], []
    for c in range(nch):
        ch, ref = eng.create_chain(irs), oracle.Chain()
        _full_chain(ch, ref, oracle, sr, taps)
        chains.append(ch)
        refs.append(ref)
    eng.set_rendezvous(nch, 2000)
    x = np.stack([synth_signal(c, frames * blocks, sr) for c in range(nch)])
    got = np.zeros_like(x)
    for b in range(blocks):
        sl = slice(b * frames, (b + 1) * frames)

        def work(c):
            got[c, sl] = chains[c].Process(x[c, sl], sr)

        threads = [threading.Thread(target=work, args=(c,)) for c in range(nch)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    assert eng.last_error() == ""
    for c in range(nch):
        want = np.concatenate([refs[c].process(x[c, b * frames:(b + 1) * frames], sr) for b in range(blocks)])
        assert rms(got[c] - want) <= TOL_RMS
    eng.close()


@pytest.mark.gpu
def test_partial_rendezvous_processes_only_the_callers(host, oracle):
    sr, frames = 48000, 512
    eng = host.Engine(3, frames)
    chains, refs = [], []
    for c in range(3):
        ch, ref = eng.create_chain(), oracle.Chain()
        i = ch.AppendUnit(11)
        ch.SetBypass(i, False)
        ref.append_unit("tone_stack")
        chains.append(ch)
        refs.append(ref)
    eng.set_rendezvous(3, 20)                                 # only one caller shows up: subset after the timeout
    x = synth_signal(1, frames * 3, sr)
    for b in range(3):
        blk = x[b * frames:(b + 1) * frames]
        got = chains[1].Process(blk, sr)
        assert rms(got - refs[1].process(blk, sr)) <= TOL_RMS
    # length mismatch is a silent no-op (signal.go:366): the output buffer is left untouched
    out = chains[0].Process(x[:frames], sr, n_out=frames - 1)
    assert np.all(np.isnan(out))
    eng.close()


@pytest.mark.gpu
def test_standalone_unit_process_and_process_all(host, oracle):
    sr, frames = 96000, 2048
    u = host.Unit(11)
    u.SetNumericValue("middle", -9)
    ref = oracle.Unit("tone_stack")
    ref.set_params([0, -9, -5, -5])
    x = synth_signal(2, frames * 2, sr)
    for b in range(2):
        blk = x[b * frames:(b + 1) * frames]
        assert rms(u.Process(blk, sr) - ref.process(blk, sr)) <= TOL_RMS
    eng = host.Engine(2, frames)
    refs = []
    for c in range(2):
        ch = eng.create_chain()
        i = ch.AppendUnit(20)
        ch.SetBypass(i, False)
        r = oracle.Chain()
        r.append_unit("cabinet")
        refs.append(r)
    xx = np.stack([synth_signal(c, frames, sr) for c in range(2)])
    got = eng.process_all(xx, sr)
    for c in range(2):
        assert rms(got[c] - refs[c].process(xx[c], sr)) <= TOL_RMS
    eng.close()


# ---- shards, spatializer.Spatializer, tuner.Tuner, power-amp reset semantics ---------------------------------------
def test_shard_routing_is_contiguous_blocks(host):
    """Channel c lives on shard c * G / N; never more shards than channels.  No GPU needed: contexts are lazy."""
    eng = host.Engine(8, devices=[0, 0, 0])
    assert eng.shards() == 3
    assert [eng.shard_of(c) for c in range(8)] == [0, 0, 1, 1, 1, 2, 2, 2]
    eng.close()
    eng = host.Engine(512, devices=list(range(8)))
    assert [eng.shard_of(c) for c in (0, 63, 64, 127, 448, 511)] == [0, 0, 1, 1, 7, 7]
    eng.close()
    eng = host.Engine(2, devices=[0, 1, 2, 3])
    assert eng.shards() == 2
    eng.close()


def test_spatializer_interface_and_error_strings(host):
    """spatializer/spatializer.go:68-413: getters, setters, range checks and their messages."""
    eng = host.Engine(3)
    sp = host.Spatializer(eng, 3)
    assert (sp.GetInputCount(), sp.GetOutputCount()) == (3, 2)
    assert (sp.GetAzimuth(1), sp.GetDistance(1), sp.GetLevel(1)) == (0.0, 0.0, 1.0)        # spatializer.go:436-447: level 1 by default
    sp.SetAzimuth(1, -35.5)
    sp.SetDistance(1, 2.5)
    sp.SetLevel(1, 0.25)
    assert (sp.GetAzimuth(1), sp.GetDistance(1), sp.GetLevel(1)) == (-35.5, 2.5, 0.25)
    with pytest.raises(host.HostError, match=r"Failed to set distance: Value must be within \[0, 10\]\."):
        sp.SetDistance(0, 10.5)
    with pytest.raises(host.HostError, match=r"Failed to set level: Value must be within \[0, 1\]\."):
        sp.SetLevel(0, -0.1)
    with pytest.raises(host.HostError, match=r"Cannot set azimuth for channel 7: Only 3 channels exist\."):
        sp.SetAzimuth(7, 1.0)
    with pytest.raises(host.HostError, match=r"Cannot get level for channel 4: Only 3 channels exist\."):
        sp.GetLevel(4)
    with pytest.raises(host.HostError, match=r"Cannot set distance for channel 5: Only 3 channels exist\."):
        sp.SetLevel(5, 0.5)                                  # the reference's SetLevel message says "distance" (spatializer.go:395)
    del sp
    eng.close()


def _every_visible_device(host):
    """One shard per visible HIP device -- the reference's actual deployment: N chains over the GPUs of one node inside ONE process
    (controller/controller.go:3262-3269, :3333-3341).  Skips on a one-GPU box, where the three-contexts-on-device-0 variants stand in."""
    pkg = entry.load_package()
    n = pkg.lib().gdg_device_count()
    if n < 2:
        pytest.skip("needs at least two HIP devices (%d visible)" % n)
    return list(range(n))


@pytest.mark.gpu
def test_sharded_engine_matches_oracle(host, oracle):
    """Three shards (three independent contexts on device 0) behind ONE rendezvous: the Go shim's N/8 routing, on one GPU."""
    _sharded_engine_case(host, oracle, [0, 0, 0], 7)


@pytest.mark.gpu
def test_sharded_engine_on_every_visible_device_matches_oracle(host, oracle):
    """The same with G = hipGetDeviceCount() shards on G REAL devices driven from one process (hipSetDevice per call, one copy pool per
    context): 2 G + 1 channels, so that the blocks are uneven."""
    devices = _every_visible_device(host)
    _sharded_engine_case(host, oracle, devices, 2 * len(devices) + 1)


def _sharded_engine_case(host, oracle, devices, nch):
    sr, frames, blocks = 48000, 1024, 3
    taps = {"Cab": synth_ir(2000, seed=3), "Room": synth_ir(5000, seed=4)}
    irs = host.ImpulseResponses()
    irs.add("Cab", sr, -20, taps["Cab"])
    irs.add("Room", sr, -10, taps["Room"])
    eng = host.Engine(nch, frames, devices=devices)
    assert eng.shards() == min(len(devices), nch)
    chains, refs = [], []
    for c in range(nch):
        ch, ref = eng.create_chain(irs), oracle.Chain()
        _full_chain(ch, ref, oracle, sr, taps)
        chains.append(ch)
        refs.append(ref)
    eng.set_rendezvous(nch, 2000)
    x = np.stack([synth_signal(c, frames * blocks, sr) for c in range(nch)])
    got = np.zeros_like(x)
    for b in range(blocks):
        sl = slice(b * frames, (b + 1) * frames)

        def work(c):
            got[c, sl] = chains[c].Process(x[c, sl], sr)

        threads = [threading.Thread(target=work, args=(c,)) for c in range(nch)]
        for t in threads:
            t.start()
        for t in threads:
            t.join()
    assert eng.last_error() == ""
    for c in range(nch):
        want = np.concatenate([refs[c].process(x[c, b * frames:(b + 1) * frames], sr) for b in range(blocks)])
        assert rms(got[c] - want) <= TOL_RMS, "channel %d (shard %d)" % (c, eng.shard_of(c))
    eng.close()


@pytest.mark.gpu
def test_power_amp_set_to_its_current_value_resets_the_filter_like_the_reference(host, oracle):
    """poweramp.go:131-181: EVERY successful Set compiles a new filter object -- fresh tail -- even if the value did not change.
    The oracle models that with set_fir(same taps); the twin (and the Go shim) count successful sets, not value changes."""
    sr, frames = 48000, 512
    taps = synth_ir(4000, seed=8)
    irs = host.ImpulseResponses()
    irs.add("Cab", sr, -20, taps)
    eng = host.Engine(1, frames)
    ch = eng.create_chain(irs)
    i = ch.AppendUnit(19)
    ch.SetDiscreteValue(i, "filter_1", "Cab")
    ch.SetNumericValue(i, "level_1", -6)
    ch.SetBypass(i, False)
    composite = oracle.Filter(taps, sr, 10.0 ** (0.05 * -20)).normalize().multiply(10.0 ** (0.05 * -6)).coefficients()
    ref = oracle.Chain()
    ref.append_unit("power_amp", fir=composite)
    x = synth_signal(0, frames * 6, sr)
    got, want = np.zeros_like(x), np.zeros_like(x)
    for b in range(6):
        sl = slice(b * frames, (b + 1) * frames)
        if b == 2:
            ch.SetNumericValue(i, "level_1", -6)              # the SAME value: the reference still replaces currentFilter
            ref.unit(0).set_fir(composite)
        if b == 4:
            ch.SetBypass(i, True)                             # bypass toggles do NOT touch the filter (signal.go:390-401)
            ch.SetBypass(i, False)
        got[sl] = ch.Process(x[sl], sr)
        want[sl] = ref.process(x[sl], sr)
    assert rms(got - want) <= TOL_RMS
    # and a version WITHOUT the reset differs audibly, i.e. the test can tell the two semantics apart
    ref2 = oracle.Chain()
    ref2.append_unit("power_amp", fir=composite)
    cont = np.concatenate([ref2.process(x[b * frames:(b + 1) * frames], sr) for b in range(6)])
    assert rms(cont - want) > 1e-4
    eng.close()


@pytest.mark.gpu
def test_spatializer_twin_matches_oracle_over_shards(host, oracle):
    """Partial mixes per shard + host sum + aux (spatializer.go:300-310); also the no-upload path that mixes the chain outputs
    still on the devices (what controller.process() feeds it, controller.go:2744-2761)."""
    sr, frames, nch, blocks = 96000, 2048, 6, 3
    rng = np.random.default_rng(11)
    eng = host.Engine(nch, frames, devices=[0, 0])
    chains, refs = [], []
    for c in range(nch):
        ch = eng.create_chain()
        i = ch.AppendUnit(11)
        ch.SetBypass(i, False)
        chains.append(ch)
        r = oracle.Chain()
        r.append_unit("tone_stack")
        refs.append(r)
    sp = host.Spatializer(eng, nch)
    sp.SetSampleRate(sr)
    ref_sp = oracle.Spatializer(nch)
    ref_sp.set_sample_rate(sr)
    for c in range(nch):
        a, d, l = float(rng.uniform(-180, 180)), float(rng.uniform(0, 10)), float(rng.uniform(0, 1))
        sp.SetAzimuth(c, a); sp.SetDistance(c, d); sp.SetLevel(c, l)
        ref_sp.set_azimuth(c, a); ref_sp.set_distance(c, d); ref_sp.set_level(c, l)
    x = np.stack([synth_signal(c, frames * blocks, sr) for c in range(nch)])
    aux = 0.1 * synth_signal(50, frames * blocks, sr)
    for b in range(blocks):
        sl = slice(b * frames, (b + 1) * frames)
        y = eng.process_all(x[:, sl], sr)
        want_y = np.stack([refs[c].process(x[c, sl], sr) for c in range(nch)])
        want_l, want_r = ref_sp.process(want_y, aux=aux[sl])
        if b == 1:
            got_l, got_r = sp.Process(y, aux[sl], reuse_chain_outputs=True)          # nothing uploaded
        else:
            got_l, got_r = sp.Process(y, aux[sl])
        assert rms(got_l - want_l) <= TOL_RMS and rms(got_r - want_r) <= TOL_RMS, "block %d" % b
    del sp
    eng.close()


@pytest.mark.gpu
def test_tuner_twin(host, oracle):
    """tuner.Tuner: Process enqueues, Analyze reports note / cents / frequency (criterion of tuner_test.go:95-106)."""
    sr = 96000
    t = host.Tuner()
    ref = oracle.Tuner()
    n = np.arange(96000)
    for freq, note in ((110.0, "A2"), (146.8324, "D3"), (329.6276, "E4")):
        tone = sum((0.5 / k) * np.sin(2 * np.pi * k * freq * n / sr) for k in (1, 2, 3))
        for at in range(0, tone.size, 8192):
            t.Process(tone[at:at + 8192], sr)
            ref.process(tone[at:at + 8192], sr)
        got, want = t.Analyze(), ref.analyze()
        assert got["note"] == note == want["note"]
        assert got["cents"] == want["cents"] and abs(got["cents"]) <= 5
        assert abs(got["frequency"] - want["frequency"]) / want["frequency"] <= 1e-9


@pytest.mark.gpu
def test_control_plane_setters_race_with_processing(host, oracle):
    """Setters run on the controller's message pump while the workers process (controller.go:3493-3498).  A unit is pushed from ONE
    locked snapshot (versions, resolved values, a copy of the taps): hammer the setters from a second thread -- nothing crashes, no
    error is reported, every block is finite, and once the hammering stops the stream follows the oracle with the final parameters."""
    sr, frames, blocks = 48000, 1024, 60
    taps = {"Cab": synth_ir(3000, seed=21), "Room": synth_ir(700, seed=22)}
    irs = host.ImpulseResponses()
    irs.add("Cab", sr, -20, taps["Cab"])
    irs.add("Room", sr, -10, taps["Room"])
    eng = host.Engine(2, frames)
    chains = [eng.create_chain(irs) for _ in range(2)]
    for ch in chains:
        for t in (5, 19, 11):                                  # compressor, power amp, tone stack
            ch.SetBypass(ch.AppendUnit(t), False)
        ch.SetDiscreteValue(1, "filter_1", "Cab")
    eng.set_rendezvous(2, 2000)
    stop = threading.Event()

    def hammer():
        k = 0
        while not stop.is_set():
            k += 1
            chains[k & 1].SetNumericValue(0, "target_level", -30 + (k % 25))
            chains[k & 1].SetNumericValue(2, "middle", -(k % 20))
            chains[(k >> 1) & 1].SetDiscreteValue(1, "filter_2", "Room" if k & 2 else "- NONE -")
            chains[k & 1].SetNumericValue(1, "level_1", -(k % 12))

    x = np.stack([synth_signal(c, frames * blocks, sr) for c in range(2)])
    th = threading.Thread(target=hammer)
    th.start()
    try:
        for b in range(blocks // 2):
            sl = slice(b * frames, (b + 1) * frames)
            outs = [None, None]

            def work(c):
                outs[c] = chains[c].Process(x[c, sl], sr)

            ws = [threading.Thread(target=work, args=(c,)) for c in range(2)]
            for w in ws:
                w.start()
            for w in ws:
                w.join()
            assert all(np.all(np.isfinite(o)) for o in outs)
    finally:
        stop.set()
        th.join()
    assert eng.last_error() == ""
    # quiesce: set final values, then compare a fresh stretch with the oracle (the final Set of the power amp resets its tail)
    refs = []
    for c, ch in enumerate(chains):
        ch.SetNumericValue(0, "target_level", -18)
        ch.SetNumericValue(2, "middle", -4)
        ch.SetDiscreteValue(1, "filter_2", "Room")
        ch.SetNumericValue(1, "level_1", -3)
    got = np.zeros((2, frames * (blocks - blocks // 2)))
    for b in range(blocks // 2, blocks):
        sl = slice(b * frames, (b + 1) * frames)
        y = eng.process_all(x[:, sl], sr)
        got[:, (b - blocks // 2) * frames:(b - blocks // 2 + 1) * frames] = y
    # the compressor's and tone stack's STATE carries over from the hammered stretch, which the oracle (started here, from zero
    # state, with the final parameters) does not know: compare where the one-pole states of both have converged -- the power
    # amp's tail is exact (fresh filter at the final Set)
    a = oracle.Filter(taps["Cab"], sr, 10.0 ** (0.05 * -20)).normalize().multiply(10.0 ** (0.05 * -3))
    bflt = oracle.Filter(taps["Room"], sr, 10.0 ** (0.05 * -10)).normalize().multiply(1.0)
    composite = oracle.Filter([], sr).add(a).add(bflt).coefficients()
    for c in range(2):
        ref = oracle.Chain()
        ref.append_unit("compressor", params=[1, 30, -18])
        ref.append_unit("power_amp", fir=composite)
        ref.append_unit("tone_stack", params=[0, -4, -5, -5])
        want = np.concatenate([ref.process(x[c, b * frames:(b + 1) * frames], sr) for b in range(blocks // 2, blocks)])
        tail = slice(28 * frames, None)                        # the level follower forgets with exp(-t / 2400 samples): e^-12 by then
        assert rms(got[c, tail] - want[tail]) <= 1e-6, "channel %d" % c
    eng.close()


@pytest.mark.gpu
def test_engine_batch_run_over_three_shards_matches_the_oracle_pipeline(host, oracle):
    _engine_batch_case(host, oracle, [0, 0, 0])


@pytest.mark.gpu
def test_engine_batch_run_on_every_visible_device_matches_the_oracle_pipeline(host, oracle):
    """Engine::BatchRun with one shard per REAL device (skipped below two devices): every shard's gdg_batch_run_shard runs on its own GPU at the
    same time, the float64 partial master mixes meet on shard 0's device."""
    _engine_batch_case(host, oracle, _every_visible_device(host)[:7])


def _engine_batch_case(host, oracle, devices):
    """Engine::BatchRun (= what the Go shim's patched controller.processFiles calls): file bytes in, file bytes out, seven channels over
    THREE shards (three contexts on device 0), windows of 4 blocks, the master finished once from the shards' float64 partial mixes +
    the metronome as aux input (spatializer.go:300-310, controller.go:3123-3219).  Against the oracle's pipeline: decode ->
    resample.Time -> pad -> per block N chains, metronome, spatializer(+aux) -> encode; 24-bit files byte for byte."""
    BLOCK = 8192
    sr, nch = 48000, 7
    rng = np.random.default_rng(21)
    taps = {"Cab": synth_ir(2000, seed=3), "Room": synth_ir(5000, seed=4)}
    irs = host.ImpulseResponses()
    irs.add("Cab", sr, -20, taps["Cab"])
    irs.add("Room", sr, -10, taps["Room"])
    eng = host.Engine(nch, BLOCK, devices=devices)
    chains, refs = [], []
    for c in range(nch):
        ch, ref = eng.create_chain(irs), oracle.Chain()
        _full_chain(ch, ref, oracle, sr, taps)
        chains.append(ch)
        refs.append(ref)
    sp = host.Spatializer(eng, nch)
    sp.SetSampleRate(sr)
    ref_sp = oracle.Spatializer(nch)
    ref_sp.set_sample_rate(sr)
    for c in range(nch):
        a, d, l = float(rng.uniform(-90, 90)), float(rng.uniform(0.3, 5)), float(rng.uniform(0.2, 1))
        sp.SetAzimuth(c, a); sp.SetDistance(c, d); sp.SetLevel(c, l)
        ref_sp.set_azimuth(c, a); ref_sp.set_distance(c, d); ref_sp.set_level(c, l)
    tick, tock = rng.uniform(-0.5, 0.5, 900), rng.uniform(-0.5, 0.5, 500)
    m0 = eng.raw_context(0)                                   # the metronome lives on shard 0
    m0.metronome_set_sounds(tick, tock)
    m0.metronome_configure(3, 200, sr)
    ref_met = oracle.Metronome()
    ref_met.tick, ref_met.tock = tick, tock
    ref_met.s.beats_per_period, ref_met.s.bpm_speed, ref_met.s.sample_rate = 3, 200, sr
    # the files: 16-bit at the target rate, one at 44.1 kHz (resampled), one empty; different lengths
    lengths = [30000, 70000, 41000, 0, 20000, 65000, 9000]
    inputs, ref_in = [], []
    for c, n in enumerate(lengths):
        if n == 0:
            inputs.append(None)
            ref_in.append(np.zeros(0))
            continue
        rate = 44100 if c == 2 else sr
        data = oracle.wave_encode("lpcm16", 0.7 * synth_signal(c, n, rate))
        inputs.append((data, "lpcm16", rate))
        xd = oracle.wave_decode("lpcm16", data)
        ref_in.append(xd if rate == sr else oracle.resample_time(xd, rate, sr))
    longest = max(len(v) for v in ref_in)
    length = BLOCK * ((longest + BLOCK - 1) // BLOCK)
    xin = np.zeros((nch, length))
    for c, v in enumerate(ref_in):
        xin[c, :len(v)] = v
    ref_out = np.zeros((nch + 3, length))
    for b in range(length // BLOCK):
        sl = slice(b * BLOCK, (b + 1) * BLOCK)
        for c in range(nch):
            ref_out[c, sl] = refs[c].process(xin[c, sl], sr)
        ref_out[nch + 2, sl] = ref_met.process(BLOCK)
        ref_out[nch, sl], ref_out[nch + 1, sl] = ref_sp.process(ref_out[:nch, sl], aux=ref_out[nch + 2, sl])
    outs = eng.batch_run(inputs, sr, "lpcm24", window=4, metronome_to_master=True)
    assert eng.last_error() == ""
    assert len(outs) == nch + 3 and all(o.size == 3 * length for o in outs)
    for r in range(nch + 3):
        np.testing.assert_array_equal(outs[r], oracle.wave_encode("lpcm24", ref_out[r]), err_msg="output %d" % r)
    del sp
    eng.close()
