"""GPU parity of the data formats either side of the hot path: the WAVE sample codecs
(bit exact), resample.Time (Lanczos-3) and the level meters, against the oracle and the reference's own golden
vectors (wave/wave_test.go, resample/resample_test.go, level/level_test.go).  Run with `pytest -m gpu`."""
import numpy as np
import pytest

from helpers import TOL_RMS, package, rms, synth_signal

pytestmark = pytest.mark.gpu

WAVE_CASES = [("lpcm8", "PCM8", 1), ("lpcm16", "PCM16", 2), ("lpcm24", "PCM24", 3), ("lpcm32", "PCM32", 4), ("ieee32", "IEEE32", 4), ("ieee64", "IEEE64", 8)]


@pytest.fixture(scope="module")
def pkg():
    p = package()
    assert p.device_count() > 0
    return p


@pytest.fixture(scope="module")
def ctx(pkg):
    return pkg.Context(1, 8192)


def awkward_samples(n, seed):
    """Samples that exercise every clipping / truncation branch: in range, beyond +-1, exact +-1, +-0, tiny, half steps."""
    rng = np.random.default_rng(seed)
    x = rng.uniform(-1.3, 1.3, n)
    special = np.array([0.0, -0.0, 1.0, -1.0, 1.0000001, -1.0000001, 0.5, -0.5, 1e-12, -1e-12, 2.0, -2.0,
                        1.0 / 127.0, -1.0 / 127.0, 32767.5 / 32767.5, 0.999999999, -0.999999999, 3.0518509e-5, -3.0518509e-5])
    x[:len(special)] = special
    k = rng.integers(-40000, 40000, n // 4)                   # values on and next to 16-bit quantisation steps
    x[n // 2:n // 2 + len(k)] = k / 32767.5 + rng.choice([0.0, 1e-13, -1e-13], len(k))
    return x


# ---- wave codecs: the reference's golden vectors, through the device path -----------------------------------------
@pytest.mark.parametrize("fmt,tag,width", WAVE_CASES)
def test_wave_export_golden_bytes(ctx, golden, fmt, tag, width):
    t = golden("wave")["tests"]["TestExport%sMono" % tag]
    samples = t["samples"]["value"]
    want = np.array(t["expectedOutput"]["value"], dtype=np.uint8)[-len(samples) * width:]
    np.testing.assert_array_equal(ctx.wave_encode(fmt, samples), want)


@pytest.mark.parametrize("fmt,tag,width", WAVE_CASES)
def test_wave_import_golden_samples(ctx, oracle, golden, fmt, tag, width):
    t = golden("wave")["tests"]["TestImport%sMono" % tag]
    want = np.array(t["expectedSamples"]["value"])
    data = np.array(t["buf"]["value"], dtype=np.uint8)[-len(want) * width:]
    tol = {"lpcm8": 0.078125, "lpcm16": 3.0518e-5, "lpcm24": 1.1921e-7, "lpcm32": 4.6567e-10, "ieee32": 1.1921e-7, "ieee64": 1.0e-16}[fmt]
    got = ctx.wave_decode(fmt, data)
    np.testing.assert_allclose(got, want, rtol=0, atol=tol)          # the reference's own tolerance
    np.testing.assert_array_equal(got, oracle.wave_decode(fmt, data))  # and bit exact with the oracle


# ---- wave codecs: bit exact with the oracle on large awkward inputs -------------------------------------------------
@pytest.mark.parametrize("fmt,tag,width", WAVE_CASES)
@pytest.mark.parametrize("n", [1, 255, 4097, 1 << 20])
def test_wave_encode_bit_exact(ctx, oracle, fmt, tag, width, n):
    x = awkward_samples(max(n, 64), 7 + n)[:n]
    got = ctx.wave_encode(fmt, x)
    assert got.size == n * width
    np.testing.assert_array_equal(got, oracle.wave_encode(fmt, x))


@pytest.mark.parametrize("fmt,tag,width", WAVE_CASES)
@pytest.mark.parametrize("n", [1, 255, 4097, 1 << 20])
def test_wave_decode_bit_exact(ctx, oracle, fmt, tag, width, n):
    rng = np.random.default_rng(11 + n)
    data = rng.integers(0, 256, n * width, dtype=np.uint8)
    if fmt in ("ieee32", "ieee64"):       # arbitrary bit patterns include NaNs: compare the bits, not the values
        got = ctx.wave_decode(fmt, data)
        np.testing.assert_array_equal(got.view(np.uint64), oracle.wave_decode(fmt, data).view(np.uint64))
    else:
        np.testing.assert_array_equal(ctx.wave_decode(fmt, data), oracle.wave_decode(fmt, data))


@pytest.mark.parametrize("fmt,tag,width", WAVE_CASES)
def test_wave_all_codes_round_trip(ctx, oracle, fmt, tag, width):
    """Every 8- and 16-bit code (and a dense sweep of the wider ones) decodes and re-encodes like the oracle."""
    if width == 1:
        data = np.arange(256, dtype=np.uint8)
    elif width == 2:
        data = np.arange(65536, dtype=np.uint16).view(np.uint8)
    else:
        data = np.random.default_rng(3).integers(0, 256, 65536 * width, dtype=np.uint8)
        if fmt.startswith("ieee"):
            v = np.linspace(-1.5, 1.5, 65536)
            data = (v.astype(np.float32) if width == 4 else v).view(np.uint8)
    dec = ctx.wave_decode(fmt, data)
    np.testing.assert_array_equal(dec, oracle.wave_decode(fmt, data))
    np.testing.assert_array_equal(ctx.wave_encode(fmt, dec), oracle.wave_encode(fmt, dec))


@pytest.mark.parametrize("channels", [2, 3, 8])
def test_wave_interleaving(ctx, oracle, channels):
    """channelsToSamples / samplesToChannels (wave.go:173-270): planar <-> interleaved."""
    per = 1000
    x = np.stack([synth_signal(c, per, 48000) for c in range(channels)])
    inter = x.T.reshape(-1)                                           # sample j of channel c at j * C + c
    for fmt in ("lpcm16", "lpcm24", "ieee64"):
        b = ctx.wave_encode(fmt, x)
        np.testing.assert_array_equal(b, oracle.wave_encode(fmt, inter))
        back = ctx.wave_decode(fmt, b, channels=channels)
        np.testing.assert_array_equal(back, oracle.wave_decode(fmt, b).reshape(per, channels).T)


def test_wave_empty_and_errors(pkg, ctx):
    assert ctx.wave_encode("lpcm16", np.zeros(0)).size == 0
    assert ctx.wave_decode("lpcm24", np.zeros(0, dtype=np.uint8)).size == 0
    with pytest.raises(pkg.GdgError):
        ctx.wave_encode(17, np.zeros(4))
    assert pkg.lib().gdg_wave_bytes_per_sample(17) == 0
    assert [pkg.lib().gdg_wave_bytes_per_sample(f) for f in range(6)] == [1, 2, 3, 4, 4, 8]


# ---- resample.Time ------------------------------------------------------------------------------------------------------
def test_resample_time_golden(ctx, golden):
    t = golden("resample")["tests"]["TestTimeSeries"]                # resample_test.go:48-99, tolerance 1e-8
    for x, up, down in zip(t["in"]["value"], t["outExpectedUp"]["value"], t["outExpectedDown"]["value"]):
        got_up = ctx.resample_time(x, 96000, 192000)
        assert len(got_up) == len(up)
        np.testing.assert_allclose(got_up, up, atol=1e-8, rtol=0)
        got_down = ctx.resample_time(x, 96000, 44100)
        assert len(got_down) == len(down)
        np.testing.assert_allclose(got_down, down, atol=1e-8, rtol=0)


@pytest.mark.parametrize("src,dst", [(44100, 96000), (96000, 44100), (48000, 192000), (192000, 48000), (44100, 44100), (22050, 192000), (96000, 88200)])
@pytest.mark.parametrize("n", [1, 7, 1000, 65537])
def test_resample_time_matches_oracle(pkg, ctx, oracle, src, dst, n):
    x = synth_signal(n % 48, n, src)
    got = ctx.resample_time(x, src, dst)
    want = oracle.resample_time(x, src, dst)
    assert len(got) == len(want) == pkg.lib().gdg_resample_time_length(n, src, dst)
    if len(want):
        assert rms(got - want) <= TOL_RMS * max(1.0, rms(want))
        np.testing.assert_allclose(got, want, rtol=0, atol=1e-12)


def test_resample_time_rejects_wrong_length(pkg, ctx):
    x = np.zeros(100)
    out = np.zeros(500)
    rc = pkg.lib().gdg_resample_time(ctx._h, x.ctypes.data, 100, 44100, 96000, out.ctypes.data, 500)
    assert rc == pkg.GDG_ERR_INVALID


# ---- level meters --------------------------------------------------------------------------------------------------------
def test_level_meter_known_readings(pkg):
    sr = 96000                                                        # level_test.go:17-224: one second of a 1 Hz sine
    a = np.sin(2.0 * np.pi * (np.arange(sr) / float(sr)))
    ctx = pkg.Context(1, 8192)
    ctx.meter_configure(3)
    ctx.meter_set_enabled(True, 0)
    ctx.meter_set_enabled(True, 1)                                    # port 2 stays disabled
    ctx.meter_process(np.stack([a, 0.5 * a, a]), sr)
    lv, pk = ctx.meter_analyze()
    assert (lv[0], pk[0]) == (-3, 0)
    assert (lv[1], pk[1]) == (-9, -6)
    assert (lv[2], pk[2]) == (-200, -200)
    ctx.meter_set_enabled(False)
    lv, pk = ctx.meter_analyze()
    assert list(lv) == [-200] * 3 and list(pk) == [-200] * 3


def meter_signals(kind, n, sr):
    rng = np.random.default_rng(sum(map(ord, kind)))
    t = np.arange(n) / float(sr)
    if kind == "noise":
        return rng.normal(0, 0.3, n)
    if kind == "burst_then_silence":                               # record, full hold, then decay phase
        x = np.zeros(n)
        x[1000:1400] = rng.normal(0, 0.5, 400)
        return x
    if kind == "plateau_ties":                                     # equal maxima: the LAST one restarts the hold
        x = 0.1 * np.sin(2 * np.pi * 50 * t)
        x[::7777] = 0.75
        x[5::9001] = -0.75
        return x
    if kind == "decay_recapture":                                  # a small tone that the decaying peak meets again
        x = np.zeros(n)
        x[10] = 1.0
        x[n // 2:] = 0.05 * np.sin(2 * np.pi * 100 * t[n // 2:])
        return x
    if kind == "zeros":
        return np.zeros(n)
    raise KeyError(kind)


@pytest.mark.parametrize("sr,frames", [(96000, 8192), (44100, 1000), (192000, 8192), (3000, 8192)])
def test_level_meter_state_matches_oracle(pkg, oracle, sr, frames):
    kinds = ["noise", "burst_then_silence", "plateau_ties", "decay_recapture", "zeros"]
    total = int(3.2 * sr) // frames * frames + frames              # past the 2 s hold into the decay
    x = np.stack([meter_signals(k, total, sr) for k in kinds])
    ctx = pkg.Context(1, 8192)
    ctx.meter_configure(len(kinds))
    ctx.meter_set_enabled(True)
    refs = [oracle.ChannelMeter() for _ in kinds]
    for r in refs:
        r.set_enabled(True)
    for b in range(0, total, frames):
        ctx.meter_process(x[:, b:b + frames], sr)
        for p, r in enumerate(refs):
            r.process(x[p, b:b + frames], sr)
        if (b // frames) % 5 == 0 or b + frames >= total:
            lv, pk = ctx.meter_analyze()
            for p, r in enumerate(refs):
                cur, peak, cnt = ctx.meter_state(p)
                rc, rp, rn = r.state
                assert cnt == rn, (kinds[p], b)
                assert abs(cur - rc) <= 1e-12 * max(rc, 1e-300), (kinds[p], b)
                assert abs(peak - rp) <= 1e-12 * max(rp, 1e-300), (kinds[p], b)
                assert (lv[p], pk[p]) == r.analyze(), (kinds[p], b)


def test_level_meter_device_rows(pkg, oracle):
    """The device entry point over the rows of a [ports][frames] buffer with a row stride."""
    sr, frames, ports = 192000, 8192, 67
    x = np.stack([synth_signal(p, frames, sr) * (0.1 + 0.01 * p) for p in range(ports)])
    ctx = pkg.Context(1, 8192)
    d = ctx.alloc(ports, frames)
    d.upload(x)
    ctx.meter_configure(ports)
    ctx.meter_set_enabled(True)
    for _ in range(3):
        ctx.meter_process_device(d, frames, frames, sr)
    lv, pk = ctx.meter_analyze()
    for p in range(ports):
        r = oracle.ChannelMeter()
        r.set_enabled(True)
        for _ in range(3):
            r.process(x[p], sr)
        assert (lv[p], pk[p]) == r.analyze()
        assert ctx.meter_state(p)[2] == r.state[2]


def test_wave_device_entry_unaligned_buffers(pkg, ctx, oracle):
    """The four-samples-per-thread fast path needs aligned buffers; odd device offsets take the scalar kernel."""
    n = 5003
    x = awkward_samples(n, 99)
    d_x = ctx.alloc(1, n + 2)
    d_b = ctx.alloc(1, n + 2)
    for fmt, f in pkg.WAVE_FORMATS.items():
        w = pkg.lib().gdg_wave_bytes_per_sample(f)
        for off_s, off_b in ((0, 0), (8, 1), (8, 2), (0, 3), (8, 4)):
            if fmt in ("lpcm32", "ieee32") and off_b % 4:
                continue                                           # 32-bit containers stay naturally aligned
            if fmt == "lpcm16" and off_b % 2:
                continue
            if fmt == "ieee64" and off_b:
                continue
            buf = np.zeros(n + 2)
            buf.view(np.uint8)[off_s:off_s + 8 * n] = x.view(np.uint8)
            d_x.upload(buf)
            ctx._check(pkg.lib().gdg_wave_encode_device(ctx._h, f, d_x.ptr + off_s, n, 1, d_b.ptr + off_b))
            got = d_b.download().view(np.uint8).reshape(-1)[off_b:off_b + n * w]
            want = oracle.wave_encode(fmt, x)
            np.testing.assert_array_equal(got, want)
            ctx._check(pkg.lib().gdg_wave_decode_device(ctx._h, f, d_b.ptr + off_b, n, 1, d_x.ptr + off_s))
            back = d_x.download().view(np.uint8).reshape(-1)[off_s:off_s + 8 * n].view(np.float64)
            np.testing.assert_array_equal(back.view(np.uint64), oracle.wave_decode(fmt, want).view(np.uint64))


# ---- metronome -------------------------------------------------------------------------------------------------------------
@pytest.mark.parametrize("sr,bpm,beats,frames", [(192000, 120, 4, 8192), (48000, 208, 3, 1024), (44100, 60, 0, 5000), (96000, 30, 7, 8192),
                                                  (1000, 60001, 2, 64), (8000, 480, 1, 1000)])
def test_metronome_matches_oracle(pkg, oracle, sr, bpm, beats, frames):
    rng = np.random.default_rng(bpm)
    tick, tock = rng.uniform(-1, 1, 1500), rng.uniform(-1, 1, 900)
    ctx = pkg.Context(1, 8192)
    ref = oracle.Metronome()
    blocks = 40
    for b in range(blocks):
        if b == 0:
            ctx.metronome_set_sounds(tick, tock)
            ref.tick, ref.tock = tick, tock
            ctx.metronome_configure(beats, bpm, sr)
            ref.s.beats_per_period, ref.s.bpm_speed, ref.s.sample_rate = beats, bpm, sr
        if b == 17:                                   # a speed change mid-stream leaves the counters where they are
            ctx.metronome_configure(beats, bpm * 3, sr)
            ref.s.bpm_speed = bpm * 3
        if b == 25:                                   # fewer beats per period than the current tick counter, and no tock sound
            nb = 2 if beats != 2 else 1
            ctx.metronome_configure(nb, bpm * 3, sr)
            ref.s.beats_per_period = nb
            ctx.metronome_set_sounds(tick, None)
            ref.tock = None
        np.testing.assert_array_equal(ctx.metronome_process(frames), ref.process(frames), err_msg="block %d" % b)
    ctx.close()


def test_metronome_defaults_and_errors(pkg, oracle):
    ctx = pkg.Context(1, 1024)
    np.testing.assert_array_equal(ctx.metronome_process(1024), np.zeros(1024))       # no sounds set: silence (metronome.go:98, :108)
    with pytest.raises(pkg.GdgError):
        ctx.metronome_configure(4, 0, 48000)
    ctx.close()
