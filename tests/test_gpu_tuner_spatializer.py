"""GPU parity of the tuner (batched 262144-point real-FFT autocorrelation) and of the spatializer's
N -> 2 mixdown against the oracle.  Run with `pytest -m gpu`."""
import numpy as np
import pytest

from helpers import TOL_RMS, package, rms, synth_signal

pytestmark = pytest.mark.gpu

# the six guitar-string notes of tuner/tuner_test.go (its WAV fixtures are missing upstream): synthetic harmonic tones
STRINGS = [("D2", 73.4162), ("A2", 110.0), ("D3", 146.8324), ("G3", 195.9978), ("H3", 246.9417), ("E4", 329.6276)]


@pytest.fixture(scope="module")
def pkg():
    p = package()
    assert p.device_count() > 0
    return p


def tone(freq, n, sr, detune_cents=0.0, phase=0.0):
    f = freq * 2.0 ** (detune_cents / 1200.0)
    t = np.arange(n) / float(sr)
    return sum(a * np.sin(2 * np.pi * f * h * t + phase * h) for h, a in ((1, 0.5), (2, 0.25), (3, 0.12), (4, 0.06)))


# 300 kHz: the lag window reaches past 4096, i.e. the long (262144-point) analysis; the other rates take the short-lag kernel
@pytest.mark.parametrize("sr,frames", [(96000, 8192), (192000, 8192), (44100, 1000), (22050, 512), (300000, 8192)])
def test_tuner_matches_oracle_on_string_tones(pkg, oracle, sr, frames):
    nch = len(STRINGS)
    total = 96000 + 3 * frames                      # wraps the ring
    x = np.stack([tone(f, total, sr, detune_cents=1.0 * (i - 2), phase=0.3 * i) for i, (_, f) in enumerate(STRINGS)])
    ctx = pkg.Context(nch, frames)
    refs = [oracle.Tuner() for _ in range(nch)]
    for b in range(0, total, frames):
        blk = x[:, b:b + frames]
        ctx.tuner_enqueue(blk, sr)
        for c in range(nch):
            refs[c].process(blk[c], sr)
    got = ctx.tuner_analyze()
    for c, (name, _) in enumerate(STRINGS):
        want = refs[c].analyze()
        assert got[c]["note"] == want["note"] == name
        assert abs(want["cents"]) <= 5                                  # tuner_test.go:95-106 criterion
        assert got[c]["cents"] == want["cents"]
        assert abs(got[c]["frequency"] - want["frequency"]) / want["frequency"] <= 1e-9
    ctx.close()


def test_tuner_analysis_split_over_workgroups_agrees(pkg, oracle):
    """Fewer channels than CUs: a channel's 24 blocks are transformed by up to 8 workgroups (runs of blocks, each repeating its predecessor
    for the cross term), the partial sums added in part order.  Every split gives the note, the cents and -- to rounding -- the frequency of
    the one-workgroup analysis and of the oracle (option "tuner_parts" forces the count; 0 = the default picks it from the channel count)."""
    sr, frames = 192000, 8192
    nch = len(STRINGS)
    total = 96000 + 5 * frames
    x = np.stack([tone(f, total, sr, detune_cents=2.0 * (i - 3), phase=0.2 * i) + 0.01 * synth_signal(i, total, sr) for i, (_, f) in enumerate(STRINGS)])
    ctx = pkg.Context(nch, frames)
    refs = [oracle.Tuner() for _ in range(nch)]
    for b in range(0, total, frames):
        blk = x[:, b:b + frames]
        ctx.tuner_enqueue(blk, sr)
        for c in range(nch):
            refs[c].process(blk[c], sr)
    results = {}
    try:
        for parts in (1, 2, 3, 5, 8):
            ctx.set_option("tuner_parts", parts)
            results[parts] = ctx.tuner_analyze()
    finally:
        ctx.set_option("tuner_parts", 0)                # process-wide: back to "by channel count" for the tests that follow
    results["auto"] = ctx.tuner_analyze()
    ctx.close()
    for c in range(nch):
        want = refs[c].analyze()
        for key, got in results.items():
            assert got[c]["note"] == want["note"] and got[c]["cents"] == want["cents"], (key, c)
            assert abs(got[c]["frequency"] - want["frequency"]) / want["frequency"] <= 1e-9, (key, c)
            assert abs(got[c]["frequency"] - results[1][c]["frequency"]) / want["frequency"] <= 1e-12, (key, c)


def test_tuner_noise_and_silence(pkg, oracle):
    sr, frames, nch = 48000, 4096, 3
    ctx = pkg.Context(nch, frames)
    refs = [oracle.Tuner() for _ in range(nch)]
    x = np.stack([synth_signal(5, 30 * frames, sr), 0.3 * np.random.default_rng(3).standard_normal(30 * frames), synth_signal(40, 30 * frames, sr)])
    for b in range(0, x.shape[1], frames):
        ctx.tuner_enqueue(x[:, b:b + frames], sr)
        for c in range(nch):
            refs[c].process(x[c, b:b + frames], sr)
    got = ctx.tuner_analyze()
    for c in range(nch):
        want = refs[c].analyze()
        assert got[c]["note_index"] == want["note_index"]
        assert abs(got[c]["frequency"] - want["frequency"]) / want["frequency"] <= 1e-9
    ctx.close()


@pytest.mark.parametrize("nch,frames,sr", [(5, 1024, 48000), (70, 8192, 192000), (3, 37, 96000)])
def test_spatializer_matches_oracle(pkg, oracle, nch, frames, sr):
    rng = np.random.default_rng(11)
    ctx = pkg.Context(nch, frames)
    ref = oracle.Spatializer(nch)
    ctx.spatializer_set_sample_rate(sr)
    ref.set_sample_rate(sr)
    for c in range(nch):
        az, dist, lvl = float(rng.uniform(-180, 180)), float(rng.uniform(0, 10)), float(rng.uniform(0, 1))
        if c == 0:
            az, dist, lvl = 0.0, 0.0, 1.0            # the defaults: gains clipped to 1, zero delay
        if c == 1:
            az, dist = 90.0, 0.05                    # largest inter-aural delay
        ctx.spatializer_set_position(c, az, dist, lvl)
        assert ref.set_azimuth(c, az) == 0 and ref.set_distance(c, dist) == 0 and ref.set_level(c, lvl) == 0
    x = np.stack([synth_signal(c, frames * 4, sr) for c in range(nch)])
    for b in range(0, x.shape[1], frames):
        blk = x[:, b:b + frames]
        gl, gr = ctx.spatialize(blk)
        wl, wr = ref.process(blk)
        assert rms(gl - wl) <= TOL_RMS and rms(gr - wr) <= TOL_RMS
    ctx.close()


def test_spatializer_rejects_out_of_range(pkg):
    ctx = pkg.Context(2, 64)
    with pytest.raises(pkg.GdgError):
        ctx.spatializer_set_position(0, 0.0, 11.0, 1.0)
    with pytest.raises(pkg.GdgError):
        ctx.spatializer_set_position(0, 0.0, 1.0, 1.5)
    with pytest.raises(pkg.GdgError):
        ctx.spatializer_set_position(2, 0.0, 1.0, 1.0)
    ctx.close()


def test_spatializer_wide_shard(pkg, oracle):
    """More channels than the spatializer stages descriptors for in LDS (512): 700 channels take the variant that reads them from HBM."""
    nch, frames, sr = 700, 512, 48000
    rng = np.random.default_rng(4)
    ctx = pkg.Context(nch, frames)
    ref = oracle.Spatializer(nch)
    ctx.spatializer_set_sample_rate(sr)
    ref.set_sample_rate(sr)
    for c in range(nch):
        a, d, l = float(rng.uniform(-180, 180)), float(rng.uniform(0, 10)), float(rng.uniform(0, 1))
        ctx.spatializer_set_position(c, a, d, l)
        ref.set_azimuth(c, a); ref.set_distance(c, d); ref.set_level(c, l)
    for b in range(3):
        x = rng.uniform(-0.5, 0.5, (nch, frames))
        gl, gr = ctx.spatialize(x)
        wl, wr = ref.process(x)
        assert rms(gl - wl) <= TOL_RMS and rms(gr - wr) <= TOL_RMS, b
    ctx.close()


def test_tuner_replace_is_twelve_enqueues_in_one(pkg, oracle):
    """gdg_tuner_replace: the whole ring of ONE channel, oldest first (what circular.Buffer.Retrieve hands the Go overlay), whatever the write
    position the context's channels share -- same analysis as feeding the samples block by block; any length but 96000 is refused."""
    sr, frames = 192000, 8192
    nch = 3
    total = 96000 + 5 * frames
    x = np.stack([tone(f, total, sr, detune_cents=3.0 * i, phase=0.1 * i) for i, (_, f) in enumerate(STRINGS[:nch])])
    ctx = pkg.Context(nch, frames)
    for b in range(0, 2 * frames + 100, frames):                       # leave the shared write position somewhere inside the ring
        ctx.tuner_enqueue(x[:, b:b + frames][:, :min(frames, 2 * frames + 100 - b)], sr)
    refs = []
    for c in range(nch):
        ring = x[c, total - 96000:]
        ctx.tuner_replace(c, ring, sr)
        t = oracle.Tuner()
        for b in range(0, 96000, frames):
            t.process(ring[b:b + frames], sr)
        refs.append(t.analyze())
    got = ctx.tuner_analyze()
    for c in range(nch):
        assert got[c]["note"] == refs[c]["note"] and got[c]["cents"] == refs[c]["cents"], c
        assert abs(got[c]["frequency"] - refs[c]["frequency"]) / refs[c]["frequency"] <= 1e-9, c
    with pytest.raises(pkg.GdgError):
        ctx.tuner_replace(0, x[0, :95999], sr)
    with pytest.raises(pkg.GdgError):
        ctx.tuner_replace(nch, x[0, :96000], sr)
    ctx.close()
