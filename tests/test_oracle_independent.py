"""Second, independent formulations of the parts of the path the reference does NOT test itself
(effects units, filter.Process for long IRs, chain, spatializer, tuner).  Each check re-derives the unit
from its mathematical definition with numpy / scipy.signal — recursive filters via
lfilter, delays via index arithmetic on the concatenated stream — never from the oracle's C code, and must
agree with the oracle to ~1e-12.  CPU only.
"""
import numpy as np
import pytest
from scipy.signal import lfilter

from helpers import synth_ir, synth_signal

SR = 48000
TOL = 1e-11


def stream(unit, x, frames, sr=SR):
    return np.concatenate([unit.process(x[i:i + frames], sr) for i in range(0, len(x), frames)])


def db(v):
    return 10.0 ** (v / 20.0)


def onepole_state(u, a, s0=0.0):
    """s[n] = s[n-1] + (u[n] - s[n-1]) a  -> returns s AFTER each sample"""
    y, _ = lfilter([a], [1.0, -(1.0 - a)], u, zi=[(1.0 - a) * s0])
    return y


def old(s, s0=0.0):
    return np.concatenate([[s0], s[:-1]])


@pytest.fixture(scope="module")
def x():
    return synth_signal(3, 6000, SR)


# ---- filter.Process: streaming y = clip(x * h), any block split ----------------------------------------------
@pytest.mark.parametrize("taps,frames", [(77, 16), (300, 256), (1000, 256), (5000, 1024), (2049, 2048)])
def test_filter_process_is_clipped_linear_convolution(oracle, taps, frames):
    h = synth_ir(taps) * 2.5
    xs = synth_signal(1, frames * 7, SR)
    f = oracle.Filter(h)
    got = np.concatenate([f.process(xs[i:i + frames]) for i in range(0, len(xs), frames)])
    want = np.clip(np.convolve(xs, h)[:len(xs)], -1.0, 1.0)
    np.testing.assert_allclose(got, want, rtol=0, atol=1e-12)


def test_filter_algebra(oracle):
    a, b = oracle.Filter([1.0, 2.0, 3.0], 48000, 0.5), oracle.Filter([1.0, 1.0], 48000, 1.0)
    np.testing.assert_array_equal(a.add(b).coefficients(), [2.0, 3.0, 3.0])
    np.testing.assert_allclose(a.normalize().coefficients(), 0.5 * np.array([1, 2, 3]) / np.sqrt(14.0), atol=1e-15)
    np.testing.assert_array_equal(a.multiply(2.0).coefficients(), [2.0, 4.0, 6.0])
    assert a.add(oracle.Filter([1.0], 44100)) is None                     # sample-rate mismatch
    assert len(oracle.Filter(synth_ir(3000), 48000, 1.0).reduce(1024).coefficients()) == 1024


# ---- memoryless units -------------------------------------------------------------------------------------------
def test_overdrive_distortion_excess(oracle, x):
    u = oracle.Unit("overdrive"); u.set_params([5, 15, 70, -3, 1, 0])
    g, d, lv = db(20), 0.7, db(-3)
    np.testing.assert_allclose(stream(u, x, 1000), lv * (d * (2.0 / (1.0 + np.exp(-g * x)) - 1.0) + (1 - d) * x), atol=TOL)
    u = oracle.Unit("overdrive"); u.set_params([0, 20, 100, 0, 0, 0])
    np.testing.assert_allclose(stream(u, x, 1000), (2 / np.pi) * np.arctan((np.pi / 4) * db(20) * x), atol=TOL)
    u = oracle.Unit("distortion"); u.set_params([10, 10, -6, 0])
    np.testing.assert_allclose(stream(u, x, 1000), db(-6) * np.clip(db(20) * x, -1, 1), atol=TOL)
    u = oracle.Unit("excess"); u.set_params([25, -2, 0])
    p = db(25) * x                                                       # triangle fold of p into [-1, 1]
    np.testing.assert_allclose(stream(u, x, 1000), db(-2) * (2 / np.pi) * np.arcsin(np.sin(np.pi * p / 2)), atol=1e-9)


# ---- recursive units through lfilter --------------------------------------------------------------------------------
def test_tone_stack(oracle, x):
    u = oracle.Unit("tone_stack"); u.set_params([-3, -1, 0, -12])
    edges = [20.0, 300.0, 3000.0, 6000.0, 20000.0]
    fac = [db(-3), db(-1), db(0), db(-12)]
    want = np.zeros_like(x)
    for j in range(4):
        aH, aL = 1 - np.exp(-2 * np.pi * edges[j] / SR), 1 - np.exp(-2 * np.pi * edges[j + 1] / SR)
        h = onepole_state(x, aH)
        l = onepole_state(x - old(h), aL)
        want += fac[j] * old(l)
    np.testing.assert_allclose(stream(u, x, 750), np.clip(want, -1, 1), atol=TOL)


def test_cabinet(oracle, x):
    u = oracle.Unit("cabinet")
    v = x.copy()
    for f in (300.0, 120.0, 80.0):
        v = v - old(onepole_state(v, 1 - np.exp(-2 * np.pi * f / SR)))
    for f in (3000.0, 4000.0, 5000.0, 6000.0):
        v = old(onepole_state(v, 1 - np.exp(-2 * np.pi * f / SR)))
    np.testing.assert_allclose(stream(u, x, 600), np.clip(v, -1, 1), atol=TOL)


def envelope(x, follow, sr):
    a = np.exp(-20.0 / sr)
    if follow == 1:
        return onepole_state(np.abs(x), 1 - a)
    e, out = 0.0, np.empty_like(x)
    for i, v in enumerate(np.abs(x)):
        e = max(e * a, v)
        out[i] = e
    return out


@pytest.mark.parametrize("follow", [0, 1])
def test_compressor(oracle, x, follow):
    u = oracle.Unit("compressor"); u.set_params([follow, 20, -12])
    with np.errstate(divide="ignore"):
        gain = np.minimum(db(-12) / envelope(x, follow, SR), db(20))
    np.testing.assert_allclose(stream(u, x, 500), np.clip(gain * x, -1, 1), atol=TOL)


@pytest.mark.parametrize("follow", [0, 1])
def test_fuzz(oracle, x, follow):
    u = oracle.Unit("fuzz"); u.set_params([follow, 40, 5, 20, 80, -3, 0])
    a = np.exp(-20.0 / SR)
    e = envelope(x, follow, SR)
    p = 0.8 * np.clip(db(25) * (x - 0.4 * e), -1, 1) + (1 - 0.8) * x
    c = onepole_state(p, 1 - a)
    np.testing.assert_allclose(stream(u, x, 1000), db(-3) * np.clip(p - c, -1, 1), atol=TOL)


def test_bandpass(oracle, x):
    u = oracle.Unit("bandpass"); u.set_params([2, 4000, 200])                 # order 6, frequencies swapped on purpose
    aH, aL = 1 - np.exp(-2 * np.pi * 200 / SR), 1 - np.exp(-2 * np.pi * 4000 / SR)
    p = x * 3.0                                                               # drive it into the inter-stage clip
    got = stream(u, p, 500)
    for _ in range(3):
        h = onepole_state(p, aH)
        l = onepole_state(p - old(h), aL)
        p = np.clip(old(l), -1, 1)
    np.testing.assert_allclose(got, p, atol=TOL)


def test_auto_wah(oracle, x):
    u = oracle.Unit("auto_wah"); u.set_params([1, -30, -6, 400, 5000])
    e = envelope(x, 1, SR)
    with np.errstate(divide="ignore"):
        level = 20 * np.log10(e)
    fc = np.where(level <= -30, 400.0, np.where(level >= -6, 5000.0, 400.0 + (5000.0 - 400.0) / 24.0 * (level + 30)))
    alpha = 1 - np.exp(-fc / SR)
    v = x.copy()
    h, l = np.zeros(8), np.zeros(8)
    out = np.empty_like(x)
    for i in range(len(x)):                                                   # time-varying: plain loop
        s = v[i]
        for j in range(8):
            d = s - h[j]
            h[j] += d * alpha[i]
            l[j] += (d - l[j]) * alpha[i]
            s = l[j]
        out[i] = s
    np.testing.assert_allclose(stream(u, x, 1000), np.clip(256 * out, -1, 1), atol=1e-10)


# ---- delay-type units: index arithmetic on the whole stream -------------------------------------------------------------
def hist(xs, idx):
    """sample at (possibly negative or fractional-free) integer index, zero before the stream"""
    idx = np.asarray(idx)
    return np.where(idx >= 0, xs[np.clip(idx, 0, len(xs) - 1)], 0.0)


def frac(xs, i, D):
    fl, ce = np.floor(D), np.ceil(D)
    return (1 - (D - fl)) * hist(xs, i - fl.astype(int)) + (1 - (ce - D)) * hist(xs, i - ce.astype(int))


def lfo_phase(n_total, frames, omega, advance, sr):
    """phase of sample i: the per-call phase advances by `advance` seconds per call (the buffer length quirk)"""
    ph = np.empty(n_total)
    prev = 0.0
    for b in range(0, n_total, frames):
        k = np.arange(min(frames, n_total - b))
        ph[b:b + len(k)] = np.fmod(prev + omega * (k / sr), 2 * np.pi)
        prev = np.fmod(prev + omega * advance, 2 * np.pi)
    return ph


def test_chorus(oracle, x):
    frames = 1000
    u = oracle.Unit("chorus"); u.set_params([60, 45])
    B = int(np.floor(0.05 * SR + 0.5))
    ph0 = lfo_phase(len(x), frames, 0.001 * np.pi * 45, B / SR, SR)
    i = np.arange(len(x))
    acc = np.zeros_like(x)
    for j in range(5):
        ph = np.fmod(ph0 + 0.4 * np.pi * j, 2 * np.pi)
        acc += 0.2 * frac(x, i, 0.001 * (40 + 6.0 * np.sin(ph)) * SR)
    np.testing.assert_allclose(stream(u, x, frames), 0.5 * x + 0.5 * acc, atol=1e-10)


@pytest.mark.parametrize("unit,params", [("flanger", [80, 20]), ("phaser", [50, 35, -30])])
def test_flanger_phaser(oracle, x, unit, params):
    frames = 500
    u = oracle.Unit(unit); u.set_params(params)
    B = int(np.floor(0.002 * SR + 0.5))
    d = params[0] / 100.0
    ph = lfo_phase(len(x), frames, 0.02 * np.pi * params[1], B * (1.0 / SR), SR)
    delayed = frac(x, np.arange(len(x)), 0.001 * (d + d * np.sin(ph)) * SR)
    if unit == "phaser":
        p = 0.5 * np.sin(np.pi / 180.0 * params[2])
        want = (1 - abs(p)) * x + p * delayed
    else:
        want = 0.5 * x + 0.5 * delayed
    np.testing.assert_allclose(stream(u, x, frames), want, atol=1e-10)


def test_delay_and_auto_yoy(oracle, x):
    u = oracle.Unit("delay"); u.set_params([37, -8, -2])
    K = int(np.floor(0.037 * SR + 0.5))
    np.testing.assert_allclose(stream(u, x, 800), np.clip(db(-2) * (x + db(-8) * hist(x, np.arange(len(x)) - K)), -1, 1), atol=TOL)
    u = oracle.Unit("auto_yoy"); u.set_params([1, -35, -5, 70])
    with np.errstate(divide="ignore"):
        level = 20 * np.log10(envelope(x, 1, SR))
    delta = np.where(level <= -35, 0.0, np.where(level >= -5, 0.7, 0.0 + (0.7 / 30.0) * (level + 35)))
    want = 0.5 * x + 0.5 * frac(x, np.arange(len(x)), 0.01 * delta * SR)
    np.testing.assert_allclose(stream(u, x, 800), want, atol=1e-10)


def test_reverb(oracle):
    sr = 22050                                                                 # short loops: several wrap-arounds
    xs = synth_signal(2, 30000, sr)
    u = oracle.Unit("reverb"); u.set_params([40])
    i = np.arange(len(xs))
    dl = sum(c * hist(xs, i - int(round(t * sr))) for t, c in zip((0.19196, 0.19996, 0.21596, 0.23204), (0.1855, 0.18325, 0.17875, 0.17425)))
    w = dl
    for t in (0.04204, 0.01348, 0.00452):
        M = int(round(t * sr)) - 1                                             # ring of D slots delays by D - 1
        b = np.zeros(M + 1); b[0], b[M] = 0.7, 1.0
        a = np.zeros(M + 1); a[0], a[M] = 1.0, 0.7
        w = lfilter(b, a, w)
    want = np.clip(0.6 * xs + 0.5 * 0.4 * (dl + w), -1, 1)
    np.testing.assert_allclose(stream(u, xs, 4096, sr), want, atol=1e-10)


# ---- counters, oscillators, FSMs ----------------------------------------------------------------------------------------------
def test_tremolo_ringmod_siggen(oracle, x):
    u = oracle.Unit("tremolo"); u.set_params([100, 30, -12])
    period = int(SR / 10.0); on = int(SR / 10.0 * 0.3); off = period - on
    att, cnt, want = False, 0, np.empty_like(x)
    for i, v in enumerate(x):
        if att and cnt >= off: att, cnt = False, 0
        elif (not att) and cnt >= on: att, cnt = True, 0
        want[i] = v * db(-12) if att else v
        cnt += 1
    np.testing.assert_allclose(stream(u, x, 777), want, atol=TOL)
    u = oracle.Unit("ring_modulator"); u.set_params([30])
    np.testing.assert_allclose(stream(u, x, 1000), x * np.sin(2 * np.pi * 30 * np.arange(len(x)) / SR), atol=1e-9)
    u = oracle.Unit("signal_generator"); u.set_params([50, -6, 3, 441, 80, -3])
    ph = np.fmod(2 * np.pi * 441 * np.arange(len(x)) / SR, 2 * np.pi)
    saw = ph / np.pi - np.where(ph > np.pi, 2.0, 0.0)
    np.testing.assert_allclose(stream(u, x, 1000), 0.5 * db(-6) * x + 0.8 * db(-3) * saw, atol=1e-9)


def test_noise_gate(oracle, x):
    u = oracle.Unit("noise_gate"); u.set_params([-6, -12, 2])
    xs = x * np.concatenate([np.ones(2000), 0.1 * np.ones(2000), np.ones(2000)])
    hold = int(np.floor(0.002 * SR + 0.5))
    gate, since, want = False, 0, np.empty_like(xs)
    for i, v in enumerate(xs):
        a = abs(v)
        gate = gate or a > db(-6)
        if a > db(-12): since = 0
        if since >= hold: gate = False
        want[i] = v if gate else 0.0
        since += 1
    np.testing.assert_allclose(stream(u, xs, 640), want, atol=0)
    assert 0 < np.count_nonzero(want) < len(want)


def test_octaver(oracle, x):
    u = oracle.Unit("octaver"); u.set_params([1, -6, -12, -20, -3, -9, -15])
    a = np.exp(-20.0 / SR)
    e = envelope(x, 1, SR)
    pp, reg, c, want = 0.0, 0, 0.0, np.empty_like(x)
    for i, v in enumerate(x):
        s = np.sign(v)
        if s != 0 and s != pp and abs(v) > e[i] * db(-15):
            reg, pp = (reg + 1) & 7, s
        d1, d2 = (-1.0 if reg & 2 else 1.0), (-1.0 if reg & 4 else 1.0)
        p = db(-12) * v + (db(-6) * v * v / e[i] if e[i] > 1e-4 else 0.0) + db(-20) * s * e[i] + db(-3) * d1 * e[i] + db(-9) * d2 * e[i]
        c += (p - c) * (1 - a)
        want[i] = min(1.0, max(-1.0, p - c))
    np.testing.assert_allclose(stream(u, x, 1000), want, atol=1e-10)


# ---- oversampling as a polyphase filter bank ------------------------------------------------------------------------
@pytest.mark.parametrize("f", [2, 4])
def test_oversampler_decimator_polyphase(oracle, f):
    xs = synth_signal(4, 4096, SR)
    osd = oracle.OversamplerDecimator(f)
    frames = 512
    up = np.concatenate([osd.oversample(xs[i:i + frames]) for i in range(0, len(xs), frames)])
    L3 = lambda t: np.where(t == 0, 1.0, 3 * np.sin(np.pi * t) * np.sin(np.pi * t / 3) / (np.pi * t) ** 2 * (np.abs(t) < 3))
    m = np.arange(len(up))
    i, r = m // f, m % f
    q = i - 4 + r / f
    want = sum(hist(xs, np.floor(q).astype(int) + j) * L3(q - (np.floor(q) + j)) for j in range(-2, 4))
    np.testing.assert_allclose(up, want, atol=1e-12)
    osd2 = oracle.OversamplerDecimator(f)
    w = np.tanh(3 * up)
    down = np.concatenate([osd2.decimate(w[i:i + f * frames]) for i in range(0, len(w), f * frames)])
    h = oracle.aa_taps(f)
    want_down = 0.9440608762859234 * np.clip(np.convolve(w, h)[:len(w)], -1, 1)[::f]
    np.testing.assert_allclose(down, want_down, atol=1e-12)


# ---- chain, spatializer, tuner ---------------------------------------------------------------------------------------------------------
def test_chain_semantics(oracle, x):
    ch = oracle.Chain()
    i0 = ch.append_unit("distortion", bypass=True, params=[0, 20, 0, 0])
    ch.append_unit("tone_stack")
    a = ch.process(x[:1000], SR)
    solo = oracle.Unit("tone_stack")
    np.testing.assert_allclose(a, solo.process(x[:1000], SR), atol=0)       # bypassed slot skipped
    ch.set_bypass(i0, False)
    b = ch.process(x[1000:2000], SR)
    d = oracle.Unit("distortion"); d.set_params([0, 20, 0, 0])
    np.testing.assert_allclose(b, solo.process(d.process(x[1000:2000], SR), SR), atol=0)
    out = ch.process(x[:10], SR, n_out=9)                                    # length mismatch: no-op
    assert np.all(out == 0.0)
    assert oracle.Chain().process(x[:10], SR).tolist() == x[:10].tolist()    # empty chain copies


def test_spatializer(oracle):
    sr, n, nch = 96000, 3000, 4
    xs = np.stack([synth_signal(c, n, sr) for c in range(nch)])
    sp = oracle.Spatializer(nch)
    pos = [(0.0, 0.0, 1.0), (90.0, 0.05, 0.8), (-40.0, 2.0, 0.5), (170.0, 7.0, 1.0)]
    L, R = np.zeros(n), np.zeros(n)
    for c, (az, dist, lvl) in enumerate(pos):
        sp.set_azimuth(c, az); sp.set_distance(c, dist); sp.set_level(c, lvl)
        xp, yp = dist * np.sin(np.radians(az)), dist * np.cos(np.radians(az))
        dl, dr = np.hypot(xp + 0.1075, yp), np.hypot(xp - 0.1075, yp)
        gl, gr = lvl * min(1.0, 1.0 / dl) if dl > 0 else lvl, lvl * min(1.0, 1.0 / dr) if dr > 0 else lvl
        tau = (6.3e-4 / 0.215) * (dl - dr)
        D = min(abs(tau) * 96000.0, 1e9)
        H = int(np.ceil(sr * 6.3e-4))
        i = np.arange(n)
        fl, ce = min(int(np.floor(D)), H - 1), min(int(np.ceil(D)), H - 1)
        delayed = (1 - (D - np.floor(D))) * hist(xs[c], i - fl) + (1 - (np.ceil(D) - D)) * hist(xs[c], i - ce)
        if tau == 0: L += gl * xs[c]; R += gr * xs[c]
        elif tau > 0: L += gl * delayed; R += gr * xs[c]
        else: L += gl * xs[c]; R += gr * delayed
    gl_, gr_ = np.zeros(n), np.zeros(n)
    for b in range(0, n, 1000):
        l, r = sp.process(xs[:, b:b + 1000])
        gl_[b:b + 1000], gr_[b:b + 1000] = l, r
    np.testing.assert_allclose(gl_, L, atol=1e-12)
    np.testing.assert_allclose(gr_, R, atol=1e-12)


def test_tuner_autocorrelation_against_numpy(oracle):
    sr = 96000
    t = np.arange(96000 + 5000) / sr
    sig = 0.6 * np.sin(2 * np.pi * 110.0 * t) + 0.3 * np.sin(2 * np.pi * 220.0 * t + 0.4)
    tu = oracle.Tuner()
    for b in range(0, len(sig), 4096):
        tu.process(sig[b:b + 4096], sr)
    res = tu.analyze()
    win = sig[-96000:]
    r = np.fft.irfft(np.abs(np.fft.rfft(win, 262144)) ** 2, 262144)
    np.testing.assert_allclose(tu.correlation()[:5000], r[:5000], rtol=0, atol=1e-7 * r[0])
    lo, hi = int(sr / 1975.5332 + 0.5), int(sr / 61.7354 + 0.5)
    k = lo + int(np.argmax(r[lo:hi]))
    shift = np.clip(0.5 * (r[k + 1] - r[k - 1]) / (2 * r[k] - (r[k + 1] + r[k - 1])), -0.5, 0.5)
    assert res["note"] == "A2" and abs(res["cents"]) <= 5
    assert abs(res["frequency"] - sr / (k + shift)) / res["frequency"] < 1e-9


# ---- metronome (metronome/metronome.go:63-131): no reference test exists; second formulation = beat arithmetic ------------------
def test_metronome_against_beat_arithmetic(oracle):
    rng = np.random.default_rng(12)
    tick, tock = rng.uniform(-1, 1, 700), rng.uniform(-1, 1, 300)
    for sr, bpm, beats, n in ((48000, 120, 4, 8192), (96000, 200, 3, 8192), (1000, 90, 1, 500), (44100, 60, 0, 10000)):
        m = oracle.Metronome()
        m.tick, m.tock = tick, tock
        m.s.sample_rate, m.s.bpm_speed, m.s.beats_per_period = sr, bpm, beats
        got = np.concatenate([m.process(n) for _ in range(12)])
        spb = (60 * sr) // bpm
        g = np.arange(got.size)
        sc, beat = g % spb, (g // spb) % max(beats, 1)
        want = np.where(beat == 0, np.where(sc < tick.size, tick[np.minimum(sc, tick.size - 1)], 0.0),
                        np.where(sc < tock.size, tock[np.minimum(sc, tock.size - 1)], 0.0))
        np.testing.assert_array_equal(got, want)
