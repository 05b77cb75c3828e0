#!/usr/bin/env python3
"""Third formulation of the units the reference holds no test vectors for: the first samples of every unit from ZERO state, written
out from the MATH of SURVEY.md Appendix B (not from oracle/*.c and not from the kernels) in plain Python floats, and frozen as
literals in units_handcalc.json.  Each case cites the reference lines whose behaviour it pins.

    python tests/golden/make_units_handcalc.py        # rewrites tests/golden/units_handcalc.json

The arithmetic follows the reference's operation order (IEEE-754 binary64, no fused multiply-add), so the oracle agrees with these
literals to the last bits; tests/test_units_handcalc.py allows 1e-12.  Parameter lists are in the table order of each create*()
function (numeric value, or index of the discrete value)."""
import json
import math
import os

X = [0.5, -0.25, 0.8, -0.6]
SR = 48000


def dB(v):
    return math.pow(10.0, 0.05 * v)


def clip(v):
    return -1.0 if v < -1.0 else (1.0 if v > 1.0 else v)


cases = []


def case(unit, params, x, sr, y, source, note, fir=None):
    c = {"unit": unit, "params": params, "sample_rate": sr, "x": list(x), "y": [float(v) for v in y], "source": source, "note": note}
    if fir is not None:
        c["fir"] = list(fir)
    cases.append(c)


# ---- memoryless -------------------------------------------------------------------------------------------------------------------
g = dB(0 + 20)
case("overdrive", [0, 20, 100, 0, 1, 0], X, SR, [1.0 * ((1.0 * ((2.0 / (1.0 + math.exp(-(g * x)))) - 1.0)) + (0.0 * x)) for x in X],
     "effects/overdrive.go:57-76", "ECC83: level (drive (2 / (1 + exp(-g x)) - 1) + (1 - drive) x), g = 10")
d = 0.01 * 70
case("overdrive", [0, 20, 70, -6, 0, 0], X, SR,
     [dB(-6) * ((d * ((2.0 / math.pi) * math.atan((0.25 * math.pi) * (g * x)))) + ((1.0 - d) * x)) for x in X],
     "effects/overdrive.go:57-76", "ECC82: (2 / pi) atan((pi / 4) g x), drive 70 %, level -6 dB")
case("distortion", [0, 20, -3, 0], X, SR, [dB(-3) * clip(g * x) for x in X], "effects/distortion.go:34-47", "level clip(g x)")


def excess(x, g, l):
    p = g * x
    if abs(p) <= 1.0:
        return l * p
    m = math.fmod(abs(p) + 1.0, 2.0)
    section = int(0.5 * math.floor(abs(p) + 1.0))
    inverted = (section % 2 != 0) != (p < 0.0)
    return l * ((1.0 - m) if inverted else (m - 1.0))


case("excess", [20, -3, 0], [0.05, -0.25, 0.33, -0.61], SR, [excess(x, dB(20), dB(-3)) for x in [0.05, -0.25, 0.33, -0.61]],
     "effects/excess.go:33-64", "triangle fold of g x: 0.5 stays linear, -2.5 / 3.3 / -6.1 fold through 1, 1 and 3 sections")

# ---- followers --------------------------------------------------------------------------------------------------------------------
a = math.exp(-20.0 / SR)
ainv = 1.0 - a


def follower(kind, xs):
    e, out = 0.0, []
    for x in xs:
        if kind == "level":
            e = e + ((abs(x) - e) * ainv)
        else:
            e = e * a
            if abs(x) > e:
                e = abs(x)
        out.append(e)
    return out


for follow, name in ((1, "level"), (0, "envelope")):
    es = follower(name, X)
    y = []
    for x, e in zip(X, es):
        gain = dB(-20) / e
        if gain > dB(30):
            gain = dB(30)
        y.append(clip(gain * x))
    case("compressor", [follow, 30, -20], X, SR, y, "effects/compressor.go:37-82",
         "follow = %s, e from 0; gain = min(dB(target) / e, dB(limit)); the level follower's first values are ~2e-4, so the limit (31.6) applies" % name)

es = follower("level", X)
c, y = 0.0, []
for x, e in zip(X, es):
    p = clip(1.0 * (x - (0.5 * e)))
    p = (1.0 * p) + (0.0 * x)
    c = c + ((p - c) * ainv)
    p = p - c
    y.append(1.0 * clip(p))
case("fuzz", [1, 50, 0, 0, 100, 0, 0], X, SR, y, "effects/fuzz.go:47-106", "level follower, bias 50 %, coupling capacitor uses the UPDATED voltage")

# ---- one-pole networks ------------------------------------------------------------------------------------------------------------
F = [20.0, 300.0, 3000.0, 6000.0, 20000.0]
levels = [0, -2, -5, -5]
h, l, y = [0.0] * 4, [0.0] * 4, []
m2pi = -(2.0 * math.pi) / SR
for x in X:
    s = 0.0
    for j in range(4):
        aH = 1.0 - math.exp(m2pi * F[j])
        aL = 1.0 - math.exp(m2pi * F[j + 1])
        dd = x - h[j]
        h[j] += dd * aH
        dd -= l[j]
        pre = l[j]
        l[j] += dd * aL
        s += dB(levels[j]) * pre
    y.append(clip(s))
case("tone_stack", [0, -2, -5, -5], X, SR, y, "effects/tonestack.go:63-99", "band output = the low-pass voltage BEFORE its update: the first output sample is 0")

v = list(X)
for f in (300.0, 120.0, 80.0):
    aa, hh = 1.0 - math.exp(m2pi * f), 0.0
    for i in range(len(v)):
        dd = v[i] - hh
        v[i] = dd
        hh += dd * aa
for f in (3000.0, 4000.0, 5000.0, 6000.0):
    aa, ll = 1.0 - math.exp(m2pi * f), 0.0
    for i in range(len(v)):
        dd = v[i] - ll
        v[i] = ll
        ll += dd * aa
case("cabinet", [0], X, SR, [clip(t) for t in v], "effects/cabinet.go:100-160",
     "three high-passes (x - h before the update), four low-passes emitting the OLD voltage: four samples of latency, so 0, 0, 0, 0")
X8 = X + [0.3, 0.7, -0.9, 0.1]
v = list(X8)
for f in (300.0, 120.0, 80.0):
    aa, hh = 1.0 - math.exp(m2pi * f), 0.0
    for i in range(len(v)):
        dd = v[i] - hh
        v[i] = dd
        hh += dd * aa
for f in (3000.0, 4000.0, 5000.0, 6000.0):
    aa, ll = 1.0 - math.exp(m2pi * f), 0.0
    for i in range(len(v)):
        dd = v[i] - ll
        v[i] = ll
        ll += dd * aa
case("cabinet", [0], X8, SR, [clip(t) for t in v], "effects/cabinet.go:100-160", "eight samples: the first non-zero output is sample 4")

aH, aL = 1.0 - math.exp(m2pi * 300.0), 1.0 - math.exp(m2pi * 3000.0)
hh = ll = 0.0
y = []
for x in X:
    dd = x - hh
    hh += dd * aH
    dd -= ll
    iv = ll
    ll += dd * aL
    y.append(clip(iv))
case("bandpass", [0, 300, 3000], X, SR, y, "effects/bandpass.go:60-96", "order 2 = one stage: high-pass into low-pass, old low-pass voltage out, clip")

# auto-wah: (level, freq) pairs (-40, 300) and (-10, 6000); alpha = 1 - exp(-fc / sr) -- no 2 pi
es = follower("level", X)
hs, ls, y = [0.0] * 8, [0.0] * 8, []
for x, e in zip(X, es):
    L = 20.0 * math.log10(e)
    if L <= -40.0:
        fc = 300.0
    elif L >= -10.0:
        fc = 6000.0
    else:
        fc = 300.0 + (((6000.0 - 300.0) / (-10.0 - -40.0)) * (L - -40.0))
    alpha = 1.0 - math.exp(-fc / SR)
    vv = x
    for j in range(8):
        dd = vv - hs[j]
        hs[j] += dd * alpha
        vv = ls[j]
        dd -= vv
        vv += dd * alpha
        ls[j] = vv
    y.append(clip(256.0 * vv))
case("auto_wah", [1, -40, -10, 300, 6000], X, SR, y, "effects/autowah.go:60-128",
     "eight stages, each passing its UPDATED low-pass voltage on; the follower is below -40 dB here, so fc = 300 Hz")

# auto-yoy: below level_1 the depth is 0 -> delay 0 samples -> floor == ceil -> BOTH weights 1 -> the sample counts twice
case("auto_yoy", [1, -40, -10, 100], X, SR, [(0.5 * x) + (0.5 * ((1.0 * x) + (1.0 * x))) for x in X], "effects/autoyoy.go:86-140",
     "quirk: integral delay (here 0) => weights 1 and 1 => y = 0.5 x + 0.5 (x + x) = 1.5 x")

# octaver
es = follower("level", X)
reg, prev, cc, y = 0, 0, 0.0, []
f_up = f_clean = f_dist = f_d1 = f_d2 = f_hyst = dB(-20)
for x, e in zip(X, es):
    s = -1 if x < 0.0 else (1 if x > 0.0 else 0)
    if s != 0 and s != prev and abs(x) > e * f_hyst:
        reg = (reg + 1) & 7
        prev = s
    d1 = -1.0 if (reg & 2) else 1.0
    d2 = -1.0 if (reg & 4) else 1.0
    p = f_clean * x
    if e > 0.0001:
        p += f_up * ((x * x) / e)
    p += f_dist * (float(s) * e)
    p += f_d1 * (d1 * e)
    p += f_d2 * (d2 * e)
    cc = cc + ((p - cc) * ainv)
    y.append(clip(p - cc))
case("octaver", [1, -20, -20, -20, -20, -20, -20], X, SR, y, "effects/octaver.go:60-137",
     "polarity register counts every sign change above the hysteresis: 1, 2, 3, 4 -> first-down flips at 2, second-down at 4")

# noise gate: open 0.1, close 0.01, hold 2400 samples
XG = [0.05, 0.5, -0.005, 0.2]
op, since, y = False, 0, []
for x in XG:
    if abs(x) > dB(-20):
        op = True
    if abs(x) > dB(-40):
        since = 0
    if since >= 2400:
        op = False
    y.append((1.0 if op else 0.0) * x)
    since += 1
case("noise_gate", [-20, -40, 50], XG, SR, y, "effects/noisegate.go:50-94", "0.05 does not open the gate (-> 0), 0.5 does, -0.005 passes while the hold runs")

# ---- delay-type units, zero history --------------------------------------------------------------------------------------------------
case("chorus", [100, 30], X, SR, [(0.5 * x) + (0.5 * 0.0) for x in X], "effects/chorus.go:86-112", "five delays of 30 .. 50 ms into an empty history: 0.5 x")
dd_ = 0.01 * 1
om = (0.02 * math.pi) * 10
y = []
for i, x in enumerate(X):
    ph = math.fmod(0.0 + (om * (float(i) * (1.0 / SR))), 2.0 * math.pi)
    D = (0.001 * (dd_ + (dd_ * math.sin(ph)))) * SR
    e_, l_ = math.floor(D), math.ceil(D)
    se = X[i - int(e_)] if i - int(e_) >= 0 else 0.0
    sl = X[i - int(l_)] if i - int(l_) >= 0 else 0.0
    fr = ((1.0 - (D - e_)) * se) + ((1.0 - (l_ - D)) * sl)
    y.append((0.5 * x) + (0.5 * fr))
case("flanger", [1, 10], X, SR, y, "effects/flanger.go:63-99", "depth 1 %: delay 0.48 (1 + sin phi) samples, linear interpolation between x[i] and x[i - 1]")
pp = 0.5 * math.sin((math.pi / 180.0) * 45.0)
y = []
for i, x in enumerate(X):
    ph = math.fmod(0.0 + (om * (float(i) * (1.0 / SR))), 2.0 * math.pi)
    D = (0.001 * (dd_ + (dd_ * math.sin(ph)))) * SR
    e_, l_ = math.floor(D), math.ceil(D)
    se = X[i - int(e_)] if i - int(e_) >= 0 else 0.0
    sl = X[i - int(l_)] if i - int(l_) >= 0 else 0.0
    fr = ((1.0 - (D - e_)) * se) + ((1.0 - (l_ - D)) * sl)
    y.append(((1.0 - abs(pp)) * x) + (pp * fr))
case("phaser", [1, 10, 45], X, SR, y, "effects/phaser.go:63-105", "as the flanger with mix (1 - |p|) x + p delayed, p = 0.5 sin 45 deg")

# tremolo at sr = 48: period = uint32(48 / 10) = 4, on = uint32(4.8 * 0.5) = 2, off = 2
att, cnt, y = False, 0, []
for x in X:
    if att and cnt >= 2:
        att, cnt = False, 0
    elif (not att) and cnt >= 2:
        att, cnt = True, 0
    y.append(x * dB(-10) if att else x)
    cnt += 1
case("tremolo", [100, 50, -10], X, 48, y, "effects/tremolo.go:33-63", "sample rate 48: two samples unattenuated, then two at -10 dB")

dl = (2.0 * math.pi) * 100.0 / SR
case("ring_modulator", [100], X, SR, [math.sin(math.fmod(0.0 + (float(i) * dl), 2.0 * math.pi)) * x for i, x in enumerate(X)],
     "effects/ringmodulator.go:30-44", "x sin(i 2 pi 100 / sr), phase 0")
# delay at sr = 2000: K = floor(0.001 * 1 * 2000 + 0.5) = 2
case("delay", [1, -6, -3], X, 2000, [clip(dB(-3) * (x + (dB(-6) * (X[i - 2] if i >= 2 else 0.0)))) for i, x in enumerate(X)],
     "effects/delay.go:45-87", "feed-FORWARD echo of the input two samples back, not a recirculating one")
case("reverb", [30], X, SR, [clip(((1.0 - 0.3) * x) + ((0.5 * 0.3) * (0.0 + 0.0))) for x in X], "effects/reverb.go:300-336",
     "taps >= 0.19 s back and empty all-pass rings: only the dry path, (1 - mix) x")
dl = (2.0 * math.pi) * (440.0 / SR)
case("signal_generator", [100, 0, 0, 440, 100, 0], X, SR,
     [((0.01 * 100.0) * dB(0)) * x + ((0.01 * 100.0) * dB(0)) * math.sin(math.fmod(0.0 + (float(i) * dl), 2.0 * math.pi)) for i, x in enumerate(X)],
     "effects/signalgenerator.go:60-150", "sine: input + sin(i 2 pi 440 / sr)")
dlt = (2.0 * math.pi) * (12000.0 / SR)
y = []
for i, x in enumerate(X):
    ph = math.fmod(0.0 + (float(i) * dlt), 2.0 * math.pi)
    tri = ((2.0 / math.pi) * ph) - 1.0 if ph < math.pi else 3.0 - ((2.0 / math.pi) * ph)
    y.append((((0.01 * 50.0) * dB(-6)) * x) + (((0.01 * 80.0) * dB(-3)) * tri))
case("signal_generator", [50, -6, 1, 12000, 80, -3], X, SR, y, "effects/signalgenerator.go:60-150", "triangle at sr / 4: -1, 0, 1, 0")

# ---- FIR ---------------------------------------------------------------------------------------------------------------------------
taps = [0.5, -0.25, 0.125]
XF = [0.5, -0.25, 0.8, -0.6, 3.0, 3.0, 0.0, 0.0]
y = []
for n in range(len(XF)):
    acc = 0.0
    for k, hk in enumerate(taps):
        if n - k >= 0:
            acc += hk * XF[n - k]
    y.append(clip(acc))
case("power_amp", [14], XF, SR, y, "filter/filter.go:342-515, effects/poweramp.go:186-216", "y[n] = clip(sum h[k] x[n - k]); sample 4 = 1.5 - ... clips at 1", fir=taps)

# ---- spatializer: one source at 30 degrees, 2 m, level 0.8 -----------------------------------------------------------------------------
az = (math.pi / 180.0) * 30.0
xp, yp = 2.0 * math.sin(az), 2.0 * math.cos(az)
dL = math.sqrt((abs(xp + (0.5 * 0.215)) ** 2) + (abs(yp) ** 2))
dR = math.sqrt((abs(xp - (0.5 * 0.215)) ** 2) + (abs(yp) ** 2))
gL, gR = 0.8 * min(1.0, 1.0 / dL), 0.8 * min(1.0, 1.0 / dR)
tau = (6.3e-4 / 0.215) * (dL - dR)
assert tau > 0.0 and abs(tau) * 96000.0 > 4.0         # the left ear is ~30 samples late: silent for the first four samples
spat = {"azimuth": 30.0, "distance": 2.0, "level": 0.8, "sample_rate": SR, "x": X, "left": [gL * 0.0 for _ in X], "right": [gR * x for x in X],
        "source": "spatializer/spatializer.go:170-298", "note": "source to the right: the LEFT ear is delayed ~30 samples (empty history), the right ear gets level min(1, 1 / dR) x"}

out = {"_comment": "generated by tests/golden/make_units_handcalc.py from the math of SURVEY.md Appendix B; zero initial state; see that script",
       "units": cases, "spatializer": spat}
path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "units_handcalc.json")
with open(path, "w") as f:
    json.dump(out, f, indent=1)
print("wrote %d unit cases to %s" % (len(cases), path))
