import os, sys
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import numpy as np
import __graft_entry__ as entry
oracle = entry.load_oracle()
have_gpu = False
try:
    pkg = entry.load_package(); have_gpu = pkg.device_count() > 0
except Exception: pass
for seed, chan in ((1061, 1), (1147, 0)):
    rng = np.random.default_rng(9500 + seed)
    nch = int(rng.integers(1, 9))
    sr = int(rng.choice([22050, 44100, 48000, 96000, 192000]))
    frames = int(rng.choice([64, 1000, 4096, 8192]))
    total = int(rng.choice([3 * frames, 96000 + 2 * frames, 50000]))
    total = max(frames, (total // frames) * frames)
    t = np.arange(total) / float(sr)
    x = np.zeros((nch, total)); info = {}
    for c in range(nch):
        kind = rng.random()
        if kind < 0.15: continue
        f0 = float(np.exp(rng.uniform(np.log(62.0), np.log(1900.0))))
        amps = [1.0] + [float(rng.uniform(0.0, 0.6)) / h for h in range(2, 6)]
        tone = sum(a * np.sin(2 * np.pi * f0 * (h + 1) * t + rng.uniform(0, 2 * np.pi)) for h, a in enumerate(amps) if f0 * (h + 1) < 0.45 * sr)
        tone = 0.5 * tone / max(np.max(np.abs(tone)), 1e-9)
        x[c] = tone + float(rng.uniform(0.0, 0.03)) * rng.standard_normal(total)
        info[c] = (f0, amps)
    print("seed", seed, "nch", nch, "sr", sr, "frames", frames, "total", total, "chan", chan, "f0/amps", info.get(chan))
    ref = oracle.Tuner()
    for b in range(0, total, frames): ref.process(x[chan, b:b + frames], sr)
    print(" oracle:", ref.analyze())
    # the ring: last 96000 samples, zero padded at the front when shorter
    ring = np.zeros(96000); n = min(total, 96000); ring[96000 - n:] = x[chan, total - n:]
    X = np.fft.rfft(ring, 262144); r = np.fft.irfft(np.abs(X) ** 2)
    lo, hi = int(sr / 1975.5 + 0.5), int(sr / 61.7 + 0.5)
    k = lo + int(np.argmax(r[lo:hi]))
    print(" numpy autocorrelation: window [%d, %d), argmax lag %d -> %.3f Hz, r = %.9g" % (lo, hi, k, sr / k, r[k]))
    for f in (381.6, 486.5, 489.4):
        L = int(round(sr / f)); print("   around %.1f Hz: lags %d..%d r = %s" % (f, L - 2, L + 2, ["%.9g" % v for v in r[L - 2:L + 3]]))
    if have_gpu:
        ctx = pkg.Context(nch, frames)
        for b in range(0, total, frames): ctx.tuner_enqueue(x[:, b:b + frames], sr)
        print(" device:", ctx.tuner_analyze()[chan]); ctx.close()
