"""Synthetic workload of SURVEY.md 8(d): inputs and impulse responses built on the reference's own pseudo random
numbers, so that every leg of bench.py and every full-size test -- HIP path, oracle, CPU baseline -- sees the same samples.

The generator is the linear congruency generator of random/random.go:23-55:
    x0 = (64979 * seed + 83) mod (2^31 - 1)    (uint64 wrap-around arithmetic, random.go:41-43)
    x <- 16807 * x mod (2^31 - 1);  NextFloat = x / (2^31 - 2)
vectorised by jump-ahead (x_n = x0 * 16807^n mod m; every product of two residues fits 64 bits).  Known answers:
random/random_test.go:53-70 (tests/golden/random.json, checked by tests/test_synth.py).
"""
import numpy as np

_M = (1 << 31) - 1
_A = 16807
_powers = np.ones(1, dtype=np.uint64)          # 16807^n mod m, n = 0 .. len - 1, grown by doubling


def _pow_table(n):
    global _powers
    while _powers.size < n + 1:
        step = np.uint64(pow(_A, int(_powers.size), _M))
        _powers = np.concatenate([_powers, (_powers * step) % np.uint64(_M)])
    return _powers


def lcg_floats(seed, n):
    """The first n values NextFloat() returns for CreatePRNG(seed)."""
    x0 = ((64979 * (int(seed) & 0xFFFFFFFFFFFFFFFF) + 83) & 0xFFFFFFFFFFFFFFFF) % _M
    x = (np.uint64(x0) * _pow_table(n)[1:n + 1]) % np.uint64(_M)
    return x.astype(np.float64) / float(_M - 1)


# seeds of SURVEY 8(d): inputs 1337 + channel; cabinet IR 4242, reverb IR 4243.  The metric's d = 1 wants every channel to own its
# spectra, so channel g's IRs are seeded 4242 + 2 g (cabinet, even) and 4243 + 2 g (reverb, odd): no two filters coincide and channel 0
# carries exactly the survey's pair (which is also what the CPU baseline runs).
INPUT_SEED = 1337
IR_SEED = {"cab": 4242, "rev": 4243}


def ir_seed(kind, index=0):
    return IR_SEED[kind] + 2 * int(index)


def synth_ir(n_taps, seed):
    """h[k] = (1 - 2 r_k) exp(-6.9 k / L), scaled to unit energy (= Normalize at 0 dB, filter/filter.go:328-336)."""
    k = np.arange(n_taps)
    h = (1.0 - 2.0 * lcg_floats(seed, n_taps)) * np.exp(-6.9 * k / float(n_taps))
    return h / np.sqrt(np.sum(h * h))


def synth_rows(n_channels, n_samples, sample_rate, channel0=0, start=0):
    """x_c[n] = 0.5 sin(2 pi f_c n / sr) + 0.25 sin(2 pi 3 f_c n / sr) + 0.05 (1 - 2 u_c[n]), f_c = 82.4069 * 2^((c mod 48) / 12),
    u_c = the LCG seeded 1337 + c; samples start .. start + n_samples - 1 of GLOBAL channels channel0 .. (a channel sounds the same
    on whichever GPU it lands)."""
    n = np.arange(start, start + n_samples)
    t = n / float(sample_rate)
    x = np.empty((n_channels, n_samples))
    for c in range(n_channels):
        g = channel0 + c
        f = 82.4069 * 2.0 ** ((g % 48) / 12.0)
        u = lcg_floats(INPUT_SEED + g, start + n_samples)[start:]
        x[c] = 0.5 * np.sin(2 * np.pi * f * t) + 0.25 * np.sin(2 * np.pi * 3 * f * t) + 0.05 * (1.0 - 2.0 * u)
    return x


def synth_block(n_channels, frames, sample_rate, channel0=0):
    return synth_rows(n_channels, frames, sample_rate, channel0=channel0)
