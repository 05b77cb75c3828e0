"""Shared helpers of the test-suite: seeded synthetic signals/IRs and a
pair builder that sets up the SAME chain on the HIP context and on the oracle."""
import numpy as np

import __graft_entry__ as entry

TOL_RMS = 1e-9          # north_star: output matches the float64 reference within 1e-9 RMS


def lcg_floats(seed, n):
    """random/random.go LCG, vectorised: x0 = (64979 seed + 83) mod (2^31-1); x <- 16807 x mod (2^31-1)."""
    mod = (1 << 31) - 1
    x = (64979 * seed + 83) % mod
    out = np.empty(n)
    for i in range(n):
        x = (16807 * x) % mod
        out[i] = x / (mod - 1)
    return out


def synth_signal(channel, n, sample_rate, seed_base=1337):
    """x_c[n] = 0.5 sin(2 pi f_c n/sr) + 0.25 sin(2 pi 3 f_c n/sr) + 0.05 u_c[n], f_c = 82.4069 * 2^((c mod 48)/12)."""
    f = 82.4069 * 2.0 ** ((channel % 48) / 12.0)
    t = np.arange(n) / float(sample_rate)
    rng = np.random.default_rng(seed_base + channel)
    u = 1.0 - 2.0 * rng.random(n)
    return 0.5 * np.sin(2 * np.pi * f * t) + 0.25 * np.sin(2 * np.pi * 3 * f * t) + 0.05 * u


def synth_ir(n_taps, seed=4242):
    """h[k] = (1 - 2 r_k) exp(-6.9 k / L), scaled to unit energy (Normalize with 0 dB compensation)."""
    rng = np.random.default_rng(seed)
    k = np.arange(n_taps)
    h = (1.0 - 2.0 * rng.random(n_taps)) * np.exp(-6.9 * k / float(n_taps))
    return h / np.sqrt(np.sum(h * h))


def rms(a):
    return float(np.sqrt(np.mean(np.square(a)))) if a.size else 0.0


class ChainPair:
    """The same chain on the HIP context (channel c) and on the oracle."""

    def __init__(self, ctx, channel, oracle):
        self.ctx, self.channel, self.oracle = ctx, channel, oracle
        self.ref = oracle.Chain()
        self.handles = []

    def append(self, unit, params=None, fir=None, bypass=False):
        self.ref.append_unit(unit, bypass=bypass, params=params, fir=fir)
        h = self.ctx.append_unit(self.channel, unit, params=params, fir=fir, bypass=bypass)
        self.handles.append(h)
        return h


def run_pairs(ctx, pairs, x, frames, sample_rate):
    """Stream x [nch][n] through ctx and the oracle in blocks of `frames`; returns (got, want)."""
    nch, n = x.shape
    got, want = np.zeros_like(x), np.zeros_like(x)
    for b in range(0, n, frames):
        blk = x[:, b:b + frames]
        got[:, b:b + frames] = ctx.process(blk, sample_rate)
        for c, p in enumerate(pairs):
            want[c, b:b + frames] = p.ref.process(blk[c], sample_rate)
    return got, want


def package():
    return entry.load_package()
