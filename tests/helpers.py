"""Shared helpers of the test-suite: seeded synthetic signals/IRs and a
pair builder that sets up the SAME chain on the HIP context and on the oracle."""
import numpy as np

import __graft_entry__ as entry

TOL_RMS = 1e-9          # north_star: output matches the float64 reference within 1e-9 RMS


def _synth():
    import importlib
    entry.load_package()
    return importlib.import_module("go_dsp_guitar_amd.synth")


def lcg_floats(seed, n):
    """random/random.go LCG (go-dsp-guitar_amd/synth.py; known answers: tests/test_synth.py)."""
    return _synth().lcg_floats(seed, n)


def synth_signal(channel, n, sample_rate, seed_base=1337):
    """x_c[n] = 0.5 sin(2 pi f_c n/sr) + 0.25 sin(2 pi 3 f_c n/sr) + 0.05 u_c[n], f_c = 82.4069 * 2^((c mod 48)/12), u_c = the
    reference LCG seeded seed_base + c."""
    f = 82.4069 * 2.0 ** ((channel % 48) / 12.0)
    t = np.arange(n) / float(sample_rate)
    u = 1.0 - 2.0 * lcg_floats(seed_base + channel, n)
    return 0.5 * np.sin(2 * np.pi * f * t) + 0.25 * np.sin(2 * np.pi * 3 * f * t) + 0.05 * u


def synth_ir(n_taps, seed=4242):
    """h[k] = (1 - 2 r_k) exp(-6.9 k / L), r = the reference LCG, scaled to unit energy (Normalize with 0 dB compensation)."""
    return _synth().synth_ir(n_taps, seed)


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
