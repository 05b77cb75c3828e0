"""Time blocking (gdg_ctx_set_window / gdg_process_window_device): W consecutive 8192-sample frames per channel and call, every power
amp reading its IR spectra and its delay line once for all W frames.  The sums keep the order of the per-frame kernels and, like
them, are formed without fused multiply-adds: the output must be BIT-IDENTICAL to W calls of gdg_process_device; against the oracle
the usual 1e-9 RMS applies."""
import numpy as np
import pytest

from helpers import TOL_RMS, package, rms, synth_ir, synth_signal

pytestmark = pytest.mark.gpu

B = 8192
RATE = 96000
PRE = [("compressor", [1, 30, -20]), ("overdrive", [0, 15, 80, -3, 1, 0]), ("chorus", None)]
POST = [("cabinet", None), ("reverb", [30])]
# channel: taps of the first amp, taps of the second amp (None: one amp only; K = 1, 3, 8 and 9 partitions, the last = two chunks at W = 8)
TAPS = [(3000, None), (20000, 5000), (65536, 65536), (70000, 100), (0, None)]




def make_ctx(pkg, irs):
    ctx = pkg.Context(len(TAPS), B)
    for c, (t1, t2) in enumerate(TAPS):
        for name, p in PRE:
            ctx.append_unit(c, name, params=p)
        ctx.append_unit(c, "power_amp", fir=irs[c][0])
        if t2 is not None:
            ctx.append_unit(c, "tone_stack")
            ctx.append_unit(c, "power_amp", fir=irs[c][1])
        for name, p in POST:
            ctx.append_unit(c, name, params=p)
    return ctx


def signals(blocks):
    return np.stack([0.6 * synth_signal(c, blocks * B, RATE) for c in range(len(TAPS))])


def impulse_responses():
    return [(synth_ir(t1, seed=70 + c) if t1 else np.zeros(0), synth_ir(t2, seed=90 + c) if t2 else None) for c, (t1, t2) in enumerate(TAPS)]


def per_frame(pkg, irs, x):
    nch, n = x.shape
    ctx = make_ctx(pkg, irs)
    d_in, d_out = ctx.alloc(nch, B), ctx.alloc(nch, B)
    out = np.zeros_like(x)
    for b in range(n // B):
        d_in.upload(np.ascontiguousarray(x[:, b * B:(b + 1) * B]))
        ctx.process_device(d_in, d_out, B, RATE)
        out[:, b * B:(b + 1) * B] = d_out.download()
    ctx.close()
    return out


@pytest.mark.parametrize("W", [2, 4, 8, 16])
def test_window_equals_single_frames(W):
    pkg = package()
    irs = impulse_responses()
    blocks = 2 * W + (W - 1)                      # two full windows, then the tail in windows of W/2, W/4, .., 1
    x = signals(blocks)
    want = per_frame(pkg, irs, x)
    nch, n = x.shape
    ctx = make_ctx(pkg, irs)
    ctx.set_window(W)
    d_in, d_out = ctx.alloc(nch, n), ctx.alloc(nch, n)         # "whole files" in HBM: the rows are the windows' row stride
    d_in.upload(x)
    done = 0
    while done < blocks:
        w = W
        while w > blocks - done:
            w //= 2
        ctx.process_window_device(d_in.ptr + 8 * done * B, d_out.ptr + 8 * done * B, n, w, RATE)
        done += w
    got = d_out.download()
    ctx.close()
    for c in range(nch):
        assert np.array_equal(got[c], want[c]), "channel %d: max diff %.3e" % (c, np.max(np.abs(got[c] - want[c])))


def test_window_against_the_oracle(oracle):
    pkg = package()
    irs = impulse_responses()
    blocks = 8
    x = signals(blocks)
    nch, n = x.shape
    ctx = make_ctx(pkg, irs)
    ctx.set_window(4)
    d_in, d_out = ctx.alloc(nch, n), ctx.alloc(nch, n)
    d_in.upload(x)
    for w0 in range(0, blocks, 4):
        ctx.process_window_device(d_in.ptr + 8 * w0 * B, d_out.ptr + 8 * w0 * B, n, 4, RATE)
    got = d_out.download()
    ctx.close()
    for c, (t1, t2) in enumerate(TAPS):
        ch = oracle.Chain()
        for name, p in PRE:
            ch.append_unit(name, params=p)
        ch.append_unit("power_amp", fir=irs[c][0])
        if t2 is not None:
            ch.append_unit("tone_stack")
            ch.append_unit("power_amp", fir=irs[c][1])
        for name, p in POST:
            ch.append_unit(name, params=p)
        want = np.concatenate([ch.process(x[c, b * B:(b + 1) * B], RATE) for b in range(blocks)])
        assert rms(got[c] - want) <= TOL_RMS, "channel %d: RMS %.3e" % (c, rms(got[c] - want))


def test_window_size_changes_mid_stream_keep_the_convolution_state():
    """Single frames, then windows of 4, then of 8, then single frames again on a context that is back at W = 1: the delay lines move
    into the larger (and back into the smaller) ring; the re-partitioning is exact to ~1e-16, not bit-exact."""
    pkg = package()
    irs = impulse_responses()
    plan = [(1, 1), (1, 1), (1, 1), (4, 4), (4, 4), (8, 8), (1, 1), (1, 1)]          # (window set on the context, frames in the call)
    blocks = sum(w for _, w in plan)
    x = signals(blocks)
    want = per_frame(pkg, irs, x)
    nch, n = x.shape
    ctx = make_ctx(pkg, irs)
    d_in, d_out = ctx.alloc(nch, n), ctx.alloc(nch, n)
    d_in.upload(x)
    done = 0
    for W, w in plan:
        ctx.set_window(W)
        ctx.process_window_device(d_in.ptr + 8 * done * B, d_out.ptr + 8 * done * B, n, w, RATE)
        done += w
    got = d_out.download()
    ctx.close()
    for c in range(nch):
        assert rms(got[c] - want[c]) <= 1e-13, "channel %d: RMS %.3e" % (c, rms(got[c] - want[c]))


def test_window_rejections():
    pkg = package()
    ctx = pkg.Context(2, B)
    d_in, d_out = ctx.alloc(2, 8 * B), ctx.alloc(2, 8 * B)
    with pytest.raises(pkg.GdgError, match="window of 3 frames: 1, 2, 4, 8 or 16"):
        ctx.set_window(3)
    with pytest.raises(pkg.GdgError, match="window of 2 frames, the context is set up for 1"):
        ctx.process_window_device(d_in, d_out, 8 * B, 2, RATE)
    ctx.set_window(8)
    with pytest.raises(pkg.GdgError, match="row stride 8192 is shorter than the window"):
        ctx.process_window_device(d_in, d_out, B, 4, RATE)
    ctx.process_window_device(d_in, d_out, 8 * B, 8, RATE)              # empty chains: a copy
    ctx.close()
    small = pkg.Context(2, 1024)
    with pytest.raises(pkg.GdgError, match="windows are made of 8192-sample frames"):
        small.set_window(2)
    small.close()


def test_window_with_shared_ir_spectra():
    """Channels with identical IRs share one copy of the spectra (cacheable loads in the multiply-accumulate): same result."""
    pkg = package()
    ir = synth_ir(30000, seed=11)
    blocks, nch = 8, 4
    x = np.stack([0.6 * synth_signal(c, blocks * B, RATE) for c in range(nch)])
    outs = []
    for W in (1, 8):
        ctx = pkg.Context(nch, B)
        ctx.share_ir_spectra(True)
        for c in range(nch):
            ctx.append_unit(c, "power_amp", fir=ir)
            ctx.append_unit(c, "cabinet")
        ctx.set_window(W)
        d_in, d_out = ctx.alloc(nch, blocks * B), ctx.alloc(nch, blocks * B)
        d_in.upload(x)
        for b in range(0, blocks, W):
            ctx.process_window_device(d_in.ptr + 8 * b * B, d_out.ptr + 8 * b * B, blocks * B, W, RATE)
        outs.append(d_out.download())
        ctx.close()
    assert np.array_equal(outs[0], outs[1])


ALL_UNITS = [
    ("noise_gate", [-30, -20, 50]), ("compressor", [1, 12, -6]), ("overdrive", [0, 20, 100, 0, 1, 2]), ("distortion", [0, 20, -3, 1]),
    ("excess", [12, -6, 2]), ("fuzz", [1, 50, 0, 20, 100, 0, 1]), ("fuzz", [0, 30, 10, 10, 70, -6, 2]), ("octaver", [0, -6, -12, -3, 0, -6, -30]),
    ("tone_stack", [-10, 0, -3, -20]), ("auto_wah", [0, -5, -45, 200, 9000]), ("auto_yoy", [0, -10, -50, 40]), ("bandpass", [3, 5000, 100]),
    ("chorus", [35, 77]), ("flanger", [40, 100]), ("phaser", [70, 33, -60]), ("delay", [3, -10, 0]), ("ring_modulator", [7]),
    ("tremolo", [10, 0, -60]), ("signal_generator", [50, -6, 1, 1000, 80, -3]), ("cabinet", None), ("reverb", [100]),
]


def test_every_unit_type_in_windows():
    """All 21 unit types (oversampled shapers, FSM scans, delay rings, the noise generator's jump-ahead) frame after frame inside windows:
    the segments' state runs through the W launches of a window exactly as through W calls -- the same bits (no convolution in between
    whose contraction could differ)."""
    pkg = package()
    blocks, W = 8, 4
    third = len(ALL_UNITS) // 3
    chains = [ALL_UNITS[:third], ALL_UNITS[third:2 * third], ALL_UNITS[2 * third:]]
    x = np.stack([0.7 * synth_signal(20 + c, blocks * B, RATE) for c in range(len(chains))])

    def make():
        ctx = pkg.Context(len(chains), B)
        for c, chain in enumerate(chains):
            for name, p in chain:
                ctx.append_unit(c, name, params=p)
        return ctx
    ctx = make()
    d_in, d_out = ctx.alloc(len(chains), B), ctx.alloc(len(chains), B)
    want = np.zeros_like(x)
    for b in range(blocks):
        d_in.upload(np.ascontiguousarray(x[:, b * B:(b + 1) * B]))
        ctx.process_device(d_in, d_out, B, RATE)
        want[:, b * B:(b + 1) * B] = d_out.download()
    ctx.close()
    ctx = make()
    ctx.set_window(W)
    d_in, d_out = ctx.alloc(len(chains), blocks * B), ctx.alloc(len(chains), blocks * B)
    d_in.upload(x)
    for b in range(0, blocks, W):
        ctx.process_window_device(d_in.ptr + 8 * b * B, d_out.ptr + 8 * b * B, blocks * B, W, RATE)
    got = d_out.download()
    ctx.close()
    for c in range(len(chains)):
        assert np.array_equal(got[c], want[c]), "chain %d: max diff %.3e" % (c, np.max(np.abs(got[c] - want[c])))


@pytest.mark.parametrize("W", [2, 4])
def test_window_chain_of_adjacent_power_amps_gives_the_same_bits(W):
    """Time blocking with two power amps in a row, a chip's worth of channels.  W = 2: one workgroup per channel walks the window, the
    inverse transform of amp 1 running into the forward transform of amp 2 (fir_inv_fwd_chain_chan_kernel).  W = 4: two launches, the inverse
    through one LDS buffer with two workgroups per CU (fir_inv13h_kernel), then the forward walk.  GDG_FIR_CHAIN=0 drops the
    chain hint from the plan; per-frame calls are the third way.  All three: identical samples."""
    import os
    pkg = package()
    nch, frames, sr, taps, blocks = 256, 8192, 96000, 20000, 8
    x = np.stack([synth_signal(c % 7, frames * blocks, sr) * (0.5 + 0.001 * c) for c in range(nch)])
    irs = [[synth_ir(taps, seed=70 + 2 * k + j) * 0.9 for j in range(2)] for k in range(3)]
    outs = {}
    for mode in ("window_chain", "window_separate", "per_frame"):
        os.environ["GDG_FIR_CHAIN"] = "0" if mode == "window_separate" else "1"
        try:
            ctx = pkg.Context(nch, frames)
        finally:
            del os.environ["GDG_FIR_CHAIN"]
        for c in range(nch):
            ctx.append_unit(c, "tone_stack")
            ctx.append_unit(c, "power_amp", fir=irs[c % 3][0])
            ctx.append_unit(c, "power_amp", fir=irs[c % 3][1])
            ctx.append_unit(c, "cabinet")
        d_in, d_out = ctx.alloc(nch, blocks * frames), ctx.alloc(nch, blocks * frames)
        d_in.upload(x)
        if mode == "per_frame":
            for b in range(blocks):
                ctx.process_window_device(d_in.ptr + 8 * b * frames, d_out.ptr + 8 * b * frames, blocks * frames, 1, sr)
        else:
            ctx.set_window(W)
            for b in range(0, blocks, W):
                ctx.process_window_device(d_in.ptr + 8 * b * frames, d_out.ptr + 8 * b * frames, blocks * frames, W, sr)
        outs[mode] = d_out.download()
        ctx.close()
    assert np.array_equal(outs["window_chain"], outs["window_separate"])
    assert np.array_equal(outs["window_chain"], outs["per_frame"])
    assert np.abs(outs["per_frame"]).max() > 0.01


@pytest.mark.parametrize("bits", [0, 47])
def test_transform_variants_give_the_same_bits(bits):
    """GDG_FFT_HALF_LDS picks the kernels of the 8192-point transforms (read once per process: a child each).  0: two LDS buffers, one
    workgroup per CU, chained inverse -> forward kernel in every window; 47: one buffer everywhere it exists, the real-time path's forward
    transform included.  Each must pass the bit-for-bit tests of this file that the default (14) passes in the parent."""
    import os
    import subprocess
    import sys
    env = dict(os.environ, GDG_FFT_HALF_LDS=str(bits))
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.abspath(__file__), "-q", "-x", "-m", "gpu", "-p", "no:cacheprovider",
                        "-k", "equals_single_frames or chain_of_adjacent or shared_ir"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-500:]
