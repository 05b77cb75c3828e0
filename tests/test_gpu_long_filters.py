"""Long filters and small hops (round-3 review, item 6).  `filter_order` goes up to 1 048 576 taps (effects/poweramp.go:303-329 -- it is the
DEFAULT), i.e. K = 128 partitions at the batch block size; at 64-sample hops a 9600-tap filter has K = 150.  Until round 4 the largest K any
GPU test ran was 47.  Each case: against the oracle (filter.Process, filter/filter.go:342-515) AND against a direct formulation
(scipy.signal.fftconvolve + the output clip), per-frame calls, windows of 16 frames where they are legal (8192-sample frames), and across a
frame-size change in mid-stream (the delay line is re-partitioned)."""
import numpy as np
import pytest
from scipy.signal import fftconvolve

from helpers import TOL_RMS, package, rms, synth_ir, synth_signal

pytestmark = pytest.mark.gpu

B = 8192


def direct(x, h):
    """y[n] = clip(sum_k h[k] x[n - k]) (filter.go:487-493: the output is clipped, the tail is not)"""
    return np.clip(fftconvolve(x, h)[:x.size], -1.0, 1.0)


@pytest.mark.parametrize("taps,blocks", [(1048576, 4), (262144, 5), (131073, 4)])
def test_long_filters_per_frame_and_in_windows(oracle, taps, blocks):
    """K = 128 / 32 / 17 partitions of 8192 taps: fused multiply-accumulate + inverse (per frame) and the time-blocked kernel in chunks of 8
    partitions (windows) -- same bits from both, oracle and direct convolution within 1e-9 RMS."""
    pkg = package()
    sr, nch = 96000, 2
    h = [synth_ir(taps, seed=300 + c) * (1.0 if c == 0 else 12.0) for c in range(nch)]         # channel 1 is driven into the output clip
    W = 16 if blocks >= 16 else 4
    n = blocks * B
    x = np.stack([0.7 * synth_signal(c, n, sr) for c in range(nch)])
    ctx = pkg.Context(nch, B)
    refs = []
    for c in range(nch):
        ctx.append_unit(c, "power_amp", fir=h[c])
        r = oracle.Chain()
        r.append_unit("power_amp", fir=h[c])
        refs.append(r)
    got = np.zeros_like(x)
    for b in range(blocks):
        got[:, b * B:(b + 1) * B] = ctx.process(np.ascontiguousarray(x[:, b * B:(b + 1) * B]), sr)
    ctx.close()
    for c in range(nch):
        want = np.concatenate([refs[c].process(x[c, b * B:(b + 1) * B], sr) for b in range(blocks)])
        assert rms(got[c] - want) <= TOL_RMS, ("oracle", c, rms(got[c] - want))
        assert rms(got[c] - direct(x[c], h[c])) <= TOL_RMS, ("direct", c)
    assert np.max(np.abs(got[1])) == min(1.0, float(np.max(np.abs(fftconvolve(x[1], h[1])[:n]))))       # clips exactly where the convolution exceeds full scale
    # the same stream in windows: one full window of W frames, then the tail frame by frame -- bit-identical to the per-frame calls
    ctx = pkg.Context(nch, B)
    for c in range(nch):
        ctx.append_unit(c, "power_amp", fir=h[c])
    ctx.set_window(W)
    d_in, d_out = ctx.alloc(nch, n), ctx.alloc(nch, n)
    d_in.upload(x)
    done = 0
    while done < blocks:
        w = W
        while w > blocks - done:
            w //= 2
        ctx.process_window_device(d_in.ptr + 8 * done * B, d_out.ptr + 8 * done * B, n, w, sr)
        done += w
    win = d_out.download()
    ctx.close()
    for c in range(nch):
        assert np.array_equal(win[c], got[c]), (c, float(np.max(np.abs(win[c] - got[c]))))


def test_one_million_taps_through_a_window_of_16(oracle):
    """The reference's default filter order at the batch run's window size: K = 128, ring of 143 slots, 16 frames per call."""
    pkg = package()
    sr, taps, blocks, W = 192000, 1048576, 18, 16
    h = synth_ir(taps, seed=411)
    x = 0.7 * synth_signal(3, blocks * B, sr)[None, :]
    ctx = pkg.Context(1, B)
    ctx.append_unit(0, "power_amp", fir=h)
    ctx.set_window(W)
    d_in, d_out = ctx.alloc(1, blocks * B), ctx.alloc(1, blocks * B)
    d_in.upload(x)
    ctx.process_window_device(d_in.ptr, d_out.ptr, blocks * B, W, sr)
    ctx.process_window_device(d_in.ptr + 8 * W * B, d_out.ptr + 8 * W * B, blocks * B, 2, sr)
    got = d_out.download()[0]
    ctx.close()
    assert rms(got - direct(x[0], h)) <= TOL_RMS
    ref = oracle.Chain()
    ref.append_unit("power_amp", fir=h)
    want = np.concatenate([ref.process(x[0, b * B:(b + 1) * B], sr) for b in range(3)])        # the oracle's 2^21-point transforms: three blocks
    assert rms(got[:3 * B] - want) <= TOL_RMS


@pytest.mark.parametrize("hop,taps", [(64, 9600), (128, 9600), (64, 4097)])
def test_many_partitions_at_small_hops(oracle, hop, taps):
    """K = 150 / 75 / 65 partitions of `hop` taps (the live path's buffer sizes with a 0.1 s IR at 96 kHz)."""
    pkg = package()
    sr, nch, blocks = 96000, 2, 200
    h = [synth_ir(taps, seed=500 + c) * (1.0 + c) for c in range(nch)]
    x = np.stack([0.7 * synth_signal(c, blocks * hop, sr) for c in range(nch)])
    ctx = pkg.Context(nch, hop)
    refs = []
    for c in range(nch):
        ctx.append_unit(c, "power_amp", fir=h[c])
        r = oracle.Chain()
        r.append_unit("power_amp", fir=h[c])
        refs.append(r)
    got = np.zeros_like(x)
    want = np.zeros_like(x)
    for b in range(blocks):
        blk = np.ascontiguousarray(x[:, b * hop:(b + 1) * hop])
        got[:, b * hop:(b + 1) * hop] = ctx.process(blk, sr)
        for c in range(nch):
            want[c, b * hop:(b + 1) * hop] = refs[c].process(blk[c], sr)
    ctx.close()
    for c in range(nch):
        assert rms(got[c] - want[c]) <= TOL_RMS, ("oracle", c, rms(got[c] - want[c]))
        assert rms(got[c] - direct(x[c], h[c])) <= TOL_RMS, ("direct", c)


@pytest.mark.parametrize("taps,sizes", [(262144, [8192, 8192, 1024, 1024, 1024, 8192, 8192, 8192]),
                                        (9600, [64] * 40 + [128] * 30 + [8192, 8192] + [64] * 20),
                                        (1048576, [8192, 4096, 4096, 8192])])
def test_long_filters_across_frame_size_changes(oracle, taps, sizes):
    """filter.Process's state depends on L only (filter.go:370-428): a stream cut into frames of changing size is ONE convolution.  K goes
    32 -> 256 -> 32 (262144 taps), 150 -> 75 -> 2 -> 150 (9600 taps), 128 -> 256 -> 128 (1048576 taps) and the delay line follows."""
    pkg = package()
    sr = 96000
    h = synth_ir(taps, seed=611) * 2.0
    x = 0.6 * synth_signal(2, sum(sizes), sr)
    ctx = pkg.Context(1, 8192)
    ctx.append_unit(0, "power_amp", fir=h)
    ref = oracle.Chain()
    ref.append_unit("power_amp", fir=h)
    got, want = np.zeros_like(x), np.zeros_like(x)
    at = 0
    for n in sizes:
        blk = np.ascontiguousarray(x[None, at:at + n])
        got[at:at + n] = ctx.process(blk, sr)[0]
        want[at:at + n] = ref.process(blk[0], sr)
        at += n
    ctx.close()
    assert rms(got - want) <= TOL_RMS, rms(got - want)
    assert rms(got - direct(x, h)) <= TOL_RMS


def test_window_change_in_mid_stream_with_a_long_filter(oracle):
    """gdg_ctx_set_window on a LIVE 262144-tap power amp: the 32-slot delay line moves into the ring of 47 (W = 16) and back."""
    pkg = package()
    sr, taps, blocks = 96000, 262144, 2 + 16 + 2
    h = synth_ir(taps, seed=733)
    x = 0.7 * synth_signal(1, blocks * B, sr)[None, :]
    ctx = pkg.Context(1, B)
    ctx.append_unit(0, "power_amp", fir=h)
    d_in, d_out = ctx.alloc(1, blocks * B), ctx.alloc(1, blocks * B)
    d_in.upload(x)
    for b in range(2):
        ctx.process_window_device(d_in.ptr + 8 * b * B, d_out.ptr + 8 * b * B, blocks * B, 1, sr)
    ctx.set_window(16)
    ctx.process_window_device(d_in.ptr + 8 * 2 * B, d_out.ptr + 8 * 2 * B, blocks * B, 16, sr)
    ctx.set_window(1)
    for b in range(18, 20):
        ctx.process_window_device(d_in.ptr + 8 * b * B, d_out.ptr + 8 * b * B, blocks * B, 1, sr)
    got = d_out.download()[0]
    ctx.close()
    assert rms(got - direct(x[0], h)) <= TOL_RMS
