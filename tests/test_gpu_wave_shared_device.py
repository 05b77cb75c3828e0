"""WAVE launches (windows of few channels: a workgroup per frame and channel, the frames of a channel meeting through counters in HBM, seg.hip)
when the device is SHARED and when a hand-off never comes (round-5 review, item 5).

(i)  two contexts on ONE device running W = 16 WAVE windows concurrently from two host threads -- what gdg_batch_run_shard does on a 4-GPU box
     with 8 shards -- must give the bits of the same contexts run one after the other: the tickets and counters belong to a context, the
     workgroups of two launches share CUs, and a frame's wait must never be satisfied (or starved) by the other context's traffic.
(ii) the bounded wait: option debug_stall_unit withholds one hand-off; with wave_spin_limit_ms lowered the launch gives up after that time,
     gdg_ctx_synchronize returns GDG_ERR_HIP with a message, the context is usable again after gdg_unit_reset, and a second context that ran
     on the device meanwhile is not disturbed.  The reference behaviour being protected: Chain.Process never hangs (signal/signal.go:361-414)."""
import threading
import time

import numpy as np
import pytest

import __graft_entry__ as entry
from helpers import synth_ir, synth_signal

pytestmark = pytest.mark.gpu

FRAMES, W = 8192, 16
CHAIN = [("compressor", [1, 30, -20]), ("overdrive", [0, 20, 100, 0, 1, 0]), ("tone_stack", None), ("chorus", None),
         ("power_amp", "ir"), ("cabinet", None), ("reverb", [50])]


@pytest.fixture(scope="module")
def pkg():
    return entry.load_package()


def build(pkg, nch, seed):
    ctx = pkg.Context(nch, FRAMES)
    for c in range(nch):
        for name, p in CHAIN:
            if p == "ir":
                ctx.append_unit(c, name, fir=synth_ir(20000, seed=seed + c))
            else:
                ctx.append_unit(c, name, params=p)
    ctx.set_window(W)
    return ctx


def run_windows(ctx, x, sr, windows, out, barrier=None):
    nch = x.shape[0]
    d_in, d_out = ctx.alloc(nch, W * FRAMES), ctx.alloc(nch, W * FRAMES)
    if barrier is not None:
        barrier.wait()
    for w in range(windows):
        d_in.upload(x[:, w * W * FRAMES:(w + 1) * W * FRAMES])
        ctx.process_window_device(d_in.ptr, d_out.ptr, W * FRAMES, W, sr)
        out[:, w * W * FRAMES:(w + 1) * W * FRAMES] = d_out.download()
    ctx.synchronize()
    d_in.free()
    d_out.free()


def test_two_contexts_share_the_device_while_both_run_wave_windows(pkg):
    sr, windows = 192000, 6
    shapes = [(24, 1), (40, 2)]                     # two shards of different size: their launches interleave unevenly
    xs = [np.stack([synth_signal(c + 10 * s, FRAMES * W * windows, sr) for c in range(n)]) for n, s in shapes]
    serial, together = [], []
    for (n, s), x in zip(shapes, xs):
        ctx = build(pkg, n, 100 * s)
        out = np.zeros_like(x)
        run_windows(ctx, x, sr, windows, out)
        serial.append(out)
        ctx.close()
    ctxs = [build(pkg, n, 100 * s) for n, s in shapes]
    outs = [np.zeros_like(x) for x in xs]
    barrier = threading.Barrier(2)
    errors = []

    def worker(i):
        try:
            run_windows(ctxs[i], xs[i], sr, windows, outs[i], barrier)
        except Exception as e:          # noqa: BLE001
            errors.append((i, repr(e)))

    threads = [threading.Thread(target=worker, args=(i,)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(timeout=300)
    assert not errors, errors
    for ctx in ctxs:
        ctx.close()
    for i in range(2):
        np.testing.assert_array_equal(outs[i], serial[i])
        assert np.isfinite(outs[i]).all() and np.abs(outs[i]).max() > 0.01


def test_a_hand_off_that_never_comes_ends_as_an_error_and_the_context_stays_usable(pkg):
    sr, nch = 96000, 8
    x = np.stack([synth_signal(c + 3, FRAMES * W * 3, sr) for c in range(nch)])
    # what an undisturbed context gives for the first window (the stream a reset context starts over with)
    ref = build(pkg, nch, 7)
    want = np.zeros_like(x[:, :W * FRAMES])
    run_windows(ref, x[:, :W * FRAMES], sr, 1, want)
    ref.close()

    ctx = build(pkg, nch, 7)
    other = build(pkg, 16, 50)                       # a bystander on the same device
    xo = np.stack([synth_signal(c + 40, FRAMES * W * 4, sr) for c in range(16)])
    want_other = np.zeros_like(xo)
    run_windows(other, xo, sr, 4, want_other)
    other.close()
    other = build(pkg, 16, 50)
    got_other = np.zeros_like(xo)
    bystander = threading.Thread(target=run_windows, args=(other, xo, sr, 4, got_other))

    ctx.set_option("wave_spin_limit_ms", 200)
    assert ctx.get_option("wave_spin_limit_ms") == 200
    tone_stack = ctx._chains[3][2][0]                # channel 3's tone stack: frame 1 of that channel waits for a counter that stays away
    ctx.set_option("debug_stall_unit", tone_stack)
    d_in, d_out = ctx.alloc(nch, W * FRAMES), ctx.alloc(nch, W * FRAMES)
    d_in.upload(x[:, :W * FRAMES])
    t0 = time.perf_counter()
    bystander.start()
    ctx.process_window_device(d_in.ptr, d_out.ptr, W * FRAMES, W, sr)          # the call itself is asynchronous and returns
    with pytest.raises(pkg.GdgError) as e:
        ctx.synchronize()
    dt = time.perf_counter() - t0
    assert e.value.code == pkg.GDG_ERR_HIP and "timed out" in str(e.value)
    assert 0.15 <= dt < 20.0, dt                     # the limit, not a hang (and not the old iteration count's several seconds per frame)
    bystander.join(timeout=300)
    np.testing.assert_array_equal(got_other, want_other)                        # nobody else on the device noticed
    other.close()

    # the units' state is undefined after the error: reset them; the context itself is usable again
    ctx.set_option("debug_stall_unit", -1)
    ctx.set_option("wave_spin_limit_ms", 1000)
    for c in range(nch):
        for h, _ in ctx._chains[c]:
            ctx.unit_reset(h)
    got = np.zeros_like(want)
    d_in.upload(x[:, :W * FRAMES])
    ctx.process_window_device(d_in.ptr, d_out.ptr, W * FRAMES, W, sr)
    ctx.synchronize()
    got[:] = d_out.download()
    np.testing.assert_array_equal(got, want)
    ctx.close()
