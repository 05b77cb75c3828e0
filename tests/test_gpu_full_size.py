"""BASELINE.json's configurations at their FULL sizes on one GPU.  The oracle is too slow to follow 512 channels of
65536-tap convolution for long, so the full-size runs are checked through size-independent properties:
  * channel independence: channels that carry the same chain and the same input give bit-identical output whatever
    their index (first / last workgroup, different XCDs), and differ from their neighbours;
  * determinism: a second context fed the same stream reproduces the output bit for bit (checksum of checksums);
  * linearity of the convolution stage: FIR(a x + b y) = a FIR(x) + b FIR(y) to rounding;
  * a sample of channels is followed by the oracle at the 1e-9 RMS bar;
  * batch mode (windows of 8 frames, time-blocked convolution) gives the samples of the per-frame calls.
Configs 3 and 5 are small enough for the oracle to follow every channel.  Run with `pytest -m gpu`."""
import hashlib

import numpy as np
import pytest

from helpers import TOL_RMS, ChainPair, package, rms, synth_ir, synth_signal
from test_gpu_parity import full_chain

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    p = package()
    assert p.device_count() > 0
    return p


def checksum_of_checksums(y):
    rows = [hashlib.sha256(np.ascontiguousarray(r).tobytes()).digest() for r in y]
    return hashlib.sha256(b"".join(rows)).hexdigest()


def build_config4(pkg, oracle, nch, frames, taps, n_distinct, followed):
    """512-channel full chain with two 65536-tap IRs; channel c uses parameter/IR set c % n_distinct."""
    ctx = pkg.Context(nch, frames)
    cab = [synth_ir(taps, seed=4242 + i) for i in range(n_distinct)]
    rev = [synth_ir(taps, seed=5242 + i) for i in range(n_distinct)]
    pairs = {}
    for c in range(nch):
        k = c % n_distinct
        if c in followed:
            p = ChainPair(ctx, c, oracle)
            full_chain(p, 0, cab[k], rev[k])
            pairs[c] = p
        else:
            ctx.append_unit(c, "compressor", params=[1, 30, -20])
            ctx.append_unit(c, "overdrive", params=[0, 20, 100, 0, 1, 0])
            ctx.append_unit(c, "tone_stack")
            ctx.append_unit(c, "chorus")
            ctx.append_unit(c, "power_amp", fir=cab[k])
            ctx.append_unit(c, "power_amp", fir=rev[k])
            ctx.append_unit(c, "cabinet")
            ctx.append_unit(c, "reverb", params=[50])
    return ctx, pairs


def test_config4_512_channels_192k_two_64k_irs(pkg, oracle):
    nch, frames, sr, taps, blocks, n_distinct = 512, 8192, 192000, 65536, 3, 8
    followed = {0, 257, 511}
    ctx, pairs = build_config4(pkg, oracle, nch, frames, taps, n_distinct, followed)
    ctx2, _ = build_config4(pkg, oracle, nch, frames, taps, n_distinct, set())
    # input: channel c carries signal (c % 16): channels c and c + 16 k * ... with equal c % 16 AND equal c % 8 are twins
    sig = np.stack([synth_signal(s, frames * blocks, sr) for s in range(16)])
    x = sig[np.arange(nch) % 16]
    got = np.empty_like(x)
    sums = []
    for b in range(blocks):
        blk = np.ascontiguousarray(x[:, b * frames:(b + 1) * frames])
        got[:, b * frames:(b + 1) * frames] = ctx.process(blk, sr)
        sums.append(checksum_of_checksums(ctx2.process(blk, sr)))
    assert np.isfinite(got).all()
    # determinism across contexts
    for b in range(blocks):
        assert checksum_of_checksums(got[:, b * frames:(b + 1) * frames]) == sums[b]
    # channel independence: twins (same signal, same parameter set) are bit identical, neighbours are not
    for c in range(16, nch):
        np.testing.assert_array_equal(got[c], got[c % 16])
    assert not np.array_equal(got[0], got[1])
    # the oracle follows three channels (first, middle, last workgroup)
    for c, p in pairs.items():
        want = np.concatenate([p.ref.process(x[c, b * frames:(b + 1) * frames], sr) for b in range(blocks)])
        assert rms(got[c] - want) <= TOL_RMS, c
    ctx.close()
    ctx2.close()


def test_config4_time_blocked_windows_at_full_size(pkg, oracle):
    """Batch mode at config 4's full size: windows of 8 frames per call (time-blocked convolution, two free-running channel groups)
    against one frame per call on a second context -- the same bits -- and against the oracle on three channels."""
    nch, frames, sr, taps, blocks, n_distinct, W = 512, 8192, 192000, 65536, 16, 8, 8
    followed = {0, 300, 511}
    ctx, pairs = build_config4(pkg, oracle, nch, frames, taps, n_distinct, followed)
    ctx1, _ = build_config4(pkg, oracle, nch, frames, taps, n_distinct, set())
    ctx.set_window(W)
    sig = np.stack([synth_signal(s, frames * blocks, sr) for s in range(16)])
    x = sig[np.arange(nch) % 16]
    d_in, d_out = ctx.alloc(nch, blocks * frames), ctx.alloc(nch, blocks * frames)
    d_in.upload(x)
    for b in range(0, blocks, W):
        ctx.process_window_device(d_in.ptr + 8 * b * frames, d_out.ptr + 8 * b * frames, blocks * frames, W, sr)
    got = d_out.download()
    d1_in, d1_out = ctx1.alloc(nch, frames), ctx1.alloc(nch, frames)
    worst = 0.0
    for b in range(blocks):
        d1_in.upload(np.ascontiguousarray(x[:, b * frames:(b + 1) * frames]))
        ctx1.process_device(d1_in, d1_out, frames, sr)
        worst = max(worst, float(np.max(np.abs(d1_out.download() - got[:, b * frames:(b + 1) * frames]))))
    assert worst == 0.0, worst
    for c in range(16, nch):                                   # twins stay bit-identical in window mode too
        np.testing.assert_array_equal(got[c], got[c % 16])
    for c, p in pairs.items():
        want = np.concatenate([p.ref.process(x[c, b * frames:(b + 1) * frames], sr) for b in range(blocks)])
        assert rms(got[c] - want) <= TOL_RMS, c
    ctx.close()
    ctx1.close()


def build_private(pkg, oracle, nch, frames, taps, channel0, followed):
    """bench.py's headline context: every channel its OWN pair of 65536-tap IRs (d = 1: private spectra, 2.3 GiB of state at
    512 channels), seeded by GLOBAL channel number like bench.py (go-dsp-guitar_amd/synth.py)."""
    import importlib
    synth = importlib.import_module("go_dsp_guitar_amd.synth")
    ctx = pkg.Context(nch, frames)
    pairs = {}
    for c in range(nch):
        g = channel0 + c
        cab, rev = synth.synth_ir(taps, synth.ir_seed("cab", g)), synth.synth_ir(taps, synth.ir_seed("rev", g))
        if c in followed:
            p = ChainPair(ctx, c, oracle)
            full_chain(p, 0, cab, rev)
            pairs[c] = p
        else:
            for name, params, fir in (("compressor", [1, 30, -20], None), ("overdrive", [0, 20, 100, 0, 1, 0], None), ("tone_stack", None, None),
                                      ("chorus", None, None), ("power_amp", None, cab), ("power_amp", None, rev), ("cabinet", None, None),
                                      ("reverb", [50], None)):
                ctx.append_unit(c, name, params=params, fir=fir)
    return ctx, pairs


@pytest.mark.parametrize("nch,channel0,groups", [(512, 0, 2), (64, 448, 1)], ids=["headline_512ch_fused_private_spectra", "per_gpu_shard_64ch_split_mac"])
def test_config4_private_irs_device_calls_follow_the_oracle(pkg, oracle, nch, channel0, groups):
    """The EXACT launch shapes bench.py times, followed by the oracle at full size:
      * 512 channels, d = 1: `fir_inv_kernel<13, 1>` (multiply-accumulate fused into the inverse transform, private spectra through
        non-temporal loads), two free-running channel groups, device-resident per-frame calls -- the headline;
      * 64 channels (the last shard of the 512-channel job over 8 GPUs, BASELINE config 4's per-GPU shape): the split path,
        `fir_mac_kernel` (bin-tiled) + `fir_inv_kernel<13, 0>`.
    16 blocks, so the whole 8-partition delay line is live and wraps; a second context runs the same stream as ONE window of 16
    frames (`fir_mac_tb_kernel<16, 8>` on private spectra) and must give the same bits; the oracle follows the first, a middle and
    the last channel of the context over all 16 blocks."""
    import importlib
    synth = importlib.import_module("go_dsp_guitar_amd.synth")
    frames, sr, taps, blocks, W = 8192, 192000, 65536, 16, 16
    followed = {0, nch // 2 - 1, nch - 1}
    ctx, pairs = build_private(pkg, oracle, nch, frames, taps, channel0, followed)
    ctxw, _ = build_private(pkg, oracle, nch, frames, taps, channel0, set())
    ctx.set_overlap(groups)
    ctxw.set_window(W)
    x = synth.synth_rows(nch, frames * blocks, sr, channel0=channel0)
    d_in, d_out = ctx.alloc(nch, frames), ctx.alloc(nch, frames)
    got = np.empty_like(x)
    for b in range(blocks):
        d_in.upload(np.ascontiguousarray(x[:, b * frames:(b + 1) * frames]))
        ctx.process_device(d_in, d_out, frames, sr)
        got[:, b * frames:(b + 1) * frames] = d_out.download()
    assert np.isfinite(got).all()
    w_in, w_out = ctxw.alloc(nch, blocks * frames), ctxw.alloc(nch, blocks * frames)
    w_in.upload(x)
    ctxw.process_window_device(w_in.ptr, w_out.ptr, blocks * frames, W, sr)
    gotw = w_out.download()
    assert np.array_equal(got, gotw), float(np.max(np.abs(got - gotw)))
    # every channel has its own filters and its own noise: no two rows coincide
    assert len({hashlib.sha256(np.ascontiguousarray(r).tobytes()).digest() for r in got}) == nch
    for c, p in pairs.items():
        want = np.concatenate([p.ref.process(x[c, b * frames:(b + 1) * frames], sr) for b in range(blocks)])
        err = rms(got[c] - want)
        assert err <= TOL_RMS, (c, err)
    for b in (d_in, d_out, w_in, w_out):
        b.free()
    ctx.close()
    ctxw.close()


def test_small_context_right_after_a_large_one_is_released(pkg):
    """a 64-channel context created right after a ~15 GB 512-channel context was destroyed ran 26-36x slow in two bench
    runs (never reproduced since, 12 attempts in round 3).  gdg_ctx_destroy must leave the device quiet: the large context's state
    comes out of a few arena chunks (released in milliseconds), and a fresh small context steps at its normal rate at once -- no
    repetition of 30 steps may take more than 3x the median."""
    import time
    import importlib
    synth = importlib.import_module("go_dsp_guitar_amd.synth")
    frames, sr, taps = 8192, 192000, 65536
    big, _ = build_private(pkg, None, 512, frames, taps, 0, set())
    big.set_window(16)
    d_in, d_out = big.alloc(512, 16 * frames), big.alloc(512, 16 * frames)
    d_in.upload(np.tile(synth.synth_rows(512, frames, sr), (1, 16)))
    big.process_window_device(d_in.ptr, d_out.ptr, 16 * frames, 16, sr)
    big.synchronize()
    t0 = time.perf_counter()
    big.close()                                             # frees d_in / d_out too (the context owns what gdg_device_alloc handed out)
    t_close = time.perf_counter() - t0
    small, _ = build_private(pkg, None, 64, frames, taps, 0, set())
    s_in, s_out = small.alloc(64, frames), small.alloc(64, frames)
    s_in.upload(synth.synth_rows(64, frames, sr))
    for _ in range(3):
        small.process_device(s_in, s_out, frames, sr)
    small.synchronize()
    reps = []
    for _ in range(8):
        t0 = time.perf_counter()
        for _ in range(30):
            small.process_device(s_in, s_out, frames, sr)
        small.synchronize()
        reps.append((time.perf_counter() - t0) / 30)
    small.close()
    med = sorted(reps)[len(reps) // 2]
    assert t_close < 0.5, "destroying the large context took %.0f ms" % (t_close * 1e3)
    assert max(reps) <= 3.0 * med, ["%.0f us" % (t * 1e6) for t in reps]
    assert med < 1e-3, "64 channels: %.0f us per step" % (med * 1e6)


def test_config4_convolution_is_linear_at_full_size(pkg):
    nch, frames, sr, taps, blocks = 512, 8192, 192000, 65536, 10            # 10 blocks: the whole 8-partition delay line is live
    irs = [synth_ir(taps, seed=777 + i) * 0.05 for i in range(4)]           # small gain: the output clip stays inactive
    ctxs = []
    for _ in range(3):
        ctx = pkg.Context(nch, frames)
        for c in range(nch):
            ctx.append_unit(c, "power_amp", fir=irs[c % 4])
        ctxs.append(ctx)
    rng = np.random.default_rng(5)
    a, b = 0.75, -0.4
    worst = 0.0
    for _ in range(blocks):
        xa = rng.uniform(-0.5, 0.5, (nch, frames))
        xb = rng.uniform(-0.5, 0.5, (nch, frames))
        ya = ctxs[0].process(xa, sr)
        yb = ctxs[1].process(xb, sr)
        yc = ctxs[2].process(a * xa + b * xb, sr)
        assert np.abs(yc).max() < 1.0
        worst = max(worst, rms(yc - (a * ya + b * yb)) / max(rms(yc), 1e-30))
    assert worst <= 1e-13, worst
    for ctx in ctxs:
        ctx.close()


def test_config3_64_channels_96k_4x_oversampling_32k_ir(pkg, oracle):
    nch, frames, sr, taps, blocks = 64, 8192, 96000, 32768, 2
    ctx = pkg.Context(nch, frames)
    pairs = []
    for c in range(nch):
        p = ChainPair(ctx, c, oracle)
        full_chain(p, 2, synth_ir(taps, seed=4242 + c % 4))
        pairs.append(p)
    x = np.stack([synth_signal(c, frames * blocks, sr) for c in range(nch)])
    for b in range(blocks):
        blk = np.ascontiguousarray(x[:, b * frames:(b + 1) * frames])
        got = ctx.process(blk, sr)
        for c, p in enumerate(pairs):
            assert rms(got[c] - p.ref.process(blk[c], sr)) <= TOL_RMS, (b, c)
    ctx.close()


def test_config2_single_channel_8k_ir_1024_frame_buffers(pkg, oracle):
    frames, sr, taps, blocks = 1024, 48000, 8192, 24                          # 24 blocks: three full turns of the delay line
    ctx = pkg.Context(1, frames)
    p = ChainPair(ctx, 0, oracle)
    full_chain(p, 0, synth_ir(taps))
    x = synth_signal(0, frames * blocks, sr)[None, :]
    got = np.concatenate([ctx.process(x[:, b * frames:(b + 1) * frames], sr)[0] for b in range(blocks)])
    want = np.concatenate([p.ref.process(x[0, b * frames:(b + 1) * frames], sr) for b in range(blocks)])
    assert rms(got - want) <= TOL_RMS
    ctx.close()


def test_config5_256_tuners_and_spatializer_192k(pkg, oracle):
    nch, frames, sr = 256, 8192, 192000
    total = 96000 + 2 * frames
    notes = [73.4162, 110.0, 146.8324, 195.9978, 246.9417, 329.6276]
    t = np.arange(total) / float(sr)
    x = np.stack([sum(a * np.sin(2 * np.pi * notes[c % 6] * 2.0 ** (((c % 7) - 3) / 1200.0) * h * t + 0.1 * c)
                      for h, a in ((1, 0.5), (2, 0.25), (3, 0.12))) for c in range(nch)])
    ctx = pkg.Context(nch, frames)
    sp = oracle.Spatializer(nch)
    sp.set_sample_rate(sr)
    ctx.spatializer_set_sample_rate(sr)
    for c in range(nch):
        az, dist, lv = -90.0 + 180.0 * c / (nch - 1), 0.5 + 0.05 * (c % 40), 0.2 + 0.003 * c
        ctx.spatializer_set_position(c, az, dist, lv)
        sp.set_azimuth(c, az); sp.set_distance(c, dist); sp.set_level(c, lv)
    tuners = [oracle.Tuner() for _ in range(nch)]
    for b in range(0, total - frames + 1, frames):
        blk = np.ascontiguousarray(x[:, b:b + frames])
        ctx.tuner_enqueue(blk, sr)
        left, right = ctx.spatialize(blk)
        wl, wr = sp.process(blk)
        assert rms(left - wl) <= TOL_RMS and rms(right - wr) <= TOL_RMS
        for c in range(nch):
            tuners[c].process(blk[c], sr)
    got = ctx.tuner_analyze()
    for c in range(nch):
        want = tuners[c].analyze()
        assert got[c]["note_index"] == want["note_index"] and got[c]["cents"] == want["cents"], c
        assert abs(got[c]["frequency"] - want["frequency"]) <= 1e-9 * want["frequency"], c
    ctx.close()


@pytest.mark.parametrize("sr,taps,os_index,two_irs", [(192000, 65536, 0, True), (96000, 32768, 2, False)])
def test_long_stream_64_blocks_does_not_drift(pkg, oracle, sr, taps, os_index, two_irs):
    """the synthetic stream is 64 blocks of 8192 frames; RMS over the WHOLE stream stays under the bar (scan-carried
    states, LFO phases, ring positions and the partition delay line all wrap many times)."""
    frames, blocks, nch = 8192, 64, 2
    ctx = pkg.Context(nch, frames)
    pairs = []
    for c in range(nch):
        p = ChainPair(ctx, c, oracle)
        full_chain(p, os_index, synth_ir(taps, seed=4242 + c), synth_ir(taps, seed=4243 + c) if two_irs else None)
        pairs.append(p)
    x = np.stack([synth_signal(c, frames * blocks, sr) for c in range(nch)])
    got, want = np.empty_like(x), np.empty_like(x)
    for b in range(blocks):
        blk = np.ascontiguousarray(x[:, b * frames:(b + 1) * frames])
        got[:, b * frames:(b + 1) * frames] = ctx.process(blk, sr)
        for c, p in enumerate(pairs):
            want[c, b * frames:(b + 1) * frames] = p.ref.process(blk[c], sr)
    for c in range(nch):
        assert rms(got[c] - want[c]) <= TOL_RMS, (c, rms(got[c] - want[c]))
        last = slice((blocks - 1) * frames, blocks * frames)
        assert rms(got[c, last] - want[c, last]) <= TOL_RMS, ("last block", c)
    print("max abs error over %d samples: %.3e" % (x.size, np.abs(got - want).max()))
    ctx.close()
