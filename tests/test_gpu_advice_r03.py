"""Round-3 advisor findings, each pinned by a test on the HIP path."""
import ctypes as C

import numpy as np
import pytest

from helpers import TOL_RMS, ChainPair, package, rms, run_pairs, synth_signal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def pkg():
    return package()


@pytest.mark.parametrize("nch", [1024, 1025, 2049, 2600])
def test_spatializer_runs_any_channel_count(pkg, oracle, nch):
    """The one-launch spatializer keeps 64 groups of 16 channels' partial sums in LDS; wider shards go through them in rounds (round 3 rejected
    more than 2048 channels with a bare `invalid argument`).  The running sums carry over the rounds: same order of additions."""
    frames, sr = 256, 96000
    rng = np.random.default_rng(nch)
    ctx = pkg.Context(nch, frames)
    ref = oracle.Spatializer(nch)
    for c in range(nch):
        a, d, l = float(rng.uniform(-180, 180)), float(rng.uniform(0, 10)), float(rng.uniform(0, 1))
        ctx.spatializer_set_position(c, a, d, l)
        ref.set_azimuth(c, a); ref.set_distance(c, d); ref.set_level(c, l)
    for b in range(2):
        x = rng.uniform(-0.5, 0.5, (nch, frames))
        gl, gr = ctx.spatialize(x)
        wl, wr = ref.process(x)
        assert rms(gl - wl) <= TOL_RMS and rms(gr - wr) <= TOL_RMS, b
        assert np.max(np.abs(gl - wl)) <= 1e-12 * nch
    ctx.close()


def test_spatializer_rate_beyond_the_history_limit_is_refused_with_a_message(pkg):
    ctx = pkg.Context(2, 64)
    with pytest.raises(pkg.GdgError, match="at most 1024"):
        ctx.spatializer_set_sample_rate(2_000_000)
    ctx.close()


def test_disabled_meter_ports_do_not_read_their_rows(pkg, oracle):
    """gdg_meter_process_device takes caller-supplied rows: the rows of DISABLED ports may be unbacked.  Here ports 1 and 3 are disabled and the
    row pointer / stride are chosen so that their rows lie outside every allocation of the test (a read there would fault or, at best, change
    nothing) -- the enabled ports' readings must match the oracle and the disabled ports stay at their initial state."""
    sr, frames, ports = 96000, 4096, 4
    ctx = pkg.Context(1, 8192)
    x = np.stack([synth_signal(p, frames, sr) * 0.3 for p in range(ports)])
    # rows 0 and 2 live in two separate device blocks a huge stride apart; rows 1 and 3 would fall between / behind them
    d0, d2 = ctx.alloc(1, frames), ctx.alloc(1, frames)
    d0.upload(x[0:1]); d2.upload(x[2:3])
    lo, hi = sorted([(d0.ptr, 0), (d2.ptr, 2)])
    stride_bytes = hi[0] - lo[0]
    if stride_bytes % 16 != 0:
        pytest.skip("allocator did not give two blocks at a usable distance")
    order = [lo[1], hi[1]]                                   # which signal sits in port 0 / port 2 of the strided view
    ctx.meter_configure(ports)
    ctx.meter_set_enabled(True)
    ctx.meter_set_enabled(False, port=1)
    ctx.meter_set_enabled(False, port=3)
    half = stride_bytes // 2 // 8                             # doubles: port p's row starts at lo + p * half
    for _ in range(3):
        ctx._check(pkg.lib().gdg_meter_process_device(ctx._h, C.c_void_p(lo[0]), C.c_size_t(half), frames, sr))
    ctx.synchronize()
    lv, pk = ctx.meter_analyze()
    for port, sig in ((0, order[0]), (2, order[1])):
        r = oracle.ChannelMeter()
        r.set_enabled(True)
        for _ in range(3):
            r.process(x[sig], sr)
        assert (lv[port], pk[port]) == r.analyze(), port
    off = oracle.ChannelMeter()
    assert (lv[1], pk[1]) == off.analyze() and (lv[3], pk[3]) == off.analyze()
    ctx.close()


def test_shard_run_refuses_metronome_to_master(pkg):
    """A shard's master mix is a partial sum: the flag used to be dropped silently; now the call says what to do instead."""
    BLOCK = 8192
    ctx = pkg.Context(1, BLOCK)
    data = np.zeros(BLOCK, dtype=np.int16).view(np.uint8)                 # one block of 16-bit samples
    arr, keep = ctx._batch_inputs([(data, "lpcm16", 48000)])
    opt = pkg.BatchOptions(48000, pkg.WAVE_FORMATS["lpcm16"], 1, 0, 0)
    out = np.zeros(2 * BLOCK, dtype=np.uint8)
    left, right = np.zeros(BLOCK), np.zeros(BLOCK)
    ptrs = (C.c_void_p * 1)(out.ctypes.data)
    so = pkg.BatchShardOut(left.ctypes.data, right.ctypes.data, None, None, 0)
    rc = pkg.lib().gdg_batch_run_shard(ctx._h, arr, 1, C.byref(opt), ptrs, C.byref(so))
    assert rc == pkg.GDG_ERR_INVALID
    assert "gdg_batch_finish_master" in pkg.lib().gdg_last_error(ctx._h).decode()
    opt.metronome_to_master = 0
    assert pkg.lib().gdg_batch_run_shard(ctx._h, arr, 1, C.byref(opt), ptrs, C.byref(so)) == pkg.GDG_OK
    ctx.close()


def test_small_context_holds_a_small_arena(pkg, capfd, monkeypatch):
    """The arena's first chunk is 4 MiB (round 3: 64 MiB for every context, however small), and a chunk is zeroed once: unit state, history
    rings and delay lines that come out of never-used space need no fill of their own."""
    monkeypatch.setenv("GDG_ARENA_TRACE", "1")
    ctx = pkg.Context(1, 1024)
    taps = np.random.default_rng(1).standard_normal(300)
    for name in ("compressor", "chorus", "reverb", "cabinet"):
        ctx.append_unit(0, name)
    ctx.append_unit(0, "power_amp", fir=taps)
    x = np.random.default_rng(2).uniform(-0.5, 0.5, (1, 1024))
    ctx.process(x, 48000)
    ctx.close()
    err = capfd.readouterr().err
    line = [l for l in err.splitlines() if l.startswith("[arena]")][-1]
    mib = float(line.split("peak ")[1].split(" MiB")[0])
    issued = int(line.split("issued ")[1].split(",")[0])
    avoided = int(line.split("avoided ")[1])
    assert mib <= 8.0, line
    assert avoided >= 10 and issued <= 2, line


def test_malformed_pcie_weights_are_refused_as_a_whole(pkg):
    """GDG_PCIE_WEIGHTS is read once per process: checked in a child process each."""
    import os
    import subprocess
    import sys
    code = (
        "import numpy as np, sys\n"
        "sys.path.insert(0, %r); sys.path.insert(0, %r)\n"
        "from helpers import package\n"
        "pkg = package()\n"
        "ctx = pkg.Context(8, 256)\n"
        "x = np.random.default_rng(0).uniform(-0.5, 0.5, (8, 256))\n"
        "ctx.append_unit(0, 'cabinet')\n"
        "y = ctx.process(x, 48000)\n"
        "print('OK', float(np.abs(y[1:] - x[1:]).max()))\n"
    ) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    for env_value, complains in (("1,0,2", True), ("1,x", True), ("2,", True), ("1,3,3,1", False)):
        env = dict(os.environ, GDG_PCIE_WEIGHTS=env_value)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0 and r.stdout.startswith("OK 0.0"), (env_value, r.stderr[-500:])
        assert ("GDG_PCIE_WEIGHTS" in r.stderr) == complains, (env_value, r.stderr[-500:])


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_state_of_new_units_is_zero_after_arena_churn(pkg, oracle, seed):
    """Unit state, delay lines and history rings come out of the arena without a fill when the space was never handed out.  Blocks that
    straddle the never-used mark (a freed temporary that coalesced with the untouched tail) must move the mark: here units and filters of random
    sizes come and go while noise runs through them, then fresh units with long memories are created and must behave like the oracle's fresh
    ones."""
    frames, sr = 512, 48000
    rng = np.random.default_rng(seed)
    ctx = pkg.Context(2, frames)
    live = []                                                   # (handle, unit type, channel)
    stateful = ["chorus", "reverb", "delay", "flanger", "phaser", "compressor", "auto_wah", "tone_stack"]

    def clear():
        for c in range(2):
            ctx.chain_set(c, [], [])

    for it in range(24):
        kind = rng.integers(0, 3)
        clear()
        if kind == 0 or not live:
            name, ch = str(rng.choice(stateful + ["power_amp"])), int(rng.integers(0, 2))
            h = ctx.unit_create(ch, name)
            if name == "power_amp":
                ctx.unit_set_fir(h, rng.standard_normal(int(rng.integers(10, 6000))))
            live.append((h, name, ch))
        elif kind == 1:
            ctx.unit_destroy(live.pop(int(rng.integers(0, len(live))))[0])
        else:
            fir = [u for u in live if u[1] == "power_amp"]
            if fir:
                ctx.unit_set_fir(fir[int(rng.integers(0, len(fir)))][0], rng.standard_normal(int(rng.integers(10, 20000))))
        for c in range(2):
            hs = [u[0] for u in live if u[2] == c]
            ctx.chain_set(c, hs, [False] * len(hs))
        ctx.process(rng.uniform(-0.9, 0.9, (2, frames)), sr)
    for c in range(2):
        ctx.chain_set(c, [], [])
    for u in live:
        ctx.unit_destroy(u[0])
    pairs = [ChainPair(ctx, c, oracle) for c in range(2)]
    for p in pairs:
        for name in ("delay", "chorus", "reverb", "flanger", "compressor"):
            p.append(name)
        p.append("power_amp", fir=rng.standard_normal(3000) * 0.01)
    x = np.zeros((2, 8 * frames))
    x[:, 0] = 0.5                                               # an impulse: whatever sits in the delay lines comes out after it
    got, want = run_pairs(ctx, pairs, x, frames, sr)
    assert rms(got - want) <= TOL_RMS
    ctx.close()


def test_chunks_of_destroyed_long_filters_go_back_to_the_device(pkg, oracle, capfd, monkeypatch):
    """Entirely free chunks (but the first and one spare) are returned: a run of 1M-tap filters does not pin its memory for the life of the
    context.  The context keeps working afterwards, new units start from zeros."""
    monkeypatch.setenv("GDG_ARENA_TRACE", "1")
    frames, sr = 8192, 192000
    ctx = pkg.Context(2, frames)
    rng = np.random.default_rng(5)
    x = rng.uniform(-0.5, 0.5, (2, frames))
    hs = [ctx.append_unit(c, "power_amp", fir=rng.standard_normal(1 << 20) * 1e-3) for c in range(2)]
    ctx.process(x, sr)
    for c in range(2):
        ctx.chain_set(c, [], [])
    for h in hs:
        ctx.unit_destroy(h)
    pairs = [ChainPair(ctx, c, oracle) for c in range(2)]
    pairs[0].append("reverb")
    pairs[1].append("power_amp", fir=rng.standard_normal(500) * 0.05)
    got, want = run_pairs(ctx, pairs, np.concatenate([x, x], axis=1), frames, sr)
    assert rms(got - want) <= TOL_RMS
    ctx.close()
    line = [l for l in capfd.readouterr().err.splitlines() if l.startswith("[arena]")][-1]
    held = float(line.split("chunks, ")[1].split(" MiB")[0])
    peak = float(line.split("peak ")[1].split(" MiB")[0])
    back = int(line.split("MiB, ")[1].split(" chunks given back")[0])
    assert peak >= 64.0 and back >= 1 and held <= peak / 2, line
