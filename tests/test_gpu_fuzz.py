"""Randomised chains: units, order, parameters (anywhere inside the reference's ranges, tests/golden/params.json), bypass flags, frame size
and sample rate drawn from a seeded generator, streamed through the HIP path and the oracle.  The cases every other test was written for
are the ones somebody thought of; this one is for the others.  Tolerance: 1e-9 RMS per channel (north_star)."""
import json
import os

import numpy as np
import pytest

import __graft_entry__ as entry
from helpers import TOL_RMS, ChainPair, package, rms, synth_ir, synth_signal

pytestmark = pytest.mark.gpu

with open(os.path.join(entry.ROOT, "tests", "golden", "params.json")) as f:
    PARAMS = json.load(f)


def random_params(rng, unit_type, allow_oversampling):
    out = []
    for p in PARAMS[str(unit_type)]["params"]:
        if p["Type"] == "PARAMETER_TYPE_DISCRETE":
            n = len(p["DiscreteValues"])
            v = int(rng.integers(0, n))
            if p["Name"] == "oversampling" and not allow_oversampling:
                v = 0
            out.append(v)
        else:
            lo, hi = int(p["Minimum"]), int(p["Maximum"])
            # the extremes are drawn more often than a uniform pick would
            out.append(int(rng.choice([lo, hi, int(rng.integers(lo, hi + 1)), int(rng.integers(lo, hi + 1))])))
    return out


def reference_panics(frames, taps):
    """filter.Process walks the frame in blocks of nextpow2(L) samples but counts them on nextpow2(N): with N not a power of two a block can
    start beyond the frame and the reference panics on the slice bounds (filter/filter.go:370-382, :443-453); the library rejects such pairs"""
    if taps <= 0 or frames <= 0:
        return False
    n_power = 1 << (frames - 1).bit_length()
    block = 1 << (taps - 1).bit_length()
    blocks = -(-n_power // block)
    return (blocks - 1) * block > frames


def _split_points(units):
    """Indices of octavers that have another (non-bypassed) unit in front of them.  The octaver's polarity logic looks at the SIGN of its
    input and compares its magnitude with the follower's (octaver.go:86-100): a discontinuous function of the input.  Where that input is
    digital silence or sits on the hysteresis threshold, the 1e-18 by which two correct implementations of the stages in front differ
    decides the octave registers, and with them the sub-octaves' polarity for the rest of the stream.  First seen behind the stages the
    reference computes by FFT (decimators, power amps: rounding noise of arbitrary sign where the signal is silent); the soak of round 3
    (profiles/probes/fuzz_soak.py, seed 5419; taken apart by profiles/probes/chain_case.py) met it behind distortion -> band pass -> ring
    modulator -> noise gate as well: stages agreeing to 3e-18, octaver outputs apart by 2.5e-7, the octaver alone on the oracle's signal
    exact to 2e-17.  So every such octaver (and what follows it) is compared on the ORACLE's intermediate signal -- same input, same
    output -- and the stages in front of it against the oracle as usual."""
    cut, seen = [], False
    for i, (name, params, bypass) in enumerate(units):
        if bypass:
            continue
        if name == "octaver" and seen:
            cut.append(i)
        seen = True
    return cut


@pytest.mark.parametrize("seed", range(64))
def test_random_chains_follow_the_oracle(oracle, seed):
    pkg = package()
    rng = np.random.default_rng(1000 + seed)
    sr = int(rng.choice([22050, 44100, 48000, 96000, 192000]))
    frames = int(rng.choice([8192, 8192, 1024, 1000, 480, 4096]))
    blocks = 3 if frames >= 4096 else 5
    nch = 3
    chains = []
    for c in range(nch):
        units = []
        for _ in range(int(rng.integers(1, 8))):
            t = int(rng.integers(0, 21))
            name = pkg.UNIT_NAMES[t]
            bypass = bool(rng.random() < 0.15)
            if name == "power_amp":
                taps = int(rng.choice([1, 77, 500, 3000, 9000]))
                while reference_panics(frames, taps):                # keep away from the pairs the reference itself panics on
                    taps *= 2
                fir = synth_ir(taps, seed=int(rng.integers(1, 10 ** 6))) * float(rng.choice([0.5, 1.0, 2.5]))
                units.append((name, fir, bypass))
            else:
                units.append((name, random_params(rng, t, allow_oversampling=True), bypass))
        chains.append(units)
    x = np.stack([synth_signal(int(rng.integers(0, 48)), frames * blocks, sr) * float(rng.choice([0.05, 0.5, 1.0])) for _ in range(nch)])

    def run(sections, xin):
        """sections[c] = the units of channel c; returns (device output, oracle output) for the stream xin"""
        ctx = pkg.Context(nch, frames)
        pairs = []
        for c in range(nch):
            p = ChainPair(ctx, c, oracle)
            for name, arg, bypass in sections[c]:
                if name == "power_amp":
                    p.append(name, fir=arg, bypass=bypass)
                else:
                    p.append(name, params=arg, bypass=bypass)
            pairs.append(p)
        got, want = np.zeros_like(xin), np.zeros_like(xin)
        for b in range(blocks):
            sl = slice(b * frames, (b + 1) * frames)
            got[:, sl] = ctx.process(np.ascontiguousarray(xin[:, sl]), sr)
            for c, p in enumerate(pairs):
                want[c, sl] = p.ref.process(xin[c, sl], sr)
        ctx.close()
        return got, want

    def describe(units):
        return [(n, (len(a) if n == "power_amp" else a), b) for n, a, b in units]

    # every channel's chain in sections: [.. up to an ill-conditioned octaver) [octaver ..) ..; section k + 1 is fed the oracle's output of section k
    cuts = [_split_points([(n, (a if n != "power_amp" else [0]), b) for n, a, b in chains[c]]) for c in range(nch)]
    n_sections = 1 + max(len(cc) for cc in cuts)
    xin = x
    for k in range(n_sections):
        sections = []
        for c in range(nch):
            bounds = [0] + cuts[c] + [len(chains[c])]
            sections.append(chains[c][bounds[k]:bounds[k + 1]] if k + 1 < len(bounds) else [])
        got, want = run(sections, xin)
        for c in range(nch):
            err = rms(got[c] - want[c])
            assert np.isfinite(got[c]).all(), (seed, c, describe(chains[c]))
            assert err <= TOL_RMS, "seed %d channel %d section %d: RMS %.3e, chain %s at %d Hz, %d frames" % (
                seed, c, k, err, describe(chains[c]), sr, frames)
        xin = want


def test_octaver_behind_a_decimator_matches_on_the_same_input(oracle):
    """The case the random chains turned up (seed 0): 4x-oversampled excess -> octaver.  Each unit alone follows the oracle; the pair does
    not, because the decimator's first outputs are silence that the reference's FFT turns into rounding noise of arbitrary sign, and the
    octaver's register state hangs on those signs.  Fed the oracle's decimator output, the device octaver agrees."""
    pkg = package()
    sr, frames, blocks = 44100, 1000, 5
    x = 0.5 * synth_signal(11, frames * blocks, sr)
    ex, oc = ("excess", [-19, 0, 2]), ("octaver", [1, 0, -23, 0, -60, -8, 0])

    def run(units, xin):
        ctx = pkg.Context(1, frames)
        ref = oracle.Chain()
        for name, params in units:
            ctx.append_unit(0, name, params=params)
            ref.append_unit(name, params=params)
        got = np.concatenate([ctx.process(xin[None, b * frames:(b + 1) * frames], sr)[0] for b in range(blocks)])
        want = np.concatenate([ref.process(xin[b * frames:(b + 1) * frames], sr) for b in range(blocks)])
        ctx.close()
        return got, want

    g1, w1 = run([ex], x)
    assert rms(g1 - w1) <= 1e-15                          # the decimated signal itself: identical to rounding
    lead = np.nonzero(np.abs(w1) > 1e-12)[0][0]
    assert lead > 0 and np.max(np.abs(w1[:lead])) > 0.0 and not np.array_equal(np.sign(g1[:lead]), np.sign(w1[:lead]))     # silence: noise of other signs
    g2, w2 = run([oc], w1)
    assert rms(g2 - w2) <= TOL_RMS                        # same input, same output


@pytest.mark.parametrize("seed", range(64))
def test_random_edits_in_mid_stream_follow_the_oracle(oracle, seed):
    """A stream during which the chain keeps changing the way a user at the web UI changes it: parameters set to new values, units bypassed
    and re-enabled, moved up and down, the block size switched between calls (power amps carry their convolution state across, like
    filter.Process), the sample rate changed (the reference re-makes the rate-dependent buffers).  Every edit goes to the HIP context and
    to the oracle's chain; every block is compared."""
    pkg = package()
    rng = np.random.default_rng(5000 + seed)
    rates = [44100, 48000, 96000]
    sr = int(rng.choice(rates))
    max_frames = int(rng.choice([8192, 2048, 1024]))
    nch = 2
    ctx = pkg.Context(nch, max_frames)
    pairs, types = [], []
    for c in range(nch):
        p = ChainPair(ctx, c, oracle)
        ts = []
        with_fft = bool(rng.random() < 0.6)
        for _ in range(int(rng.integers(2, 7))):
            while True:
                t = int(rng.integers(0, 21))
                name = pkg.UNIT_NAMES[t]
                if name == "power_amp" and not with_fft:
                    continue
                if name == "octaver":                       # discontinuous in its input (see _split_points): not this test's subject
                    continue
                break
            if name == "power_amp":
                p.append(name, fir=synth_ir(int(rng.choice([33, 700, 2500])), seed=int(rng.integers(1, 10 ** 6))), bypass=bool(rng.random() < 0.2))
            else:
                p.append(name, params=random_params(rng, t, allow_oversampling=with_fft), bypass=bool(rng.random() < 0.2))
            ts.append(t)
        pairs.append(p)
        types.append(ts)
    bypass = [[b for _, b in ctx._chains[c]] for c in range(nch)]
    order = [list(range(len(types[c]))) for c in range(nch)]          # order[c][slot] = index into pairs[c].handles / types[c]
    log = []
    pos = 0
    for blk in range(10):
        # ---- edits -------------------------------------------------------------------------------------------------------
        for c in range(nch):
            p = pairs[c]
            r = rng.random()
            if r < 0.25:                                              # a parameter gets a new value
                slot = int(rng.integers(0, len(order[c])))
                u = order[c][slot]
                name = pkg.UNIT_NAMES[types[c][u]]
                if name != "power_amp":
                    newp = random_params(rng, types[c][u], allow_oversampling=any(pkg.UNIT_NAMES[t] == "power_amp" for t in types[c]) or name != "octaver")
                    if name == "octaver" or not any(pkg.UNIT_NAMES[t] == "octaver" for t in types[c]):
                        idx = int(rng.integers(0, len(newp)))
                        ctx.unit_set_param(p.handles[u], idx, newp[idx])
                        p.ref.unit(slot).set_param(idx, newp[idx])
                        log.append((blk, c, "param", name, idx, newp[idx]))
            elif r < 0.45:                                            # bypass toggled
                slot = int(rng.integers(0, len(order[c])))
                bypass[c][slot] = not bypass[c][slot]
                ctx.chain_set(c, [p.handles[u] for u in order[c]], bypass[c])
                p.ref.set_bypass(slot, bypass[c][slot])
                log.append((blk, c, "bypass", slot, bypass[c][slot]))
            elif r < 0.6 and len(order[c]) > 1:                       # a unit moves up (its state travels with it)
                slot = int(rng.integers(1, len(order[c])))
                order[c][slot - 1], order[c][slot] = order[c][slot], order[c][slot - 1]
                bypass[c][slot - 1], bypass[c][slot] = bypass[c][slot], bypass[c][slot - 1]
                ctx.chain_set(c, [p.handles[u] for u in order[c]], bypass[c])
                p.ref.move_up(slot)
                log.append((blk, c, "move_up", slot))
        if rng.random() < 0.12:
            sr = int(rng.choice(rates))
            log.append((blk, "rate", sr))
        frames = max_frames if rng.random() < 0.6 else int(rng.choice([max_frames // 2, max_frames // 4, 1000 if max_frames >= 1000 else 480]))
        # frame sizes that are not a power of two with a long filter: the reference panics (filter.go:443-453) -- keep to the others there
        if frames & (frames - 1) and any(pkg.UNIT_NAMES[t] == "power_amp" for ts in types for t in ts):
            frames = max_frames // 2
        # ---- one block --------------------------------------------------------------------------------------------------------
        x = np.stack([synth_signal(3 + 5 * c, pos + frames, sr)[pos:] * 0.6 for c in range(nch)])
        pos += frames
        got = ctx.process(np.ascontiguousarray(x), sr)
        for c in range(nch):
            want = pairs[c].ref.process(x[c], sr)
            err = rms(got[c] - want)
            assert err <= TOL_RMS, "seed %d block %d channel %d (%d frames at %d Hz): RMS %.3e; units %s; edits %s" % (
                seed, blk, c, frames, sr, err, [pkg.UNIT_NAMES[types[c][u]] for u in order[c]], log)
    ctx.close()


@pytest.mark.parametrize("seed", range(32))
def test_random_chains_in_windows_give_the_bits_of_per_frame_calls(seed):
    """Batch mode: the same random chains walked W frames per call (time-blocked convolution, one segment launch per window, the window size
    changing from call to call like at the tail of a file) and one frame per call: identical samples, whatever the units."""
    pkg = package()
    rng = np.random.default_rng(9000 + seed)
    sr = int(rng.choice([44100, 96000, 192000]))
    frames, nch, blocks = 8192, 3, 14
    chains = []
    for c in range(nch):
        units = []
        for _ in range(int(rng.integers(1, 7))):
            t = int(rng.integers(0, 21))
            name = pkg.UNIT_NAMES[t]
            if name == "power_amp":
                units.append((name, synth_ir(int(rng.choice([100, 8192, 20000, 40000])), seed=int(rng.integers(1, 10 ** 6))) * 0.8, bool(rng.random() < 0.1)))
            else:
                units.append((name, random_params(rng, t, allow_oversampling=True), bool(rng.random() < 0.1)))
        chains.append(units)
    x = np.stack([synth_signal(int(rng.integers(0, 48)), frames * blocks, sr) * 0.7 for _ in range(nch)])
    W = int(rng.choice([2, 4, 8, 16]))
    plan, left = [], blocks                                     # windows of W, then the tail in windows of W/2 ... 1
    w = W
    while left > 0:
        while w > left:
            w //= 2
        plan.append(w)
        left -= w
    outs = {}
    for mode in ("windows", "frames"):
        ctx = pkg.Context(nch, frames)
        for c in range(nch):
            for name, arg, bypass in chains[c]:
                if name == "power_amp":
                    ctx.append_unit(c, name, fir=arg, bypass=bypass)
                else:
                    ctx.append_unit(c, name, params=arg, bypass=bypass)
        d_in, d_out = ctx.alloc(nch, blocks * frames), ctx.alloc(nch, blocks * frames)
        d_in.upload(x)
        if mode == "windows":
            ctx.set_window(W)
        b = 0
        for w in (plan if mode == "windows" else [1] * blocks):
            ctx.process_window_device(d_in.ptr + 8 * b * frames, d_out.ptr + 8 * b * frames, blocks * frames, w, sr)
            b += w
        outs[mode] = d_out.download()
        ctx.close()
    assert np.isfinite(outs["frames"]).all()
    assert np.array_equal(outs["windows"], outs["frames"]), (seed, W, plan, [[(n, b) for n, _, b in ch] for ch in chains],
                                                             float(np.max(np.abs(outs["windows"] - outs["frames"]))))


FORMATS = ["lpcm8", "lpcm16", "lpcm24", "lpcm32", "ieee32", "ieee64"]
BLOCK = 8192


@pytest.mark.parametrize("seed", range(16))
def test_random_batch_runs_follow_the_oracle_pipeline(oracle, seed):
    """gdg_batch_run on random jobs: 2-5 inputs in random sample formats, lengths (odd ones included), rates (some need resample.Time),
    interleaved files of which one channel is taken, empty inputs; random chains; a random window size; the metronome in or out of the
    master mix; a random output format.  Against the oracle's pipeline (decode -> resample.Time -> pad -> per block: chains, metronome,
    spatializer + aux -> encode): 8-, 16- and 24-bit containers byte for byte except samples that sit on a code boundary (see below), 32-bit
    codes within the float tolerance, float containers within 1e-9 RMS."""
    pkg = package()
    rng = np.random.default_rng(7000 + seed)
    rate = int(rng.choice([44100, 48000, 96000]))
    nch = int(rng.integers(2, 6))
    W = int(rng.choice([1, 2, 4, 8]))
    out_fmt = str(rng.choice(FORMATS))
    to_master = bool(rng.random() < 0.5)
    # ---- the files ------------------------------------------------------------------------------------------------------------
    inputs, decoded = [None] * nch, {}
    for c in range(nch):
        if rng.random() < 0.15:
            continue                                             # "leaving channel empty"
        fmt = str(rng.choice(FORMATS))
        r = int(rng.choice([rate, rate, 44100, 22050]))
        n = int(rng.choice([1, 7, 5000, BLOCK, BLOCK + 1, 3 * BLOCK - 5, 20000]))
        chans = int(rng.choice([1, 1, 2, 3]))
        take = int(rng.integers(0, chans))
        w = pkg.lib().gdg_wave_bytes_per_sample(pkg.WAVE_FORMATS[fmt])
        per_chan = [oracle.wave_encode(fmt, 0.8 * synth_signal(7 * c + k, n, r)).reshape(n, w) for k in range(chans)]
        data = np.ascontiguousarray(np.stack(per_chan, axis=1)).reshape(-1)
        inputs[c] = (data, fmt, r, chans, take)
        xd = oracle.wave_decode(fmt, per_chan[take].reshape(-1))
        decoded[c] = xd if r == rate else oracle.resample_time(xd, r, rate)
    longest = max([len(v) for v in decoded.values()] + [0])
    length = BLOCK * ((longest + BLOCK - 1) // BLOCK)
    # ---- chains, spatializer, metronome on both sides ----------------------------------------------------------------------------
    ctx = pkg.Context(nch, BLOCK)
    ctx.set_window(W)
    refs = []
    for c in range(nch):
        p = ChainPair(ctx, c, oracle)
        fft = bool(rng.random() < 0.5)
        for _ in range(int(rng.integers(0, 5))):
            while True:
                t = int(rng.integers(0, 21))
                name = pkg.UNIT_NAMES[t]
                if (name == "power_amp" and not fft) or (name == "octaver" and (fft or p.handles)):      # an octaver only at the head (see _split_points)
                    continue
                break
            if name == "power_amp":
                p.append(name, fir=synth_ir(int(rng.choice([50, 3000, 12000])), seed=int(rng.integers(1, 10 ** 6))) * 0.7)
            else:
                p.append(name, params=random_params(rng, t, allow_oversampling=fft))
        refs.append(p.ref)
    ref_sp = oracle.Spatializer(nch)
    ctx.spatializer_set_sample_rate(rate)
    ref_sp.set_sample_rate(rate)
    for c in range(nch):
        a, d, l = float(rng.uniform(-180, 180)), float(rng.uniform(0.1, 10)), float(rng.uniform(0, 1))
        ctx.spatializer_set_position(c, a, d, l)
        ref_sp.set_azimuth(c, a); ref_sp.set_distance(c, d); ref_sp.set_level(c, l)
    tick, tock = rng.uniform(-0.5, 0.5, 700), rng.uniform(-0.5, 0.5, 300)
    beats, bpm = int(rng.integers(1, 8)), int(rng.integers(40, 360))
    ctx.metronome_set_sounds(tick, tock)
    ctx.metronome_configure(beats, bpm, rate)
    ref_met = oracle.Metronome()
    ref_met.tick, ref_met.tock = tick, tock
    ref_met.s.beats_per_period, ref_met.s.bpm_speed, ref_met.s.sample_rate = beats, bpm, rate
    # ---- oracle pipeline ---------------------------------------------------------------------------------------------------------
    xin = np.zeros((nch, length))
    for c, v in decoded.items():
        xin[c, :len(v)] = v
    ref_out = np.zeros((nch + 3, length))
    for b in range(length // BLOCK):
        sl = slice(b * BLOCK, (b + 1) * BLOCK)
        for c in range(nch):
            ref_out[c, sl] = refs[c].process(xin[c, sl], rate)
        ref_out[nch + 2, sl] = ref_met.process(BLOCK)
        ref_out[nch, sl], ref_out[nch + 1, sl] = ref_sp.process(ref_out[:nch, sl], aux=(ref_out[nch + 2, sl] if to_master else None))
    # ---- device: one call -----------------------------------------------------------------------------------------------------------
    outs = ctx.batch_run(inputs, rate, out_fmt, metronome_to_master=to_master)
    ctx.close()
    wo = pkg.lib().gdg_wave_bytes_per_sample(pkg.WAVE_FORMATS[out_fmt])
    assert len(outs) == nch + 3 and all(o.size == length * wo for o in outs), (seed, [o.size for o in outs], length)
    for r in range(nch + 3):
        want = oracle.wave_encode(out_fmt, ref_out[r]) if length else np.zeros(0, dtype=np.uint8)
        if out_fmt in ("ieee32", "ieee64"):
            err = rms(oracle.wave_decode(out_fmt, outs[r]) - oracle.wave_decode(out_fmt, want)) if length else 0.0
            assert err <= TOL_RMS, (seed, r, out_fmt, err)
        elif out_fmt == "lpcm32":
            # a 32-bit code is 4.7e-10 wide: the chains' legitimate 1e-16 .. 1e-15 differences (device exp / sin / log10 against glibc's, scan
            # association) move a sample across a truncation boundary about once in 10^6 samples (seed 2056 of profiles/probes/fuzz_soak.py).
            # The encoder itself is bit exact on equal input (test_random_codec_and_resampler_jobs...): here a few codes may be off by one.
            got_i = outs[r].view("<i4").astype(np.int64) if length else np.zeros(0, dtype=np.int64)
            want_i = want.view("<i4").astype(np.int64) if length else np.zeros(0, dtype=np.int64)
            d = np.abs(got_i - want_i)
            # behind stages the reference computes by FFT (oversampled units, power amps) the float difference reaches 1e-13 and one sample in a
            # few thousand moves (seed 5421 of the soak: 10 of 24576); the bound is the float tolerance expressed in codes
            assert d.size == 0 or (d.max() <= 2 and float(np.sqrt(np.mean((d / 2147483648.0) ** 2))) <= TOL_RMS), (seed, r, int(d.max(initial=0)), int(np.count_nonzero(d)))
        else:
            if not np.array_equal(outs[r], want):
                # a sample that sits ON a code boundary may take either code: the compressor with a peak follower and a 0 dB target (or the
                # auto-yoy at full level) puts its peaks at +-1 (1 +- one ulp) -- exactly the encoder's top boundary, where 254 or 255 is a matter
                # of the last bit (soak seeds 20198, 20445).  Every differing sample must be the code of the oracle's value moved by <= 1e-9.
                g = outs[r].reshape(length, wo)
                ok = np.all(g == want.reshape(length, wo), axis=1)
                for delta in (-1e-9, 1e-9):
                    ok |= np.all(g == oracle.wave_encode(out_fmt, ref_out[r] + delta).reshape(length, wo), axis=1)
                bad = int(np.count_nonzero(~ok))
                assert bad == 0, "seed %d output %d (%s, W = %d): %d samples differ by more than a boundary case" % (seed, r, out_fmt, W, bad)


@pytest.mark.parametrize("seed", range(24))
def test_random_spatializer_jobs_follow_the_oracle(oracle, seed):
    """spatializer.Process on random shards: channel counts either side of the kernel's limits (one lane group of 32, the 512 descriptors kept in
    LDS), any rate, frame sizes that change from call to call, positions that move between calls (history carried over, spatializer.go:313-331)."""
    pkg = package()
    rng = np.random.default_rng(9000 + seed)
    nch = int(rng.choice([1, 2, 3, 17, 32, 33, 64, 130, 513]))
    max_frames = int(rng.choice([64, 1000, 8192]))
    sr = int(rng.choice([22050, 44100, 48000, 96000, 192000]))
    ctx = pkg.Context(nch, max_frames)
    ref = oracle.Spatializer(nch)
    ctx.spatializer_set_sample_rate(sr)
    ref.set_sample_rate(sr)

    def place(c):
        a = float(rng.choice([-180.0, -90.0, 0.0, 90.0, 180.0, rng.uniform(-180, 180)]))
        d = float(rng.choice([0.0, 0.05, 10.0, rng.uniform(0, 10)]))
        l = float(rng.choice([0.0, 1.0, rng.uniform(0, 1)]))
        ctx.spatializer_set_position(c, a, d, l)
        assert ref.set_azimuth(c, a) == 0 and ref.set_distance(c, d) == 0 and ref.set_level(c, l) == 0

    for c in range(nch):
        if rng.random() < 0.8:
            place(c)                                             # the others keep the defaults
    for call in range(int(rng.integers(3, 7))):
        frames = int(rng.choice([1, 2, max_frames, int(rng.integers(1, max_frames + 1))]))
        x = rng.uniform(-1.0, 1.0, (nch, frames))
        if rng.random() < 0.3:
            x[int(rng.integers(0, nch))] = 0.0
        gl, gr = ctx.spatialize(x)
        wl, wr = ref.process(x)
        assert rms(gl - wl) <= TOL_RMS and rms(gr - wr) <= TOL_RMS, (seed, call, nch, frames, sr, rms(gl - wl), rms(gr - wr))
        for _ in range(int(rng.integers(0, 3))):
            place(int(rng.integers(0, nch)))
        if rng.random() < 0.2:
            sr = int(rng.choice([44100, 96000, 192000]))
            ctx.spatializer_set_sample_rate(sr)
            ref.set_sample_rate(sr)
    ctx.close()


@pytest.mark.parametrize("seed", range(16))
def test_random_tuner_signals_follow_the_oracle(oracle, seed):
    """tuner.Process / Analyze on random harmonic tones (fundamental anywhere in the tuner's range, random partials and phases, noise 20 dB or more
    below the tone), silence and short fills of the ring, at random rates and frame sizes: note, cents, and the frequency to 1e-9."""
    pkg = package()
    rng = np.random.default_rng(9500 + seed)
    nch = int(rng.integers(1, 9))
    sr = int(rng.choice([22050, 44100, 48000, 96000, 192000]))
    frames = int(rng.choice([64, 1000, 4096, 8192]))
    total = int(rng.choice([3 * frames, 96000 + 2 * frames, 50000]))
    # at least four times the longest lag the analysis looks at: with fewer samples than the lag the autocorrelation is identically zero there
    # and the arg-max picks among rounding noise (1e-16) -- the reference's own result is then an accident of its FFT's rounding (seeds 1061
    # and 1147 of profiles/probes/fuzz_soak.py: 192 samples at 192 kHz, a tone whose autocorrelation is negative at every lag that overlaps)
    total = max(total, 4 * int(sr / 61.7 + 1.5))
    total = max(frames, (total // frames) * frames)
    t = np.arange(total) / float(sr)
    x = np.zeros((nch, total))
    for c in range(nch):
        kind = rng.random()
        if kind < 0.15:
            continue                                             # silence
        f0 = float(np.exp(rng.uniform(np.log(62.0), np.log(1900.0))))
        amps = [1.0] + [float(rng.uniform(0.0, 0.6)) / h for h in range(2, 6)]
        tone = sum(a * np.sin(2 * np.pi * f0 * (h + 1) * t + rng.uniform(0, 2 * np.pi)) for h, a in enumerate(amps) if f0 * (h + 1) < 0.45 * sr)
        tone = 0.5 * tone / max(np.max(np.abs(tone)), 1e-9)
        x[c] = tone + float(rng.uniform(0.0, 0.03)) * rng.standard_normal(total)
    ctx = pkg.Context(nch, frames)
    refs = [oracle.Tuner() for _ in range(nch)]
    for b in range(0, total, frames):
        ctx.tuner_enqueue(x[:, b:b + frames], sr)
        for c in range(nch):
            refs[c].process(x[c, b:b + frames], sr)
    got = ctx.tuner_analyze()
    ctx.close()
    for c in range(nch):
        want = refs[c].analyze()
        assert got[c]["note_index"] == want["note_index"] and got[c]["cents"] == want["cents"], (seed, c, got[c], want)
        if np.isnan(want["frequency"]):                          # silence: 0 / 0 in the reference's interpolation (tuner.go:452-470)
            assert np.isnan(got[c]["frequency"]), (seed, c, got[c], want)
        elif want["frequency"] > 0.0:
            assert abs(got[c]["frequency"] - want["frequency"]) / want["frequency"] <= 1e-9, (seed, c, got[c], want)
        else:
            assert got[c]["frequency"] == want["frequency"], (seed, c, got[c], want)


@pytest.mark.parametrize("seed", range(16))
def test_random_host_calls_follow_the_oracle(oracle, seed):
    """The host-buffer entry points on random jobs: gdg_process (all channels), gdg_process_subset and gdg_process_staged (random channel
    subsets in random order -- the channels left out keep their state untouched), frame sizes changing from call to call, one context."""
    pkg = package()
    rng = np.random.default_rng(9900 + seed)
    nch = int(rng.integers(2, 10))
    max_frames = int(rng.choice([512, 2048, 8192]))
    sr = int(rng.choice([44100, 48000, 96000, 192000]))
    ctx = pkg.Context(nch, max_frames)
    pairs = []
    for c in range(nch):
        p = ChainPair(ctx, c, oracle)
        fft = bool(rng.random() < 0.4)
        for _ in range(int(rng.integers(0, 4))):
            while True:
                t = int(rng.integers(0, 21))
                name = pkg.UNIT_NAMES[t]
                if (name == "power_amp" and not fft) or (name == "octaver" and (fft or p.handles)):      # an octaver only at the head (see _split_points)
                    continue
                break
            if name == "power_amp":
                taps = int(rng.choice([64, 900, 5000]))
                p.append(name, fir=synth_ir(taps, seed=int(rng.integers(1, 10 ** 6))) * 0.7)
            else:
                p.append(name, params=random_params(rng, t, allow_oversampling=fft))
        pairs.append(p)
    pos = 0
    for call in range(int(rng.integers(4, 9))):
        frames = int(rng.choice([max_frames, max_frames // 2, max_frames // 8]))           # powers of two: no reference-panic pairs
        how = rng.random()
        if how < 0.4:
            chans = list(range(nch))
        else:
            k = int(rng.integers(1, nch + 1))
            chans = [int(c) for c in rng.permutation(nch)[:k]]
        x = np.stack([0.7 * synth_signal(31 * c + 1, pos + frames, sr)[pos:] for c in chans])
        if how < 0.4:
            y = ctx.process(x, sr)
        elif how < 0.7:
            y = ctx.process_subset(chans, x, sr)
        else:
            y = ctx.process_staged(chans, x, sr)
        for i, c in enumerate(chans):
            want = pairs[c].ref.process(x[i], sr)
            err = rms(y[i] - want)
            assert err <= TOL_RMS, (seed, call, c, frames, err)
        pos += frames
    ctx.close()


@pytest.mark.parametrize("seed", range(16))
def test_random_meter_streams_follow_the_oracle(oracle, seed):
    """level.channelMeter on random streams cut into calls of random lengths (1 .. 8192, so segments that are not multiples of the kernel's chunk,
    shorter than a wave, a single sample), with quantised samples (many equal maxima: the LAST one restarts the hold), bursts, silences, and rates
    low enough for the two-second hold to run out inside the stream: the state triple (current value, peak, counter) and the integer dB readings
    after every few calls."""
    pkg = package()
    rng = np.random.default_rng(12000 + seed)
    sr = int(rng.choice([4000, 8000, 22050, 48000, 192000]))
    ports = int(rng.integers(1, 6))
    total = int(rng.choice([3 * sr, 30000, 70000]))
    x = np.zeros((ports, total))
    for p in range(ports):
        kind = rng.integers(0, 4)
        if kind == 0:
            x[p] = np.round(rng.normal(0, 0.3, total) * 8) / 8                  # few distinct levels: ties everywhere
        elif kind == 1:
            a, b = sorted(rng.integers(0, total, 2))
            x[p, a:b] = rng.uniform(-1, 1, b - a)                               # one burst, silence around it
        elif kind == 2:
            x[p] = 0.5 * np.sin(2 * np.pi * rng.uniform(20, 2000) * np.arange(total) / sr) * np.exp(-np.arange(total) / (0.3 * total))
        # kind 3: silence
    ctx = pkg.Context(1, 8192)
    ctx.meter_configure(ports)
    ctx.meter_set_enabled(True)
    refs = [oracle.ChannelMeter() for _ in range(ports)]
    for r in refs:
        r.set_enabled(True)
    pos, call = 0, 0
    while pos < total:
        n = int(rng.choice([1, 2, 63, 64, 1000, 8192, int(rng.integers(1, 8193))]))
        n = min(n, total - pos)
        ctx.meter_process(x[:, pos:pos + n], sr)
        for p, r in enumerate(refs):
            r.process(x[p, pos:pos + n], sr)
        pos += n
        call += 1
        if call % 4 == 0 or pos >= total:
            lv, pk = ctx.meter_analyze()
            for p, r in enumerate(refs):
                cur, peak, cnt = ctx.meter_state(p)
                rc, rp, rn = r.state
                assert cnt == rn, (seed, p, pos, n, cnt, rn)
                assert abs(cur - rc) <= 1e-12 * max(rc, 1e-300), (seed, p, pos, cur, rc)
                assert abs(peak - rp) <= 1e-12 * max(rp, 1e-300), (seed, p, pos, peak, rp)
                assert (lv[p], pk[p]) == r.analyze(), (seed, p, pos)
    ctx.close()


@pytest.mark.parametrize("seed", range(16))
def test_random_metronome_settings_follow_the_oracle(oracle, seed):
    """metronome.Process with random tick / tock sounds (long, short, empty, none), beats per period (0 included: the reference counts it as 1),
    speeds, rates and a random frame size per call, reconfigured in mid-stream: every block bit for bit."""
    pkg = package()
    rng = np.random.default_rng(13000 + seed)
    ctx = pkg.Context(1, 8192)
    ref = oracle.Metronome()

    def sound():
        k = rng.integers(0, 5)
        if k == 0:
            return None
        if k == 1:
            return np.zeros(0)
        return rng.uniform(-1, 1, int(rng.choice([1, 50, 3000, 30000])))

    def configure():
        tick, tock = sound(), sound()
        beats, bpm, sr = int(rng.integers(0, 9)), int(rng.choice([1, 40, 120, 360, 1000])), int(rng.choice([8000, 44100, 192000]))
        ctx.metronome_set_sounds(tick, tock)
        ctx.metronome_configure(beats, bpm, sr)
        ref.tick, ref.tock = tick, tock
        ref.s.beats_per_period, ref.s.bpm_speed, ref.s.sample_rate = beats, bpm, sr

    configure()
    for call in range(40):
        n = int(rng.choice([1, 17, 1024, 8192, int(rng.integers(1, 8193))]))
        got, want = ctx.metronome_process(n), ref.process(n)
        assert np.array_equal(got, want), (seed, call, n, int(np.count_nonzero(got != want)))
        if rng.random() < 0.15:
            configure()
    ctx.close()


@pytest.mark.parametrize("seed", range(12))
def test_random_codec_and_resampler_jobs_follow_the_oracle(oracle, seed):
    """The data formats either side of the path on random jobs: every sample format, random lengths (0 and 1 included), interleaved files of
    1-4 channels, values beyond full scale (the encoders clip), then resample.Time between random rate pairs: encoders and the integer decoders
    byte / bit exact, the resampler to 1e-9."""
    pkg = package()
    rng = np.random.default_rng(14000 + seed)
    ctx = pkg.Context(1, 64)
    for _ in range(6):
        fmt = str(rng.choice(FORMATS))
        n = int(rng.choice([0, 1, 3, 4, 5, 1023, 4096, int(rng.integers(1, 20000))]))
        chans = int(rng.integers(1, 5))
        x = rng.uniform(-1.3, 1.3, (chans, n))
        if n:
            x[0, 0] = float(rng.choice([-1.0, 1.0, 0.0, -2.0, 2.0]))
        # encode every channel (planar), interleave on the host, decode the interleaved file on the device
        enc = [ctx.wave_encode(fmt, x[c]) for c in range(chans)]
        want = [oracle.wave_encode(fmt, x[c]) for c in range(chans)]
        for c in range(chans):
            assert np.array_equal(enc[c], want[c]), (seed, fmt, n, c)
        w = pkg.lib().gdg_wave_bytes_per_sample(pkg.WAVE_FORMATS[fmt])
        inter = np.ascontiguousarray(np.stack([e.reshape(n, w) for e in enc], axis=1)).reshape(-1) if n else np.zeros(0, dtype=np.uint8)
        dec = np.asarray(ctx.wave_decode(fmt, inter, channels=chans)).reshape(chans, n)
        for c in range(chans):
            ref = oracle.wave_decode(fmt, want[c])
            assert dec.shape == (chans, n) and np.array_equal(dec[c], ref), (seed, fmt, n, c)
    for _ in range(4):
        src, dst = int(rng.choice([8000, 22050, 44100, 48000, 96000])), int(rng.choice([44100, 48000, 96000, 192000]))
        n = int(rng.choice([1, 2, 7, 500, int(rng.integers(1, 6000))]))
        x = rng.uniform(-1, 1, n)
        got, want = ctx.resample_time(x, src, dst), oracle.resample_time(x, src, dst)
        assert got.shape == want.shape, (seed, src, dst, n, got.shape, want.shape)
        if want.size:
            assert rms(got - want) <= TOL_RMS, (seed, src, dst, n, rms(got - want))
    ctx.close()


@pytest.mark.parametrize("seed", range(12))
def test_random_power_amp_compiles_follow_the_oracle(oracle, seed):
    """effects/poweramp.go:25-127 on random slot lists: up to 8 slots -- absent, empty, 1 tap to tens of thousands --, random gain compensations and
    levels, target orders below, between and above the slots' lengths (0 = keep): the composite tap by tap."""
    from test_gpu_compile import oracle_compile
    pkg = package()
    rng = np.random.default_rng(15000 + seed)
    ctx = pkg.Context(1, 256)
    h = ctx.append_unit(0, "power_amp")
    for _ in range(3):
        filters = []
        for _s in range(int(rng.integers(1, 9))):
            k = rng.integers(0, 6)
            if k == 0:
                filters.append((None, 1.0, 0))
            elif k == 1:
                filters.append((np.zeros(0), 1.0, 0))
            else:
                n = int(rng.choice([1, 2, 100, 777, 4096, 5000, 20000, int(rng.integers(1, 40000))]))
                filters.append((synth_ir(n, seed=int(rng.integers(1, 10 ** 6))) * float(rng.uniform(0.1, 2.0)),
                                10.0 ** (0.05 * float(rng.uniform(-30, 0))), int(rng.integers(-30, 7))))
        order = int(rng.choice([0, 1, 2, 3, 64, 1000, 1024, 4096, 30000, 65536]))
        ctx.unit_compile_fir(h, filters, order)
        got = ctx.unit_get_fir(h)
        want = oracle_compile(oracle, filters, order)
        assert len(got) == len(want), (seed, order, len(got), len(want))
        if len(want):
            assert rms(got - want) <= TOL_RMS * max(rms(want), 1e-300), (seed, order, rms(got - want), rms(want))
            np.testing.assert_allclose(got, want, rtol=0, atol=1e-12 * max(1.0, float(np.max(np.abs(want)))))
    ctx.close()
