"""The batch run end to end ON THE DEVICE against the oracle: what controller.processFiles does between "the WAV data sections are
in memory" and "the output data sections are in memory" (controller/controller.go:2884-3219):

    bytesToSamples (six sample formats) -> resample.Time to the target rate -> zero-pad to a multiple of BLOCK_SIZE = 8192 ->
    per block: copy in, N x Chain.Process, metronome, spatializer N -> 2, level meters over the 2N+3 ports, copy out ->
    samplesToBytes of the N + 3 outputs.

Every stage runs as a HIP kernel through the *_device entry points of the C-ABI; nothing touches host memory between the upload of
the file bytes and the download of the result bytes.  The oracle runs the same stages on the CPU."""
import ctypes as C

import numpy as np
import pytest

from helpers import TOL_RMS, package, rms, synth_ir, synth_signal

pytestmark = pytest.mark.gpu

BLOCK = 8192                                             # controller/controller.go:36
FORMATS = ["lpcm8", "lpcm16", "lpcm24", "lpcm32", "ieee32", "ieee64"]
CHAIN = [("compressor", [1, 30, -20]), ("overdrive", [0, 15, 80, -3, 1, 0]), ("tone_stack", None), ("chorus", None),
         ("power_amp", "ir"), ("cabinet", None), ("reverb", [30])]


class DevBytes:
    """A byte buffer in the context's device memory."""

    def __init__(self, ctx, pkg, nbytes):
        self.ctx, self.pkg, self.nbytes = ctx, pkg, nbytes
        p = C.c_void_p()
        ctx._check(pkg.lib().gdg_device_alloc(ctx._h, max(nbytes, 8), C.byref(p)))
        self.ptr = p.value

    def upload(self, a):
        a = np.ascontiguousarray(a)
        self.ctx._check(self.pkg.lib().gdg_copy_to_device(self.ctx._h, self.ptr, a.ctypes.data, a.nbytes))

    def download(self, dtype=np.uint8):
        out = np.empty(self.nbytes // np.dtype(dtype).itemsize, dtype=dtype)
        self.ctx._check(self.pkg.lib().gdg_copy_to_host(self.ctx._h, out.ctypes.data, self.ptr, out.nbytes))
        return out


def test_batch_run_on_device_matches_oracle(oracle):
    pkg = package()
    lib = pkg.lib()
    nch, src_rate, rate, n_src = len(FORMATS), 44100, 48000, 30000
    rng = np.random.default_rng(3)
    irs = [synth_ir(3000, seed=20 + c) for c in range(nch)]
    positions = [(float(rng.uniform(-90, 90)), float(rng.uniform(0.3, 5)), float(rng.uniform(0.2, 1))) for _ in range(nch)]
    tick, tock = rng.uniform(-0.5, 0.5, 1200), rng.uniform(-0.5, 0.5, 700)
    # ---- the "files": one mono data section per input, each in another sample format -----------------------------------
    sources = [0.8 * synth_signal(c, n_src, src_rate) for c in range(nch)]
    file_bytes = [oracle.wave_encode(FORMATS[c], sources[c]) for c in range(nch)]
    n_out = lib.gdg_resample_time_length(n_src, src_rate, rate)
    length = BLOCK * ((n_out + BLOCK - 1) // BLOCK)           # controller.go:3014-3016
    blocks = length // BLOCK
    ports = 2 * nch + 3

    # ---- oracle -------------------------------------------------------------------------------------------------------------
    ref_in = np.zeros((nch, length))
    for c in range(nch):
        ref_in[c, :n_out] = oracle.resample_time(oracle.wave_decode(FORMATS[c], file_bytes[c]), src_rate, rate)
    chains = []
    for c in range(nch):
        ch = oracle.Chain()
        for name, p in CHAIN:
            ch.append_unit(name, fir=irs[c]) if p == "ir" else ch.append_unit(name, params=p)
        chains.append(ch)
    ref_sp = oracle.Spatializer(nch)
    ref_sp.set_sample_rate(rate)
    for c, (a, d, l) in enumerate(positions):
        ref_sp.set_azimuth(c, a); ref_sp.set_distance(c, d); ref_sp.set_level(c, l)
    ref_met = oracle.Metronome()
    ref_met.tick, ref_met.tock = tick, tock
    ref_met.s.beats_per_period, ref_met.s.bpm_speed, ref_met.s.sample_rate = 4, 140, rate
    ref_meters = [oracle.ChannelMeter() for _ in range(ports)]
    for m in ref_meters:
        m.set_enabled(True)
    ref_out = np.zeros((nch + 3, length))
    for b in range(blocks):
        sl = slice(b * BLOCK, (b + 1) * BLOCK)
        for c in range(nch):
            ref_out[c, sl] = chains[c].process(ref_in[c, sl], rate)
        ref_out[nch + 2, sl] = ref_met.process(BLOCK)                                    # controller.go:2721-2742
        ref_out[nch, sl], ref_out[nch + 1, sl] = ref_sp.process(ref_out[:nch, sl])       # :2744-2761, metronome not in the master mix
        rows = [ref_in[c, sl] for c in range(nch)] + [ref_out[c, sl] for c in range(nch)] + [ref_out[nch + 2, sl], ref_out[nch, sl], ref_out[nch + 1, sl]]
        for m, r in zip(ref_meters, rows):                                               # :2707-2781: inputs, outputs, metronome, master L/R
            m.process(r, rate)

    # ---- device -------------------------------------------------------------------------------------------------------------
    ctx = pkg.Context(nch, BLOCK)
    for c in range(nch):
        for name, p in CHAIN:
            ctx.append_unit(c, name, fir=irs[c]) if p == "ir" else ctx.append_unit(c, name, params=p)
    ctx.spatializer_set_sample_rate(rate)
    for c, (a, d, l) in enumerate(positions):
        ctx.spatializer_set_position(c, a, d, l)
    ctx.metronome_set_sounds(tick, tock)
    ctx.metronome_configure(4, 140, rate)
    ctx.meter_configure(ports)
    ctx.meter_set_enabled(True)
    d_inputs = ctx.alloc(nch, length)                         # whole (padded) files, resident in HBM
    d_outputs = ctx.alloc(nch + 3, length)
    d_inputs.upload(np.zeros((nch, length)))
    d_src = ctx.alloc(1, n_src)
    for c in range(nch):
        raw = DevBytes(ctx, pkg, file_bytes[c].nbytes)
        raw.upload(file_bytes[c])
        ctx._check(lib.gdg_wave_decode_device(ctx._h, pkg.WAVE_FORMATS[FORMATS[c]], raw.ptr, n_src, 1, d_src.ptr))
        ctx._check(lib.gdg_resample_time_device(ctx._h, d_src.ptr, n_src, src_rate, rate, d_inputs.ptr + 8 * c * length, n_out))
    d_in_blk, d_out_blk, d_meter_blk = ctx.alloc(nch, BLOCK), ctx.alloc(nch + 3, BLOCK), ctx.alloc(ports, BLOCK)
    rows = lambda buf, r: buf.ptr + 8 * r * BLOCK
    for b in range(blocks):
        off = 8 * b * BLOCK
        ctx._check(lib.gdg_copy_rows_device(ctx._h, d_in_blk.ptr, BLOCK, d_inputs.ptr + off, length, BLOCK, nch))        # copy in
        ctx.process_device(d_in_blk, rows(d_out_blk, 0), BLOCK, rate)                                                  # N x Chain.Process
        ctx._check(lib.gdg_metronome_process_device(ctx._h, rows(d_out_blk, nch + 2), BLOCK))
        ctx._check(lib.gdg_spatialize_device(ctx._h, rows(d_out_blk, 0), rows(d_out_blk, nch), BLOCK))                 # -> rows N, N + 1
        ctx._check(lib.gdg_copy_rows_device(ctx._h, rows(d_meter_blk, 0), BLOCK, d_in_blk.ptr, BLOCK, BLOCK, nch))
        ctx._check(lib.gdg_copy_rows_device(ctx._h, rows(d_meter_blk, nch), BLOCK, rows(d_out_blk, 0), BLOCK, BLOCK, nch))
        ctx._check(lib.gdg_copy_rows_device(ctx._h, rows(d_meter_blk, 2 * nch), BLOCK, rows(d_out_blk, nch + 2), BLOCK, BLOCK, 1))
        ctx._check(lib.gdg_copy_rows_device(ctx._h, rows(d_meter_blk, 2 * nch + 1), BLOCK, rows(d_out_blk, nch), BLOCK, BLOCK, 2))
        ctx.meter_process_device(d_meter_blk, BLOCK, BLOCK, rate)
        ctx._check(lib.gdg_copy_rows_device(ctx._h, d_outputs.ptr + off, length, d_out_blk.ptr, BLOCK, BLOCK, nch + 3))  # copy out
    # ---- the output files: out_i in the input's own format, masters as 24-bit and 64-bit float, metronome as 16-bit --------
    out_formats = FORMATS + ["lpcm24", "ieee64", "lpcm16"]
    got_bytes = []
    for r, fmt in enumerate(out_formats):
        w = lib.gdg_wave_bytes_per_sample(pkg.WAVE_FORMATS[fmt])
        enc = DevBytes(ctx, pkg, w * length)
        ctx._check(lib.gdg_wave_encode_device(ctx._h, pkg.WAVE_FORMATS[fmt], d_outputs.ptr + 8 * r * length, length, 1, enc.ptr))
        got_bytes.append(enc.download())
    lv, pk = ctx.meter_analyze()
    got_out = d_outputs.download()
    ctx.close()

    # ---- compare ---------------------------------------------------------------------------------------------------------------
    for r in range(nch + 3):
        assert rms(got_out[r] - ref_out[r]) <= TOL_RMS, "output %d: RMS %.3e" % (r, rms(got_out[r] - ref_out[r]))
    for r, fmt in enumerate(out_formats):
        want = oracle.wave_encode(fmt, ref_out[r])
        if fmt in ("ieee32", "ieee64"):
            # float containers keep the 1e-16 differences of the device transcendentals: compare the decoded samples
            assert rms(oracle.wave_decode(fmt, got_bytes[r]) - oracle.wave_decode(fmt, want)) <= TOL_RMS, fmt
        else:
            # integer containers: identical bytes (a 1e-15 difference moves a truncation with probability ~1e-10 per sample)
            np.testing.assert_array_equal(got_bytes[r], want, err_msg="output %d (%s)" % (r, fmt))
    for p, m in enumerate(ref_meters):
        assert (lv[p], pk[p]) == m.analyze(), "meter port %d" % p


def _interleave(rows):
    return np.ascontiguousarray(np.stack(rows, axis=1)).reshape(-1)


def _batch_case(oracle, pkg):
    """A small batch with every special case of controller.processFiles (controller.go:2809-3219) and its oracle result: a stereo file
    whose second channel is taken and whose rate already is the target (no resampling, :2993), an empty input (:2935), inputs at two
    other rates, the metronome in the master mix (metrMasterOutput), meters on, tuner fed."""
    rate, nch = 48000, 5
    rng = np.random.default_rng(11)
    irs = [synth_ir(2000, seed=40 + c) for c in range(nch)]
    positions = [(float(rng.uniform(-90, 90)), float(rng.uniform(0.3, 5)), float(rng.uniform(0.2, 1))) for _ in range(nch)]
    tick, tock = rng.uniform(-0.5, 0.5, 900), rng.uniform(-0.5, 0.5, 500)
    ports = 2 * nch + 3
    # channel: (format, rate, samples, file channels, channel taken) -- channel 3 stays empty
    files = {0: ("lpcm16", 48000, 20000, 2, 1), 1: ("lpcm24", 44100, 62000, 1, 0), 2: ("ieee32", 96000, 50000, 3, 2), 4: ("lpcm32", 48000, 9000, 1, 0)}
    inputs, decoded = [None] * nch, {}
    for c, (fmt, r, n, chans, take) in files.items():
        chan_samples = [0.7 * synth_signal(10 * c + k, n, r) for k in range(chans)]
        w = pkg.lib().gdg_wave_bytes_per_sample(pkg.WAVE_FORMATS[fmt])
        per_chan = [oracle.wave_encode(fmt, s).reshape(n, w) for s in chan_samples]
        data = np.ascontiguousarray(np.stack(per_chan, axis=1)).reshape(-1)           # interleaved frames (wave.go:237-270)
        inputs[c] = (data, fmt, r, chans, take)
        x = oracle.wave_decode(fmt, per_chan[take].reshape(-1))
        decoded[c] = x if r == rate else oracle.resample_time(x, r, rate)
    longest = max(len(x) for x in decoded.values())
    length = BLOCK * ((longest + BLOCK - 1) // BLOCK)
    ref_in = np.zeros((nch, length))
    for c, x in decoded.items():
        ref_in[c, :len(x)] = x

    # ---- oracle ---------------------------------------------------------------------------------------------------------
    chains = []
    for c in range(nch):
        ch = oracle.Chain()
        for name, p in CHAIN:
            ch.append_unit(name, fir=irs[c]) if p == "ir" else ch.append_unit(name, params=p)
        chains.append(ch)
    ref_sp = oracle.Spatializer(nch)
    ref_sp.set_sample_rate(rate)
    for c, (a, d, l) in enumerate(positions):
        ref_sp.set_azimuth(c, a); ref_sp.set_distance(c, d); ref_sp.set_level(c, l)
    ref_met = oracle.Metronome()
    ref_met.tick, ref_met.tock = tick, tock
    ref_met.s.beats_per_period, ref_met.s.bpm_speed, ref_met.s.sample_rate = 3, 200, rate
    ref_meters = [oracle.ChannelMeter() for _ in range(ports)]
    for m in ref_meters:
        m.set_enabled(True)
    ref_tuners = [oracle.Tuner() for _ in range(nch)]
    ref_out = np.zeros((nch + 3, length))
    for b in range(length // BLOCK):
        sl = slice(b * BLOCK, (b + 1) * BLOCK)
        for c in range(nch):
            ref_tuners[c].process(ref_in[c, sl], rate)
            ref_out[c, sl] = chains[c].process(ref_in[c, sl], rate)
        ref_out[nch + 2, sl] = ref_met.process(BLOCK)
        ref_out[nch, sl], ref_out[nch + 1, sl] = ref_sp.process(ref_out[:nch, sl], aux=ref_out[nch + 2, sl])
        rows = [ref_in[c, sl] for c in range(nch)] + [ref_out[c, sl] for c in range(nch)] + [ref_out[nch + 2, sl], ref_out[nch, sl], ref_out[nch + 1, sl]]
        for m, r in zip(ref_meters, rows):
            m.process(r, rate)

    def configured(first=0, count=nch):
        """a context carrying channels first .. first + count - 1 of the job (the whole job by default)"""
        ctx = pkg.Context(count, BLOCK)
        for c in range(count):
            for name, p in CHAIN:
                ctx.append_unit(c, name, fir=irs[first + c]) if p == "ir" else ctx.append_unit(c, name, params=p)
        ctx.spatializer_set_sample_rate(rate)
        for c in range(count):
            ctx.spatializer_set_position(c, *positions[first + c])
        ctx.metronome_set_sounds(tick, tock)
        ctx.metronome_configure(3, 200, rate)
        ctx.meter_configure(2 * count + 3)
        ctx.meter_set_enabled(True)
        return ctx

    assert length == 9 * BLOCK
    from types import SimpleNamespace
    return SimpleNamespace(rate=rate, nch=nch, inputs=inputs, length=length, ref_out=ref_out, ref_meters=ref_meters, ref_tuners=ref_tuners,
                           configured=configured)


def test_batch_run_one_call_matches_oracle(oracle):
    """gdg_batch_run: the batch of _batch_case in ONE call of the C-ABI."""
    pkg = package()
    case = _batch_case(oracle, pkg)
    rate, nch, inputs, length, ref_out, ref_meters, ref_tuners, configured = (case.rate, case.nch, case.inputs, case.length, case.ref_out,
                                                                              case.ref_meters, case.ref_tuners, case.configured)
    # W: frames per step (time blocking, gdg_ctx_set_window): 9 blocks = 8 + 1 at W = 8, 2 + 2 + 2 + 2 + 1 at W = 2
    for out_fmt, W in (("ieee64", 1), ("lpcm24", 8), ("ieee64", 2)):
        ctx = configured()
        ctx.set_window(W)
        outs = ctx.batch_run(inputs, rate, out_fmt, metronome_to_master=True, run_meters=True, tuner_enqueue=True)
        lv, pk = ctx.meter_analyze()
        tuned = ctx.tuner_analyze()
        ctx.close()
        assert len(outs) == nch + 3
        for r in range(nch + 3):
            want = oracle.wave_encode(out_fmt, ref_out[r])
            assert outs[r].size == want.size == length * (8 if out_fmt == "ieee64" else 3)
            if out_fmt == "ieee64":
                err = rms(outs[r].view(np.float64) - ref_out[r])
                assert err <= TOL_RMS, "output %d: RMS %.3e" % (r, err)
            else:
                np.testing.assert_array_equal(outs[r], want, err_msg="output %d" % r)
        for p, m in enumerate(ref_meters):
            assert (lv[p], pk[p]) == m.analyze(), "meter port %d" % p
        for c in range(nch):
            want = ref_tuners[c].analyze()
            assert tuned[c]["note_index"] == want["note_index"] and tuned[c]["cents"] == want["cents"], (c, tuned[c], want)
            if np.isnan(want["frequency"]):                                    # the empty input: an all-zero correlation, 0/0 in the reference too
                assert np.isnan(tuned[c]["frequency"])
            else:
                assert abs(tuned[c]["frequency"] - want["frequency"]) <= 1e-9 * max(1.0, abs(want["frequency"]))


@pytest.mark.parametrize("W", [1, 4])
def test_batch_run_sharded_over_three_contexts_matches_oracle(oracle, W):
    """The same job split over THREE contexts in contiguous channel blocks (here all on one GPU): every shard runs
    gdg_batch_run_shard on its channels -- in windows of W frames -- and hands out its PARTIAL master mix as float64; the master is
    finished once (gdg_batch_finish_master): shard partials added in shard order, then the metronome as the aux input, then the
    encoder with its clip -- spatializer/spatializer.go:300-310, controller/controller.go:3123-3219.  Encoding each shard's partial
    mix instead would clip and truncate before the sum."""
    import importlib
    pkg = package()
    shard = importlib.import_module("go_dsp_guitar_amd.shard")
    case = _batch_case(oracle, pkg)
    rate, nch, length, ref_out = case.rate, case.nch, case.length, case.ref_out
    G = 3
    blocks = [shard.channel_shard(nch, G, g) for g in range(G)]
    assert [b[1] for b in blocks] == [1, 2, 2]
    ctxs = [case.configured(first, count) for first, count in blocks]
    for ctx in ctxs:
        ctx.set_window(W)
    job = max(ctx.batch_length(case.inputs[f:f + n], rate) for ctx, (f, n) in zip(ctxs, blocks))
    assert job == length
    for out_fmt in ("lpcm24", "ieee64"):
        if out_fmt == "ieee64":                      # a second job on fresh state
            for ctx in ctxs:
                ctx.close()
            ctxs = [case.configured(first, count) for first, count in blocks]
            for ctx in ctxs:
                ctx.set_window(W)
        lefts, rights, outs, metro_bytes, metro = [], [], [], None, None
        for g, (ctx, (first, count)) in enumerate(zip(ctxs, blocks)):
            o, l, r, mb, mf = ctx.batch_run_shard(case.inputs[first:first + count], rate, out_fmt, job_samples=job, metronome=(g == 0),
                                                  run_meters=True, tuner_enqueue=True)
            outs += o
            lefts.append(l)
            rights.append(r)
            if g == 0:
                metro_bytes, metro = mb, mf
        ml, mr = ctxs[0].batch_finish_master(out_fmt, lefts, rights, aux=metro, sample_rate=rate, run_meters=True)
        got = outs + [ml, mr, metro_bytes]
        for r in range(nch + 3):
            want = oracle.wave_encode(out_fmt, ref_out[r])
            assert got[r].size == want.size
            if out_fmt == "ieee64":
                err = rms(got[r].view(np.float64) - ref_out[r])
                assert err <= TOL_RMS, "output %d: RMS %.3e" % (r, err)
            else:
                np.testing.assert_array_equal(got[r], want, err_msg="output %d" % r)
        # a partial mix is NOT the master: the shards' partial sums add up to the mix before the aux input
        assert rms(sum(lefts) + metro - ref_out[nch]) <= TOL_RMS and rms(sum(rights) + metro - ref_out[nch + 1]) <= TOL_RMS
        if out_fmt == "lpcm24":
            # meters: every shard fed its inputs and outputs, shard 0 the metronome, the finishing context the master
            for g, (ctx, (first, count)) in enumerate(zip(ctxs, blocks)):
                lv, pk = ctx.meter_analyze()
                for c in range(count):
                    assert (lv[c], pk[c]) == case.ref_meters[first + c].analyze(), "input meter of channel %d" % (first + c)
                    assert (lv[count + c], pk[count + c]) == case.ref_meters[nch + first + c].analyze(), "output meter of channel %d" % (first + c)
                if g == 0:
                    assert (lv[2 * count], pk[2 * count]) == case.ref_meters[2 * nch].analyze(), "metronome meter"
                    for k in (1, 2):
                        assert (lv[2 * count + k], pk[2 * count + k]) == case.ref_meters[2 * nch + k].analyze(), "master meter %d" % k
            for g, (ctx, (first, count)) in enumerate(zip(ctxs, blocks)):
                tuned = ctx.tuner_analyze()
                for c in range(count):
                    want = case.ref_tuners[first + c].analyze()
                    assert tuned[c]["note_index"] == want["note_index"] and tuned[c]["cents"] == want["cents"]
    # rejections
    with pytest.raises(pkg.GdgError, match="at least this shard's"):
        ctxs[1].batch_run_shard(case.inputs[1:3], rate, "lpcm16", job_samples=BLOCK)
    for ctx in ctxs:
        ctx.close()


def test_batch_run_rejections_and_empty_batch():
    pkg = package()
    ctx = pkg.Context(2, BLOCK)
    data = np.zeros(200, dtype=np.uint8)
    with pytest.raises(pkg.GdgError, match="the batch has 1 inputs, the context 2 channels"):
        ctx.batch_run([(data, "lpcm16", 48000)], 48000, "lpcm16")
    with pytest.raises(pkg.GdgError, match="channel 2 of 2"):
        ctx.batch_run([(data, "lpcm16", 48000, 2, 2), None], 48000, "lpcm16")
    with pytest.raises(pkg.GdgError, match="level meters: 0 ports configured, the batch needs 2 N \\+ 3 = 7"):
        ctx.batch_run([(data, "lpcm16", 48000), None], 48000, "lpcm16", run_meters=True)
    outs = ctx.batch_run([None, None], 48000, "lpcm16")              # every channel left empty: outputs of 0 samples
    assert [o.size for o in outs] == [0] * 5
    outs = ctx.batch_run([(data, "lpcm16", 48000), None], 48000, "lpcm16")      # 100 samples -> one block, chains empty = pass-through
    assert all(o.size == 2 * BLOCK for o in outs)
    np.testing.assert_array_equal(outs[0], np.zeros(2 * BLOCK, dtype=np.uint8))
    ctx.close()
    small = pkg.Context(2, 1024)
    with pytest.raises(pkg.GdgError, match="blocks of 8192 frames"):
        small.batch_run([(data, "lpcm16", 48000), None], 48000, "lpcm16")
    small.close()


def test_batch_run_streams_every_format_bit_exact(oracle):
    """Mono inputs at the target rate go up step by step beside the block loop and are decoded piece-wise (wave_decode_rows_kernel):
    with empty chains the float64 outputs are the decoded files, bit for bit, whatever the format, the file length and the window."""
    pkg = package()
    rate = 44100
    lengths = [3 * BLOCK + 17, 1, BLOCK, 5 * BLOCK - 1, 2 * BLOCK + 4096, 7]
    files, want = [], []
    for c, fmt in enumerate(FORMATS):
        data = oracle.wave_encode(fmt, 0.9 * synth_signal(50 + c, lengths[c], rate))
        files.append((data, fmt, rate))
        want.append(oracle.wave_decode(fmt, data))
    total = 5 * BLOCK
    for W in (1, 2, 4):
        ctx = pkg.Context(len(FORMATS), BLOCK)
        ctx.set_window(W)
        outs = ctx.batch_run(files, rate, "ieee64")
        ctx.close()
        for c in range(len(FORMATS)):
            got = outs[c].view(np.float64)
            assert got.size == total
            assert np.array_equal(got[:lengths[c]], want[c]), (W, FORMATS[c])
            assert not got[lengths[c]:].any(), (W, FORMATS[c])


@pytest.mark.parametrize("W,blocks", [(8, 31), (16, 50), (4, 14)])
def test_long_batch_run_in_windows_gives_the_bytes_of_the_block_by_block_run(oracle, W, blocks):
    """A run long enough for everything the batch loop does to keep the device busy (api_batch.cpp): it opens with a quarter and a half window,
    the inputs of step i + 3 are gathered on a helper thread while step i is scattered, every step comes down in four pieces with the scatter
    behind each, the tail runs in windows of W/2 .. 1.  None of that may change a byte: the N + 3 output files, the meters and the tuners are
    those of the same context run block by block (W = 1: the reference's loop, compared with the oracle in the tests above).  Files of
    different formats and lengths (zero padding from different steps on), metronome in the master mix."""
    pkg = package()
    rate, nch = 48000, 7                                                   # 7 + 3 rows: the four-piece download needs >= 8
    rng = np.random.default_rng(100 + W)
    irs = [synth_ir(1500, seed=60 + c) for c in range(nch)]
    fmts = ["lpcm16", "lpcm24", "ieee64", "lpcm8", "lpcm32", "ieee32", "lpcm16"]
    lengths = [blocks * BLOCK, blocks * BLOCK - 5, (blocks - 3) * BLOCK + 100, 2 * BLOCK + 1, blocks * BLOCK - BLOCK // 2, 9 * BLOCK, blocks * BLOCK]
    files = []
    for c in range(nch):
        x = 0.6 * synth_signal(70 + c, lengths[c], rate)
        files.append((oracle.wave_encode(fmts[c], x), fmts[c], rate))
    tick, tock = rng.uniform(-0.5, 0.5, 900), rng.uniform(-0.5, 0.5, 500)

    def run(window):
        ctx = pkg.Context(nch, BLOCK)
        for c in range(nch):
            for name, p in CHAIN:
                ctx.append_unit(c, name, fir=irs[c]) if p == "ir" else ctx.append_unit(c, name, params=p)
        ctx.spatializer_set_sample_rate(rate)
        for c in range(nch):
            ctx.spatializer_set_position(c, 25.0 * (c - 3), 0.5 + 0.7 * c, 0.3 + 0.1 * c)
        ctx.metronome_set_sounds(tick, tock)
        ctx.metronome_configure(4, 170, rate)
        ctx.meter_configure(2 * nch + 3)
        ctx.meter_set_enabled(True)
        ctx.set_window(window)
        outs = [ctx.batch_run(files, rate, "lpcm24", metronome_to_master=True, run_meters=True, tuner_enqueue=True)]
        outs.append(ctx.batch_run(files, rate, "lpcm24", metronome_to_master=True, run_meters=False, tuner_enqueue=False))   # state carries on
        meters, tuned = ctx.meter_analyze(), ctx.tuner_analyze()
        # ... and as ONE SHARD of a larger job (its partial master mix and the metronome leave the device as float64 behind the last piece)
        shard = ctx.batch_run_shard(files, rate, "lpcm16", metronome=True)
        ctx.close()
        return outs, meters, tuned, shard

    (a1, a2), am, at, ash = run(W)
    (b1, b2), bm, bt, bsh = run(1)
    for r in range(nch):
        np.testing.assert_array_equal(ash[0][r], bsh[0][r], err_msg="shard output %d" % r)
    for k in (1, 2, 3, 4):                                                 # partial left, right, metronome bytes, metronome float64
        np.testing.assert_array_equal(ash[k], bsh[k], err_msg="shard part %d" % k)
    assert ash[1].any() and ash[3].any() and ash[4].any()
    assert len(a1) == nch + 3
    for got, want in ((a1, b1), (a2, b2)):
        for r in range(nch + 3):
            assert got[r].size == blocks * BLOCK * 3
            np.testing.assert_array_equal(got[r], want[r], err_msg="output %d" % r)
    assert any(a1[r].any() for r in range(nch)) and a1[nch].any() and a1[nch + 2].any()
    assert all(np.array_equal(x, y) for x, y in zip(am, bm))
    assert [(t["note_index"], t["cents"]) for t in at] == [(t["note_index"], t["cents"]) for t in bt]
    assert np.array_equal([t["frequency"] for t in at], [t["frequency"] for t in bt], equal_nan=True)      # the silent channel has no pitch


def test_batch_release_and_reuse():
    """The batch run keeps its device buffers between runs (larger, then smaller batches reuse them); gdg_batch_release frees them and the
    next run allocates again.  Same bytes every time."""
    pkg = package()
    ctx = pkg.Context(2, BLOCK)
    rng = np.random.default_rng(8)
    long_file = rng.integers(-9000, 9000, 3 * BLOCK, dtype=np.int16).view(np.uint8)
    short_file = long_file[:2 * 5000].copy()
    a = ctx.batch_run([(long_file, "lpcm16", 48000), None], 48000, "lpcm16")
    b = ctx.batch_run([(short_file, "lpcm16", 48000), None], 48000, "lpcm16")
    ctx.batch_release()
    c = ctx.batch_run([(long_file, "lpcm16", 48000), None], 48000, "lpcm16")
    ctx.close()
    assert a[0].size == long_file.size and a[0].any()
    np.testing.assert_array_equal(c[0], a[0])                      # after the release: the same bytes
    np.testing.assert_array_equal(b[0][:short_file.size], a[0][:short_file.size])      # the smaller batch in the larger buffers
    assert b[0].size == 2 * BLOCK and not b[0][short_file.size:].any()
