#!/usr/bin/env python3
"""Soak of the in-launch hand-offs (seg.hip WAVE, os_tiles_kernel incl. its absorbed compressor, the premac, and round 6's: a frame on two workgroups
with the scans' carries through HBM granules, the chorus's tile hand-off, the reverbs' wet paths as extra workgroups) under UNEVEN load: the same streams through a context that uses them
and one that does not (walk, in-segment shaper, no premac), bit for bit, window after window and call after call, while a third context keeps the
chip busy from another thread with 512-channel steps (HBM streams, both segment kernels) -- the situation in which a missing release or acquire
shows (MI355X_MICROARCH.md: idle chips and L1-cold consumers hide such failures).

    SECONDS=20 python profiles/probes/wave_soak.py
"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import numpy as np
import bench
import __graft_entry__ as entry
pkg = entry.load_package()
frames, sr, taps = 8192, 192000, 65536
seconds = float(os.environ.get("SECONDS", "20"))
rng = np.random.default_rng(11)
stop = False


def load():
    ctx = bench.make_context(pkg, 512, frames, 0, taps, n_distinct=8)
    d_in, d_out = ctx.alloc(512, frames), ctx.alloc(512, frames)
    d_in.upload(bench.synth_block(512, frames, sr))
    n = 0
    while not stop:
        for _ in range(8):
            ctx.process_device(d_in, d_out, frames, sr)
        ctx.synchronize()
        n += 8
        if n % 64 == 0:
            time.sleep(0.002 * (n // 64 % 5))            # uneven: bursts and pauses
    ctx.close()
    print("load thread: %d steps of 512 channels beside the soak" % n, flush=True)


def pair(nch, chain, W):
    out = []
    for fancy in (True, False):
        ctx = bench.make_context(pkg, nch, frames, 0, taps, chain=chain, n_distinct=4)
        if not fancy:
            ctx.set_option("seg_wave_max_channels", 0)
            ctx.set_option("seg_os_tiles_max_channels", 0)
            ctx.set_option("fir_premac", 0)
            ctx.set_option("seg_tile_max_channels", 0)
            ctx.set_option("seg_reverb_ahead_max_channels", 0)
            ctx.set_option("seg_os_tiles_prefix", 0)
        if W > 1:
            ctx.set_window(W)
        out.append((ctx, ctx.alloc(nch, W * frames), ctx.alloc(nch, W * frames)))
    return out


chain_os = [(n, ([0, 20, 100, 0, 1, 2] if n == "overdrive" else p)) for n, p in bench.CHAIN]
cases = [("64 ch, W = 16, bench chain", pair(64, bench.CHAIN, 16), 64, 16),
         ("160 ch, W = 16, bench chain (two-per-CU build)", pair(160, bench.CHAIN, 16), 160, 16),
         ("48 ch, W = 8, 4x oversampled overdrive", pair(48, chain_os, 8), 48, 8),
         ("64 ch, per-frame calls (premac, two workgroups per frame, reverbs beside the first segment)", pair(64, bench.CHAIN, 1), 64, 1),
         ("24 ch, per-frame calls, bench chain", pair(24, bench.CHAIN, 1), 24, 1),
         ("64 ch, per-frame calls, 4x oversampled overdrive behind a compressor (absorbed into the tiles' launch)", pair(64, chain_os, 1), 64, 1)]
th = threading.Thread(target=load)
th.start()
t0, rounds, bad = time.time(), 0, 0
while time.time() - t0 < seconds:
    for name, ((a, ai, ao), (b, bi, bo)), nch, W in cases:
        x = 0.8 * rng.uniform(-1, 1, (nch, W * frames))
        ai.upload(x); bi.upload(x)
        reps = 1 if W > 1 else 6
        for r in range(reps):
            if W > 1:
                a.process_window_device(ai.ptr, ao.ptr, W * frames, W, sr)
                b.process_window_device(bi.ptr, bo.ptr, W * frames, W, sr)
            else:
                a.process_device(ai, ao, frames, sr)
                b.process_device(bi, bo, frames, sr)
        ga, gb = ao.download(), bo.download()
        if not np.array_equal(ga, gb):
            bad += 1
            d = np.abs(ga - gb)
            print("MISMATCH round %d, %s: max %.3e in channel %d" % (rounds, name, d.max(), int(np.argmax(d.max(axis=1)))), flush=True)
    rounds += 1
stop = True
th.join()
for _, ((a, ai, ao), (b, bi, bo)), _, _ in cases:
    a.close(); b.close()
print("%d rounds of %d cases in %.0f s under load: %d mismatches" % (rounds, len(cases), time.time() - t0, bad))
sys.exit(1 if bad else 0)
