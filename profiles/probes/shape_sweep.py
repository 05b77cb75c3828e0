#!/usr/bin/env python3
"""Do the library's launch-shape thresholds (include/gdg.h, gdg_ctx_set_option) pick the faster shape on chains the bench never runs?

For four chains -- no reverb; 96 kHz with a two-partition filter and a 4 x oversampled overdrive; flanger + delay + octaver (units the
two-per-CU kernel does not run); the bench's own (two power amps per channel) -- and channel counts 32 .. 512, the library's DEFAULT is timed against every single option flipped to the
other side of its threshold (the shape it would take if the threshold were elsewhere):

    per-frame calls:  fir_split_max_channels, fir_premac, seg_two_per_cu_min_channels, seg_os_tiles_max_channels, seg_reverb_ahead_max_channels,
                      seg_tile_max_channels, seg_os_tiles_prefix
    windows of 16:    seg_wave_max_channels, seg_two_per_cu_min_channels, seg_os_tiles_max_channels

A cell is a VIOLATION when the default is more than TOL (5 %) slower than an alternative -- after the pair has been measured again (the
violation must persist).  `sweep()` is what tests/test_gpu_shape_sweep.py runs on a subset; run as a script it prints the whole table
(profiles/shape_sweep_r06.txt).

    python profiles/probes/shape_sweep.py [CHANNELS=32,64,...] [CHAINS=a,b,c]
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import bench  # noqa: E402
import __graft_entry__ as entry  # noqa: E402

TOL = 0.05
BIG = 1 << 20
CHAINS = {
    # name: (sample rate, taps, chain)
    "a_no_reverb": (192000, 65536, [("compressor", [1, 30, -20]), ("overdrive", [0, 20, 100, 0, 1, 0]), ("tone_stack", None), ("chorus", None),
                                    ("power_amp", "cab"), ("cabinet", None)]),
    "b_96k_K2_os4": (96000, 16384, [("compressor", [1, 30, -20]), ("overdrive", [0, 20, 100, 0, 1, 2]), ("tone_stack", None),
                                    ("power_amp", "cab"), ("cabinet", None), ("reverb", [50])]),
    "c_flanger_delay_octaver": (192000, 65536, [("flanger", None), ("delay", None), ("octaver", None), ("power_amp", "cab"), ("phaser", None)]),
    "d_bench_two_amps": (192000, 65536, list(bench.CHAIN)),          # the bench's own chain: TWO power amps per channel (the other limit of the split convolution)
}
DEFAULTS = {"fir_split_max_channels": 192, "fir_split_max_channels_one_amp": 112, "fir_premac": 1, "seg_two_per_cu_min_channels": 128,
            "seg_os_tiles_max_channels": 192, "seg_reverb_ahead_max_channels": 72, "seg_wave_max_channels": 448, "seg_wave_release_max_channels": 112,
            "seg_tile_max_channels": 112, "seg_os_tiles_prefix": 1}
RELEASE_UNITS = {"flanger", "phaser", "delay", "fuzz", "auto_yoy", "auto_wah", "bandpass", "octaver", "noise_gate"}


def flips(nch, window, defaults, chain):
    """(label, {option: value}) for every single launch-shape decision moved to the other side of its threshold at this channel count.  Where
    two options share a decision (the split convolution's limit with one amp per channel, the WAVE limit of segments that hand their state
    on through a write-back) both move together."""
    out = []
    amps = sum(1 for n, _ in chain if n == "power_amp")
    release = any(n in RELEASE_UNITS for n, _ in chain)

    def other(keys, thr, below_is_on):
        on = (nch <= thr) if below_is_on else (nch >= thr)
        v = (0 if on else BIG) if below_is_on else (BIG if on else 0)
        return {k: v for k in keys}
    if not window:
        split_keys = ["fir_split_max_channels"] + (["fir_split_max_channels_one_amp"] if amps < 2 else [])
        split_thr = min(defaults[k] for k in split_keys)
        out.append(("fir_split_max_channels", other(split_keys, split_thr, True)))
        if nch <= split_thr:
            out.append(("fir_premac", {"fir_premac": 0 if defaults["fir_premac"] else 1}))
        taps_parts = None
        out.append(("seg_reverb_ahead_max_channels", None))          # resolved by the caller (needs the plan's premac state)
        out.append(("seg_tile_max_channels", None))                  # resolved by the caller (the launch's workgroup budget)
        if nch <= defaults["seg_os_tiles_max_channels"] and any(n == "overdrive" and p and p[5] for n, p in chain) and chain[0][0] == "compressor":
            out.append(("seg_os_tiles_prefix", {"seg_os_tiles_prefix": 0}))       # the compressor in front of the shaper as a launch of its own again
    else:
        wave_keys = ["seg_wave_max_channels"] + (["seg_wave_release_max_channels"] if release else [])
        out.append(("seg_wave_max_channels", other(wave_keys, min(defaults[k] for k in wave_keys), True)))
    out.append(("seg_two_per_cu_min_channels", other(["seg_two_per_cu_min_channels"], defaults["seg_two_per_cu_min_channels"], False)))
    out.append(("seg_os_tiles_max_channels", other(["seg_os_tiles_max_channels"], defaults["seg_os_tiles_max_channels"], True)))
    return out


def sweep(pkg, chains, channels, modes=("frame", "window"), log=print, tol=TOL, device=0):
    """Returns (rows, violations): rows = (chain, nch, mode, label, us_default, us_flipped, ratio default / flipped)"""
    rows, violations = [], []
    frames, W = 8192, 16
    for cname in chains:
        sr, taps, chain = CHAINS[cname]
        for nch in channels:
            ctx = bench.make_context(pkg, nch, frames, device, taps, chain=chain, second_amp=any(p == "rev" for _, p in chain))
            defaults = {k: ctx.get_option(k) for k in DEFAULTS}
            d_in1, d_out1 = ctx.alloc(nch, frames), ctx.alloc(nch, frames)
            d_in1.upload(bench.synth_block(nch, frames, sr))
            blocks = 2 * W
            d_inw = d_outw = None

            def time_frame():
                def run():
                    for _ in range(20):
                        ctx.process_device(d_in1, d_out1, frames, sr)
                return bench.robust_time(run, ctx.synchronize, units=20, reps=5)["median"]

            def time_window():
                def run():
                    for b in range(0, blocks, W):
                        ctx.process_window_device(d_inw.ptr + 8 * b * frames, d_outw.ptr + 8 * b * frames, blocks * frames, W, sr)
                return bench.robust_time(run, ctx.synchronize, units=blocks, reps=3)["median"]

            for mode in modes:
                window = mode == "window"
                if window:
                    ctx.set_window(W)
                    d_inw, d_outw = ctx.alloc(nch, blocks * frames), ctx.alloc(nch, blocks * frames)
                    d_inw.upload(np.tile(bench.synth_block(nch, frames, sr), (1, blocks)))
                timer = time_window if window else time_frame

                def with_options(opts):
                    for k, v in opts.items():
                        ctx.set_option(k, v)
                    t = timer()
                    for k in opts:
                        ctx.set_option(k, defaults[k])
                    return t

                t_def = with_options({})
                for label, opts in flips(nch, window, defaults, chain):
                    has_os = any(n == "overdrive" and p and p[5] for n, p in chain)
                    has_rev = any(n == "reverb" for n, _ in chain)
                    if label == "seg_os_tiles_max_channels" and not has_os:
                        continue
                    if label == "seg_reverb_ahead_max_channels":
                        if not has_rev:
                            continue
                        # the limit in force: the option when the call also sums convolution terms ahead (premac), 127 otherwise (include/gdg.h)
                        K = -(-taps // frames)
                        amps = sum(1 for n, _ in chain if n == "power_amp")
                        split_thr = min(defaults["fir_split_max_channels"], defaults["fir_split_max_channels_one_amp"]) if amps < 2 else defaults["fir_split_max_channels"]
                        premac = defaults["fir_premac"] and nch <= split_thr and K >= 2 and nch * K >= (320 if amps >= 2 else 384)
                        thr = defaults[label] if premac else max(defaults[label], 127)
                        opts = {label: 0 if nch <= thr else BIG}
                    if label == "seg_tile_max_channels":
                        # tiled while 2 x channels + the hosted reverbs fit 224 workgroups (ctx.h GDG_TILE_WORKGROUP_BUDGET) and channels <= the option;
                        # the other side: the option 0.  (Where the budget says no, there is no option that forces it: nothing to flip.)
                        hosted = nch if (has_rev and nch <= 127) else 0
                        if nch > defaults[label] or 2 * nch + hosted > 224:
                            continue
                        opts = {label: 0}
                    t_alt = with_options(opts)
                    t_d = t_def
                    if t_d > (1.0 + tol) * t_alt:
                        # measure the pair again, twice: a violation must persist
                        for _ in range(2):
                            t_d = min(t_d, with_options({}))
                            t_alt = max(t_alt, with_options(opts))
                    ratio = t_d / t_alt
                    row = (cname, nch, mode, label, list(opts.values())[0], t_d * 1e6, t_alt * 1e6, ratio)
                    rows.append(row)
                    bad = ratio > 1.0 + tol
                    if bad:
                        violations.append(row)
                    log("%-24s %4d ch %-6s default %8.1f us | %-30s = %-8d %8.1f us | default / flipped %.3f%s"
                        % (cname, nch, mode, t_d * 1e6, label, row[4], t_alt * 1e6, ratio, "   <-- VIOLATION" if bad else ""))
                if window:
                    d_inw.free()
                    d_outw.free()
                    ctx.set_window(1)
            d_in1.free()
            d_out1.free()
            ctx.close()
    return rows, violations


if __name__ == "__main__":
    pkg = entry.load_package()
    channels = [int(v) for v in os.environ.get("CHANNELS", "32,64,96,128,192,256,448,512").split(",")]
    chains = [c for c in CHAINS if c[0] in os.environ.get("CHAINS", "a,b,c").split(",")]
    print("# default vs each launch-shape option flipped to the other side of its threshold; us per frame (per-frame calls / windows of 16)")
    rows, bad = sweep(pkg, chains, channels, modes=tuple(os.environ.get("MODES", "frame,window").split(",")))
    print("# %d cells, %d violations (default more than %.0f %% slower than an alternative, measured three times)" % (len(rows), len(bad), TOL * 100))
