#!/usr/bin/env python3
"""Per-unit cost inside the fused segment kernel: 512 channels @ 192 kHz, 8192-frame blocks, one unit per
chain, device-resident frames, HIP-event time of seg_kernel.  Run on the GPU box:

    python profiles/seg_breakdown.py [unit ...] > gpurun_out/seg_breakdown.txt
"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import numpy as np  # noqa: E402

import __graft_entry__ as entry  # noqa: E402
from helpers import synth_signal  # noqa: E402

CASES = [
    ("(copy)", []),
    ("compressor", [("compressor", None)]),
    ("overdrive", [("overdrive", [0, 20, 100, 0, 1, 0])]),
    ("overdrive 4x", [("overdrive", [0, 20, 100, 0, 1, 2])]),
    ("tone_stack", [("tone_stack", None)]),
    ("chorus", [("chorus", None)]),
    ("cabinet", [("cabinet", None)]),
    ("reverb", [("reverb", None)]),
    ("flanger", [("flanger", None)]),
    ("delay", [("delay", None)]),
    ("noise_gate", [("noise_gate", None)]),
    ("octaver", [("octaver", None)]),
    ("auto_wah", [("auto_wah", None)]),
    ("fuzz", [("fuzz", None)]),
    ("fuzz 4x", [("fuzz", [1, 50, 0, 20, 100, 0, 2])]),
    ("bandpass 8", [("bandpass", [3, 300, 3000])]),
    ("seg0 of bench", [("compressor", None), ("overdrive", [0, 20, 100, 0, 1, 0]), ("tone_stack", None), ("chorus", None)]),
    ("seg1 of bench", [("cabinet", None), ("reverb", None)]),
]


def main():
    pkg = entry.load_package()
    nch, frames, sr, steps = int(os.environ.get("NCH", "512")), 8192, 192000, 10
    args = [a for a in sys.argv[1:] if a != "--cold" and not a.startswith("--window=") and not a.startswith("--pad=")]
    pad = max([int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("--pad=")] + [0])      # window mode: extra doubles per row (row stride = W * frames + pad)
    W = max([int(a.split("=")[1]) for a in sys.argv[1:] if a.startswith("--window=")] + [0])      # > 0: gdg_process_window_device, W frames per launch
    cold = "--cold" in sys.argv[1:]          # evict L2 / MALL between launches (as after the 1 GB MAC stream in the bench)
    only = set(args)
    x = np.stack([synth_signal(c, frames, sr) for c in range(nch)])
    print("%-16s %10s %12s %12s%s" % ("chain", "avg_us", "Msamples/s", "GB/s@16B", "   (cold caches)" if cold else ""))
    ctx_flush = pkg.Context(1, 64) if cold else None
    flush = None
    for name, chain in CASES:
        if only and name.split()[0] not in only and name not in only:
            continue
        ctx = pkg.Context(nch, frames)
        for c in range(nch):
            for unit, params in chain:
                ctx.append_unit(c, unit, params=params)
        if W:
            ctx.set_window(W)
            ctx.set_overlap(1)
            stride = W * frames + pad
            d_in, d_out = ctx.alloc(nch, stride), ctx.alloc(nch, stride)
            d_in.upload(np.concatenate([np.tile(x, (1, W)), np.zeros((nch, pad))], axis=1))
            for _ in range(2):
                ctx.process_window_device(d_in.ptr, d_out.ptr, stride, W, sr)
            ctx.synchronize()
            ctx.profile_enable(True)
            for _ in range(3):
                ctx.process_window_device(d_in.ptr, d_out.ptr, stride, W, sr)
            ctx.synchronize()
            ms, n = ctx.profile_read(pkg.K_SEGMENT)
            us = ms / n * 1e3 / W
            print("%-16s %10.1f %12.0f %12.0f   per frame of a window of %d" % (name, us, nch * frames / us, nch * frames * 16 / us / 1e3, W), flush=True)
            ctx.close()
            continue
        d_in, d_out = ctx.alloc(nch, frames), ctx.alloc(nch, frames)
        d_in.upload(x)
        for _ in range(2):
            ctx.process_device(d_in, d_out, frames, sr)
        ctx.synchronize()
        if cold and flush is None:
            n_flush = 1 << 27
            flush = (ctx_flush.alloc(1, n_flush), ctx_flush.alloc(1, n_flush), n_flush)
        ctx.profile_enable(True)
        for _ in range(steps):
            if cold:
                ctx.synchronize()
                ctx_flush._check(pkg.lib().gdg_wave_decode_device(ctx_flush._h, 5, flush[0].ptr, flush[2], 1, flush[1].ptr))
                ctx_flush.synchronize()
            ctx.process_device(d_in, d_out, frames, sr)
        ctx.synchronize()
        ms, n = ctx.profile_read(pkg.K_SEGMENT)
        us = ms / n * 1e3
        print("%-16s %10.1f %12.0f %12.0f" % (name, us, nch * frames / us, nch * frames * 16 / us / 1e3))
        ctx.close()


if __name__ == "__main__":
    main()
