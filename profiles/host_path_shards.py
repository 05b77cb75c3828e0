"""The reference's deployment is G shards inside ONE process (controller/controller.go:3262-3269, :3333-3341): G contexts, one per GPU, each
driven by its own host thread at the same time.  This probe runs the 512-channel bench job as G = 1, 2, 4, 8 contexts -- on G devices when
the box has them, else all on device 0 -- through the host-buffer entry points (gdg_process: caller's pageable rows; gdg_process_staged:
pinned slab already written), one thread per context, all started together, and prints the job's rate.  Round 4 gave every context its
own copy pool (before: one process-wide pool that only ONE of the concurrently copying shards got; the others copied on their caller's
thread alone).      python profiles/host_path_shards.py > gpurun_out/host_path_shards_r04.txt"""
import ctypes as C
import os
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
import __graft_entry__ as ge  # noqa: E402

pkg = ge.load_package()
ge.load_package()
from go_dsp_guitar_amd import shard  # noqa: E402

lib = pkg.lib()
total, frames, sr, taps, steps = 512, 8192, 192000, 65536, 8
n_dev = lib.gdg_device_count()
print("# %d HIP device(s) visible; job = %d channels @ %d Hz, %d-frame blocks, 2 x %d-tap IRs per channel" % (n_dev, total, sr, frames, taps))
print("shards,devices,entry point,ms_per_block (slowest shard),job Msamples/s,GB/s over PCIe (in+out, all shards)")
for G in (1, 2, 4, 8):
    ctxs, bufs = [], []
    for g in range(G):
        c0, cnt = shard.channel_shard(total, G, g)
        dev = g % n_dev
        ctx = bench.make_context(pkg, cnt, frames, dev, taps, channel0=c0, n_distinct=8)
        x = bench.synth_block(cnt, frames, sr, channel0=c0)
        out = np.empty_like(x)
        ins = (C.c_void_p * cnt)(*[x[c].ctypes.data for c in range(cnt)])
        outs = (C.c_void_p * cnt)(*[out[c].ctypes.data for c in range(cnt)])
        carr = (C.c_int * cnt)(*range(cnt))
        ctx.process_staged(list(range(cnt)), x, sr)
        ctxs.append(ctx)
        bufs.append((x, out, ins, outs, carr, cnt))
    for name in ("gdg_process", "gdg_process_staged"):
        def work(g, n, res):
            ctx = ctxs[g]
            x, out, ins, outs, carr, cnt = bufs[g]
            t0 = time.perf_counter()
            for _ in range(n):
                if name == "gdg_process":
                    ctx._check(lib.gdg_process(ctx._h, ins, outs, frames, sr))
                else:
                    ctx._check(lib.gdg_process_staged(ctx._h, carr, cnt, frames, sr))
            res[g] = (time.perf_counter() - t0) / n

        def run(n):
            res = [0.0] * G
            ths = [threading.Thread(target=work, args=(g, n, res)) for g in range(G)]
            for t in ths:
                t.start()
            for t in ths:
                t.join()
            return max(res)
        run(2)
        dt = sorted(run(steps) for _ in range(3))[1]
        print("%d,%s,%s,%.3f,%.0f,%.1f" % (G, "+".join(str(g % n_dev) for g in range(G)), name, dt * 1e3, total * frames / dt / 1e6, 2 * total * frames * 8 / dt / 1e9))
    for ctx in ctxs:
        ctx.close()
