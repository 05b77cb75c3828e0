"""The N > 1 path of bench.py on CPU: two processes, gloo backend, 127.0.0.1 rendez-vous.
Checks the sharding (no channel lost or duplicated), the barrier + max-over-ranks timing and the
whole-job aggregation; plus the host-side combination of per-shard spatializer partials (oracle only)."""
import os
import socket
import time

import numpy as np
import pytest

import __graft_entry__ as entry
from helpers import synth_signal


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, q):
    import torch.distributed as dist
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    entry.load_package()
    from go_dsp_guitar_amd import shard
    start, count = shard.channel_shard(512, world, rank)
    sleep = 0.01 * (1 + 2 * rank)                       # rank 1 is three times slower: the job time is ITS time
    elapsed = shard.timed_steps(lambda: time.sleep(sleep), 5, lambda: None, dist)
    # the sharded batch run's one exchange: float64 partial master mixes to rank 0's host, in rank order
    n = 3 * 8192
    left = np.full(n, 1.0 + rank) * np.linspace(0.0, 1.0, n)
    right = -left
    lefts, rights = shard.gather_master_partials(left, right, dist, dst=0)
    if rank == 0:
        gathered = bool(len(lefts) == world and all(np.array_equal(lefts[r], np.full(n, 1.0 + r) * np.linspace(0.0, 1.0, n)) for r in range(world))
                        and all(np.array_equal(rights[r], -lefts[r]) for r in range(world)))
    else:
        gathered = lefts is None and rights is None
    q.put((rank, start, count, elapsed, gathered))
    dist.destroy_process_group()


def test_two_rank_gloo_timing_and_sharding():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    (r0, s0, c0, e0, g0), (r1, s1, c1, e1, g1) = res
    assert g0 and g1                                     # partial master mixes: gathered on rank 0 in rank order, nowhere else
    assert (s0, c0, s1, c1) == (0, 256, 256, 256)
    assert e0 == e1                                      # every rank reports the max over ranks
    assert 0.15 <= e0 < 1.0                              # >= 5 x 30 ms of the slow rank
    entry.load_package()
    from go_dsp_guitar_amd import shard
    assert shard.aggregate_throughput(256 * 8192, 2, 5, e0) == pytest.approx(2 * 256 * 8192 * 5 / e0)


def test_shards_partition_any_channel_count():
    entry.load_package()
    from go_dsp_guitar_amd import shard
    for total in (1, 7, 8, 64, 256, 512, 513):
        for world in (1, 2, 4, 8):
            blocks = [shard.channel_shard(total, world, r) for r in range(world)]
            covered = [c for s, n in blocks for c in range(s, s + n)]
            assert covered == list(range(total))


def test_spatializer_partials_add_up_to_the_full_mix(oracle):
    """8 shards of 4 channels each, mixed separately and added on the host, equal one 32-channel spatializer."""
    entry.load_package()
    from go_dsp_guitar_amd import shard
    nch, n, sr, world = 32, 2048, 96000, 8
    rng = np.random.default_rng(5)
    x = np.stack([synth_signal(c, n, sr) for c in range(nch)])
    pos = [(float(rng.uniform(-180, 180)), float(rng.uniform(0, 10)), float(rng.uniform(0, 1))) for _ in range(nch)]
    aux = 0.1 * synth_signal(99, n, sr)
    full = oracle.Spatializer(nch)
    for c, (a, d, l) in enumerate(pos):
        full.set_azimuth(c, a); full.set_distance(c, d); full.set_level(c, l)
    want_l, want_r = full.process(x, aux=aux)
    partials = []
    for r in range(world):
        s, cnt = shard.channel_shard(nch, world, r)
        sp = oracle.Spatializer(cnt)
        for i in range(cnt):
            a, d, l = pos[s + i]
            sp.set_azimuth(i, a); sp.set_distance(i, d); sp.set_level(i, l)
        partials.append(sp.process(x[s:s + cnt]))
    got_l, got_r = shard.combine_spatializer_partials(partials, aux)
    np.testing.assert_allclose(got_l, want_l, rtol=0, atol=1e-13)
    np.testing.assert_allclose(got_r, want_r, rtol=0, atol=1e-13)


# ---- `bench.py --gpus N` must become N ranks however it is started (round-3 review: a plain `python bench.py --gpus 8` ran one rank) ----

def test_launch_plan_decides_run_spawn_or_error():
    entry.load_package()
    from go_dsp_guitar_amd import shard
    assert shard.launch_plan(1, {}) == "run"
    assert shard.launch_plan(8, {}) == "spawn"                                   # no launcher: re-exec under torch.distributed.run
    assert shard.launch_plan(2, {"WORLD_SIZE": ""}) == "spawn"
    assert shard.launch_plan(8, {"WORLD_SIZE": "8", "RANK": "3", "LOCAL_RANK": "3"}) == "run"
    assert shard.launch_plan(1, {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}) == "run"
    assert shard.launch_plan(1, {"WORLD_SIZE": "1"}) == "run"                     # a lone WORLD_SIZE=1: a single process
    for gpus, env in ((8, {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}), (1, {"WORLD_SIZE": "8", "RANK": "0", "LOCAL_RANK": "0"}),
                      (2, {"WORLD_SIZE": "two"}), (0, {}), (4, {"WORLD_SIZE": "4"})):
        with pytest.raises(shard.LaunchError) as e:
            shard.launch_plan(gpus, env)
        assert e.value.code == 2 and e.value.msg
    cmd = shard.spawn_command("/usr/bin/python3", "/x/bench.py", ["--gpus", "4", "--steps", "7"], 4, 29511)
    assert cmd[:3] == ["/usr/bin/python3", "-m", "torch.distributed.run"]
    assert "--nproc-per-node" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29511"
    assert cmd[-5:] == ["/x/bench.py", "--gpus", "4", "--steps", "7"]
    assert shard.pick_device(8, 5, 8, False) == 5                                # every rank sees the node's eight devices
    assert shard.pick_device(8, 5, 1, False) == 0                                # the launcher gave every rank ONE visible device
    assert shard.pick_device(2, 1, 1, True) == 0                                 # harness self-test: every rank on device 0
    with pytest.raises(shard.LaunchError):
        shard.pick_device(8, 5, 4, False)                                         # neither shape
    with pytest.raises(shard.LaunchError):
        shard.pick_device(2, 0, 0, True)
    shard.check_distinct(["0000:05:00.0", "0000:15:00.0"], False)
    shard.check_distinct(["0000:05:00.0", "0000:05:00.0"], True)
    shard.check_distinct([None, None], False)                                     # ids unknown: nothing to say
    with pytest.raises(shard.LaunchError) as e:
        shard.check_distinct(["0000:05:00.0", "0000:05:00.0"], False)             # two ranks, one GPU: not a 2-GPU measurement
    assert "share 1 device" in e.value.msg


def _bench(args, env_extra, timeout=180):
    import subprocess
    import sys
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env.update(env_extra)
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    return subprocess.run([sys.executable, os.path.join(root, "bench.py")] + args, env=env, capture_output=True, text=True, timeout=timeout)


def test_bench_refuses_a_launcher_whose_world_size_is_not_gpus():
    r = _bench(["--gpus", "8"], {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and "WORLD_SIZE=1" in r.stderr and r.stdout.strip() == ""
    r = _bench(["--gpus", "1"], {"WORLD_SIZE": "2", "RANK": "0", "LOCAL_RANK": "0"})
    assert r.returncode == 2 and r.stdout.strip() == ""


def test_plain_python_bench_gpus_2_spawns_two_ranks():
    """No launcher, --gpus 2: the process re-executes itself under torch.distributed.run; here (no GPU) both ranks then stop at the device
    check -- with a non-zero exit and WITHOUT a JSON line (on a GPU box the same command prints "n_gpus": 2)."""
    import torch
    if torch.cuda.is_available():
        pytest.skip("the GPU variant of this test is tests/test_gpu_bench_launch.py")
    r = _bench(["--gpus", "2", "--steps", "1", "--warmup", "1", "--no-extras", "--no-cpu-baseline", "--no-parity"], {})
    assert r.returncode != 0
    assert "torch.distributed.run" in r.stderr and "--nproc-per-node 2" in r.stderr
    assert "rank 0 of 2 started" in r.stderr and "rank 1 of 2 started" in r.stderr, r.stderr[-2000:]      # BOTH ranks ran
    assert r.stderr.count("no HIP device visible") >= 1, r.stderr[-2000:]    # (the launcher may kill the second rank before it says so too)
    assert '"n_gpus"' not in r.stdout


def test_job_shape_defaults_to_the_strong_split_of_the_fixed_job():
    """round-5 review: `bench.py --gpus 8` ran 8 x 512 channels (weak scaling); BASELINE config 4 is 512 channels SPLIT over the GPUs"""
    entry.load_package()
    from go_dsp_guitar_amd import shard
    assert shard.job_shape(1, 0, 512) == {"scaling": "strong", "total_channels": 512, "channel0": 0, "channels_per_gpu": 512}
    blocks = [shard.job_shape(8, r, 512) for r in range(8)]
    assert all(b["scaling"] == "strong" and b["total_channels"] == 512 and b["channels_per_gpu"] == 64 for b in blocks)
    assert [b["channel0"] for b in blocks] == list(range(0, 512, 64))
    assert shard.job_shape(4, 3, 512, weak=True) == {"scaling": "weak", "total_channels": 2048, "channel0": 1536, "channels_per_gpu": 512}
    assert shard.job_shape(4, 3, 512, total_channels=0)["scaling"] == "weak"
    assert shard.job_shape(2, 1, 512, total_channels=100) == {"scaling": "strong", "total_channels": 100, "channel0": 50, "channels_per_gpu": 50}
    with pytest.raises(shard.LaunchError):
        shard.job_shape(8, 0, 4)


def test_two_rank_bench_reports_the_strong_split_of_512_channels():
    """`python bench.py --gpus 2 --plan-only`: the real launch path (self re-exec under torch.distributed.run, gloo group, the ranks' own
    job_shape, rank 0's line) without touching a device -- the line a 2-GPU SCALE run would carry says strong / 512 / 256"""
    import json
    r = _bench(["--gpus", "2", "--plan-only", "--steps", "20", "--warmup", "5"], {})
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["scaling"] == "strong"
    assert line["config"]["total_channels"] == 512 and line["config"]["channels_per_gpu"] == 256
    assert line["config"]["channel_blocks"] == [[0, 256], [256, 256]]
    assert "512ch@192kHz" in line["metric"]
    r = _bench(["--gpus", "2", "--plan-only", "--weak"], {})
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][0])
    assert line["scaling"] == "weak" and line["config"]["total_channels"] == 1024 and line["config"]["channels_per_gpu"] == 512
