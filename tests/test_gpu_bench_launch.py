"""`bench.py --gpus N` started WITHOUT a launcher must run N ranks (round-3 review: it ran one and printed "n_gpus": 1).  On the one-GPU box
both ranks share device 0 (GDG_BENCH_ONE_DEVICE=1): what is tested is the launch path, the rank bookkeeping and the line, not scaling."""
import json
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_plain_python_bench_gpus_2_prints_a_two_rank_line():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT")}
    env["GDG_BENCH_ONE_DEVICE"] = "1"
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "4", "--warmup", "1", "--channels", "64",
                        "--no-extras", "--no-cpu-baseline"], env=env, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, r.stdout[-2000:]                      # rank 0 alone prints
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["n_gpus_requested"] == 2
    assert [d["rank"] for d in line["devices"]] == [0, 1] and all(d["pci_bus_id"] for d in line["devices"])
    # the default job is FIXED (the reference's `-channels T`, controller.go:3262-3269): --channels 64 over two ranks = 32 each, strong scaling
    assert line["scaling"] == "strong" and line["config"]["total_channels"] == 64 and line["config"]["channels_per_gpu"] == 32
    assert line["settled"]["steps"] == 40 and line["settled"]["value"] > 0          # --steps 4: the settled figure rides on the same line
    assert line["parity"]["ok"] and line["parity"]["rms_max_all_ranks"] <= 1e-9
    assert line["value"] > 0 and line["roofline"]["frac"] is not None


@pytest.mark.gpu
def test_two_ranks_on_one_gpu_are_refused_unless_asked_for():
    """Without GDG_BENCH_ONE_DEVICE the same command on a one-GPU box must NOT print a line that says "n_gpus": 2: the ranks exchange the PCI bus ids
    of their devices and stop when they are not distinct (on a box with two or more GPUs the ranks get a device each and the run succeeds)."""
    import torch
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT", "GDG_BENCH_ONE_DEVICE")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--channels", "8", "--taps", "8192",
                        "--no-extras", "--no-cpu-baseline", "--no-parity"], env=env, capture_output=True, text=True, timeout=900)
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    if torch.cuda.device_count() >= 2:
        assert r.returncode == 0 and len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2
        assert len({d["pci_bus_id"] for d in json.loads(lines[0])["devices"]}) == 2
    else:
        assert r.returncode != 0 and not lines, r.stdout[-500:]
        assert "share 1 device" in r.stderr
