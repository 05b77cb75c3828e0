"""Multi-GPU plumbing: channels are independent (controller/controller.go:3262-3269), so N GPUs are N
shards with NO data-path collective.  torch.distributed -- over gloo, on the GPU box as in the CPU tests: the job
contains no RCCL / xGMI traffic -- is used only for the barrier around the timed region, the max-over-ranks of the
elapsed time and, for the sharded batch run, the gather of the shards' partial master mixes on rank 0's host.
"""
import time

import numpy as np


def channel_shard(n_total, world, rank):
    """Contiguous block of channels of shard `rank`: [rank*N/world, (rank+1)*N/world)."""
    start = (rank * n_total) // world
    stop = ((rank + 1) * n_total) // world
    return start, stop - start


def job_shape(world, rank, channels, total_channels=-1, weak=False):
    """What `bench.py --gpus N` runs on rank `rank`.  The reference's `-channels T` is a FIXED job (controller/controller.go:3262-3269): the
    default is therefore the strong split of `channels` (BASELINE's 512) over the N GPUs in contiguous blocks -- config 4 at N = 8, the
    headline configuration itself at N = 1.  `weak` (or total_channels = 0) puts `channels` on EVERY GPU instead; total_channels > 0 names
    another fixed job.  Returns scaling, total_channels, channel0 and channels_per_gpu of this rank."""
    if total_channels == 0:
        weak = True
    if weak:
        return {"scaling": "weak", "total_channels": world * channels, "channel0": rank * channels, "channels_per_gpu": channels}
    total = channels if total_channels < 0 else total_channels
    if total < world:
        raise LaunchError("%d channels cannot be split over %d GPUs" % (total, world))
    start, count = channel_shard(total, world, rank)
    return {"scaling": "strong", "total_channels": total, "channel0": start, "channels_per_gpu": count}


def timed_steps(step, steps, synchronize, dist=None, device=None):
    """Barrier + synchronize, run `step` exactly `steps` times, synchronize; returns the MAX over ranks of the
    elapsed seconds (the same number on every rank)."""
    import torch
    if dist is not None:
        dist.barrier()
    synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    synchronize()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], dtype=torch.float64, device=device if device is not None else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
        dist.barrier()
    return elapsed


def aggregate_throughput(units_per_rank_step, world, steps, elapsed):
    """Whole-job units per second: every rank processed the same amount (weak scaling)."""
    return world * units_per_rank_step * steps / elapsed


def combine_spatializer_partials(partials, aux=None):
    """Host-side sum of the per-shard (left, right) pairs in shard order, then the aux input
    (spatializer/spatializer.go:300-310)."""
    left = np.zeros_like(partials[0][0])
    right = np.zeros_like(partials[0][1])
    for l, r in partials:
        left += l
        right += r
    if aux is not None:
        left += aux
        right += aux
    return left, right


def gather_master_partials(left, right, dist, dst=0):
    """The sharded batch run's only exchange: every rank's float64 partial master mix (left, right: numpy arrays of the job's length) to
    rank `dst`'s HOST over the control-plane group (gloo) -- SURVEY.md 8e: "the host adds the partials".  Returns (lefts, rights) in rank order
    on `dst` (to be handed to gdg_batch_finish_master, which adds them in that order, then the aux input, then encodes), (None, None) elsewhere."""
    import torch
    lt, rt = torch.from_numpy(np.ascontiguousarray(left)), torch.from_numpy(np.ascontiguousarray(right))
    world, rank = dist.get_world_size(), dist.get_rank()
    gl = [torch.empty_like(lt) for _ in range(world)] if rank == dst else None
    gr = [torch.empty_like(rt) for _ in range(world)] if rank == dst else None
    dist.gather(lt, gl, dst=dst)
    dist.gather(rt, gr, dst=dst)
    if rank != dst:
        return None, None
    return [g.numpy() for g in gl], [g.numpy() for g in gr]


# ---- how `bench.py --gpus N` becomes N ranks, however it is started -------------------------------------------------------------

class LaunchError(SystemExit):
    """--gpus and the launcher disagree: the run must not print a line for a job it did not run (exit code 2)."""

    def __init__(self, msg):
        super().__init__(2)
        self.msg = msg


def launch_plan(gpus, environ):
    """What a process started as `bench.py --gpus N` has to do:
      "run"    -- it IS the job: N = 1 without a launcher, or one of the N ranks of a launcher (WORLD_SIZE == N);
      "spawn"  -- N > 1 and no launcher (a plain `python bench.py --gpus N`): re-exec under torch.distributed.run with N ranks,
                  one device per rank -- silently running ONE rank and printing "n_gpus": 1 is the failure this guards against;
    LaunchError when a launcher is there and its WORLD_SIZE is not N (the line would carry another N than was asked for)."""
    if gpus < 1:
        raise LaunchError("--gpus %d: at least one" % gpus)
    world = environ.get("WORLD_SIZE")
    if world is None or world == "":
        return "run" if gpus == 1 else "spawn"
    try:
        world = int(world)
    except ValueError:
        raise LaunchError("WORLD_SIZE=%r is not a number" % world)
    if world != gpus:
        raise LaunchError("--gpus %d but the launcher started WORLD_SIZE=%d ranks" % (gpus, world))
    for key in ("RANK", "LOCAL_RANK"):
        if world > 1 and environ.get(key) in (None, ""):       # a lone WORLD_SIZE=1 is simply a single process
            raise LaunchError("WORLD_SIZE=%d is set but %s is not: not a torch.distributed.run environment" % (world, key))
    return "run"


def free_port():
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def spawn_command(python, script, argv, gpus, port):
    """The command line of the driver's own N > 1 launch (one node, one rank per GPU, rendezvous on 127.0.0.1)."""
    return [python, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus), "--master-addr", "127.0.0.1",
            "--master-port", str(port), script] + list(argv)


def pick_device(gpus, local_rank, visible, one_device):
    """The HIP device index of this rank.  Normally local_rank (every rank sees all the node's devices).  A launcher may instead give
    every rank ONE visible device of its own (HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES per rank): then it is device 0, and
    check_distinct() makes sure afterwards that the ranks really sit on different devices.  GDG_BENCH_ONE_DEVICE (harness self-test on
    a one-GPU box): every rank on device 0, on purpose."""
    if visible < 1:
        raise LaunchError("no HIP device visible")
    if one_device:
        return 0
    if local_rank < visible:
        return local_rank
    if visible == 1:
        return 0
    raise LaunchError("--gpus %d, local rank %d, but %d HIP devices are visible" % (gpus, local_rank, visible))


def check_distinct(pci_bus_ids, one_device):
    """After the ranks have exchanged the PCI bus ids of their devices: N ranks on fewer than N devices is not an N-GPU measurement
    (unless the harness self-test asked for exactly that)."""
    if one_device:
        return
    ids = [i for i in pci_bus_ids if i]
    if len(ids) == len(pci_bus_ids) and len(set(ids)) != len(ids):
        raise LaunchError("%d ranks share %d device(s) (%s): every rank needs a GPU of its own (GDG_BENCH_ONE_DEVICE=1 shares device 0 "
                          "for a harness self-test)" % (len(ids), len(set(ids)), ", ".join(sorted(set(ids)))))
