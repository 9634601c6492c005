"""Data-parallel exchange step of the GAN update: one process per GPU, the flat gradient vector
of the net being updated is all-reduced (sum) with torch.distributed -- backend "nccl" is RCCL over xGMI on the
GPU box, "gloo" in the CPU tests -- and averaged by the 1/world factor folded into the fused optimizer pass.
The reference has no multi-GPU path (single process, cutorch.setDevice, train.lua:79); this is new functionality
required by BASELINE configs 3 and 5."""
import os

import torch
import torch.distributed as dist


def init(backend=None):
    """Initialise from the torchrun environment (RANK / LOCAL_RANK / WORLD_SIZE / MASTER_*).  Returns (rank, world)."""
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if world == 1:
        return 0, 1
    if not dist.is_initialized():
        backend = backend or ("nccl" if torch.cuda.is_available() else "gloo")
        kw = {}
        if backend == "nccl":
            os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
            kw["device_id"] = torch.device("cuda", int(os.environ.get("LOCAL_RANK", "0")))
        dist.init_process_group(backend, **kw)
    return dist.get_rank(), dist.get_world_size()


def allreduce_sum_(flat):
    """In-place SUM all-reduce of a flat gradient vector (no-op for a single process)."""
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    return flat


def grad_scale():
    """1/world: BCECriterion averages over the LOCAL batch, so the mean over the global batch is sum/world."""
    return 1.0 / (dist.get_world_size() if dist.is_initialized() else 1)


def shard(t, rank=None, world=None):
    """This rank's contiguous shard of a global batch tensor (host RNG draws the global batch once)."""
    rank = dist.get_rank() if rank is None else rank
    world = dist.get_world_size() if world is None else world
    n = t.shape[0] // world
    return t[rank * n:(rank + 1) * n]


def broadcast_(flat, src=0):
    if dist.is_initialized() and dist.get_world_size() > 1:
        dist.broadcast(flat, src)
    return flat
