"""Data-parallel exchange step of the GAN update: one process per GPU, the flat gradient vector of the
net being updated is sum-all-reduced over xGMI and averaged by the 1/world factor folded into the fused optimizer pass.
The reference has no multi-GPU path (single process, cutorch.setDevice, train.lua:79); this is new functionality required
by BASELINE configs 3 and 5.

Two carriers behind one small interface (`Collective`):
  * FgCollective   -- the library's own RCCL communicator behind the C ABI (fg_comm_create / fg_allreduce_sum[_async] /
                      fg_comm_wait, include/facegen_hip.h).  This is the path a Lua host uses too; only the 128-byte
                      bootstrap id travels through the host's own channel (here: a torch.distributed broadcast).
  * TorchCollective -- torch.distributed ("nccl" == RCCL on the GPU box, "gloo" in the CPU tests and in the two-process
                      one-GPU tests, where RCCL refuses two ranks on one device).
"""
import ctypes
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


class Collective:
    """What the trainers need from a carrier.  All calls are in place and ordered on the context's stream."""
    name = "none"
    world, rank = 1, 0
    fallback = False        # True: fg_comm was asked for and could not be used (make_collective fell back to torch.distributed)
    ranks_seen = None       # the self-test's sum of ones over the carrier actually used (== world when it works)

    def get_world_size(self):
        return self.world

    def get_rank(self):
        return self.rank

    def allreduce_sum_(self, t):
        """Blocking (stream-ordered) sum of `t` (fp32 / fp64 / int32) across ranks."""
        return t

    def allreduce_sum_async_(self, t):
        """Start the sum of fp32 `t` off the critical path; `wait()` orders every later launch after it."""
        return self.allreduce_sum_(t)

    def wait(self):
        pass

    def broadcast_(self, t, src=0):
        return t

    def describe(self):
        return self.name

    def close(self):
        pass

    # -- control-flow agreement (first-run insurance for N > 1: a branch that one rank takes alone is a hang on the hardware) --
    def _gather_int(self, v, what="int"):
        """-> list of every rank's int `v` (one int32 all-reduce of a one-hot vector: the carriers only sum)."""
        if self.world == 1:
            return [int(v)]
        t = torch.zeros(self.world, dtype=torch.int32, device=self._device())
        t[self.rank] = int(v)
        self.allreduce_sum_(t)
        return [int(x) for x in t.tolist()]

    def _device(self):
        return getattr(getattr(self, "ctx", None), "device", "cpu")

    def agree(self, flag):
        """True iff EVERY rank passed True.  The maxAccuracyD gate (adversarial.lua:167-178) is evaluated by each rank on the global
        confusion counts and must come out the same everywhere; if host arithmetic ever made one rank disagree, the ranks would
        issue different collectives from here on.  With the vote they all skip."""
        votes = self._gather_int(1 if flag else 0, "vote")
        return sum(votes) == self.world

    def min_int(self, v):
        """The smallest `v` over the ranks (iterations of an epoch: a tail batch only one rank has must not be trained alone)."""
        return min(self._gather_int(v, "min"))


class TorchCollective(Collective):
    def __init__(self, d=dist):
        self.d = d
        self.world, self.rank = d.get_world_size(), d.get_rank()
        self.name = "torch.distributed (%s)" % d.get_backend()
        self._works = []

    def _device(self):
        return torch.device("cuda", torch.cuda.current_device()) if self.d.get_backend() == "nccl" else "cpu"

    def allreduce_sum_(self, t):
        if self.world > 1:
            self.d.all_reduce(t, op=self.d.ReduceOp.SUM)
        return t

    def allreduce_sum_async_(self, t):
        if self.world > 1:
            self._works.append(self.d.all_reduce(t, op=self.d.ReduceOp.SUM, async_op=True))
        return t

    def wait(self):
        for w in self._works:
            w.wait()
        self._works = []

    def broadcast_(self, t, src=0):
        if self.world > 1:
            self.d.broadcast(t, src)
        return t


class FgCollective(Collective):
    """fg_comm_* of libfacegen_hip.so: RCCL bound by the library itself, exchange on the communicator's own stream."""

    def __init__(self, ctx, rank, world, id_bytes):
        self.ctx, self.lib = ctx, ctx.lib
        self.world, self.rank = world, rank
        h = ctypes.c_void_p()
        ctx.check(self.lib.fg_comm_create(ctx.h, id_bytes, len(id_bytes), rank, world, ctypes.byref(h)))
        self.h = h
        self.name = "fg_comm (RCCL via the C ABI: %s)" % self.lib.fg_comm_library().decode()

    @staticmethod
    def unique_id(ctx):
        buf = ctypes.create_string_buffer(128)
        ctx.check(ctx.lib.fg_comm_unique_id(ctx.h, buf, 128))
        return buf.raw

    def allreduce_sum_(self, t):
        fn = {torch.float32: self.lib.fg_allreduce_sum, torch.float64: self.lib.fg_allreduce_sum_f64,
              torch.int32: self.lib.fg_allreduce_sum_i32}[t.dtype]
        assert t.is_contiguous()
        self.ctx.check(fn(self.h, t.data_ptr(), t.numel()))
        return t

    def allreduce_sum_async_(self, t):
        assert t.dtype == torch.float32 and t.is_contiguous()
        self.ctx.check(self.lib.fg_allreduce_sum_async(self.h, t.data_ptr(), t.numel()))
        return t

    def wait(self):
        self.ctx.check(self.lib.fg_comm_wait(self.h))

    def broadcast_(self, t, src=0):
        assert t.dtype == torch.float32 and t.is_contiguous()
        self.ctx.check(self.lib.fg_broadcast(self.h, t.data_ptr(), t.numel(), src))
        return t

    def close(self):
        if self.h is not None:
            self.lib.fg_comm_destroy(self.h)
            self.h = None


class DryCollective(FgCollective):
    """fg_comm_create_dry: rank `rank` of a `world`-rank job with NO transport underneath -- every collective the step entries
    would issue is recorded (order, dtype, count, stream) and skipped.  With a planning-only context (runtime.get_context(-1))
    not even a GPU is needed: bench.py --dry-collective walks fg_step_D / fg_step_G's exchange path for every rank of an N-GPU
    job in one process and checks that all ranks issue the same schedule (a mismatch would be a hang on the hardware)."""

    def __init__(self, ctx, rank, world):
        self.ctx, self.lib = ctx, ctx.lib
        self.world, self.rank = world, rank
        h = ctypes.c_void_p()
        ctx.check(self.lib.fg_comm_create_dry(ctx.h, rank, world, ctypes.byref(h)))
        self.h = h
        self.name = "fg_comm (dry: schedule only, rank %d of %d)" % (rank, world)

    # no transport: the all-reduce of the one-hot vector is recorded in the schedule and skipped.  `peers` stands for what the other
    # ranks would have contributed (fault injection in the CPU tests): a callable (rank, my value, "vote" | "min") -> int, default "as me"
    peers = None

    def _device(self):
        return "cpu"

    def _gather_int(self, v, what="int"):
        t = torch.zeros(self.world, dtype=torch.int32)
        t[self.rank] = int(v)
        self.allreduce_sum_(t)                    # recorded: "<seq> allreduce i32 <world> compute"
        return [int(v) if (r == self.rank or self.peers is None) else int(self.peers(r, int(v), what)) for r in range(self.world)]

    def schedule(self, reset=True):
        """-> list of "<seq> <op> <dtype> <count> <stream>" lines since the last reset."""
        buf = ctypes.create_string_buffer(1 << 20)
        self.ctx.check(self.lib.fg_comm_schedule(self.h, buf, len(buf), 1 if reset else 0))
        return buf.value.decode().splitlines()


def as_collective(d):
    """Trainer argument -> Collective: None (single process), a Collective, or the torch.distributed module."""
    if d is None:
        return None
    if isinstance(d, Collective):
        return d
    return TorchCollective(d)


def _self_test(coll, ctx):
    """Sum of ones over the carrier: `ranks_seen` (reported by bench.py as rccl_ranks_seen)."""
    probe = torch.ones(4, dtype=torch.float32, device=ctx.device)
    coll.allreduce_sum_(probe)
    coll.ranks_seen = int(round(probe[0].item()))
    return coll


def make_collective(ctx, d=dist, prefer="fg_comm", strict=False):
    """The carrier for an initialised torch.distributed job: the library's own communicator unless `prefer` says torch.
    The 128-byte RCCL id is broadcast from rank 0 over `d`; a one-element self-test (sum of ones == world) runs on every
    rank, and unless `strict` a failure falls back to torch.distributed on ALL ranks (the decision is itself reduced)."""
    if prefer != "fg_comm":
        return _self_test(TorchCollective(d), ctx)
    rank, world = d.get_rank(), d.get_world_size()
    err, coll = "", None
    # (1) local, non-collective: can this rank bind librccl at all?  Decided jointly BEFORE the collective create, so a
    #     rank without RCCL cannot leave the others waiting inside ncclCommInitRank.
    try:
        my_id = FgCollective.unique_id(ctx)
    except Exception as e:        # noqa: BLE001
        my_id, err = None, str(e)
    bad = torch.tensor([1.0 if err else 0.0], device=ctx.device)
    d.all_reduce(bad, op=d.ReduceOp.SUM)
    if float(bad.item()) != 0.0:
        if strict:
            raise RuntimeError("fg_comm unavailable on %d rank(s): %s" % (int(bad.item()), err or "(another rank)"))
        t = TorchCollective(d)
        t.name += " [fg_comm fell back: %s]" % (err or "another rank cannot bind librccl")[:160]
        t.fallback = True
        _self_test(t, ctx)
        return t
    try:
        idt = torch.zeros(128, dtype=torch.uint8, device=ctx.device)
        if rank == 0:
            idt.copy_(torch.frombuffer(bytearray(my_id), dtype=torch.uint8))
        d.broadcast(idt, 0)
        coll = FgCollective(ctx, rank, world, bytes(idt.cpu().numpy().tobytes()))
        probe = torch.ones(4, dtype=torch.float32, device=ctx.device)
        coll.allreduce_sum_(probe)
        torch.cuda.synchronize()
        coll.ranks_seen = int(round(probe[0].item()))
        if probe.tolist() != [float(world)] * 4:
            raise RuntimeError("self-test: sum of ones = %s, expected %d" % (probe.tolist(), world))
    except Exception as e:        # noqa: BLE001 -- reported, and decided collectively below
        err = str(e)
    bad = torch.tensor([1.0 if err else 0.0], device=ctx.device)
    d.all_reduce(bad, op=d.ReduceOp.SUM)
    if float(bad.item()) == 0.0:
        return coll
    if strict:
        raise RuntimeError("fg_comm unavailable on %d rank(s): %s" % (int(bad.item()), err or "(another rank)"))
    if coll is not None:
        coll.close()
    t = TorchCollective(d)
    t.name += " [fg_comm fell back: %s]" % (err or "another rank failed")[:160]
    t.fallback = True
    _self_test(t, ctx)
    return t


# ---- helpers kept for the host loops -----------------------------------------------------------------------------
def allreduce_sum_(flat):
    """In-place SUM all-reduce of a flat gradient vector over torch.distributed (no-op for a single process)."""
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
