"""GPU-side check of the data-parallel code path with a world of ONE rank over RCCL (backend "nccl"): the exchange
step (all-reduce + 1/world + fused Adam) must be the identity on the single-GPU result, and bench.py must run under
torch.distributed.run.  (True multi-GPU runs belong to the driver; the N=2 logic is covered on CPU with gloo.)"""
import json
import os
import subprocess
import sys

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_trainer_with_rccl_world1_matches_plain_trainer():
    import torch.distributed as dist
    from face_generator_amd import models, nn_utils, adversarial
    from face_generator_amd.runtime import get_context
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT="29533", RANK="0", WORLD_SIZE="1", LOCAL_RANK="0")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    ctx = get_context(0)
    dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
    try:
        outs = []
        for use_dist in (False, True):
            gen = torch.Generator().manual_seed(3)
            G = models.create_G((3, 32, 32), 100); D = models.create_D((3, 32, 32))
            nn_utils.initializeWeights(D, 0.05, 0.01, gen=gen); nn_utils.initializeWeights(G, 0.05, 0.01, gen=gen)
            G.cuda(ctx, max_batch=8); D.cuda(ctx, max_batch=8)
            D.device_net.mask_seed = 5
            tr = adversarial.Trainer(ctx, G, D, dict(batchSize=8), dist=dist if use_dist else None)
            real = ctx.uniform((4, 32, 32, 3), 0.0, 1.0, seed=9)
            if use_dist:             # force the N > 1 code path (async all-reduce + deferred D update) on one rank:
                tr.world, tr.gscale = 2, 1.0     # a 1-rank sum all-reduce is the identity, so keep the scale at 1
            tr.step_D(real, ctx.uniform((4, 100), -1.0, 1.0, seed=10))
            assert (tr._pending_D is not None) == use_dist
            tr.step_G(ctx.uniform((8, 100), -1.0, 1.0, seed=11))
            assert tr._pending_D is None
            if use_dist:    # bucketed, backward-overlapped all-reduce of G: >= 2 buckets covering the whole flat vector
                bk = tr._buckets_G()
                assert len(bk) >= 2 and bk[0][0] == G.device_net.lib.fg_net_num_stages(G.device_net.h) - 1 and bk[-1][1] == 0
                assert sorted((lo, hi) for (_, _, lo, hi) in bk)[0][0] == 0
                assert sum(hi - lo for (_, _, lo, hi) in bk) == G.getParameters()[0].numel()
            outs.append((G.getParameters()[0].cpu().numpy().copy(), D.getParameters()[0].cpu().numpy().copy()))
        assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
    finally:
        dist.destroy_process_group()


def test_bench_runs_under_torchrun_single_rank():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3",
           "--warmup", "1", "--batch", "16", "--no-cpu-baseline"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 1 and j["value"] > 0 and j["unit"] == "images/sec" and "roofline" in j
