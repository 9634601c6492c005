"""World-size-2 gloo test of the data-parallel exchange step on CPU: gradients of two half-batch
shards, all-reduced (sum) and scaled by 1/world, equal the single-process gradient on the concatenated batch; after
the same Adam step the replicas are bit-identical.  Compute is the oracle (no GPU here); the exchange logic under
test is face_generator_amd.distributed, the one the GPU Trainer uses."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.multiprocessing as mp

from oracle import torch7_nn as O


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close()
    return p


def _worker(rank, world, port, q):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1",
                      MASTER_PORT=str(port))
    torch.set_num_threads(2)
    from face_generator_amd import distributed as D
    r, w = D.init("gloo")
    assert (r, w) == (rank, world)
    rng = np.random.default_rng(7)                       # identical replicas + the same GLOBAL batch on every rank
    Dn = O.create_D32b((1, 32, 32), rng)
    Gn = O.create_G32((1, 32, 32), 100, rng)
    st = O.GanState(Gn, Dn)
    Bg = 8
    x = rng.uniform(0, 1, (Bg, 1, 32, 32)).astype(np.float32)
    tg = (rng.random(Bg) < 0.5).astype(np.float32)
    masks = [(rng.random((Bg, c)) < 0.8).astype(np.float32) for c in (64, 128, 256, 512)] + \
            [(rng.random((Bg, 512)) < 0.5).astype(np.float32) for _ in range(2)]
    # single-process reference on the whole batch
    O.set_dropout_masks(st.D, masks)
    f_ref, g_ref, _, _ = O.feval_D(st, x, tg)
    g_ref = g_ref.copy()
    # this rank's shard
    xs = D.shard(torch.tensor(x)).numpy(); ts = D.shard(torch.tensor(tg)).numpy()
    O.set_dropout_masks(st.D, [D.shard(torch.tensor(m)).numpy() for m in masks])
    st.opt.update(D_L2=0.0, D_clamp=0.0)                 # raw gradient: penalty/clamp come AFTER the reduce
    st.gD[...] = 0
    out = st.D.forward(xs)
    st.D.backward(xs, st.crit.backward(out, ts))
    flat = torch.tensor(st.gD.copy())
    D.allreduce_sum_(flat)
    g = flat.numpy() * np.float32(D.grad_scale())
    g = g + st.pD * np.float32(1e-4)                     # adversarial.lua:109
    g = np.clip(g, -1, 1)                                # adversarial.lua:121-123, after the reduce
    p = st.pD.copy()
    O.interruptable_adam(lambda _: (0.0, g), p, {}, {})
    q.put((rank, float(np.abs(g - g_ref).max()), float(np.abs(g_ref).max()), p))
    torch.distributed.barrier()
    torch.distributed.destroy_process_group()


@pytest.mark.timeout(300)
def test_sharded_D_step_matches_global_batch_and_replicas_agree():
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted([q.get(timeout=240) for _ in range(world)], key=lambda t: t[0])
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for (rank, err, scale, _) in res:
        assert err <= 2e-6 * max(scale, 1.0), (rank, err, scale)     # == the global-batch gradient (D has no BatchNorm)
    assert np.array_equal(res[0][3], res[1][3])                      # replicas bit-identical after the update


def test_shard_and_scale_single_process():
    from face_generator_amd import distributed as D
    t = torch.arange(12).reshape(6, 2)
    assert D.shard(t, 1, 3).tolist() == [[4, 5], [6, 7]]
    assert D.grad_scale() == 1.0
    assert D.allreduce_sum_(torch.ones(3)).tolist() == [1, 1, 1]
