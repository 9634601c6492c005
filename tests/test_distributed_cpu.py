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
    Gn = O.create_G32((1, 32, 32), 100, rng, weight_init_=False)
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


def _dry(cmd_prefix, gpus, env=None):
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    cmd = cmd_prefix + [os.path.join(root, "bench.py"), "--gpus", str(gpus), "--dry-collective"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=root, env=dict(os.environ, **(env or {})))
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1
    return json.loads(lines[0])


def _check_schedule(j, world):
    """What fg_step_D / fg_step_G must issue, per combination (csrc/step.hip): D's flat gradient (2 863 239 floats) once per
    update -- on the side stream when overlapped, never when the gate holds the update --, G's (2 470 406) whole or in buckets that
    add up to it, the gate's global confusion counts (4 x int32) and the ranks' vote on the gate's answer (world x int32) only when a
    gate is given, and with sync-BN one fp64 exchange of
    2C + 1 sums per BatchNorm pass of G (C = 256, 128: three forwards -- B/2 fakes, B samples -- and one backward each)."""
    assert j["ranks_agree"] is True and j["ranks_walked"] == list(range(world)) and len(j["schedule"]) == 12
    nD, nG = 2863239, 2470406
    for name, lines in j["schedule"].items():
        ops = [l.split() for l in lines]
        f32 = [(int(o[3]), o[4]) for o in ops if o[1] == "allreduce" and o[2] == "f32"]
        f64 = [int(o[3]) for o in ops if o[2] == "f64"]
        i32 = [int(o[3]) for o in ops if o[2] == "i32"]
        overlap, sync, gate = "overlap=1" in name, "sync_bn=1" in name, name.split()[0].split("=")[1]
        d_red = [x for x in f32 if x[0] == nD]
        g_red = [x for x in f32 if x[0] != nD]
        assert len(d_red) == (0 if gate == "hold" else 1), (name, lines)
        assert sum(n for n, _ in g_red) == nG, (name, lines)
        if d_red:
            assert d_red[0][1] == ("side" if overlap else "compute"), (name, lines)
        assert all(st == ("side" if overlap else "compute") for _, st in g_red), (name, lines)
        assert (len(g_red) > 1) == overlap, (name, lines)                     # bucketed under the backward only when overlapped
        # a gated D-step: the global confusion counts, then (N > 1) the one-hot vote that makes every rank take the same branch
        assert i32 == ([] if gate == "none" else [4, world]), (name, lines)
        assert f64 == ([513, 257, 513, 257, 257, 513] if sync else []), (name, lines)
        waits = [o for o in ops if o[1] == "wait"]
        assert len(waits) == ((1 if d_red else 0) + 1 if overlap else 0), (name, lines)


@pytest.mark.timeout(600)
def test_dry_collective_schedule_is_identical_on_every_rank_gloo_world_2():
    """two real processes (torch.distributed.run, gloo, no GPU) each walk their own rank of a 2-GPU job through
    the library's step entries with planning-only contexts; rank 0 gathers and compares the schedules for
    {gate none / passes / holds} x {sync_bn} x {overlap}."""
    import sys
    j = _dry([sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
              "--master-port", str(_free_port())], 2)
    _check_schedule(j, 2)


@pytest.mark.timeout(600)
def test_dry_collective_schedule_eight_ranks_in_one_process():
    import sys
    _check_schedule(_dry([sys.executable], 8), 8)


@pytest.mark.timeout(600)
@pytest.mark.parametrize("env", [dict(FG_ADAM_PACK="1"), dict(FG_DEFER_WFINISH="0"), dict(FG_ADAM_PACK="1", FG_DEFER_WFINISH="0")])
def test_dry_collective_schedule_does_not_depend_on_the_optimizer_fusions(env):
    """fg_set_fusion bits 4 / 8 (batched weight-gradient sums, Adam inside the re-pack launch) change which kernels the host queues
    around the exchange, never the exchange: the same schedule for every setting (and the host side of both paths -- the deferred
    jobs, the update-only pack jobs -- runs without a GPU)."""
    import sys
    _check_schedule(_dry([sys.executable], 2, env=env), 2)


def test_planning_only_context_excludes_device_contexts():
    """include/facegen_hip.h: a process holds either planning-only contexts or real ones; host 'device' buffers are plain memory."""
    import ctypes
    from face_generator_amd import _lib
    lib = _lib.load_library()
    h = ctypes.c_void_p()
    if lib.fg_ctx_create(-1, ctypes.byref(h)) != 0:
        assert b"device context" in lib.fg_last_error(None)
        pytest.skip("this process already holds a device context")
    try:
        h2 = ctypes.c_void_p()
        assert lib.fg_ctx_create(0, ctypes.byref(h2)) < 0 and b"planning-only" in lib.fg_last_error(None)
        p = ctypes.c_void_p()
        assert lib.fg_malloc(h, 64, ctypes.byref(p)) == 0
        src = (ctypes.c_float * 4)(1, 2, 3, 4)
        dst = (ctypes.c_float * 4)()
        assert lib.fg_h2d(h, p, src, 16) == 0 and lib.fg_d2h(h, dst, p, 16) == 0 and list(dst) == [1, 2, 3, 4]
        assert lib.fg_fill(h, p, 7.0, 4) == 0                                  # a launch: skipped, not an error
        assert lib.fg_free(h, p) == 0
        c = ctypes.c_void_p()
        assert lib.fg_comm_create_dry(h, 1, 4, ctypes.byref(c)) == 0 and lib.fg_comm_world(c) == 4 and lib.fg_comm_rank(c) == 1
        assert lib.fg_allreduce_sum_async(c, p, 10) == 0 and lib.fg_comm_wait(c) == 0 and lib.fg_allreduce_sum_i32(c, p, 4) == 0
        buf = ctypes.create_string_buffer(256)
        assert lib.fg_comm_schedule(c, buf, 256, 1) == 0
        assert buf.value.decode().splitlines() == ["0 allreduce f32 10 side", "1 wait - 1 compute", "2 allreduce i32 4 compute"]
        assert lib.fg_comm_destroy(c) == 0
    finally:
        lib.fg_ctx_destroy(h)


@pytest.mark.timeout(600)
def test_exchange_schedule_survives_rank_skew():
    """First-run insurance for configs[2] / [4] (no hardware needed): every rank of a 4-rank job is walked through
    fg_step_D / fg_step_G / fg_gan_finish_pending with dry communicators while rank skew is injected (tests/dist_skew_worker.py).
    A branch that one rank takes alone is a hang on the hardware, so the schedule TEXT must stay identical on all ranks:
      * one rank's maxAccuracyD gate says "hold" while the others say "train" (adversarial.lua:167-178): the vote
        (distributed.Collective.agree) makes every rank hold -- no rank issues D's gradient all-reduce in that iteration;
      * one rank's shard is a half-batch longer (adversarial.lua:54-56): the epoch length is agreed (Collective.min_int), every rank
        runs the same number of iterations and the epoch ends with the same deferred-D wait."""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    r = subprocess.run([sys.executable, os.path.join(root, "tests", "dist_skew_worker.py")], capture_output=True, text=True, timeout=560, cwd=root)
    assert r.returncode == 0, r.stderr[-3000:]
    j = json.loads([l for l in r.stdout.splitlines() if l.startswith("{")][-1])
    nD = 2863239
    for scen in ("gate_skew", "tail_batch"):
        ranks = sorted(j[scen])
        assert len(ranks) == 4
        for k in ranks[1:]:
            assert j[scen][k] == j[scen][ranks[0]], "%s: rank %s issues a different schedule than rank %s" % (scen, k, ranks[0])
    ops = [l.split() for l in j["gate_skew"]["0"]]
    d_reduces = [o for o in ops if o[1] == "allreduce" and o[2] == "f32" and int(o[3]) == nD]
    votes = [o for o in ops if o[1] == "allreduce" and o[2] == "i32" and int(o[3]) == 4]
    assert len(d_reduces) == 2, "three gated iterations, one held on EVERY rank -> two D gradient all-reduces"
    assert len(votes) == 6, "per gated D-step: the global confusion counts and the vote (4 ranks -> 4 ints each)"
    # the held iteration (the second): counts, vote, then straight to G's buckets -- no D all-reduce in between
    seq = [(o[1], o[2], int(o[3]) if o[3].isdigit() else o[3]) for o in ops]
    second = seq.index(("allreduce", "i32", 4), 2)
    assert seq[second + 1] == ("allreduce", "i32", 4) and seq[second + 2][2] != nD
    assert set(j["tail_iterations"].values()) == {3}, "every rank runs the shortest shard's three iterations, not its own four"
    tail = [l.split() for l in j["tail_batch"]["0"]]
    assert sum(1 for o in tail if o[1] == "allreduce" and o[2] == "f32" and int(o[3]) == nD) == 3
    assert tail[0][1:4] == ["allreduce", "i32", "4"], "the epoch opens with the agreement on its length"
    assert tail[-1][1] == "wait" or tail[-2][1] == "wait", "fg_gan_finish_pending closes the epoch on every rank"
