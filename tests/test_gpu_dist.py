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
        from face_generator_amd import distributed
        fgc = distributed.make_collective(ctx, dist, prefer="fg_comm", strict=True)     # the library's own RCCL communicator
        assert fgc.describe().startswith("fg_comm") and fgc.get_world_size() == 1
        for use_dist in (False, True, fgc):
            gen = torch.Generator().manual_seed(3)
            G = models.create_G((3, 32, 32), 100); D = models.create_D((3, 32, 32))
            nn_utils.initializeWeights(D, 0.05, 0.01, gen=gen); nn_utils.initializeWeights(G, 0.05, 0.01, gen=gen)
            G.cuda(ctx, max_batch=8); D.cuda(ctx, max_batch=8)
            D.device_net.mask_seed = 5
            tr = adversarial.Trainer(ctx, G, D, dict(batchSize=8), dist=(dist if use_dist is True else use_dist) or None)
            real = ctx.uniform((4, 32, 32, 3), 0.0, 1.0, seed=9)
            if use_dist is True:     # force the N > 1 code path (async all-reduce + deferred D update) on one rank:
                assert tr.gan is None            # torch.distributed carrier -> host-driven closures
                tr.world, tr.gscale = 2, 1.0     # a 1-rank sum all-reduce is the identity, so keep the scale at 1
            elif use_dist is fgc:    # the same inside fg_step_D / fg_step_G: overlap = 2 takes the exchange path on one rank
                assert tr.gan is not None
                tr.gan.set_comm(fgc, overlap=2)
            pending = (lambda: tr.gan.pending()) if tr.gan is not None else (lambda: tr._pending_D is not None)
            # explicit masks: the fused closure and the host-driven one draw their own masks from different Philox layouts
            mk = lambda s0: [ctx.bernoulli((8 * c,), 0.8, s0, i * 100000) for i, c in enumerate((64, 128, 256, 512))] + \
                            [ctx.bernoulli((8 * 512,), 0.5, s0 + 1, i * 100000) for i in range(2)]
            tr.step_D(real, ctx.uniform((4, 100), -1.0, 1.0, seed=10), mk(20))
            assert pending() == bool(use_dist)
            tr.step_G(ctx.uniform((8, 100), -1.0, 1.0, seed=11), mk(30))
            assert not pending()
            if use_dist is True:    # bucketed, backward-overlapped all-reduce of G: >= 2 buckets covering the whole flat vector
                bk = tr._buckets_G()
                assert len(bk) >= 2 and bk[0][0] == G.device_net.lib.fg_net_num_stages(G.device_net.h) - 1 and bk[-1][1] == 0
                assert sorted((lo, hi) for (_, _, lo, hi) in bk)[0][0] == 0
                assert sum(hi - lo for (_, _, lo, hi) in bk) == G.getParameters()[0].numel()
            outs.append((G.getParameters()[0].cpu().numpy().copy(), D.getParameters()[0].cpu().numpy().copy()))
        assert np.array_equal(outs[0][0], outs[1][0]) and np.array_equal(outs[0][1], outs[1][1])
        assert np.array_equal(outs[0][0], outs[2][0]) and np.array_equal(outs[0][1], outs[2][1])     # fg_comm carrier
        fgc.close()
    finally:
        dist.destroy_process_group()


def test_bench_runs_under_torchrun_single_rank():
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "1", "--master-addr",
           "127.0.0.1", "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "3",
           "--warmup", "1", "--batch", "16", "--no-cpu-baseline", "--no-live-traffic", "--c2f-steps", "2"]
    r = subprocess.run(cmd, capture_output=True, text=True, timeout=600, cwd=ROOT)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    j = json.loads(line)
    assert j["n_gpus"] == 1 and j["value"] > 0 and j["unit"] == "images/sec" and "roofline" in j
    # the default line carries BASELINE configs[3] as a driver-timed sub-record
    c = j["c2f"]
    assert "error" not in c, c
    assert c["value"] > 0 and c["ms_per_step"] > 0 and c["config"]["workload"].startswith("configs[3]") and "roofline" in c
    assert c["config"]["step_entry"].startswith("fg_step_D")
    # round 6: the line says which figure is which -- pipe utilisation of the dominant kernel (`frac`, live tiles x
    # live channels), SURVEY 8(d)'s algorithmic fraction beside it, the clock granted to the dominant launch ITSELF next to the
    # iteration average, the biggest launch on its own, and the shared object the run loaded
    r = j["roofline"]
    assert 0 < r["frac"] < 1 and r["bound"] == "mfma" and r["survey_8d_frac"] == j["step_roofline"]["algorithmic_frac_of_f32_mfma_peak"]
    assert r["granted_clock_ghz"] is None or (0.5 < r["granted_clock_ghz"] < 2.6 and "ONLY the dominant launch" in r["granted_clock_source"])
    assert "iteration_average_clock_ghz" in r and r["dominant_launch"]["label"].startswith(r["kernel"])
    assert 0 < r["dominant_launch"]["frac"] < 1 and r["dominant_launch"]["executed_tflops"] < 157.3
    assert j["library"].endswith("libfacegen_hip.so") and os.path.isfile(j["library"])


def test_fg_comm_c_abi_world1():
    """fg_comm_* (include/facegen_hip.h): RCCL bound by the library at run time; a one-rank communicator must be the
    identity for sum / broadcast in fp32, fp64 and int32, blocking and overlapped (side stream + fg_comm_wait)."""
    import ctypes
    from face_generator_amd.runtime import get_context
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    ctx = get_context(0)
    lib = ctx.lib
    buf = ctypes.create_string_buffer(128)
    ctx.check(lib.fg_comm_unique_id(ctx.h, buf, 128))
    assert lib.fg_comm_unique_id(ctx.h, buf, 64) < 0                       # short buffer -> FG_ERR_INVALID, not a crash
    h = ctypes.c_void_p()
    ctx.check(lib.fg_comm_create(ctx.h, buf.raw, 128, 0, 1, ctypes.byref(h)))
    assert lib.fg_comm_rank(h) == 0 and lib.fg_comm_world(h) == 1 and b"rccl" in lib.fg_comm_library()
    x = ctx.uniform((1 << 20,), -1.0, 1.0, seed=3)
    ref = x.clone()
    ctx.check(lib.fg_allreduce_sum(h, x.data_ptr(), x.numel()))
    ctx.check(lib.fg_allreduce_sum_async(h, x.data_ptr(), x.numel()))
    ctx.check(lib.fg_allreduce_sum_async(h, x[:1000].data_ptr(), 1000))
    ctx.check(lib.fg_comm_wait(h))
    ctx.check(lib.fg_broadcast(h, x.data_ptr(), x.numel(), 0))
    d64 = torch.arange(7, dtype=torch.float64, device=ctx.device)
    i32 = torch.arange(4, dtype=torch.int32, device=ctx.device)
    ctx.check(lib.fg_allreduce_sum_f64(h, d64.data_ptr(), 7))
    ctx.check(lib.fg_allreduce_sum_i32(h, i32.data_ptr(), 4))
    torch.cuda.synchronize()
    assert torch.equal(x, ref) and d64.tolist() == list(range(7)) and i32.tolist() == [0, 1, 2, 3]
    assert lib.fg_broadcast(h, x.data_ptr(), 4, 3) < 0                     # root outside the communicator
    assert lib.fg_comm_create(ctx.h, buf.raw, 128, 2, 1, ctypes.byref(ctypes.c_void_p())) < 0
    ctx.check(lib.fg_comm_destroy(h))
