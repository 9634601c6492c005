"""Worker of tests/test_distributed_cpu.py::test_exchange_schedule_survives_rank_skew (runs in its own process: a planning-only
context excludes device contexts).  Walks every rank of a 4-rank job through the library's step entries with fg_comm_create_dry
communicators and NO device, injecting rank skew, and prints one JSON object {scenario: {rank: [schedule lines]}}.

  gate_skew   three gated D/G iterations; in the second one rank 2's maxAccuracyD gate (adversarial.lua:167-178) says "hold" while
              every other rank says "train" -- as if host arithmetic had rounded differently on that rank.
  tail_batch  adversarial.train over one epoch where rank 1's shard is one half-batch longer than the others' (adversarial.lua:54-56:
              the loop advances B/2 per iteration), overlap on, then the epoch-end fg_gan_finish_pending.
"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    import torch
    from face_generator_amd import models, nn_utils, adversarial, distributed
    from face_generator_amd.runtime import get_context
    from face_generator_amd.state import S
    ctx = get_context(-1)
    world, B = 4, 8
    out = {"gate_skew": {}, "tail_batch": {}, "tail_iterations": {}}
    gen = torch.Generator().manual_seed(1)
    G = models.create_G((3, 32, 32), 100)
    D = models.create_D((3, 32, 32))
    nn_utils.initializeWeights(D, gen=gen)
    nn_utils.initializeWeights(G, gen=gen)
    G.cuda(ctx, max_batch=B)
    D.cuda(ctx, max_batch=B)
    real = ctx.zeros(B // 2, 32, 32, 3)
    for r in range(world):
        coll = distributed.DryCollective(ctx, r, world)
        # ---- a gate that disagrees on one rank
        tr = adversarial.Trainer(ctx, G, D, dict(batchSize=B, noiseDim=100), dist=coll)
        assert tr.gan is not None
        tr.gan.set_comm(coll, sync_bn=False, overlap=1)
        coll.schedule(reset=True)
        for it in range(3):
            mine = not (it == 1 and r == 2)                                   # rank 2 alone says "hold" in iteration 1
            # what the transport would have delivered from the other ranks: their votes for THIS iteration
            coll.peers = (lambda it: (lambda rank, v, what: 0 if (it == 1 and rank == 2) else 1))(it)
            res = tr.step_D(real, None, gate=lambda acc, mine=mine: mine)
            assert res["trained"] == (it != 1), (r, it, res["trained"])       # EVERY rank holds when one does
            tr.step_G(B)
        tr.finish_pending()
        out["gate_skew"][r] = coll.schedule(reset=True)
        del tr
        # ---- an epoch whose shard is longer on one rank
        S.reset()
        S.OPT.update(batchSize=B, noiseDim=100, N_epoch=-1, saveFreq=10 ** 9)
        S.MODEL_G, S.MODEL_D, S.IMG_DIMENSIONS = G, D, (3, 32, 32)
        S.dist = coll
        sizes = [3 * (B // 2), 4 * (B // 2), 3 * (B // 2), 3 * (B // 2)]       # rank 1: one more half-batch

        class Shard(list):
            def size(self):
                return len(self)
        data = Shard([torch.zeros(3, 32, 32) for _ in range(sizes[r])])
        coll.peers = lambda rank, v, what: sizes[rank] if what == "min" else 1       # everybody's gate passes
        steps = {"n": 0}
        orig = adversarial.Trainer.step_D

        def counted(self, *a, **k):
            steps["n"] += 1
            return orig(self, *a, **k)
        adversarial.Trainer.step_D = counted
        try:
            import io
            import contextlib
            with contextlib.redirect_stdout(io.StringIO()):
                adversarial.train(data, maxAccuracyD=0.9, accsInterval=20)
        finally:
            adversarial.Trainer.step_D = orig
        out["tail_batch"][r] = coll.schedule(reset=True)
        out["tail_iterations"][r] = steps["n"]
        S.reset()
        coll.close()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
