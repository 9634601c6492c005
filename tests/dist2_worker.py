"""Worker of tests/test_gpu_dist2.py: rank r of a 2-rank data-parallel job (both ranks share the one GPU of the test box,
collectives over gloo).  Runs one D-step + one G-step on its shard and dumps the resulting flat vectors."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def build(ctx, B):
    from face_generator_amd import models, nn_utils
    gen = torch.Generator().manual_seed(3)
    G = models.create_G((3, 32, 32), 100); D = models.create_D((3, 32, 32))
    nn_utils.initializeWeights(D, 0.05, 0.01, gen=gen); nn_utils.initializeWeights(G, 0.05, 0.01, gen=gen)
    G.cuda(ctx, max_batch=B); D.cuda(ctx, max_batch=B)
    return G, D


def data(B):
    g = torch.Generator().manual_seed(77)
    real = torch.rand(B // 2, 32, 32, 3, generator=g)
    nz1 = torch.rand(B // 2, 100, generator=g) * 2 - 1
    nz2 = torch.rand(B, 100, generator=g) * 2 - 1
    masks1 = [(torch.rand(B, c, generator=g) < 0.8).float() for c in (64, 128, 256, 512)] + \
             [(torch.rand(B, 512, generator=g) < 0.5).float() for _ in range(2)]
    masks2 = [(torch.rand(B, c, generator=g) < 0.8).float() for c in (64, 128, 256, 512)] + \
             [(torch.rand(B, 512, generator=g) < 0.5).float() for _ in range(2)]
    return real, nz1, nz2, masks1, masks2


def run(ctx, dist, rank, world, B, out_prefix, sync_bn=True):
    from face_generator_amd import adversarial
    d = ctx.device
    G, D = build(ctx, B // world)
    opt = dict(batchSize=B // world, noiseDim=100, D_L1=0.0, D_L2=0.0, G_L1=0.0, G_L2=0.0, D_clamp=0.0, G_clamp=0.0,
               sync_bn=(world > 1 and sync_bn))
    tr = adversarial.Trainer(ctx, G, D, opt, dist=dist if world > 1 else None)
    real, nz1, nz2, masks1, masks2 = data(B)
    h = B // 2
    hs, bs = h // world, B // world
    # D-step batch layout is [real | fake]: rank r owns real[r*hs:(r+1)*hs] and the fakes of nz1[r*hs:(r+1)*hs]; the
    # matching rows of the GLOBAL dropout masks are the real rows then the fake rows of that shard
    rows1 = list(range(rank * hs, (rank + 1) * hs)) + list(range(h + rank * hs, h + (rank + 1) * hs))
    m1 = [m[rows1].reshape(-1).to(d) for m in masks1]
    tr.step_D(real[rank * hs:(rank + 1) * hs].to(d).contiguous(), nz1[rank * hs:(rank + 1) * hs].to(d).contiguous(), m1)
    rows2 = list(range(rank * bs, (rank + 1) * bs))
    m2 = [m[rows2].reshape(-1).to(d) for m in masks2]
    tr.step_G(nz2[rank * bs:(rank + 1) * bs].to(d).contiguous(), m2)
    tr.finish_pending()
    torch.cuda.synchronize()
    np.savez(out_prefix + "_%d_of_%d.npz" % (rank, world), pG=G.getParameters()[0].cpu().numpy(), pD=D.getParameters()[0].cpu().numpy(),
             gG=G.getParameters()[1].cpu().numpy(), gD=D.getParameters()[1].cpu().numpy())


def main():
    rank, world, B, out_prefix = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), sys.argv[4]
    sync_bn = (sys.argv[6] != "0") if len(sys.argv) > 6 else True
    # carrier (argv[7]): "gloo" = both ranks on device 0, torch.distributed over gloo (the one-GPU test box);
    # "fg_comm" / "nccl" = one GPU per rank, the library's own RCCL communicator / torch.distributed over RCCL (>= 2 GPUs)
    carrier = sys.argv[7] if len(sys.argv) > 7 else "gloo"
    from face_generator_amd.runtime import get_context
    from face_generator_amd import distributed
    import torch.distributed as dist
    one_gpu_each = world > 1 and carrier in ("fg_comm", "nccl")
    ctx = get_context(rank if one_gpu_each else 0)
    carrier_obj = dist
    if world > 1:
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=sys.argv[5], RANK=str(rank), WORLD_SIZE=str(world))
        if one_gpu_each:
            torch.cuda.set_device(rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
            carrier_obj = distributed.make_collective(ctx, dist, prefer="fg_comm" if carrier == "fg_comm" else "torch", strict=True)
            assert not getattr(carrier_obj, "fallback", False)
            if carrier == "fg_comm":
                assert isinstance(carrier_obj, distributed.FgCollective), carrier_obj.describe()
        else:
            dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        run(ctx, carrier_obj, rank, world, B, out_prefix, sync_bn)
    finally:
        if world > 1:
            dist.destroy_process_group()


if __name__ == "__main__":
    main()
