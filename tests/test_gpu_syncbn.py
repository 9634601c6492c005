"""sync-BN: two data-parallel replicas of G, each on half of a batch, driven in lock-step inside one
process with an in-process sum "all-reduce" of the fp64 BatchNorm sums at every pause, must reproduce the
single-replica result on the whole batch: images, running statistics and (after summing the replicas' gradients)
the flat gradient vector.  This is the exact-parity mode for sharded runs; throughput runs use per-GPU statistics."""
import numpy as np
import pytest
import torch

from oracle import torch7_nn as O
from gpu_util import nhwc, nchw, dev, close
from test_gpu_net import check_flat_grads, draw_kink_safe

pytestmark = pytest.mark.gpu


def lockstep(gens):
    """Advance all generators together; at every pause sum their fp64 buffers in place (the all-reduce)."""
    while True:
        bufs = []
        for g in gens:
            try:
                bufs.append(next(g))
            except StopIteration:
                bufs.append(None)
        if all(b is None for b in bufs):
            return
        assert all(b is not None for b in bufs), "replicas must pause at the same BatchNorm"
        total = torch.stack(bufs).sum(0)
        for b in bufs:
            b.copy_(total)


def test_two_replicas_with_sync_bn_equal_global_batch():
    from face_generator_amd import models
    from face_generator_amd.runtime import get_context
    ctx = get_context(0)
    d = ctx.device
    B, C = 8, 3
    rng = np.random.default_rng(900)
    G = O.create_G32((C, 32, 32), 100, rng, weight_init_=False)
    for m in G.modules:
        if isinstance(m, O.SpatialBatchNormalization):
            m.bias[...] = rng.standard_normal(m.bias.shape).astype(np.float32) * 0.2
            m.weight[...] = rng.uniform(0.5, 1.5, m.weight.shape).astype(np.float32)
        if isinstance(m, O.PReLU):
            m.weight[0] = np.float32(rng.uniform(0.1, 0.4))
    pG, gG = G.getParameters()
    noise, img = draw_kink_safe(rng, lambda: rng.uniform(-1, 1, (B, 100)).astype(np.float32), G.forward, [G])
    gy = rng.standard_normal(img.shape).astype(np.float32)
    gG[...] = 0
    G.backward(noise, gy)
    reps = []
    for r in range(2):
        Gd = models.create_G((C, 32, 32), 100).cuda(ctx, max_batch=B // 2)
        Gd.getParameters()[0].copy_(torch.tensor(pG))
        Gd.device_net.params_changed()
        Gd.device_net.enable_sync_bn(None)       # externally driven (generator protocol)
        reps.append(Gd)
    h = B // 2
    lockstep([reps[r].device_net.forward_steps(dev(noise[r * h:(r + 1) * h], d)) for r in range(2)])
    out = np.concatenate([nchw(reps[r].device_net._output_view()) for r in range(2)], 0)
    close(out, img, atol=1e-5, what="sync-BN images == global-batch images")
    for r in range(2):      # running statistics are the GLOBAL-batch ones on every replica
        buf = reps[r].device_net.buffers.cpu().numpy()
        close(buf[:256], G.modules[5].running_mean, atol=1e-6, what="running_mean")
        close(buf[256:512], G.modules[5].running_var, atol=0, rtol=1e-5, what="running_var")
    lockstep([reps[r].device_net.backward_steps(nhwc(gy[r * h:(r + 1) * h], d)) for r in range(2)])
    gsum = sum(reps[r].device_net.grads.cpu().numpy().astype(np.float64) for r in range(2)).astype(np.float32)
    check_flat_grads(gsum, G, "sync-BN summed replica grads")
    # without sync the half-batch statistics differ (sanity: the test would notice a no-op implementation)
    reps[0].device_net.disable_sync_bn()
    y_local = reps[0].device_net.forward(dev(noise[:h], d))
    assert np.abs(nchw(y_local) - img[:h]).max() > 1e-4
