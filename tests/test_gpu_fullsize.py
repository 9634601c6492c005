"""BASELINE.json full sizes (B = 128, 32x32x3) and the edge cases the reference's loop produces
(adversarial.lua:54-76: shrinking tail batches, even sizes >= 4).  Full-size checks use the oracle forward where it
finishes in seconds plus size-independent properties: D is sample-independent (no BatchNorm), backward is linear in
the output gradient, a zero gradient gives zero parameter gradients, Adam with zero gradient is the identity."""
import numpy as np
import pytest
import torch

from oracle import torch7_nn as O
from gpu_util import nhwc, nchw, dev, close
from test_gpu_net import build, d_masks

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from face_generator_amd.runtime import get_context
    return get_context(0)


def test_full_batch_128_forward_matches_oracle(ctx):
    B, C = 128, 3
    st, Gd, Dd, rng = build(ctx, C, B, seed=800)
    noise = rng.uniform(-1, 1, (B, 100)).astype(np.float32)
    img = st.G.forward(noise)                                   # oracle forward only (seconds on the host cores)
    y = Gd.device_net.forward(dev(noise, ctx.device))
    close(nchw(y), img, atol=1e-5, what="G images at B=128")     # bar 1e-4 (north_star)
    masks = d_masks(rng, B)
    O.set_dropout_masks(st.D, masks)
    p = st.D.forward(img)
    pd = Dd.device_net.forward(y.clone(), masks=[dev(m.reshape(-1), ctx.device) for m in masks])
    close(pd.cpu().numpy(), p, atol=1e-5, what="D probabilities at B=128")


def test_size_independent_properties_at_full_batch(ctx):
    B, C = 128, 3
    st, Gd, Dd, rng = build(ctx, C, B, seed=801)
    d = ctx.device
    x = ctx.uniform((B, 32, 32, C), 0.0, 1.0, seed=3)
    masks = [ctx.bernoulli((B * c,), 0.8, 11, i * 100000) for i, c in enumerate((64, 128, 256, 512))] + \
            [ctx.bernoulli((B * 512,), 0.5, 12, i * 100000) for i in range(2)]
    dn = Dd.device_net
    full = dn.forward(x, masks=masks).clone()
    # (1) D is sample-independent: the first 4 rows of the B=128 answer equal a B=4 run on those rows
    sub_masks = [m.view(B, -1)[:4].contiguous().view(-1) for m in masks]
    part = dn.forward(x[:4].contiguous(), masks=sub_masks).clone()
    close(part.cpu().numpy(), full[:4].cpu().numpy(), atol=2e-6, what="D sample independence")
    # (2) backward is linear in the output gradient; zero gradient -> zero parameter gradients
    dn.forward(x, masks=masks)
    gy = ctx.normal((B, 1), 0.0, 1.0, seed=5)
    gx1 = dn.backward(gy, param_grads=True, input_grad=True).clone(); g1 = dn.grads.clone()
    dn.forward(x, masks=masks)
    gx2 = dn.backward(gy * 2.0, param_grads=True, input_grad=True).clone(); g2 = dn.grads.clone()
    close(g2.cpu().numpy(), 2 * g1.cpu().numpy(), atol=2e-5 * float(g1.abs().max()) + 1e-9, what="linearity of param grads")
    close(gx2.cpu().numpy(), 2 * gx1.cpu().numpy(), atol=2e-5 * float(gx1.abs().max()) + 1e-12, what="linearity of input grad")
    dn.forward(x, masks=masks)
    dn.backward(torch.zeros(B, 1, device=d), param_grads=True)
    assert float(dn.grads.abs().max()) == 0.0
    # (3) fused Adam with a zero gradient and no penalty leaves the parameters untouched
    p0 = dn.params.clone()
    m, v = torch.zeros_like(p0), torch.zeros_like(p0)
    ctx.check(ctx.lib.fg_adam_fused(ctx.h, dn.params.data_ptr(), dn.grads.data_ptr(), m.data_ptr(), v.data_ptr(),
                                    p0.numel(), 1.0, 0.0, 0.0, 1.0, 1e-3, 0.9, 0.999, 1e-8, 1, None))
    assert torch.equal(dn.params, p0)


@pytest.mark.parametrize("B", [4, 6, 10, 34, 130])
def test_ragged_tail_batches(ctx, B):
    """adversarial.lua:56: thisBatchSize shrinks at the end of an epoch (even, >= 4); M is then not a tile multiple."""
    from face_generator_amd import adversarial
    st, Gd, Dd, rng = build(ctx, 3, B, seed=820 + B)
    tr = adversarial.Trainer(ctx, Gd, Dd, dict(batchSize=B))
    real = rng.uniform(0, 1, (B // 2, 3, 32, 32)).astype(np.float32)
    nz = rng.uniform(-1, 1, (B // 2, 100)).astype(np.float32)
    masks = d_masks(rng, B)
    if B <= 34:
        ref = O.step_D(st, real, nz, masks)
        got = tr.step_D(nhwc(real, ctx.device), dev(nz, ctx.device), [dev(m.reshape(-1), ctx.device) for m in masks])
        close(got["outputs"].cpu().numpy().reshape(-1), ref["out"].reshape(-1), atol=1e-5, what="tail batch D outputs")
        assert (got["confusion"].cpu().numpy().reshape(2, 2) == ref["conf"]).all()
    else:   # large ragged batch: finite + probabilities in (0,1) + the step runs end to end
        got = tr.step_D(nhwc(real, ctx.device), dev(nz, ctx.device), [dev(m.reshape(-1), ctx.device) for m in masks])
        o = got["outputs"].cpu().numpy()
        assert np.isfinite(o).all() and (o > 0).all() and (o < 1).all() and o.shape[0] == B
        r2 = tr.step_G(dev(rng.uniform(-1, 1, (B, 100)).astype(np.float32), ctx.device))
        assert np.isfinite(r2["samples"].cpu().numpy()).all()


def test_invalid_arguments_return_errors_not_crashes(ctx):
    from face_generator_amd import FgError
    from face_generator_amd.runtime import DeviceNet
    with pytest.raises(FgError):
        DeviceNet(ctx, [("LINEAR", 100, 64), ("VIEW", 3, 4, 4)], (100, 1, 1), 4)          # View size mismatch
    with pytest.raises(FgError):
        DeviceNet(ctx, [("CONV", 3, 64, 4, 1)], (3, 8, 8), 4)                               # even kernel
    dn = DeviceNet(ctx, [("CONV", 3, 64, 3, 1), ("PRELU",), ("SPATIAL_DROPOUT", 0, 0, 0, 0, 0.2), ("AVGPOOL2",)], (3, 8, 8), 4)
    with pytest.raises(FgError):
        dn.forward(ctx.zeros(2, 8, 8, 3), masks=[])                                          # missing dropout mask
    with pytest.raises(FgError):
        dn._batch = 2; dn._x = ctx.zeros(2, 8, 8, 3)
        dn.backward(ctx.zeros(2, 4, 4, 64))                                                  # backward without forward
