"""GPU: closed-form properties of the two train-mode semantics SURVEY Appendix A flags as version-sensitive -- nn.BCECriterion's eps and
nn.Dropout's v2 scaling / nn.SpatialDropout's train-unscaled, eval-scaled rule -- checked on the HIP path DIRECTLY against numbers
derived by hand from the documented formulas, not against the oracle (a second, independent pin that does not
pass through oracle/torch7_nn.py).

  nn.BCECriterion (train.lua:148): f = -(1/n) sum[t log(x + eps) + (1 - t) log(1 - x + eps)], eps = 1e-12, so a fully wrong, fully
  saturated prediction costs -log(1e-12) = 27.6310211 per element -- NOT 100 (PyTorch's log clamp) and not inf; its gradient is
  -(1/n) (t - x) / ((1 - x + eps)(x + eps)) = -+ 1e12 / n there.
  nn.Dropout(p) v2 (models.lua:408, 411): train y = x * Bernoulli(1 - p) / (1 - p) -- every output is exactly 0 or x / (1 - p); evaluate
  y = x.  nn.SpatialDropout(p) (models.lua:387-402): train y = x * Bernoulli(1 - p) per (sample, channel) plane, NOT rescaled; evaluate
  y = (1 - p) x."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from face_generator_amd.runtime import get_context
    return get_context(0)


def test_bce_saturates_at_minus_log_eps_not_at_100(ctx):
    from face_generator_amd.nn import BCECriterion
    d = ctx.device
    sat = -math.log(1e-12)                                    # 27.631021...
    assert abs(sat - 27.6310211) < 1e-6
    for B in (1, 7, 128):
        # all wrong and saturated, both directions: x = 1e-20 ~ 0 with t = 1, x = 1 with t = 0
        for x, t in ((1e-20, 1.0), (1.0, 0.0)):
            p = torch.full((B,), x, device=d)
            tt = torch.full((B,), t, device=d)
            loss, grad, conf = BCECriterion().forward_backward_device(ctx, p, tt)
            assert abs(loss.item() - sat) <= 2e-6 * sat, (B, x, loss.item())
            want_g = (-1.0 if t == 1.0 else 1.0) * 1e12 / B
            assert np.allclose(grad.cpu().numpy(), want_g, rtol=2e-6, atol=0)
        # all right and saturated: the loss is -log(1 + eps) = 0 to fp32, the gradient -(1/n)(t - x)/(...) = 0
        p = torch.ones(B, device=d)
        loss, grad, _ = BCECriterion().forward_backward_device(ctx, p, torch.ones(B, device=d))
        assert abs(loss.item()) <= 1e-7 and np.abs(grad.cpu().numpy()).max() <= 1e-6 / B
    # mid-range closed forms: x = 1/2 -> log 2 either way; x = 1/4, t = 1 -> log 4; the mean over a mixed batch
    p = torch.tensor([0.5, 0.5, 0.25, 0.75], device=d)
    t = torch.tensor([1.0, 0.0, 1.0, 1.0], device=d)
    loss, grad, conf = BCECriterion().forward_backward_device(ctx, p, t)
    want = (math.log(2) + math.log(2) + math.log(4) + math.log(4.0 / 3.0)) / 4
    assert abs(loss.item() - want) <= 1e-6 * want
    want_g = np.array([-(1 - 0.5) / (0.5 * 0.5), -(0 - 0.5) / (0.5 * 0.5), -(1 - 0.25) / (0.75 * 0.25), -(1 - 0.75) / (0.25 * 0.75)]) / 4
    assert np.allclose(grad.cpu().numpy(), want_g, rtol=1e-6, atol=0)
    # confusion[pred][target] with pred = x > 0.5 (adversarial.lua:112-127): 0.5 itself counts as "fake"
    assert conf.cpu().numpy().tolist() == [1, 2, 0, 1]


def test_dropout_v2_scales_in_train_and_is_identity_in_evaluate(ctx):
    from face_generator_amd import nn
    d = ctx.device
    x = torch.randn(64, 512, device=d, generator=torch.Generator(device=d).manual_seed(3))
    for p in (0.5, 0.2):
        m = nn.Dropout(p)
        m.training()
        y = m.updateOutput(x)
        kept = y != 0
        scaled = x * np.float32(1.0 / (1.0 - p))
        assert torch.equal(y[kept], scaled[kept]), "train: kept units are x / (1 - p), bit for bit"
        assert (y[~kept] == 0).all()
        n = x.numel()
        frac = kept.float().mean().item()
        assert abs(frac - (1 - p)) <= 5 * math.sqrt(p * (1 - p) / n), frac
        assert abs(y.mean().item() - x.mean().item()) < 0.02        # the rescale keeps the expectation
        g = torch.randn_like(x)
        gx = m.updateGradInput(x, g)
        assert torch.equal(gx != 0, kept) and torch.equal(gx[kept], (g * np.float32(1.0 / (1.0 - p)))[kept])   # same mask, same scale
        m.evaluate()
        assert torch.equal(m.updateOutput(x), x), "evaluate: identity (v2)"


def test_spatial_dropout_is_unscaled_in_train_and_scaled_in_evaluate(ctx):
    from face_generator_amd import nn, FgError
    d = ctx.device
    B, H, W, C = 128, 8, 8, 64
    x = torch.randn(B, H, W, C, device=d, generator=torch.Generator(device=d).manual_seed(4)) + 3.0       # no exact zeros
    m = nn.SpatialDropout(0.2)
    m.training()
    y = m.updateOutput(x)
    plane_kept = (y != 0).reshape(B, H * W, C).any(dim=1)                 # [B][C]
    full = plane_kept[:, None, None, :].expand(B, H, W, C)
    assert torch.equal(y[full], x[full]), "train: a kept plane is x itself -- no 1 / (1 - p) rescale"
    assert (y[~full] == 0).all(), "a dropped (sample, channel) plane is zero everywhere"
    frac = plane_kept.float().mean().item()
    assert abs(frac - 0.8) <= 5 * math.sqrt(0.2 * 0.8 / (B * C)), frac
    g = torch.randn_like(x)
    gx = m.updateGradInput(x, g)
    assert torch.equal(gx[full], g[full]) and (gx[~full] == 0).all()
    m.evaluate()
    assert torch.equal(m.updateOutput(x), x * np.float32(0.8)), "evaluate: y = (1 - p) x"
    with pytest.raises(FgError):
        m.updateGradInput(x, g)                                            # upstream errors in evaluate mode


def test_net_level_dropout_masks_follow_the_same_rules(ctx):
    """The same two rules inside a compiled net (fg_net, masks drawn by the library's Philox launch): an identity Linear in front of
    nn.Dropout, an identity-free nn.SpatialDropout -> pooling chain."""
    from face_generator_amd import nn
    d = ctx.device
    B, K = 32, 64
    net = nn.Sequential()
    lin = nn.Linear(K, K)
    lin.weight.copy_(torch.eye(K)); lin.bias.zero_()
    net.add(lin).add(nn.Dropout(0.5))
    net.input_dims = (K, 1, 1)
    net.cuda(ctx, max_batch=B)
    net.training()
    x = torch.randn(B, K, device=d, generator=torch.Generator(device=d).manual_seed(5))
    y = net.device_net.forward(x)
    kept = y != 0
    assert torch.equal(y[kept], (x * 2.0)[kept]) and abs(kept.float().mean().item() - 0.5) < 5 * math.sqrt(0.25 / (B * K))
    net.evaluate()
    assert torch.equal(net.device_net.forward(x), x)
