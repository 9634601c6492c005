"""Module level (SURVEY 8(b) level i): every nn-protocol class forwards to one C entry per call.  The module-by-module
chain (nn.Sequential:updateOutput / :backward of upstream nn) must give the oracle's results and agree with the fused
device plan; module fields (.weight / .gradWeight ...) are views into the flat getParameters() vectors."""
import numpy as np
import pytest
import torch

from oracle import torch7_nn as O
from gpu_util import nhwc, nchw, dev, close
from test_gpu_net import build, check_flat_grads, d_masks, draw_kink_safe

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from face_generator_amd.runtime import get_context
    return get_context(0)


def test_module_by_module_chain_matches_oracle_and_plan(ctx):
    B, C = 4, 3
    st, Gd, Dd, rng = build(ctx, C, B, seed=700)
    d = ctx.device
    # --- G
    noise, img = draw_kink_safe(rng, lambda: rng.uniform(-1, 1, (B, 100)).astype(np.float32), st.G.forward, [st.G])
    gy = rng.standard_normal(img.shape).astype(np.float32)
    st.gG[...] = 0
    st.G.backward(noise, gy)
    Gd.zeroGradParameters()
    y = Gd.forward_modules(dev(noise, d))
    close(nchw(y), img, atol=1e-5, what="G module chain images")
    close(nchw(Gd.modules[6].output), st.G.modules[6].output, atol=5e-5, what="G modules[7].output (PReLU)")
    Gd.backward_modules(nhwc(gy, d))
    pG, gG = Gd.getParameters()
    check_flat_grads(gG.cpu().numpy(), st.G, "G (module chain)")
    assert Gd.modules[4].gradWeight.data_ptr() == gG[819200 + 8192 + 1:].data_ptr()      # views into the flat vector
    # the fused plan gives the same image from the same parameters
    yp = Gd.device_net.forward(dev(noise, d))
    close(nchw(yp), nchw(y), atol=2e-6, what="plan vs module chain")
    # --- D with injected masks (SpatialDropout / Dropout modules)
    masks = d_masks(rng, B)
    O.set_dropout_masks(st.D, masks)
    x, out = draw_kink_safe(rng, lambda: rng.uniform(0, 1, (B, C, 32, 32)).astype(np.float32), st.D.forward, [st.D])
    go = rng.standard_normal(out.shape).astype(np.float32)
    st.gD[...] = 0
    gin = st.D.backward(x, go)
    it = iter(masks)
    from face_generator_amd import nn
    for m in Dd.modules:
        if isinstance(m, (nn.SpatialDropout, nn.Dropout)):
            m.set_mask(dev(next(it), d))
    Dd.zeroGradParameters()
    yd = Dd.forward_modules(nhwc(x, d))
    close(yd.cpu().numpy(), out, atol=1e-5, what="D module chain probabilities")
    gx = Dd.backward_modules(dev(go, d))
    close(nchw(gx), gin, atol=1e-4 * np.abs(gin).max() + 1e-8, what="D module chain gradInput")
    check_flat_grads(Dd.getParameters()[1].cpu().numpy(), st.D, "D (module chain)")
    # Torch's accumulate semantics: a second backward without zeroing doubles the gradients
    Dd.forward_modules(nhwc(x, d)); Dd.backward_modules(dev(go, d))
    g2 = Dd.getParameters()[1].cpu().numpy()
    close(g2, 2 * st.gD, atol=2e-4 * np.abs(st.gD).max() + 1e-7, what="accGradParameters accumulates")


def test_host_tensor_boundary_and_errors(ctx):
    from face_generator_amd import models, nn_utils, FgError
    G = models.create_G((3, 32, 32), 100)
    with pytest.raises(FgError):
        G.modules[0].forward(torch.zeros(2, 100))          # host tensor into a module: no CPU path
    Gc = nn_utils.activateCuda(G, max_batch=4)
    assert nn_utils.isInCudaMode(Gc) and nn_utils.activateCuda(Gc) is Gc     # idempotent (nn_utils.lua:331)
    img = Gc.forward(torch.rand(4, 100) * 2 - 1)
    assert tuple(img.shape) == (4, 3, 32, 32) and not img.is_cuda and float(img.min()) >= 0 and float(img.max()) <= 1
    gi = Gc.backward(torch.rand(4, 100), torch.randn(4, 3, 32, 32))
    assert tuple(gi.shape) == (4, 100) and Gc.modules[0].gradInput is gi      # adversarial.lua:210 access pattern
    inner = nn_utils.deactivateCuda(Gc)
    assert inner is G and not G.is_cuda() and not G.modules[0].weight.is_cuda
