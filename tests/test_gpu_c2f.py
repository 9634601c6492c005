"""Coarse-to-fine nets (BASELINE configs 4-5: models_c2f.lua G_d / D_c, adversarial_c2f.lua) on the device plan vs
the oracle: forward, flat gradients, full D-step / G-step with Adam."""
import numpy as np
import pytest
import torch

from oracle import torch7_nn as O
from gpu_util import nhwc, nchw, dev, close, close_after_first_adam_step
from test_gpu_net import check_flat_grads, draw_kink_safe

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from face_generator_amd.runtime import get_context
    return get_context(0)


def build(ctx, S, B, seed):
    from face_generator_amd import models_c2f
    rng = np.random.default_rng(seed)
    G = O.create_G_d((3, S, S), rng)
    D = O.create_D_c((3, S, S), rng)
    for net in (G, D):
        for m in net.modules:
            if isinstance(m, O.PReLU):
                m.weight[0] = np.float32(rng.uniform(0.1, 0.4))
    st = O.GanState(G, D, O.C2F_OPT)
    Gd = models_c2f.create_G((3, S, S), cuda=True, max_batch=B)
    Dd = models_c2f.create_D((3, S, S), cuda=True, max_batch=B)
    pG, _ = Gd.getParameters(); pD, _ = Dd.getParameters()
    assert pG.numel() == st.pG.size and pD.numel() == st.pD.size
    pG.copy_(torch.tensor(st.pG)); pD.copy_(torch.tensor(st.pD))
    Gd.inner.device_net.params_changed(); Dd.inner.device_net.params_changed()
    return st, Gd, Dd, rng


def masks_for(rng, B, S):
    m4 = (rng.random((B, 256, S // 4, S // 4)) < 0.5).astype(np.float32)      # Dropout on the 4-D tensor (NCHW order)
    m2 = (rng.random((B, 512)) < 0.5).astype(np.float32)
    return [m4, m2]


def dev_masks(masks, device):
    m4 = torch.tensor(masks[0], device=device).permute(0, 2, 3, 1).contiguous().reshape(-1)   # -> internal NHWC order
    return [m4, torch.tensor(masks[1].reshape(-1), device=device)]


@pytest.mark.parametrize("S,B", [(16, 4), (32, 6)])
def test_c2f_G_and_D_forward_backward(ctx, S, B):
    st, Gd, Dd, rng = build(ctx, S, B, seed=500 + S)
    d = ctx.device
    cond = rng.uniform(0, 1, (B, 3, S, S)).astype(np.float32)
    noise, diff = draw_kink_safe(rng, lambda: rng.uniform(-1, 1, (B, 1, S, S)).astype(np.float32),
                                 lambda nz: st.G.forward([nz, cond]), [st.G.inner])
    gy = rng.standard_normal(diff.shape).astype(np.float32)
    st.gG[...] = 0
    st.G.backward([noise, cond], gy)
    dn = Gd.inner.device_net
    y = dn.forward(Gd.combine_device(ctx, nhwc(noise, d), nhwc(cond, d)))
    close(nchw(y), diff, atol=2e-5 * max(1, np.abs(diff).max()), what="c2f G diff image")
    dn.backward(nhwc(gy, d), param_grads=True)
    check_flat_grads(dn.grads.cpu().numpy(), st.G.inner, "c2f G")
    # D
    masks = masks_for(rng, B, S)
    O.set_dropout_masks(st.D, masks)
    x, out = draw_kink_safe(rng, lambda: rng.uniform(-1, 1, (B, 3, S, S)).astype(np.float32),
                            lambda xx: st.D.forward([xx, cond]), [st.D.inner])
    gyo = rng.standard_normal(out.shape).astype(np.float32)
    st.gD[...] = 0
    gin = st.D.backward([x, cond], gyo)
    dnD = Dd.inner.device_net
    yd = dnD.forward(Dd.combine_device(ctx, nhwc(x, d), nhwc(cond, d)), masks=dev_masks(masks, d))
    close(yd.cpu().numpy(), out, atol=1e-5, what="c2f D probabilities")
    gx = dnD.backward(dev(gyo, d), param_grads=True, input_grad=True)
    close(nchw(gx), gin[0], atol=1e-4 * np.abs(gin[0]).max() + 1e-8, what="c2f D gradInput[1]")
    check_flat_grads(dnD.grads.cpu().numpy(), st.D.inner, "c2f D")


def test_c2f_full_steps(ctx):
    from face_generator_amd import adversarial_c2f
    S, B = 16, 4
    st, Gd, Dd, rng = build(ctx, S, B, seed=600)
    d = ctx.device
    tr = adversarial_c2f.TrainerC2F(ctx, Gd, Dd, dict(batchSize=B))
    diff_r = rng.uniform(-1, 1, (B // 2, 3, S, S)).astype(np.float32)
    cond_r = rng.uniform(0, 1, (B // 2, 3, S, S)).astype(np.float32)
    cond_f = rng.uniform(0, 1, (B // 2, 3, S, S)).astype(np.float32)
    nz = rng.uniform(-1, 1, (B // 2, 1, S, S)).astype(np.float32)
    masks = masks_for(rng, B, S)
    ref = O.step_D_c2f(st, diff_r, cond_r, nz, cond_f, masks)
    got = tr.step_D(nhwc(diff_r, d), nhwc(cond_r, d), nhwc(nz, d), nhwc(cond_f, d), dev_masks(masks, d), keep_grad=True)
    close(got["outputs"].cpu().numpy().reshape(-1), ref["out"].reshape(-1), atol=1e-5, what="c2f D-step outputs")
    assert abs(got["loss"].item() - ref["f_bce"]) <= 1e-5 * abs(ref["f_bce"])
    close(got["grad"].cpu().numpy(), ref["grad"], atol=1e-4 * np.abs(ref["grad"]).max() + 1e-7, what="c2f D-step grad")
    close_after_first_adam_step(Dd.getParameters()[0].cpu().numpy(), st.pD, got["grad"].cpu().numpy(), ref["grad"],
                                "c2f D params after Adam")
    nz2 = rng.uniform(-1, 1, (B, 1, S, S)).astype(np.float32)
    cond2 = rng.uniform(0, 1, (B, 3, S, S)).astype(np.float32)
    masks2 = masks_for(rng, B, S)
    ref = O.step_G_c2f(st, nz2, cond2, masks2)
    got = tr.step_G(nhwc(nz2, d), nhwc(cond2, d), dev_masks(masks2, d), keep_grad=True)
    close(nchw(got["samples"]), ref["samples"], atol=2e-5, what="c2f G-step samples")
    close(got["outputs"].cpu().numpy().reshape(-1), ref["out"].reshape(-1), atol=1e-5, what="c2f G-step D outputs")
    close(got["grad"].cpu().numpy(), ref["grad"], atol=1e-4 * np.abs(ref["grad"]).max() + 1e-7, what="c2f G-step grad")
    close_after_first_adam_step(Gd.getParameters()[0].cpu().numpy(), st.pG, got["grad"].cpu().numpy(), ref["grad"],
                                "c2f G params after Adam")


def test_maxpool_dropout_concat_ops(ctx):
    rng = np.random.default_rng(31)
    d, lib = ctx.device, ctx.lib
    B, C, H, W = 3, 8, 6, 10
    x = rng.standard_normal((B, C, H, W)).astype(np.float32)
    x[0, 0, 0, 0] = x[0, 0, 0, 1] = 5.0                      # tie: first max in scan order wins
    mp = O.SpatialMaxPooling()
    y = mp.forward(x)
    gy = rng.standard_normal(y.shape).astype(np.float32)
    gx = mp.backward(x, gy)
    xd = nhwc(x, d)
    yd = ctx.empty(B, H // 2, W // 2, C)
    ctx.check(lib.fg_maxpool2x2_forward(ctx.h, xd.data_ptr(), yd.data_ptr(), B, H, W, C))
    close(nchw(yd), y, atol=0, what="maxpool fwd")
    gxd = ctx.empty(B, H, W, C)
    ctx.check(lib.fg_maxpool2x2_backward(ctx.h, xd.data_ptr(), nhwc(gy, d).data_ptr(), gxd.data_ptr(), B, H, W, C))
    close(nchw(gxd), gx, atol=0, what="maxpool bwd")
    a = rng.standard_normal((B, 1, H, W)).astype(np.float32)
    out = ctx.empty(B, H, W, C + 1)
    ctx.check(lib.fg_concat_channels(ctx.h, nhwc(a, d).data_ptr(), xd.data_ptr(), out.data_ptr(), B * H * W, 1, C))
    close(nchw(out), np.concatenate([a, x], 1), atol=0, what="JoinTable")
