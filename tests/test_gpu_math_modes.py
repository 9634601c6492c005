"""fg_set_math: the large contractions either on the native fp32 MFMA (mode 0) or emulated on the bf16 matrix pipe with six
exact split-plane products (mode 6, include/facegen_hip.h).  Mode 6 must be an fp32-equivalent drop-in: (1) against an fp64
reference its error is not larger than the fp32 MFMA's, on the layer shapes where the bf16x6 kernels are actually selected;
(2) the full-size forward and a whole D-step + G-step match the oracle at the same tolerances as mode 0."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import torch7_nn as O
from gpu_util import nhwc, nchw, dev, close
from test_gpu_net import build, d_masks, check_flat_grads

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from face_generator_amd.runtime import get_context
    c = get_context(0)
    prev = c.get_math()
    yield c
    c.set_math(prev)


def rel_rms(a, ref):
    return float((a.double() - ref).pow(2).mean().sqrt() / ref.pow(2).mean().sqrt())


# (B, H, W, Cin, Cout, k, folded nearest-x2): G's two big layers (models.lua:64-69), D's 128->256 (models.lua:395), and a
# small batch that reaches the bf16x6 kernels through split-K
@pytest.mark.parametrize("shape", [(128, 16, 16, 256, 128, 5, 1), (128, 8, 8, 128, 256, 5, 1), (128, 8, 8, 128, 256, 3, 0),
                                   (16, 16, 16, 256, 128, 5, 1)])
def test_bf16x6_contractions_are_not_less_accurate_than_fp32_mfma(ctx, shape):
    from face_generator_amd import ops
    B, H, W, Cin, Cout, k, up = shape
    d = ctx.device
    g = torch.Generator().manual_seed(B + Cin)
    f = 2 if up else 1
    x = torch.randn(B, H, W, Cin, generator=g); w = torch.randn(Cout, Cin, k, k, generator=g) * 0.05
    b = torch.randn(Cout, generator=g); gy = torch.randn(B, H * f, W * f, Cout, generator=g)
    nb = 2                                      # fp64 reference: first images (forward / data-gradient are per-sample)
    xr = x[:nb].permute(0, 3, 1, 2).double()
    if up: xr = F.interpolate(xr, scale_factor=2, mode="nearest")
    yref = F.conv2d(xr, w.double(), b.double(), padding=k // 2).permute(0, 2, 3, 1)
    gxr = F.conv_transpose2d(gy[:nb].permute(0, 3, 1, 2).double(), w.double(), padding=k // 2)
    if up: gxr = gxr.reshape(nb, Cin, H, 2, W, 2).sum((3, 5))
    gxr = gxr.permute(0, 2, 3, 1)
    gwr = None
    if B <= 16:                                 # the weight gradient reduces over the whole batch: fp64 only for the small one
        xa = x.permute(0, 3, 1, 2).double()
        if up: xa = F.interpolate(xa, scale_factor=2, mode="nearest")
        gwr = torch.nn.grad.conv2d_weight(xa, (Cout, Cin, k, k), gy.permute(0, 3, 1, 2).double(), padding=k // 2)
    xd, wd, bd, gyd = x.to(d), w.to(d), b.to(d), gy.to(d)
    err, outs = {}, {}
    # fg_set_math selects the arithmetic of the implicit-GEMM contractions; the forward / data gradient of 3x3, 5x5 and
    # folded up-convolution layers run as Winograd F(2x2, 3x3) on the fp32 pipe in EITHER mode (FG_FUSE_WINOGRAD*, round 5), so the
    # comparison is made with those bits cleared
    fusion = ctx.get_fusion()
    ctx.set_fusion(fusion & ~(32 | 64 | 128))
    for mode in (0, 6):
        ctx.set_math(mode)
        y = ops.conv2d_forward(xd, wd, bd, upsample2x=bool(up))
        gx = ops.conv2d_backward_data(gyd, wd, (H, W), upsample2x=bool(up))
        gw = ops.conv2d_backward_weight(xd, gyd, k, upsample2x=bool(up))
        gw = gw[0] if isinstance(gw, (tuple, list)) else gw
        outs[mode] = (y.cpu(), gx.cpu(), gw.cpu())
        err[mode] = [rel_rms(y[:nb].cpu(), yref), rel_rms(gx[:nb].cpu(), gxr)] + ([rel_rms(gw.cpu(), gwr)] if gwr is not None else [])
    ctx.set_math(0)
    ctx.set_fusion(fusion)
    for e0, e6, what in zip(err[0], err[6], ("forward", "data gradient", "weight gradient")):
        assert e6 <= max(1.5 * e0, 5e-7), "%s: bf16x6 rel rms error %.3e vs fp32 MFMA %.3e" % (what, e6, e0)
        assert e6 < 3e-6
    # the two modes are different roundings of the same fp32 result
    for a0, a6, what in zip(outs[0], outs[6], ("forward", "data gradient", "weight gradient")):
        assert float((a0 - a6).abs().max()) <= 2e-5 * float(a0.abs().max()), what
    assert not torch.equal(outs[0][0], outs[6][0]), "mode 6 did not select the bf16x6 kernels for this shape"


def test_full_batch_forward_matches_oracle_in_bf16x6(ctx):
    B, C = 128, 3
    st, Gd, Dd, rng = build(ctx, C, B, seed=810)
    noise = rng.uniform(-1, 1, (B, 100)).astype(np.float32)
    img = st.G.forward(noise)
    ctx.set_math(6)
    try:
        y = Gd.device_net.forward(dev(noise, ctx.device))
        close(nchw(y), img, atol=1e-5, what="G images at B=128 (bf16x6)")
        masks = d_masks(rng, B)
        O.set_dropout_masks(st.D, masks)
        p = st.D.forward(img)
        pd = Dd.device_net.forward(y.clone(), masks=[dev(m.reshape(-1), ctx.device) for m in masks])
        close(pd.cpu().numpy(), p, atol=1e-5, what="D probabilities at B=128 (bf16x6)")
    finally:
        ctx.set_math(0)


def test_train_step_matches_oracle_in_bf16x6(ctx):
    """adversarial.lua:240-288 at B=32 (every big layer runs the bf16x6 kernels through split-K): outputs, loss, flat
    gradients of the D-step; samples and D outputs of the G-step."""
    from face_generator_amd import adversarial
    B, C = 32, 3
    st, Gd, Dd, rng = build(ctx, C, B, seed=811, init="reference")
    ctx.set_math(6)
    try:
        tr = adversarial.Trainer(ctx, Gd, Dd, dict(batchSize=B, noiseDim=100, D_L1=0.0, D_L2=1e-4, G_L1=0.0, G_L2=0.0,
                                                  D_clamp=1.0, G_clamp=5.0))
        real = rng.uniform(0, 1, (B // 2, C, 32, 32)).astype(np.float32)
        nz = rng.uniform(-1, 1, (B // 2, 100)).astype(np.float32)
        masks = d_masks(rng, B)
        ref = O.step_D(st, real, nz, masks)
        got = tr.step_D(nhwc(real, ctx.device), dev(nz, ctx.device), [dev(m.reshape(-1), ctx.device) for m in masks],
                        keep_grad=True)
        close(got["outputs"].cpu().numpy().reshape(-1), ref["out"].reshape(-1), atol=1e-5, what="D-step D outputs (bf16x6)")
        assert abs(got["loss"].item() - ref["f_bce"]) <= 1e-5 * abs(ref["f_bce"])
        close(got["grad"].cpu().numpy(), ref["grad"], atol=1e-4 * np.abs(ref["grad"]).max() + 1e-7, what="D-step flat grad (bf16x6)")
        nz2 = rng.uniform(-1, 1, (B, 100)).astype(np.float32)
        masks2 = d_masks(rng, B)
        ref = O.step_G(st, nz2, masks2)
        got = tr.step_G(dev(nz2, ctx.device), [dev(m.reshape(-1), ctx.device) for m in masks2])
        close(nchw(got["samples"]), ref["samples"], atol=1e-5, what="G-step samples (bf16x6)")
        close(got["outputs"].cpu().numpy().reshape(-1), ref["out"].reshape(-1), atol=1e-5, what="G-step D outputs (bf16x6)")
    finally:
        ctx.set_math(0)


def test_set_math_rejects_unknown_modes(ctx):
    from face_generator_amd._lib import FgError
    with pytest.raises(FgError):
        ctx.set_math(3)
    assert ctx.get_math() == 0
