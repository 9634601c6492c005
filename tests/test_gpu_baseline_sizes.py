"""Oracle parity AT THE BASELINE.json SIZES: the kernels the production sizes select -- the
wave-specialised 256x128 contraction (forward and data gradient), the weight gradient at its production split counts, the
deferred final reductions -- only run when B is a multiple of 64 / 128, so the small-batch parity tests never see them.

  * cfg2 (configs[1]): one full D-step + G-step at 32x32x3, B = 128 -- outputs, loss, flat gradients, post-Adam
    parameters -- against the fp32 oracle, and the flat gradients against the oracle run in float64 at the SURVEY 8(c)
    bars without any widening.
  * PReLU kinks / max-pool ties: the device step runs first and the oracle adopts its branch decisions
    (gpu_util.adopt_device_branches), so gradients are compared on identical branches -- no flip allowance.
  * configs[3] (c2f 64x64): G_d / D_c forward + backward at S = 64 (B = 8 selects the production kernels there), forward at
    B = 128, and one D-step + G-step.
"""
import copy

import numpy as np
import pytest
import torch

from oracle import torch7_nn as O
from gpu_util import nhwc, nchw, dev, close, close_after_first_adam_step, adopt_device_branches, count_branch_flips
from test_gpu_net import build, d_masks, check_flat_grads
import test_gpu_c2f as C2F

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from face_generator_amd.runtime import get_context
    return get_context(0)


def f64_state(st, opt=None):
    """The same nets / parameters as `st`, every tensor and every accumulation in float64 (the error-budget oracle of
    SURVEY 8(c): weight gradients reduce over up to 131 072 pixels)."""
    G, D = copy.deepcopy(st.G), copy.deepcopy(st.D)
    G.astype(np.float64); D.astype(np.float64)
    return O.GanState(G, D, opt if opt is not None else dict(st.opt))


def check_vs_f64(name, g_dev, g32, g64, net64):
    """Tight bars: whole flat vector <= 1e-4 * max|g| + 1e-7; every weight tensor (conv / linear / BN gamma)
    <= 1e-4 * max|g_tensor| + 1e-7 against the float64 oracle.  Also reports how far the fp32 ORACLE itself is from float64,
    so a device error is read against the rounding budget of the reference formulation."""
    g_dev = g_dev.astype(np.float64)
    tol = 1e-4 * np.abs(g64).max() + 1e-7
    close(g_dev, g64, atol=tol, what="%s flat gradient vs the float64 oracle" % name)
    off, msgs = 0, []
    for (m, pn, gn) in net64.parameters():
        r = getattr(m, gn).reshape(-1)
        e = np.abs(g_dev[off:off + r.size] - r).max()
        e32 = np.abs(g32[off:off + r.size].astype(np.float64) - r).max()
        t = 1e-4 * np.abs(r).max() + 1e-7
        if pn == 'weight' and not isinstance(m, O.PReLU) and e > t:
            msgs.append("%s %s.%s: device err %.3e (fp32 oracle err %.3e) tol %.3e" % (name, type(m).__name__, pn, e, e32, t))
        off += r.size
    assert not msgs, "\n".join(msgs)


@pytest.mark.parametrize("init", ["default", "reference"])
def test_cfg2_full_step_at_batch_128(ctx, init):
    """configs[1]: adversarial.lua:240-288 at 32x32x3, B = 128, Adam -- the headline configuration itself."""
    from face_generator_amd import adversarial
    B, C = 128, 3
    st, Gd, Dd, rng = build(ctx, C, B, seed=1400, init=init)
    st64 = f64_state(st) if init == "default" else None
    twinD, twinG = ([st64.D], [st64.G]) if st64 is not None else ((), ())
    tr = adversarial.Trainer(ctx, Gd, Dd, dict(batchSize=B, noiseDim=100))
    d = ctx.device
    dnG, dnD = Gd.device_net, Dd.device_net
    real = rng.uniform(0, 1, (B // 2, C, 32, 32)).astype(np.float32)
    nz = rng.uniform(-1, 1, (B // 2, 100)).astype(np.float32)
    masks = d_masks(rng, B)
    got = tr.step_D(nhwc(real, d), dev(nz, d), [dev(m.reshape(-1), d) for m in masks], keep_grad=True)
    adopt_device_branches(ctx, dnD, st.D, also=twinD)
    ref = O.step_D(st, real, nz, masks)
    print("cfg2 B=128 D-step [%s]: %d PReLU units decided differently by the device" % (init, count_branch_flips(st.D)))
    close(got["outputs"].cpu().numpy().reshape(-1), ref["out"].reshape(-1), atol=1e-5, what="D-step D outputs (B=128)")
    assert abs(got["loss"].item() - ref["f_bce"]) <= 1e-5 * abs(ref["f_bce"])
    assert abs(got["f"] - ref["f"]) <= 1e-5 * abs(ref["f"])
    assert (got["confusion"].cpu().numpy().reshape(2, 2) == ref["conf"]).all()
    gD = got["grad"].cpu().numpy()
    close(gD, ref["grad"], atol=1e-4 * np.abs(ref["grad"]).max() + 1e-7, what="D-step flat gradient (B=128)")
    close_after_first_adam_step(Dd.getParameters()[0].cpu().numpy(), st.pD, gD, ref["grad"], "D params after Adam (B=128)")
    if st64 is not None:
        r64 = O.step_D(st64, real.astype(np.float64), nz.astype(np.float64), masks)
        check_vs_f64("D-step", gD, ref["grad"], r64["grad"], st64.D)
        st64.pG[...] = st.pG; st64.pD[...] = st.pD          # the next step starts from the fp32 oracle's state
    adopt_device_branches(ctx, dnD, st.D, clear=True, also=twinD)
    # G-step on the updated D (the oracle's D and the device's D agree to the Adam bar above)
    Dd.getParameters()[0].copy_(torch.tensor(st.pD)); dnD.params_changed()
    pG_before = dnG.params.clone()
    nz2 = rng.uniform(-1, 1, (B, 100)).astype(np.float32)
    masks2 = d_masks(rng, B)
    got = tr.step_G(dev(nz2, d), [dev(m.reshape(-1), d) for m in masks2], keep_grad=True)
    adopt_device_branches(ctx, dnD, st.D, also=twinD)
    adopt_device_branches(ctx, dnG, st.G, params=pG_before, also=twinG)
    ref = O.step_G(st, nz2, masks2)
    print("cfg2 B=128 G-step [%s]: %d (D) + %d (G) PReLU units decided differently by the device"
          % (init, count_branch_flips(st.D), count_branch_flips(st.G)))
    close(nchw(got["samples"]), ref["samples"], atol=1e-5, what="G-step samples (B=128)")
    close(got["outputs"].cpu().numpy().reshape(-1), ref["out"].reshape(-1), atol=1e-5, what="G-step D outputs (B=128)")
    assert abs(got["loss"].item() - ref["f_bce"]) <= 1e-5 * abs(ref["f_bce"])
    gG = got["grad"].cpu().numpy()
    close(gG, ref["grad"], atol=1e-4 * np.abs(ref["grad"]).max() + 1e-7, what="G-step flat gradient (B=128)")
    close_after_first_adam_step(Gd.getParameters()[0].cpu().numpy(), st.pG, gG, ref["grad"], "G params after Adam (B=128)")
    if st64 is not None:
        r64 = O.step_G(st64, nz2.astype(np.float64), masks2)
        check_vs_f64("G-step", gG, ref["grad"], r64["grad"], st64.G)


def test_c2f_S64_forward_backward(ctx):
    """configs[3] nets at fineSize 64 (models_c2f.lua:113-145, 237-278): B = 8 fills the chip with whole rounds of 256x128
    tiles at 64x64 (524 288-pixel GEMMs at B = 128 use the same kernels), Linear(65536, 512), the 7x7 head."""
    S, B = 64, 8
    st, Gd, Dd, rng = C2F.build(ctx, S, B, seed=564)
    d = ctx.device
    cond = rng.uniform(0, 1, (B, 3, S, S)).astype(np.float32)
    noise = rng.uniform(-1, 1, (B, 1, S, S)).astype(np.float32)
    gy = rng.standard_normal((B, 3, S, S)).astype(np.float32)
    dn = Gd.inner.device_net
    y = dn.forward(Gd.combine_device(ctx, nhwc(noise, d), nhwc(cond, d)))
    adopt_device_branches(ctx, dn, st.G)
    diff = st.G.forward([noise, cond])
    st.gG[...] = 0
    st.G.backward([noise, cond], gy)
    print("c2f-64 G: %d PReLU units decided differently by the device" % count_branch_flips(st.G))
    close(nchw(y), diff, atol=2e-5 * max(1, np.abs(diff).max()), what="c2f-64 G diff image")
    dn.backward(nhwc(gy, d), param_grads=True)
    # bf16x6 mode (opt-in, FG_MATH=6): its dropped plane products are a ONE-SIDED truncation (<= 2^-24 relative each); over the
    # 4 M-term cancelling sum of a PReLU slope gradient at this size the bias adds up coherently (5.9e-4 of the gradient
    # measured: 8.85e-4 on 1.51, the same value with every fusion / tiling switch of round 2 off), where independent fp32
    # roundings average out -- in that mode the slope gradients get a 1e-3 relative bar; the default fp32-MFMA mode keeps the
    # plain SURVEY 8(c) bar (+ 32 ulp of the sum's condition scale)
    check_flat_grads(dn.grads.cpu().numpy(), st.G.inner, "c2f-64 G", prelu_rtol=1e-3 if ctx.get_math() == 6 else 0.0)
    for m in st.G.inner.modules:
        m.finput = None                                     # the oracle's im2col buffers are GBs at this size
    masks = C2F.masks_for(rng, B, S)
    O.set_dropout_masks(st.D, masks)
    x = rng.uniform(-1, 1, (B, 3, S, S)).astype(np.float32)
    gyo = rng.standard_normal((B, 1)).astype(np.float32)
    dnD = Dd.inner.device_net
    yd = dnD.forward(Dd.combine_device(ctx, nhwc(x, d), nhwc(cond, d)), masks=C2F.dev_masks(masks, d))
    adopt_device_branches(ctx, dnD, st.D)
    out = st.D.forward([x, cond])
    st.gD[...] = 0
    gin = st.D.backward([x, cond], gyo)
    print("c2f-64 D: %d PReLU units decided differently by the device" % count_branch_flips(st.D))
    close(yd.cpu().numpy(), out, atol=1e-5, what="c2f-64 D probabilities")
    gx = dnD.backward(dev(gyo, d), param_grads=True, input_grad=True)
    close(nchw(gx), gin[0], atol=1e-4 * np.abs(gin[0]).max() + 1e-8, what="c2f-64 D gradInput[1]")
    check_flat_grads(dnD.grads.cpu().numpy(), st.D.inner, "c2f-64 D")


def test_c2f_S64_forward_at_batch_128(ctx):
    """configs[3] at its own batch size: G_d and D_c have no BatchNorm, so the oracle walks the batch in chunks of 8."""
    S, B, CH = 64, 128, 8
    st, Gd, Dd, rng = C2F.build(ctx, S, B, seed=565)
    d = ctx.device
    cond = rng.uniform(0, 1, (B, 3, S, S)).astype(np.float32)
    noise = rng.uniform(-1, 1, (B, 1, S, S)).astype(np.float32)
    masks = C2F.masks_for(rng, B, S)
    diff = np.empty((B, 3, S, S), np.float32)
    prob = np.empty((B, 1), np.float32)
    for i in range(0, B, CH):
        sl = slice(i, i + CH)
        diff[sl] = st.G.forward([noise[sl], cond[sl]])
        O.set_dropout_masks(st.D, [m[sl] for m in masks])
        prob[sl] = st.D.forward([diff[sl], cond[sl]])
    y = Gd.inner.device_net.forward(Gd.combine_device(ctx, nhwc(noise, d), nhwc(cond, d)))
    close(nchw(y), diff, atol=2e-5 * max(1, np.abs(diff).max()), what="c2f-64 G diff image (B=128)")
    p = Dd.inner.device_net.forward(Dd.combine_device(ctx, y.clone(), nhwc(cond, d)), masks=C2F.dev_masks(masks, d))
    close(p.cpu().numpy(), prob, atol=1e-5, what="c2f-64 D probabilities (B=128)")


def test_c2f_S64_full_steps(ctx):
    """adversarial_c2f.lua:123-187 at fineSize 64: D-step + G-step with optim.adam, B = 8."""
    from face_generator_amd import adversarial_c2f
    S, B = 64, 8
    st, Gd, Dd, rng = C2F.build(ctx, S, B, seed=566)
    d = ctx.device
    dnG, dnD = Gd.inner.device_net, Dd.inner.device_net
    tr = adversarial_c2f.TrainerC2F(ctx, Gd, Dd, dict(batchSize=B))
    diff_r = rng.uniform(-1, 1, (B // 2, 3, S, S)).astype(np.float32)
    cond_r = rng.uniform(0, 1, (B // 2, 3, S, S)).astype(np.float32)
    cond_f = rng.uniform(0, 1, (B // 2, 3, S, S)).astype(np.float32)
    nz = rng.uniform(-1, 1, (B // 2, 1, S, S)).astype(np.float32)
    masks = C2F.masks_for(rng, B, S)
    got = tr.step_D(nhwc(diff_r, d), nhwc(cond_r, d), nhwc(nz, d), nhwc(cond_f, d), C2F.dev_masks(masks, d), keep_grad=True)
    adopt_device_branches(ctx, dnD, st.D)
    ref = O.step_D_c2f(st, diff_r, cond_r, nz, cond_f, masks)
    print("c2f-64 D-step: %d units decided differently by the device" % count_branch_flips(st.D))
    close(got["outputs"].cpu().numpy().reshape(-1), ref["out"].reshape(-1), atol=1e-5, what="c2f-64 D-step outputs")
    assert abs(got["loss"].item() - ref["f_bce"]) <= 1e-5 * abs(ref["f_bce"])
    close(got["grad"].cpu().numpy(), ref["grad"], atol=1e-4 * np.abs(ref["grad"]).max() + 1e-7, what="c2f-64 D-step grad")
    close_after_first_adam_step(Dd.getParameters()[0].cpu().numpy(), st.pD, got["grad"].cpu().numpy(), ref["grad"],
                                "c2f-64 D params after Adam")
    adopt_device_branches(ctx, dnD, st.D, clear=True)
    Dd.getParameters()[0].copy_(torch.tensor(st.pD)); dnD.params_changed()
    nz2 = rng.uniform(-1, 1, (B, 1, S, S)).astype(np.float32)
    cond2 = rng.uniform(0, 1, (B, 3, S, S)).astype(np.float32)
    masks2 = C2F.masks_for(rng, B, S)
    got = tr.step_G(nhwc(nz2, d), nhwc(cond2, d), C2F.dev_masks(masks2, d), keep_grad=True)
    adopt_device_branches(ctx, dnD, st.D)
    adopt_device_branches(ctx, dnG, st.G)
    ref = O.step_G_c2f(st, nz2, cond2, masks2)
    print("c2f-64 G-step: %d (D) + %d (G) units decided differently by the device" % (count_branch_flips(st.D), count_branch_flips(st.G)))
    close(nchw(got["samples"]), ref["samples"], atol=2e-5 * max(1, np.abs(ref["samples"]).max()), what="c2f-64 G-step samples")
    close(got["outputs"].cpu().numpy().reshape(-1), ref["out"].reshape(-1), atol=1e-5, what="c2f-64 G-step D outputs")
    close(got["grad"].cpu().numpy(), ref["grad"], atol=1e-4 * np.abs(ref["grad"]).max() + 1e-7, what="c2f-64 G-step grad")
    close_after_first_adam_step(Gd.getParameters()[0].cpu().numpy(), st.pG, got["grad"].cpu().numpy(), ref["grad"],
                                "c2f-64 G params after Adam")
