"""Net-level and step-level parity of the device plan (fg_net_*) against the oracle's restatement of
models.lua / adversarial.lua, on the same seeded inputs, weights and injected dropout masks."""
import numpy as np
import pytest
import torch

from oracle import torch7_nn as O
from gpu_util import nhwc, nchw, dev, close, close_after_first_adam_step, oracle_backward_on_device_branches

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from face_generator_amd.runtime import get_context
    return get_context(0)


def build(ctx, C, B, seed, init="default"):
    """oracle nets + device nets holding identical parameters."""
    from face_generator_amd import models
    rng = np.random.default_rng(seed)
    G = O.create_G32((C, 32, 32), 100, rng, weight_init_=False)
    D = O.create_D32b((C, 32, 32), rng)
    if init == "reference":      # train.lua:137-138 -> N(0,0.005^2) / N(0,0.001^2), incl. BN gamma and PReLU slope
        O.initialize_weights(G, rng=rng); O.initialize_weights(D, rng=rng)
    else:
        for net in (G, D):       # well-conditioned init + non-trivial BN beta / PReLU slopes
            for m in net.modules:
                if isinstance(m, O.SpatialBatchNormalization):
                    m.bias[...] = rng.standard_normal(m.bias.shape).astype(np.float32) * 0.2
                    m.weight[...] = rng.uniform(0.5, 1.5, m.weight.shape).astype(np.float32)
                if isinstance(m, O.PReLU):
                    m.weight[0] = np.float32(rng.uniform(0.1, 0.4))
    st = O.GanState(G, D)
    Gd = models.create_G((C, 32, 32), 100).cuda(ctx, max_batch=B)
    Dd = models.create_D((C, 32, 32)).cuda(ctx, max_batch=B)
    pG, gG = Gd.getParameters(); pD, gD = Dd.getParameters()
    assert pG.numel() == st.pG.size and pD.numel() == st.pD.size
    pG.copy_(torch.tensor(st.pG)); pD.copy_(torch.tensor(st.pD))
    Gd.device_net.params_changed(); Dd.device_net.params_changed()
    return st, Gd, Dd, rng


def check_flat_grads(g, net, name, prelu_ulps=32, prelu_rtol=0.0):
    """Flat gradient vector vs the oracle, reported per parameter tensor (all failures listed).
    Tolerance: 1e-4 * max|g| + 1e-7 per tensor; a bias uses its module's weight-grad scale as well,
    because the bias grad of a conv feeding a BatchNorm is pure rounding noise around an exact zero; the single PReLU
    slope gradient is a cancelling sum over the whole tensor, so it also gets 32 ulp of its condition scale ||x*gy||_2
    (an fp32 GEMM that rounds differently from the oracle's -- e.g. the bf16x6 mode -- moves it by that much)."""
    offs, msgs = 0, []
    for i, m in enumerate(net.modules):
        wscale = 0.0
        for (mm, pn, gn) in m.parameters():
            ref = getattr(mm, gn).reshape(-1)
            scale = np.abs(ref).max()
            if pn == 'weight':
                wscale = scale
            got = g[offs:offs + ref.size]
            tol = 1e-4 * max(scale, wscale if pn == 'bias' else 0.0) + 1e-7
            if isinstance(mm, O.PReLU):     # scalar sum with cancellation: 32 fp32 ulps of its condition scale
                tol += prelu_ulps * 6e-8 * getattr(mm, "gw_cond", 0.0)
                tol = max(tol, prelu_rtol * scale)
            err = np.abs(got.astype(np.float64) - ref)
            if not (err <= tol).all():
                msgs.append("%s module %d %s %s: %d/%d off, max err %.3g (tol %.3g, max|ref| %.3g)"
                            % (name, i + 1, type(m).__name__, pn, (err > tol).sum(), ref.size, err.max(), tol, scale))
            offs += ref.size
    assert not msgs, "\n".join(msgs)


def prelu_margin(net):
    """min |x| over the inputs of every PReLU of the oracle's last forward, relative to the tensor scale.
    PReLU is non-differentiable at 0: an input within a few ulp of 0 makes d/dx (1 vs slope) depend on the last bit
    of the producing GEMM / BatchNorm, so a gradient comparison is only meaningful away from the kink."""
    m = 1.0
    for mod, xin in zip(net.modules, net._inputs):
        if isinstance(mod, O.PReLU):
            m = min(m, float(np.abs(xin).min() / max(np.abs(xin).max(), 1e-30)))
        if isinstance(mod, O.SpatialMaxPooling):      # near-tie in a 2x2 block: the argmax (gradient routing) is ill-defined
            n, c, h, w = xin.shape
            blk = np.sort(xin.reshape(n, c, h // 2, 2, w // 2, 2).transpose(0, 1, 2, 4, 3, 5).reshape(n, c, h // 2, w // 2, 4), -1)
            m = min(m, float((blk[..., 3] - blk[..., 2]).min() / max(np.abs(xin).max(), 1e-30)))
    return m


def draw_kink_safe(rng, draw, forward, nets, tries=80, margin=1e-7):
    """Re-draw the random input until no PReLU input of the oracle sits within `margin` (relative) of zero."""
    bns = [m for n in nets for m in n.modules if isinstance(m, O.SpatialBatchNormalization)]
    saved = [(m.running_mean.copy(), m.running_var.copy()) for m in bns]
    for _ in range(tries):
        for m, (rm, rv) in zip(bns, saved):          # every try starts from the same running statistics
            m.running_mean, m.running_var = rm.copy(), rv.copy()
        x = draw()
        y = forward(x)
        if min(prelu_margin(n) for n in nets) > margin:
            return x, y
    raise RuntimeError("could not draw a kink-safe input")


def d_masks(rng, B):
    return [(rng.random((B, c)) < 0.8).astype(np.float32) for c in (64, 128, 256, 512)] + \
           [(rng.random((B, 512)) < 0.5).astype(np.float32) for _ in range(2)]


@pytest.mark.parametrize("C,B", [(3, 4), (1, 6), (3, 18)])
def test_G_forward_backward(ctx, C, B):
    st, Gd, Dd, rng = build(ctx, C, B, seed=100 + C + B)
    noise, img = draw_kink_safe(rng, lambda: rng.uniform(-1, 1, (B, 100)).astype(np.float32), st.G.forward, [st.G])
    gy = rng.standard_normal(img.shape).astype(np.float32)
    dn = Gd.device_net
    y = dn.forward(dev(noise, ctx.device))
    oracle_backward_on_device_branches(ctx, dn, st.G, noise, gy, st.gG)      # the oracle's backward on the device's PReLU decisions
    close(nchw(dn.layer_output(2)), st.G.modules[2].output, atol=2e-5, what="G prelu(view(linear))")
    close(nchw(dn.layer_output(6)), st.G.modules[6].output, atol=5e-5, what="G conv5+bn+prelu")
    close(nchw(dn.layer_output(10)), st.G.modules[10].output, atol=5e-5, what="G conv9+bn+prelu")
    close(nchw(y), img, atol=1e-5, what="G images")                    # bar: 1e-4 (north_star)
    dn.backward(nhwc(gy, ctx.device), param_grads=True, input_grad=False)
    check_flat_grads(dn.grads.cpu().numpy(), st.G, "G")
    # BN running statistics (evaluate-mode state) follow the THNN update
    rm = dn.buffers.cpu().numpy()
    close(rm[:256], st.G.modules[5].running_mean, atol=1e-6, what="running_mean")
    close(rm[256:512], st.G.modules[5].running_var, atol=0, rtol=1e-5, what="running_var")


@pytest.mark.parametrize("C,B", [(3, 4), (1, 6), (3, 18)])
def test_D_forward_backward(ctx, C, B):
    st, Gd, Dd, rng = build(ctx, C, B, seed=200 + C + B)
    masks = d_masks(rng, B)
    O.set_dropout_masks(st.D, masks)
    x, out = draw_kink_safe(rng, lambda: rng.uniform(0, 1, (B, C, 32, 32)).astype(np.float32), st.D.forward, [st.D])
    gy = rng.standard_normal(out.shape).astype(np.float32)
    dn = Dd.device_net
    y = dn.forward(nhwc(x, ctx.device), masks=[dev(m.reshape(-1), ctx.device) for m in masks])
    gx = oracle_backward_on_device_branches(ctx, dn, st.D, x, gy, st.gD)
    close(nchw(dn.layer_output(3)), st.D.modules[3].output, atol=1e-5, what="D block1")
    close(nchw(dn.layer_output(15)), st.D.modules[15].output, atol=2e-5, what="D block4")
    close(y.cpu().numpy(), out, atol=1e-5, what="D probabilities")      # bar: 1e-4
    gxd = dn.backward(dev(gy, ctx.device), param_grads=True, input_grad=True)
    close(nchw(gxd), gx, atol=1e-4 * np.abs(gx).max() + 1e-8, what="D input grad")
    check_flat_grads(dn.grads.cpu().numpy(), st.D, "D")


def test_evaluate_mode_forward(ctx):
    """sample.lua / visualizeProgress path: BN running stats, SpatialDropout x(1-p), Dropout identity."""
    st, Gd, Dd, rng = build(ctx, 3, 8, seed=300)
    noise = rng.uniform(-1, 1, (8, 100)).astype(np.float32)
    st.G.forward(noise)                    # one train-mode pass to move the running stats
    Gd.device_net.forward(dev(noise, ctx.device))
    st.G.evaluate(); st.D.evaluate(); Gd.evaluate(); Dd.evaluate()
    img = st.G.forward(noise)
    p = st.D.forward(img)
    y = Gd.device_net.forward(dev(noise, ctx.device))
    close(nchw(y), img, atol=2e-5, what="G eval images")
    pd = Dd.device_net.forward(y.clone())
    close(pd.cpu().numpy(), p, atol=2e-5, what="D eval probabilities")


@pytest.mark.parametrize("init", ["default", "reference"])
def test_full_D_step_and_G_step(ctx, init):
    """adversarial.lua:240-288 with Adam: outputs, loss, flat gradients, post-Adam parameters."""
    from face_generator_amd import adversarial
    B, C = 8, 3
    st, Gd, Dd, rng = build(ctx, C, B, seed=400, init=init)
    tr = adversarial.Trainer(ctx, Gd, Dd, dict(batchSize=B, noiseDim=100, D_L1=0.0, D_L2=1e-4, G_L1=0.0, G_L2=0.0,
                                              D_clamp=1.0, G_clamp=5.0))
    real = rng.uniform(0, 1, (B // 2, C, 32, 32)).astype(np.float32)
    nz = rng.uniform(-1, 1, (B // 2, 100)).astype(np.float32)
    masks = d_masks(rng, B)
    ref = O.step_D(st, real, nz, masks)
    got = tr.step_D(nhwc(real, ctx.device), dev(nz, ctx.device), [dev(m.reshape(-1), ctx.device) for m in masks],
                    keep_grad=True)
    close(got["outputs"].cpu().numpy().reshape(-1), ref["out"].reshape(-1), atol=1e-5, what="D-step D outputs")
    assert abs(got["loss"].item() - ref["f_bce"]) <= 1e-5 * abs(ref["f_bce"])      # loss: rel 1e-5
    assert abs(got["f"] - ref["f"]) <= 1e-5 * abs(ref["f"])                           # incl. the L2 penalty term
    close(got["grad"].cpu().numpy(), ref["grad"], atol=1e-4 * np.abs(ref["grad"]).max() + 1e-7, what="D-step flat grad")
    close_after_first_adam_step(Dd.getParameters()[0].cpu().numpy(), st.pD, got["grad"].cpu().numpy(), ref["grad"],
                                "D params after Adam")
    assert (got["confusion"].cpu().numpy().reshape(2, 2) == ref["conf"]).all()
    # G-step on the updated D
    nz2 = rng.uniform(-1, 1, (B, 100)).astype(np.float32)
    masks2 = d_masks(rng, B)
    ref = O.step_G(st, nz2, masks2)
    got = tr.step_G(dev(nz2, ctx.device), [dev(m.reshape(-1), ctx.device) for m in masks2], keep_grad=True)
    close(nchw(got["samples"]), ref["samples"], atol=1e-5, what="G-step samples")
    close(got["outputs"].cpu().numpy().reshape(-1), ref["out"].reshape(-1), atol=1e-5, what="G-step D outputs")
    assert abs(got["loss"].item() - ref["f_bce"]) <= 1e-5 * abs(ref["f_bce"])
    close(got["grad"].cpu().numpy(), ref["grad"], atol=1e-4 * np.abs(ref["grad"]).max() + 1e-7, what="G-step flat grad")
    close_after_first_adam_step(Gd.getParameters()[0].cpu().numpy(), st.pG, got["grad"].cpu().numpy(), ref["grad"],
                                "G params after Adam")




def test_G16_forward_backward(ctx):
    """models.lua:27-51 create_G_decoder_upsampling16: same kernels from a 4x4 map."""
    from face_generator_amd import models
    B, C = 6, 3
    rng = np.random.default_rng(950)
    G = O.create_G16((C, 16, 16), 100, rng, weight_init_=False)
    for m in G.modules:
        if isinstance(m, O.SpatialBatchNormalization):
            m.bias[...] = rng.standard_normal(m.bias.shape).astype(np.float32) * 0.2
            m.weight[...] = rng.uniform(0.5, 1.5, m.weight.shape).astype(np.float32)
        if isinstance(m, O.PReLU):
            m.weight[0] = np.float32(rng.uniform(0.1, 0.4))
    pG, gG = G.getParameters()
    Gd = models.create_G((C, 16, 16), 100).cuda(ctx, max_batch=B)
    p, g = Gd.getParameters()
    assert p.numel() == pG.size
    p.copy_(torch.tensor(pG)); Gd.device_net.params_changed()
    noise, img = draw_kink_safe(rng, lambda: rng.uniform(-1, 1, (B, 100)).astype(np.float32), G.forward, [G])
    assert img.shape == (B, C, 16, 16)
    gy = rng.standard_normal(img.shape).astype(np.float32)
    y = Gd.device_net.forward(dev(noise, ctx.device))
    oracle_backward_on_device_branches(ctx, Gd.device_net, G, noise, gy, gG)
    close(nchw(y), img, atol=1e-5, what="G16 images")
    Gd.device_net.backward(nhwc(gy, ctx.device), param_grads=True)
    check_flat_grads(g.cpu().numpy(), G, "G16")


@pytest.mark.parametrize("noiseDim", [50, 99])
def test_G_with_a_noise_dimension_that_is_not_a_multiple_of_4(ctx, noiseDim):
    """train.lua `--noiseDim` takes any value (NN_UTILS.createNoiseInputs, nn_utils.lua:35-39); the first Linear's ragged
    in_features run on a zero-padded copy of the noise batch (these used to be refused)."""
    from face_generator_amd import models
    B, C = 6, 3
    rng = np.random.default_rng(970 + noiseDim)
    G = O.create_G32((C, 32, 32), noiseDim, rng, weight_init_=False)
    for m in G.modules:
        if isinstance(m, O.PReLU):
            m.weight[0] = np.float32(rng.uniform(0.1, 0.4))
    pG, gG = G.getParameters()
    Gd = models.create_G((C, 32, 32), noiseDim).cuda(ctx, max_batch=B)
    p, g = Gd.getParameters()
    assert p.numel() == pG.size
    p.copy_(torch.tensor(pG)); Gd.device_net.params_changed()
    noise, img = draw_kink_safe(rng, lambda: rng.uniform(-1, 1, (B, noiseDim)).astype(np.float32), G.forward, [G])
    gy = rng.standard_normal(img.shape).astype(np.float32)
    y = Gd.device_net.forward(dev(noise, ctx.device))
    gin = oracle_backward_on_device_branches(ctx, Gd.device_net, G, noise, gy, gG)
    close(nchw(y), img, atol=1e-5, what="G images, noiseDim %d" % noiseDim)
    gx = Gd.device_net.backward(nhwc(gy, ctx.device), param_grads=True, input_grad=True)
    close(gx.cpu().numpy().reshape(B, noiseDim), gin.reshape(B, noiseDim), atol=1e-4 * np.abs(gin).max() + 1e-8, what="noise gradient")
    check_flat_grads(g.cpu().numpy(), G, "G(noiseDim %d)" % noiseDim)


def _fill_nontrivial(net, rng):
    for m in O.walk_modules(net):
        if isinstance(m, O.PReLU):
            m.weight[0] = np.float32(rng.uniform(0.1, 0.4))


def test_D16_d_forward_backward(ctx):
    """models.lua:279-316 create_D16_d: ConcatTable{conv branch with two 3x3 stride-2 convs, dense
    branch} -> JoinTable(2) -> Linear -> Sigmoid; three compiled nets on one flat parameter vector."""
    from face_generator_amd import models
    B, C = 8, 3
    rng = np.random.default_rng(960)
    D = O.create_D16_d((C, 16, 16), rng)
    _fill_nontrivial(D, rng)
    pD, gD = D.getParameters()
    Dd = models.create_D((C, 16, 16)).cuda(ctx, max_batch=B)
    p, g = Dd.getParameters()
    assert p.numel() == pD.size
    p.copy_(torch.tensor(pD)); Dd.device_net.params_changed()
    masks = [(rng.random((B, 1024)) < 0.5).astype(np.float32), (rng.random((B, 128)) < 0.5).astype(np.float32)]
    O.set_dropout_masks(D, masks)

    def fwd(x):
        return D.forward(x)
    x = rng.uniform(0, 1, (B, C, 16, 16)).astype(np.float32)
    out = fwd(x)
    gy = rng.standard_normal(out.shape).astype(np.float32)
    gD[...] = 0
    gin = D.backward(x, gy)
    dm = [dev(m.reshape(-1), ctx.device) for m in masks]
    y = Dd.device_net.forward(nhwc(x, ctx.device), masks=dm, train=True)
    close(y.cpu().numpy(), out, atol=1e-5, what="D16_d probabilities")
    gx = Dd.device_net.backward(dev(gy, ctx.device), param_grads=True, input_grad=True)
    close(nchw(gx), gin, atol=1e-4 * np.abs(gin).max() + 1e-7, what="D16_d input gradient")
    got, ref = g.cpu().numpy(), gD
    # per parameter tensor, in flat order.  Floor of 2e-6 of the net's largest gradient entry (the bar of the BASELINE-size tests,
    # tests/test_gpu_baseline_sizes.py::check_every_tensor): a tensor whose own gradient is small inherits the rounding noise of the
    # signal propagated to it -- here the first layer's weight gradient (max 3e-3) behind the 128 -> 128 3x3 layer, whose data
    # gradient is a Winograd F(2x2, 3x3) contraction (1.7 x the direct convolution's fp32 rounding error, tests/test_gpu_wino.py)
    off, gmax = 0, np.abs(ref).max()
    for (m, pn, gn) in D.parameters():
        r = getattr(m, gn).reshape(-1)
        e = np.abs(got[off:off + r.size] - r).max()
        tol = 1e-4 * np.abs(r).max() + 1e-7 + (32 * 6e-8 * getattr(m, "gw_cond", 0.0) if isinstance(m, O.PReLU) else 0.0)
        tol = max(tol, 2e-6 * gmax)
        assert e <= tol, "D16_d %s.%s: err %.3e tol %.3e (max|g_tensor| %.3e, max|g_net| %.3e)" % (type(m).__name__, pn, e, tol, np.abs(r).max(), gmax)
        off += r.size
    # evaluate mode: dropout off / SpatialDropout scaling
    D.evaluate(); Dd.evaluate()
    close(Dd.device_net.forward(nhwc(x, ctx.device)).cpu().numpy(), D.forward(x), atol=1e-5, what="D16_d evaluate")


def test_train_step_16px(ctx):
    """--scale 16 path: one D-step + G-step with G16 / D16_d against the oracle step."""
    from face_generator_amd import models, adversarial
    B, C = 8, 3
    rng = np.random.default_rng(961)
    G = O.create_G16((C, 16, 16), 100, rng, weight_init_=False); D = O.create_D16_d((C, 16, 16), rng)
    st = O.GanState(G, D)
    Gd = models.create_G((C, 16, 16), 100).cuda(ctx, max_batch=B)
    Dd = models.create_D((C, 16, 16)).cuda(ctx, max_batch=B)
    Gd.getParameters()[0].copy_(torch.tensor(st.pG)); Dd.getParameters()[0].copy_(torch.tensor(st.pD))
    Gd.device_net.params_changed(); Dd.device_net.params_changed()
    tr = adversarial.Trainer(ctx, Gd, Dd, dict(batchSize=B, noiseDim=100))
    real = rng.uniform(0, 1, (B // 2, C, 16, 16)).astype(np.float32)
    nz = rng.uniform(-1, 1, (B // 2, 100)).astype(np.float32)
    masks = [(rng.random((B, 1024)) < 0.5).astype(np.float32), (rng.random((B, 128)) < 0.5).astype(np.float32)]
    ref = O.step_D(st, real, nz, masks)
    got = tr.step_D(nhwc(real, ctx.device), dev(nz, ctx.device), [dev(m.reshape(-1), ctx.device) for m in masks], keep_grad=True)
    close(got["outputs"].cpu().numpy().reshape(-1), ref["out"].reshape(-1), atol=1e-5, what="16px D-step outputs")
    assert abs(got["loss"].item() - ref["f_bce"]) <= 1e-5 * abs(ref["f_bce"])
    close(got["grad"].cpu().numpy(), ref["grad"], atol=1e-4 * np.abs(ref["grad"]).max() + 1e-7, what="16px D-step flat grad")
    nz2 = rng.uniform(-1, 1, (B, 100)).astype(np.float32)
    ref = O.step_G(st, nz2, masks)
    got = tr.step_G(dev(nz2, ctx.device), [dev(m.reshape(-1), ctx.device) for m in masks])
    close(nchw(got["samples"]), ref["samples"], atol=1e-5, what="16px G-step samples")
    close(got["outputs"].cpu().numpy().reshape(-1), ref["out"].reshape(-1), atol=1e-5, what="16px G-step D outputs")


@pytest.mark.parametrize("C,H,W,nout", [(20, 5, 3, 12), (12, 4, 4, 36), (64, 3, 7, 8)])   # (outputs: multiples of 4, the weight-gradient kernel's alignment)
def test_linear_behind_a_view_with_ragged_feature_dims(ctx, C, H, W, nout):
    """The re-pack of a Linear behind a View moves 16 x 16 x 16 bricks (out, channel, pixel) of the weight (pack mode 10): channel
    counts, map sizes and output counts that are NOT multiples of 16 exercise every edge of the bricks and the zero padding of
    both packs.  conv 3x3 -> View(C*H*W) -> Linear (models.lua:404-406 is this shape with C = 512, 4 x 4, 512 outputs)."""
    from face_generator_amd import nn
    B, cin = 6, 4
    rng = np.random.default_rng(1200 + C + nout)
    net = O.Sequential()
    net.add(O.SpatialConvolution(cin, C, 3, 3, 1, 1, 1, rng=rng))
    net.add(O.View(C * H * W))
    net.add(O.Linear(C * H * W, nout, rng=rng))
    pO, gO = net.getParameters()
    dn = nn.Sequential()
    dn.add(nn.SpatialConvolution(cin, C, 3, 3, 1, 1, 1))
    dn.add(nn.View(C * H * W))
    dn.add(nn.Linear(C * H * W, nout))
    dn.input_dims = (cin, H, W)
    dn.cuda(ctx, max_batch=B)
    p, g = dn.getParameters()
    assert p.numel() == pO.size
    for it in range(2):                                   # second round: weights changed -> the re-pack runs again
        if it:
            pO += rng.standard_normal(pO.shape).astype(np.float32) * 0.05
        p.copy_(torch.tensor(pO)); dn.device_net.params_changed()
        x = rng.uniform(-1, 1, (B, cin, H, W)).astype(np.float32)
        out = net.forward(x)
        gy = rng.standard_normal(out.shape).astype(np.float32)
        gO[...] = 0
        gin = net.backward(x, gy)
        y = dn.device_net.forward(nhwc(x, ctx.device))
        close(y.cpu().numpy(), out, atol=1e-5 * max(1.0, np.abs(out).max()), what="Linear behind a View: outputs")
        gx = dn.device_net.backward(dev(gy, ctx.device), param_grads=True, input_grad=True)
        close(nchw(gx), gin, atol=1e-5 * np.abs(gin).max() + 1e-7, what="Linear behind a View: input gradient")
        close(g.cpu().numpy(), gO, atol=1e-5 * np.abs(gO).max() + 1e-7, what="Linear behind a View: parameter gradients")
