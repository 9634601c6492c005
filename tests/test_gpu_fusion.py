"""fg_set_fusion (include/facegen_hip.h FG_FUSE_*): the optional kernel fusions give the results of the un-fused path.
* FG_FUSE_PRELU -- an nn.PReLU between two contraction layers of the coarse-to-fine nets (models_c2f.lua:118-130, 242-255)
  rides on the neighbouring kernels' epilogues.  Forward outputs, input gradients and every weight / bias gradient are the
  SAME fp32 operations in both paths (bit-identical); only a PReLU slope gradient is summed in a different order.
* FG_FUSE_THIN_SLAB -- the 3x3 thin-output convolution (models.lua:73 forward, :385 data gradient) on the matrix pipe vs
  the sliding-window VALU kernel: different summation order, compared at the SURVEY 8(c) forward bar, and both against the
  oracle."""
import numpy as np
import pytest
import torch

from oracle import torch7_nn as O
from gpu_util import nhwc, nchw, dev, close
from test_gpu_c2f import build, masks_for, dev_masks

pytestmark = pytest.mark.gpu

FG_FUSE_PRELU, FG_FUSE_THIN_SLAB, FG_FUSE_WFINISH_BATCH, FG_FUSE_ADAM_PACK, FG_FUSE_THIN_BIAS, FG_FUSE_WINOGRAD, FG_FUSE_ALL, FG_FUSE_DEFAULT = 1, 2, 4, 8, 16, 32, 511, 503


@pytest.fixture(scope="module")
def ctx():
    from face_generator_amd.runtime import get_context
    c = get_context(0)
    yield c
    c.set_fusion(FG_FUSE_DEFAULT)


def test_fusion_flags_roundtrip_and_reject_unknown_bits(ctx):
    from face_generator_amd._lib import FgError
    ctx.set_fusion(FG_FUSE_THIN_SLAB)
    assert ctx.get_fusion() == FG_FUSE_THIN_SLAB
    with pytest.raises(FgError):
        ctx.set_fusion(512)
    ctx.set_fusion(FG_FUSE_DEFAULT)
    assert ctx.get_fusion() == FG_FUSE_DEFAULT


@pytest.mark.parametrize("S,B", [(16, 4), (64, 8)])
def test_prelu_in_epilogue_equals_separate_passes(ctx, S, B):
    """c2f G_d and D_c forward + backward with the PReLUs folded into the contraction epilogues vs as passes of their own.
    (64, 8) selects the wave-specialised kernels of the BASELINE size, (16, 4) the 64x64-tile kernels."""
    st, Gd, Dd, rng = build(ctx, S, B, seed=900 + S)
    d = ctx.device
    cond = rng.uniform(0, 1, (B, 3, S, S)).astype(np.float32)
    noise = rng.uniform(-1, 1, (B, 1, S, S)).astype(np.float32)
    x = rng.uniform(-1, 1, (B, 3, S, S)).astype(np.float32)
    gy = rng.standard_normal((B, 3, S, S)).astype(np.float32)
    gyo = rng.standard_normal((B, 1)).astype(np.float32)
    masks = dev_masks(masks_for(rng, B, S), d)
    res = {}
    for flags in (FG_FUSE_ALL, FG_FUSE_ALL & ~FG_FUSE_PRELU):
        ctx.set_fusion(flags)
        dn = Gd.inner.device_net
        y = dn.forward(Gd.combine_device(ctx, nhwc(noise, d), nhwc(cond, d))).clone()
        dn.backward(nhwc(gy, d), param_grads=True)
        gG = dn.grads.clone()
        dnD = Dd.inner.device_net
        yd = dnD.forward(Dd.combine_device(ctx, nhwc(x, d), nhwc(cond, d)), masks=masks).clone()
        gx = dnD.backward(dev(gyo, d), param_grads=True, input_grad=True).clone()
        gD = dnD.grads.clone()
        res[flags] = (y, gG, yd, gx, gD)
    ctx.set_fusion(FG_FUSE_DEFAULT)
    a, b = res[FG_FUSE_ALL], res[FG_FUSE_ALL & ~FG_FUSE_PRELU]
    assert torch.equal(a[0], b[0]), "G output"
    assert torch.equal(a[2], b[2]), "D output"
    assert torch.equal(a[3], b[3]), "D gradInput"
    for name, net, ga, gb in (("G", st.G.inner, a[1], b[1]), ("D", st.D.inner, a[4], b[4])):
        off = 0
        for m in net.modules:
            for (mm, pn, gn) in m.parameters():
                n = getattr(mm, pn).size
                da, db = ga[off:off + n], gb[off:off + n]
                if isinstance(mm, O.PReLU):     # a 10^5..10^7-term cancelling sum, reduced in a different order
                    scale = float(max(da.abs().max(), db.abs().max(), 1e-30))
                    assert float((da - db).abs().max()) <= 2e-3 * scale + 1e-6, (name, type(mm).__name__, float(da), float(db))
                else:
                    assert torch.equal(da, db), (name, type(mm).__name__, pn, off)
                off += n
        assert off == ga.numel()


@pytest.mark.parametrize("B,H,W,Cw,Cs,flip", [(3, 32, 32, 128, 3, 0), (128, 32, 32, 128, 3, 0), (5, 16, 16, 64, 1, 0),
                                               (2, 64, 64, 64, 3, 1), (4, 8, 8, 128, 3, 1)])
def test_thin_output_3x3_slab_kernel_equals_window_kernel_and_oracle(ctx, B, H, W, Cw, Cs, flip):
    """flip = 0: forward of a Cw -> Cs convolution; flip = 1: the data gradient of a Cs -> Cw convolution (the same kernel
    with mirrored taps), through the module-level entries."""
    from face_generator_amd import ops
    rng = np.random.default_rng(B * 1000 + H + Cw + flip)
    d = ctx.device
    if not flip:
        x = rng.standard_normal((B, Cw, H, W)).astype(np.float32)
        w = (rng.standard_normal((Cs, Cw, 3, 3)) * 0.1).astype(np.float32)
        bias = rng.standard_normal(Cs).astype(np.float32)
        conv = O.SpatialConvolution(Cw, Cs, 3, 3, 1, 1, 1, 1, rng)
        conv.weight[...] = w; conv.bias[...] = bias
        ref = conv.forward(x)
        run = lambda: nchw(ops.conv2d_forward(nhwc(x, d), dev(w, d), dev(bias, d)))
    else:
        g = rng.standard_normal((B, Cw, H, W)).astype(np.float32)
        w = (rng.standard_normal((Cw, Cs, 3, 3)) * 0.1).astype(np.float32)
        conv = O.SpatialConvolution(Cs, Cw, 3, 3, 1, 1, 1, 1, rng)
        conv.weight[...] = w
        xin = rng.standard_normal((B, Cs, H, W)).astype(np.float32)
        conv.forward(xin)
        ref = conv.backward(xin, g)
        run = lambda: nchw(ops.conv2d_backward_data(nhwc(g, d), dev(w, d), (H, W)))
    out = {}
    for flags in (FG_FUSE_ALL, FG_FUSE_ALL & ~FG_FUSE_THIN_SLAB):
        ctx.set_fusion(flags)
        out[flags] = run()
    ctx.set_fusion(FG_FUSE_DEFAULT)
    tol = 2e-5 * max(1.0, float(np.abs(ref).max()))
    close(out[FG_FUSE_ALL], ref, atol=tol, what="slab kernel vs oracle")
    close(out[FG_FUSE_ALL & ~FG_FUSE_THIN_SLAB], ref, atol=tol, what="window kernel vs oracle")
    close(out[FG_FUSE_ALL], out[FG_FUSE_ALL & ~FG_FUSE_THIN_SLAB], atol=tol, what="slab vs window kernel")


@pytest.mark.parametrize("B", [6, 128])
def test_cfg2_nets_fused_equal_unfused(ctx, B):
    """The 32x32 nets (models.lua:57-81, 382-416): the PReLU [+ Dropout] backward behind D's Linear layers and G's first
    Linear rides on the pass that sums the next layer's split-K data-gradient partials; G's first Linear runs un-split with
    its PReLU in the epilogue.  Same operations on the same values: everything but the slope gradients is bit-identical."""
    from test_gpu_net import build as build32, d_masks
    st, Gd, Dd, rng = build32(ctx, 3, B, seed=700 + B)
    d = ctx.device
    noise = rng.uniform(-1, 1, (B, 100)).astype(np.float32)
    x = rng.uniform(0, 1, (B, 3, 32, 32)).astype(np.float32)
    gy = rng.standard_normal((B, 3, 32, 32)).astype(np.float32)
    gyo = rng.standard_normal((B, 1)).astype(np.float32)
    masks = [dev(m.reshape(-1), d) for m in d_masks(rng, B)]
    res = {}
    for flags in (FG_FUSE_ALL, FG_FUSE_ALL & ~FG_FUSE_PRELU):
        ctx.set_fusion(flags)
        dn = Gd.device_net
        y = dn.forward(dev(noise, d)).clone()
        dn.backward(nhwc(gy, d), param_grads=True, input_grad=False)
        gG = dn.grads.clone()
        dnD = Dd.device_net
        yd = dnD.forward(nhwc(x, d), masks=masks).clone()
        gx = dnD.backward(dev(gyo, d), param_grads=True, input_grad=True).clone()
        res[flags] = (y, gG, yd, gx, dnD.grads.clone())
    ctx.set_fusion(FG_FUSE_DEFAULT)
    a, b = res[FG_FUSE_ALL], res[FG_FUSE_ALL & ~FG_FUSE_PRELU]
    assert torch.equal(a[0], b[0]) and torch.equal(a[2], b[2]) and torch.equal(a[3], b[3])
    for name, net, ga, gb in (("G", st.G, a[1], b[1]), ("D", st.D, a[4], b[4])):
        off = 0
        for m in net.modules:
            for (mm, pn, gn) in m.parameters():
                n = getattr(mm, pn).size
                da, db = ga[off:off + n], gb[off:off + n]
                if isinstance(mm, O.PReLU):
                    scale = float(max(da.abs().max(), db.abs().max(), 1e-30))
                    assert float((da - db).abs().max()) <= 2e-3 * scale + 1e-6, (name, off, float(da), float(db))
                else:
                    assert torch.equal(da, db), (name, type(mm).__name__, pn, off)
                off += n
        assert off == ga.numel()


@pytest.mark.parametrize("B", [8, 64])
def test_batched_weight_gradient_sums_and_adam_in_the_repack_are_bit_identical(ctx, B):
    """FG_FUSE_WFINISH_BATCH (all split-K / parity sums of a backward pass in one launch, partials kept in the workspace) and
    FG_FUSE_ADAM_PACK (penalty + clamp + Adam inside the re-pack launch) change neither the order of any addition nor any
    rounding: three whole iterations of the 32x32 step (adversarial.lua:69-257) give the same bits with each of them off."""
    from test_gpu_step_abi import make32, masks32
    opt = dict(D_L1=1e-5, D_L2=1e-4, G_L2=1e-5)
    outs = {}
    for flags in (FG_FUSE_ALL, FG_FUSE_ALL & ~FG_FUSE_WFINISH_BATCH, FG_FUSE_ALL & ~FG_FUSE_ADAM_PACK,
                  FG_FUSE_ALL & ~(FG_FUSE_WFINISH_BATCH | FG_FUSE_ADAM_PACK)):
        ctx.set_fusion(flags)
        tr, G, D = make32(ctx, B, opt, True, seed=11)
        assert tr.gan is not None
        real = ctx.uniform((B // 2, 32, 32, 3), 0.0, 1.0, seed=9)
        for it in range(3):
            r1 = tr.step_D(real, ctx.uniform((B // 2, 100), -1.0, 1.0, seed=10 + it), masks32(ctx, B, 20 + it))
            d_loss = r1["loss"].clone()
            r2 = tr.step_G(ctx.uniform((B, 100), -1.0, 1.0, seed=40 + it), masks32(ctx, B, 50 + it))
        tr.finish_pending()
        # one more forward through the RE-PACKED weights: a wrong pack would not show in the flat vectors
        y = G.device_net.forward(ctx.uniform((B, 100), -1.0, 1.0, seed=77)).clone()
        outs[flags] = dict(pG=G.getParameters()[0].clone(), pD=D.getParameters()[0].clone(), gG=G.getParameters()[1].clone(),
                           gD=D.getParameters()[1].clone(), d_loss=d_loss, g_loss=r2["loss"].clone(), samples=r2["samples"].clone(), y=y)
    ctx.set_fusion(FG_FUSE_DEFAULT)
    ref = outs[FG_FUSE_ALL]
    for flags, o in outs.items():
        for k in ref:
            assert torch.equal(ref[k], o[k]), "fusion flags %d: %s differs from the all-on run" % (flags, k)


def test_c2f_batched_weight_gradient_sums_and_adam_in_the_repack_are_bit_identical(ctx):
    """the same on the coarse-to-fine nets (thin 3-channel layers with their own packs, a Linear behind a View, PReLU slopes
    and biases that no pack reads: the update-only jobs of the fused launch)"""
    from face_generator_amd import models_c2f, adversarial_c2f
    S, B = 16, 8
    outs = {}
    for flags in (FG_FUSE_ALL, FG_FUSE_ALL & ~FG_FUSE_WFINISH_BATCH, FG_FUSE_ALL & ~FG_FUSE_ADAM_PACK):
        ctx.set_fusion(flags)
        gen = torch.Generator().manual_seed(5)
        G = models_c2f.create_G((3, S, S), gen=gen).cuda(ctx, max_batch=B)
        D = models_c2f.create_D((3, S, S), gen=gen).cuda(ctx, max_batch=B)
        tr = adversarial_c2f.TrainerC2F(ctx, G, D, dict(batchSize=B))
        assert tr.gan is not None
        u = lambda shape, lo, hi, seed: ctx.uniform(shape, lo, hi, seed=seed)
        masks = [ctx.bernoulli((B * 256 * (S // 4) ** 2,), 0.5, 7), ctx.bernoulli((B * 512,), 0.5, 8)]
        for it in range(2):
            tr.step_D(u((B // 2, S, S, 3), -1, 1, 11), u((B // 2, S, S, 3), 0, 1, 12), u((B // 2, S, S, 1), -1, 1, 13 + it),
                      u((B // 2, S, S, 3), 0, 1, 14), masks)
            r2 = tr.step_G(u((B, S, S, 1), -1, 1, 15 + it), u((B, S, S, 3), 0, 1, 16), masks)
        outs[flags] = dict(pG=G.getParameters()[0].clone(), pD=D.getParameters()[0].clone(), gG=G.getParameters()[1].clone(),
                           gD=D.getParameters()[1].clone(), samples=r2["samples"].clone(), g_out=r2["outputs"].clone())
    ctx.set_fusion(FG_FUSE_DEFAULT)
    ref = outs[FG_FUSE_ALL]
    for flags, o in outs.items():
        for k in ref:
            assert torch.equal(ref[k], o[k]), "fusion flags %d: %s differs from the all-on run (c2f)" % (flags, k)


@pytest.mark.parametrize("which,S,B", [("cfg2", 32, 16), ("cfg2", 32, 128), ("c2f", 64, 8)])
def test_bias_gradient_from_the_thin_weight_gradient_kernel(ctx, which, S, B):
    """FG_FUSE_THIN_BIAS (round 4): the bias gradient of a convolution with <= 4 input channels (D's first layer, models.lua:385;
    both first layers of the c2f nets, models_c2f.lua:123, 244) is row k*k*Cin of the weight-gradient slabs -- the idle column of
    the (tap, channel) axis multiplied by 1 -- instead of a separate column-sum pass over the output gradient.  Every other entry of
    the flat gradient is bit-identical either way; the bias gradient agrees to fp32 summation-order rounding (both ways end in the
    same fp64 final) and with the oracle at the plain bar (the full-size parity tests run with the bit on)."""
    outs = {}
    for flags in (FG_FUSE_ALL, FG_FUSE_ALL & ~FG_FUSE_THIN_BIAS):
        ctx.set_fusion(flags)
        if which == "cfg2":
            from test_gpu_step_abi import make32, masks32
            tr, G, D = make32(ctx, B, dict(D_L1=0.0, D_L2=0.0), True, seed=21)
            real = ctx.uniform((B // 2, 32, 32, 3), 0.0, 1.0, seed=9)
            r = tr.step_D(real, ctx.uniform((B // 2, 100), -1.0, 1.0, seed=10), masks32(ctx, B, 20), keep_grad=True)
            nets = [D]
        else:
            from face_generator_amd import models_c2f, adversarial_c2f
            gen = torch.Generator().manual_seed(5)
            G = models_c2f.create_G((3, S, S), gen=gen).cuda(ctx, max_batch=B)
            D = models_c2f.create_D((3, S, S), gen=gen).cuda(ctx, max_batch=B)
            tr = adversarial_c2f.TrainerC2F(ctx, G, D, dict(batchSize=B))
            u = lambda shape, lo, hi, seed: ctx.uniform(shape, lo, hi, seed=seed)
            masks = [ctx.bernoulli((B * 256 * (S // 4) ** 2,), 0.5, 7), ctx.bernoulli((B * 512,), 0.5, 8)]
            pD0 = D.getParameters()[0].clone()
            tr.step_D(u((B // 2, S, S, 3), -1, 1, 11), u((B // 2, S, S, 3), 0, 1, 12), u((B // 2, S, S, 1), -1, 1, 13),
                      u((B // 2, S, S, 3), 0, 1, 14), masks, keep_grad=True)
            # the G closure must see the SAME discriminator either way: D's bias gradient legitimately differs in the last bits
            # between the two modes, its Adam step would carry that into every gradient of G
            D.getParameters()[0].copy_(pD0); D.inner.device_net.params_changed()
            tr.step_G(u((B, S, S, 1), -1, 1, 15), u((B, S, S, 3), 0, 1, 16), masks, keep_grad=True)
            nets = [D, G]
        outs[flags] = [n.getParameters()[1].clone() for n in nets] + [nets]
    ctx.set_fusion(FG_FUSE_DEFAULT)
    on, off = outs[FG_FUSE_ALL], outs[FG_FUSE_ALL & ~FG_FUSE_THIN_BIAS]
    for g_on, g_off, net in zip(on[:-1], off[:-1], on[-1]):
        inner = getattr(net, "inner", net)
        first = inner.modules[0]
        dn = inner.device_net
        wo, wn, bo, bn = dn.param_offsets(0)
        assert bn == first.bias.numel() and first.nInputPlane <= 4
        same = torch.ones_like(g_on, dtype=torch.bool)
        same[bo:bo + bn] = False
        assert torch.equal(g_on[same], g_off[same]), "a gradient entry outside the first layer's bias moved"
        b_on, b_off = g_on[bo:bo + bn].double(), g_off[bo:bo + bn].double()
        scale = float(b_off.abs().max())
        assert scale > 0 and float((b_on - b_off).abs().max()) <= 2e-6 * scale + 1e-12, (float((b_on - b_off).abs().max()), scale)
