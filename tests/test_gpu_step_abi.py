"""Step-level C entries (fg_step_D / fg_step_G, include/facegen_hip.h) against the host-driven closures built from the
net-level entries: same kernels, same order, so the results must be BIT-identical on the same inputs, noise and masks;
plus what only the fused path does -- noise and every dropout mask of a closure from one Philox launch, G's last stage
writing straight into D's batch, the gate's two-phase update (FG_STEP_NO_UPDATE + fg_gan_update)."""
import ctypes

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    from face_generator_amd.runtime import get_context
    return get_context(0)


def make32(ctx, B, opt=None, fused=True, seed=3):
    from face_generator_amd import models, nn_utils, adversarial
    gen = torch.Generator().manual_seed(seed)
    G = models.create_G((3, 32, 32), 100); D = models.create_D((3, 32, 32))
    nn_utils.initializeWeights(D, 0.05, 0.01, gen=gen); nn_utils.initializeWeights(G, 0.05, 0.01, gen=gen)
    G.cuda(ctx, max_batch=B); D.cuda(ctx, max_batch=B)
    tr = adversarial.Trainer(ctx, G, D, dict(dict(batchSize=B, noiseDim=100), **(opt or {})))
    if not fused:
        tr.gan = None
    return tr, G, D


def masks32(ctx, B, seed):
    return [ctx.bernoulli((B * c,), 0.8, seed, i * 100000) for i, c in enumerate((64, 128, 256, 512))] + \
           [ctx.bernoulli((B * 512,), 0.5, seed + 1, i * 100000) for i in range(2)]


@pytest.mark.parametrize("method", ["adam", "sgd", "adagrad"])
def test_fused_closures_equal_host_driven_closures(ctx, method):
    B = 8
    opt = dict(D_optmethod=method, G_optmethod=method, D_SGD_momentum=0.5, G_SGD_momentum=0.5, D_L1=1e-5, D_L2=1e-4, G_L2=1e-5)
    outs = []
    for fused in (True, False):
        tr, G, D = make32(ctx, B, opt, fused)
        assert (tr.gan is not None) == fused
        real = ctx.uniform((B // 2, 32, 32, 3), 0.0, 1.0, seed=9)
        for it in range(3):      # three iterations: optimizer state carries over (Adam t, momentum buffer, variance)
            r1 = tr.step_D(real, ctx.uniform((B // 2, 100), -1.0, 1.0, seed=10 + it), masks32(ctx, B, 20 + it))
            d_out, d_loss, d_conf = r1["outputs"].clone(), r1["loss"].clone(), r1["confusion"].clone()
            r2 = tr.step_G(ctx.uniform((B, 100), -1.0, 1.0, seed=40 + it), masks32(ctx, B, 50 + it))
        tr.finish_pending()
        outs.append(dict(pG=G.getParameters()[0].clone(), pD=D.getParameters()[0].clone(), d_out=d_out, d_loss=d_loss, d_conf=d_conf,
                         g_out=r2["outputs"].clone(), g_loss=r2["loss"].clone(), samples=r2["samples"].clone(),
                         bn=G.device_net.buffers.clone()))
    for k in outs[0]:
        assert torch.equal(outs[0][k], outs[1][k]), "fused and host-driven closures differ in %s (%s)" % (k, method)


def test_fused_c2f_closures_equal_host_driven_closures(ctx):
    from face_generator_amd import models_c2f, adversarial_c2f
    S, B = 16, 4
    outs = []
    for fused in (True, False):
        gen = torch.Generator().manual_seed(5)
        G = models_c2f.create_G((3, S, S), gen=gen).cuda(ctx, max_batch=B)
        D = models_c2f.create_D((3, S, S), gen=gen).cuda(ctx, max_batch=B)
        tr = adversarial_c2f.TrainerC2F(ctx, G, D, dict(batchSize=B))
        if not fused:
            tr.gan = None
        assert (tr.gan is not None) == fused
        u = lambda shape, lo, hi, seed: ctx.uniform(shape, lo, hi, seed=seed)
        masks = [ctx.bernoulli((B * 256 * (S // 4) ** 2,), 0.5, 7), ctx.bernoulli((B * 512,), 0.5, 8)]
        for it in range(2):
            r1 = tr.step_D(u((B // 2, S, S, 3), -1, 1, 11), u((B // 2, S, S, 3), 0, 1, 12), u((B // 2, S, S, 1), -1, 1, 13 + it),
                           u((B // 2, S, S, 3), 0, 1, 14), masks)
            d_out = r1["outputs"].clone()
            r2 = tr.step_G(u((B, S, S, 1), -1, 1, 15 + it), u((B, S, S, 3), 0, 1, 16), masks)
        outs.append(dict(pG=G.getParameters()[0].clone(), pD=D.getParameters()[0].clone(), d_out=d_out,
                         g_out=r2["outputs"].clone(), samples=r2["samples"].clone()))
    for k in outs[0]:
        assert torch.equal(outs[0][k], outs[1][k]), "fused and host-driven c2f closures differ in %s" % k


def test_library_drawn_noise_and_masks(ctx):
    """noise = masks = NULL: one Philox launch per closure; seeded, advancing, correctly distributed."""
    B = 64
    runs = []
    for rep in range(2):
        tr, G, D = make32(ctx, B)
        # the default dropout-mask key is unique per net of the process (runtime.DeviceNet: tag + construction count), so that no two
        # nets ever share a stream; "the same seeds" means the same keys handed to fg_gan_set_seeds
        from face_generator_amd.state import S
        tr.gan.set_seeds(S.noise_seed, S.noise_offset, 777, 0)
        real = ctx.uniform((B // 2, 32, 32, 3), 0.0, 1.0, seed=9)
        r = tr.step_D(real, None)
        nz1 = r["noise"].clone(); m1 = [m.clone() for m in r["masks"]]
        r = tr.step_G(B)
        nz2 = r["noise"].clone(); m2 = [m.clone() for m in r["masks"]]
        runs.append((nz1, m1, nz2, m2, D.getParameters()[0].clone()))
    nz1, m1, nz2, m2, _ = runs[0]
    assert nz1.numel() == B // 2 * 100 and nz2.numel() == B * 100
    for nz in (nz1, nz2):
        assert float(nz.min()) >= -1 and float(nz.max()) < 1 and abs(float(nz.mean())) < 0.05 and abs(float(nz.var()) - 1 / 3) < 0.03
    assert not torch.equal(nz1, nz2[:nz1.numel()])                                   # the stream advances
    for i, (a, b) in enumerate(zip(m1, m2)):
        keep = 0.8 if i < 4 else 0.5
        for m in (a, b):
            assert set(m.unique().tolist()) <= {0.0, 1.0} and abs(float(m.mean()) - keep) < 0.03
        assert not torch.equal(a, b)
    assert [m.numel() for m in m1] == [B * c for c in (64, 128, 256, 512, 512, 512)]
    for x, y in zip(runs[0][:1] + runs[0][2:3], runs[1][:1] + runs[1][2:3]):          # same seeds -> the same draws
        assert torch.equal(x, y)
    assert torch.equal(runs[0][4], runs[1][4])


def test_gate_two_phase_update_and_errors(ctx):
    from face_generator_amd import FgError
    B = 8
    tr, G, D = make32(ctx, B)
    gan = tr.gan
    real = ctx.uniform((B // 2, 32, 32, 3), 0.0, 1.0, seed=9)
    p0 = D.getParameters()[0].clone()
    gan.step_D(B, real, None, None, None, None, gan.NO_UPDATE)
    conf = gan.view("CONFUSION").view(torch.int32).tolist()
    assert sum(conf[:4]) == B and conf[:4] == conf[4:8]                             # one rank: global == local counts
    assert torch.equal(D.getParameters()[0], p0) and gan.steps(0) == 0                # no update, no step count
    g_before = D.getParameters()[1].clone()
    gan.update(0)
    assert not torch.equal(D.getParameters()[0], p0) and gan.steps(0) == 1
    assert torch.equal(D.getParameters()[1], g_before)                                # Adam does not touch the gradient vector
    with pytest.raises(FgError):
        gan.update(0)                                                                 # nothing pending any more
    with pytest.raises(FgError):
        gan.step_D(B + 2, real)                                                       # larger than the step object
    assert ctx.lib.fg_step_D(gan.h, 7, real.data_ptr(), None, None, None, None, 0) < 0     # odd batch
    assert ctx.lib.fg_step_D(gan.h, B, None, None, None, None, None, 0) < 0               # no real images
    assert ctx.lib.fg_gan_set_optimizer(gan.h, 0, 5, -1.0, 0.9, 0.999, 1e-8, 0.0, -1.0, 0.0, 0.0, 0) < 0
    off, cnt = ctypes.c_longlong(), ctypes.c_longlong()
    assert ctx.lib.fg_gan_buffer(gan.h, 99, ctypes.byref(off), ctypes.byref(cnt)) < 0
    # samples of a G-step are D's batch itself (G's last stage wrote them there)
    r = tr.step_G(B)
    assert r["samples"].data_ptr() == gan.view("D_INPUT").data_ptr()
    img = r["samples"]
    assert float(img.min()) > 0 and float(img.max()) < 1 and tuple(img.shape) == (B, 32, 32, 3)
