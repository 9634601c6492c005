"""PyTorch-CPU restatement of the GAN iteration -- the CPU BASELINE of SURVEY.md 8(d), not a second oracle.

TEST INFRASTRUCTURE ONLY (like oracle/torch7_nn.py): imported by bench.py's `cpu_baseline` leg and by tests/ only.

The reference's Lua/Torch `nn` CPU path cannot be executed here (no Lua, un-vendored rocks -- SURVEY F6).  ATen is the
direct descendant of TH/THNN, so the same iteration is run through `torch.nn.functional` on the host cores (fp32, oneDNN
convolutions, `torch.set_num_threads(nproc)`), with the Torch7 semantics that differ from PyTorch's restated by hand:
BCECriterion's eps = 1e-12 (train.lua:148), SpatialDropout without the 1/(1-p) rescale (models.lua:387), Torch7 Adam with eps
added before the bias correction (interruptable_optimizers.lua:49-94), penalty / clamp on the flat gradient
(adversarial.lua:103-123).  The nets are taken from an oracle net (same module list, same parameter order), so
`tests/test_oracle.py` can pin this file against the numpy oracle on identical parameters, inputs and masks.
"""
import numpy as np
import torch
import torch.nn.functional as F

from . import torch7_nn as O


class TorchNet:
    """An oracle Sequential / TableNet executed with torch ops; parameters in Module:parameters() order."""

    def __init__(self, net):
        self.first = net.first if isinstance(net, O.TableNet) else None
        self.modules = list(net.inner.modules if isinstance(net, O.TableNet) else net.modules)
        self.params = []
        self.slots = {}
        for mi, m in enumerate(self.modules):
            for (mm, pn, gn) in m.parameters():
                t = torch.tensor(np.asarray(getattr(mm, pn), np.float32).copy(), requires_grad=True)
                self.slots[(id(mm), pn)] = len(self.params)
                self.params.append(t)
        self.bn = {id(m): (torch.tensor(m.running_mean.copy()), torch.tensor(m.running_var.copy()))
                   for m in self.modules if isinstance(m, O.SpatialBatchNormalization)}
        self.n_params = sum(p.numel() for p in self.params)

    def p(self, m, name):
        return self.params[self.slots[(id(m), name)]]

    def flat(self, grads=False):
        return torch.cat([(p.grad if grads else p.detach()).reshape(-1) for p in self.params])

    def set_flat(self, v):
        off = 0
        with torch.no_grad():
            for p in self.params:
                p.copy_(v[off:off + p.numel()].view_as(p))
                off += p.numel()

    def zero_grad(self):
        for p in self.params:
            p.grad = None

    def forward(self, x, masks=None, train=True):
        if self.first is not None:
            x = torch.cat(list(x), 1) if isinstance(self.first, O.JoinTable) else x[0] + x[1]
        mi = iter(masks) if masks is not None else None
        for m in self.modules:
            if isinstance(m, O.Linear):
                x = F.linear(x, self.p(m, 'weight'), self.p(m, 'bias'))
            elif isinstance(m, O.View):
                x = x.reshape((x.shape[0],) + tuple(m.shape))
            elif isinstance(m, O.PReLU):
                x = F.prelu(x, self.p(m, 'weight'))
            elif isinstance(m, O.SpatialUpSamplingNearest):
                x = F.interpolate(x, scale_factor=2, mode='nearest')
            elif isinstance(m, O.SpatialConvolution):          # incl. SpatialConvolutionUpsample with factor 1
                x = F.conv2d(x, self.p(m, 'weight'), self.p(m, 'bias'), stride=(m.dh, m.dw), padding=(m.padh, m.padw))
                f = getattr(m, 'factor', 1)
                if f != 1:
                    n, c, h, w = x.shape
                    x = x.reshape(n, c // (f * f), h * f, w * f)
            elif isinstance(m, O.SpatialBatchNormalization):
                rm, rv = self.bn[id(m)]
                x = F.batch_norm(x, rm, rv, self.p(m, 'weight'), self.p(m, 'bias'), training=train, momentum=m.momentum, eps=m.eps)
            elif isinstance(m, O.SpatialDropout):
                if train:
                    k = next(mi) if mi is not None else (torch.rand(x.shape[:2]) < (1 - m.p)).float()
                    x = x * torch.as_tensor(k, dtype=torch.float32).reshape(x.shape[0], x.shape[1], 1, 1)
                else:
                    x = x * (1 - m.p)
            elif isinstance(m, O.Dropout):
                if train:
                    k = next(mi) if mi is not None else (torch.rand(x.shape) < (1 - m.p)).float()
                    x = x * (torch.as_tensor(k, dtype=torch.float32).reshape(x.shape) / (1 - m.p))
            elif isinstance(m, O.SpatialAveragePooling):
                x = F.avg_pool2d(x, 2, 2)
            elif isinstance(m, O.SpatialMaxPooling):
                x = F.max_pool2d(x, 2, 2)
            elif isinstance(m, O.Sigmoid):
                x = torch.sigmoid(x)
            else:
                raise TypeError("torch_cpu: no restatement for %s" % type(m).__name__)
        return x


def bce(x, t, eps=1e-12):
    """nn.BCECriterion (train.lua:148): eps inside the logs, mean over the batch."""
    x = x.reshape(-1)
    return -(t * torch.log(x + eps) + (1 - t) * torch.log(1 - x + eps)).mean()


def torch7_adam(p, g, state, lr=1e-3, b1=0.9, b2=0.999, eps=1e-8):
    """interruptable_optimizers.lua:49-94 on flat tensors."""
    if 'm' not in state:
        state.update(t=0, m=torch.zeros_like(g), v=torch.zeros_like(g))
    state['t'] += 1
    state['m'].mul_(b1).add_(g, alpha=1 - b1)
    state['v'].mul_(b2).addcmul_(g, g, value=1 - b2)
    denom = state['v'].sqrt().add_(eps)
    step = lr * np.sqrt(1 - b2 ** state['t']) / (1 - b1 ** state['t'])
    p.addcdiv_(state['m'], denom, value=-step)


class GanCPU:
    """train.lua:134-191 state + the two closures of adversarial.lua:83-231 / adversarial_c2f.lua:40-119."""

    def __init__(self, G, D, opt=None):
        self.G, self.D = TorchNet(G), TorchNet(D)
        self.opt = dict(D_L1=0.0, D_L2=1e-4, G_L1=0.0, G_L2=0.0, D_clamp=1.0, G_clamp=5.0)
        self.opt.update(opt or {})
        self.adamD, self.adamG = {}, {}

    def _finish(self, net, which, adam):
        o = self.opt
        g = net.flat(grads=True)
        p = net.flat()
        l1, l2 = o[which + '_L1'], o[which + '_L2']
        if l1 != 0 or l2 != 0:
            g = g + torch.sign(p) * (l1 if which == 'D' else l2) + p * l2        # adversarial.lua:109 / :223 (quirk C4)
        if o[which + '_clamp'] != 0:
            g = g.clamp(-o[which + '_clamp'], o[which + '_clamp'])
        torch7_adam(p, g, adam)
        net.set_flat(p)
        return g

    def step_D(self, real, g_in, cond=None, masks=None):
        """real: [B/2,...] (c2f: the real diffs); g_in: G's input for the fake half (c2f: [noise, cond_fake]);
        cond: c2f only, [B, C, S, S] conditions of the whole D batch."""
        with torch.no_grad():
            fake = self.G.forward(g_in, train=True)                         # C5: train mode, no backward
        x = torch.cat([real, fake], 0)
        t = torch.cat([torch.ones(real.shape[0]), torch.zeros(fake.shape[0])])
        self.D.zero_grad()
        out = self.D.forward([x, cond] if cond is not None else x, masks)
        loss = bce(out, t)
        loss.backward()
        g = self._finish(self.D, 'D', self.adamD)
        return dict(out=out.detach(), f_bce=float(loss.detach()), grad=g)

    def step_G(self, g_in, cond=None, masks=None):
        self.G.zero_grad(); self.D.zero_grad()
        samples = self.G.forward(g_in, train=True)
        out = self.D.forward([samples, cond] if cond is not None else samples, masks)
        loss = bce(out, torch.ones(out.shape[0]))
        loss.backward()                                                     # incl. D's weight gradients, like :209
        g = self._finish(self.G, 'G', self.adamG)
        return dict(out=out.detach(), f_bce=float(loss.detach()), grad=g, samples=samples.detach())


def usable_cores():
    import os
    try:
        return len(os.sched_getaffinity(0))
    except AttributeError:
        return os.cpu_count() or 1


def time_iterations(workload, batch, min_seconds=10.0, max_iters=50, threads=None):
    """One iteration = D-step + G-step at `batch` (cfg2: 32x32x3 G32 / D32b; c2f: 64x64 G_d / D_c), reference init,
    synthetic inputs.  -> dict(images_per_sec, iters, seconds, threads).  First iteration is warm-up."""
    import os
    import time
    threads = threads or usable_cores()
    torch.set_num_threads(threads)
    rng = np.random.default_rng(1)
    if workload == "c2f":
        S = 64
        gan = GanCPU(O.create_G_d((3, S, S), rng), O.create_D_c((3, S, S), rng), O.C2F_OPT)
        fine = torch.rand(batch, 3, S, S)
        coarse = F.interpolate(F.avg_pool2d(fine, 2), scale_factor=2)
        diff = fine - coarse
        h = batch // 2

        def iteration():
            gan.step_D(diff[:h], [torch.rand(h, 1, S, S) * 2 - 1, coarse[h:]], cond=coarse)
            gan.step_G([torch.rand(batch, 1, S, S) * 2 - 1, coarse], cond=coarse)
    else:
        G = O.create_G32((3, 32, 32), 100, rng, weight_init_=False)
        D = O.create_D32b((3, 32, 32), rng)
        O.initialize_weights(G, rng=rng); O.initialize_weights(D, rng=rng)
        gan = GanCPU(G, D)
        real = torch.rand(batch // 2, 3, 32, 32)

        def iteration():
            gan.step_D(real, torch.rand(batch // 2, 100) * 2 - 1)
            gan.step_G(torch.rand(batch, 100) * 2 - 1)
    iteration()
    n, t0 = 0, time.perf_counter()
    while n < max_iters and (n == 0 or time.perf_counter() - t0 < min_seconds):
        iteration()
        n += 1
    dt = time.perf_counter() - t0
    return dict(images_per_sec=batch * n / dt, iters=n, seconds=dt, threads=threads)
