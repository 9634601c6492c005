"""CPU oracle: restatement of the Torch7 `nn` module semantics used by the GAN hot path.

TEST INFRASTRUCTURE ONLY.  Nothing in the product path (face_generator_amd/) may import
this file; only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg do.

PARITY UNPINNED: the reference (aleju/face-generator) ships no tests, golden vectors or
fixtures for this path, and its arithmetic lives in un-vendored,
un-pinned luarocks (torch7 / nn / cunn / cudnn.torch R3 / optim, era Dec-2015..Feb-2016)
that cannot be run here (no Lua in the image).  This file restates the *published*
algorithms of those modules and is anchored on the reference's
own call sites, cited per class below.  It is cross-checked in tests/test_oracle.py
against PyTorch-CPU functional ops wherever the semantics coincide, and against closed
forms where they do not (BCE eps, SpatialDropout scaling, Torch7 Adam).

Layout is the reference's: NCHW contiguous activations, conv weights [O][I][kH][kW],
linear weights [out][in].  Modules follow the torch `nn.Module` protocol
(updateOutput / updateGradInput / accGradParameters, .output / .gradInput,
.weight / .bias / .gradWeight / .gradBias) so that nets read like models.lua.
"""
import numpy as np

F32 = np.float32


class Module:
    """nn.Module protocol (upstream torch/nn Module.lua)."""

    def __init__(self):
        self.train = True
        self.output = None
        self.gradInput = None
        self.dtype = F32

    # -- protocol -------------------------------------------------------
    def forward(self, x):
        return self.updateOutput(x)

    def backward(self, x, gy, scale=1.0):
        gx = self.updateGradInput(x, gy)
        self.accGradParameters(x, gy, scale)
        return gx

    def updateOutput(self, x):
        raise NotImplementedError

    def updateGradInput(self, x, gy):
        raise NotImplementedError

    def accGradParameters(self, x, gy, scale=1.0):
        pass

    def parameters(self):
        """-> (list of params, list of grads), weight then bias (Module.lua:parameters)."""
        w = getattr(self, 'weight', None)
        b = getattr(self, 'bias', None)
        ps, gs = [], []
        if w is not None:
            ps.append('weight'); gs.append('gradWeight')
        if b is not None:
            ps.append('bias'); gs.append('gradBias')
        return [(self, p, g) for p, g in zip(ps, gs)]

    def training(self):
        self.train = True

    def evaluate(self):
        self.train = False

    def astype(self, dtype):
        self.dtype = dtype
        for n in ('weight', 'bias', 'gradWeight', 'gradBias', 'running_mean', 'running_var'):
            v = getattr(self, n, None)
            if v is not None:
                setattr(self, n, v.astype(dtype))
        return self

    def zeroGradParameters(self):
        for (m, p, g) in self.parameters():
            getattr(m, g)[...] = 0


class Sequential(Module):
    """nn.Sequential: chain; backward visits EVERY module incl. the first
    (its gradInput is what adversarial.lua:210 reads)."""

    def __init__(self, *mods):
        super().__init__()
        self.modules = list(mods)

    def add(self, m):
        self.modules.append(m)
        return self

    def get(self, i):
        return self.modules[i - 1]  # 1-based like Lua

    def updateOutput(self, x):
        self._inputs = []
        for m in self.modules:
            self._inputs.append(x)
            x = m.updateOutput(x)
        self.output = x
        return x

    def backward(self, x, gy, scale=1.0):
        for m, xi in zip(reversed(self.modules), reversed(self._inputs)):
            gy = m.backward(xi, gy, scale)
        self.gradInput = gy
        return gy

    def updateGradInput(self, x, gy):
        for m, xi in zip(reversed(self.modules), reversed(self._inputs)):
            gy = m.updateGradInput(xi, gy)
        self.gradInput = gy
        return gy

    def parameters(self):
        out = []
        for m in self.modules:
            out.extend(m.parameters())
        return out

    def training(self):
        self.train = True
        for m in self.modules:
            m.training()

    def evaluate(self):
        self.train = False
        for m in self.modules:
            m.evaluate()

    def astype(self, dtype):
        self.dtype = dtype
        for m in self.modules:
            m.astype(dtype)
        return self

    def listModules(self):
        out = [self]
        for m in self.modules:
            out.extend(m.listModules() if isinstance(m, Sequential) else [m])
        return out

    def getParameters(self):
        """Module:getParameters() (train.lua:151-152): re-home all params (and grads)
        into ONE contiguous vector each, module order, weight then bias; the module
        fields become views."""
        plist = self.parameters()
        n = sum(getattr(m, p).size for (m, p, g) in plist)
        dt = self.dtype
        flat = np.zeros(n, dt)
        gflat = np.zeros(n, dt)
        off = 0
        for (m, p, g) in plist:
            w = getattr(m, p)
            k = w.size
            flat[off:off + k] = w.reshape(-1)
            gflat[off:off + k] = getattr(m, g).reshape(-1)
            setattr(m, p, flat[off:off + k].reshape(w.shape))
            setattr(m, g, gflat[off:off + k].reshape(w.shape))
            off += k
        return flat, gflat


class Linear(Module):
    """nn.Linear (models.lua:59, 406-412): y = x W^T + b; W [out][in]."""

    def __init__(self, nin, nout, rng=None):
        super().__init__()
        rng = rng or np.random.default_rng(0)
        s = 1.0 / np.sqrt(nin)  # reset(): U(+-1/sqrt(in))
        self.weight = rng.uniform(-s, s, (nout, nin)).astype(F32)
        self.bias = rng.uniform(-s, s, (nout,)).astype(F32)
        self.gradWeight = np.zeros_like(self.weight)
        self.gradBias = np.zeros_like(self.bias)

    def updateOutput(self, x):
        self.output = x @ self.weight.T + self.bias
        return self.output

    def updateGradInput(self, x, gy):
        self.gradInput = gy @ self.weight
        return self.gradInput

    def accGradParameters(self, x, gy, scale=1.0):
        self.gradWeight += scale * (gy.T @ x)
        self.gradBias += scale * gy.sum(0)


class View(Module):
    """nn.View (models.lua:60, 405): reshape keeping batch dim, NCHW flatten order."""

    def __init__(self, *shape):
        super().__init__()
        self.shape = tuple(int(s) for s in shape)

    def updateOutput(self, x):
        self.output = x.reshape((x.shape[0],) + self.shape)
        return self.output

    def updateGradInput(self, x, gy):
        self.gradInput = gy.reshape(x.shape)
        return self.gradInput


class PReLU(Module):
    """nn.PReLU() single shared slope (models.lua:61-71, 386-410). a=0.25 default.
    gradWeight = sum over the WHOLE tensor of x*gy where x<=0."""

    def __init__(self):
        super().__init__()
        self.weight = np.full((1,), 0.25, F32)
        self.gradWeight = np.zeros((1,), F32)

    # Test hook: `pos_override` (bool array shaped like the input, or None) replaces the branch decision x > 0.  PReLU is not
    # differentiable at 0; a unit whose pre-activation is within fp32 rounding of 0 may legitimately take either branch
    # depending on the summation order of the producing GEMM.  The GPU parity tests at the BASELINE sizes (10^7 units per
    # step) copy the device's decisions in here so that the gradients are compared on IDENTICAL branches.
    pos_override = None

    def _pos(self, x):
        if self.pos_override is not None:
            return self.pos_override.reshape(x.shape)
        return x > 0

    def updateOutput(self, x):
        self.output = np.where(self._pos(x), x, self.weight[0] * x).astype(x.dtype)
        return self.output

    def updateGradInput(self, x, gy):
        self.gradInput = np.where(self._pos(x), gy, self.weight[0] * gy).astype(x.dtype)
        return self.gradInput

    def accGradParameters(self, x, gy, scale=1.0):
        # fp64 accumulate then round: the reduction order upstream is unspecified
        t = np.where(self._pos(x), 0.0, x.astype(np.float64) * gy.astype(np.float64))
        s = np.sum(t)
        # conditioning of this (heavily cancelling) scalar sum, for the parity tests: an eps-relative perturbation of x
        # moves the sum by ~ eps * ||t||_2, whatever |s| is
        self.gw_cond = float(np.sqrt(np.sum(t * t))) * abs(scale)
        self.gradWeight += self.gradWeight.dtype.type(scale * s)


class LeakyReLU(Module):
    """In-tree LeakyReLU.lua:7-31: y = max(x,0) + s*min(x,0), s default 0.333;
    x == 0 takes the positive branch in backward (LeakyReLU.lua:21-31)."""

    def __init__(self, s=0.333):
        super().__init__()
        self.s = s

    def updateOutput(self, x):
        dt = x.dtype.type
        ax = np.abs(x)
        neg = (ax - x) * dt(-self.s / 2)
        self.output = (ax + x) / dt(2) + neg
        return self.output

    def updateGradInput(self, x, gy):
        dt = x.dtype.type
        self.gradInput = np.where(x >= 0, gy, gy * dt(self.s)).astype(x.dtype)
        return self.gradInput


class SpatialUpSamplingNearest(Module):
    """nn.SpatialUpSamplingNearest(2) (models.lua:63, 68)."""

    def __init__(self, f=2):
        super().__init__()
        self.f = f

    def updateOutput(self, x):
        f = self.f
        self.output = np.repeat(np.repeat(x, f, axis=2), f, axis=3)
        return self.output

    def updateGradInput(self, x, gy):
        f = self.f
        n, c, h, w = x.shape
        self.gradInput = gy.reshape(n, c, h, f, w, f).sum(axis=(3, 5)).astype(x.dtype)
        return self.gradInput


def _im2col(x, kh, kw, ph, pw, dh=1, dw=1, out=None):
    """THNN SpatialConvolutionMM's unfolded `finput`: [N][C*kh*kw][Ho*Wo]; Ho = floor((H + 2 ph - kh) / dh) + 1.
    A pure copy, one strided block per tap; `out` (a buffer of a previous call with the same geometry) is re-used --
    fresh GB-sized allocations cost more than the copy at the BASELINE sizes (page faults)."""
    n, c, h, w = x.shape
    ho, wo = (h + 2 * ph - kh) // dh + 1, (w + 2 * pw - kw) // dw + 1
    xp = np.zeros((n, c, h + 2 * ph, w + 2 * pw), x.dtype)
    xp[:, :, ph:ph + h, pw:pw + w] = x
    if out is None or out.shape != (n, c * kh * kw, ho * wo) or out.dtype != x.dtype:
        out = np.empty((n, c * kh * kw, ho * wo), x.dtype)
    cols = out.reshape(n, c, kh, kw, ho, wo)
    for ky in range(kh):
        for kx in range(kw):
            cols[:, :, ky, kx] = xp[:, :, ky:ky + (ho - 1) * dh + 1:dh, kx:kx + (wo - 1) * dw + 1:dw]
    return out, ho, wo


class SpatialConvolution(Module):
    """nn.SpatialConvolution / cudnn.SpatialConvolution (models.lua:64-73, 385-400):
    cross-correlation, zero pad, + bias; stride 1 on the hot path, stride 2 in create_D16_d (models.lua:289-291).
    CPU algorithm = THNN SpatialConvolutionMM: im2col into `finput` + sgemm per sample."""

    def __init__(self, nin, nout, kw, kh, dw=1, dh=1, padw=0, padh=None, rng=None):
        super().__init__()
        self.dw, self.dh = int(dw), int(dh)
        rng = rng or np.random.default_rng(0)
        self.nin, self.nout, self.kw, self.kh = nin, nout, kw, kh
        self.padw = int(padw)
        self.padh = int(padw if padh is None else padh)
        s = 1.0 / np.sqrt(kw * kh * nin)
        self.weight = rng.uniform(-s, s, (nout, nin, kh, kw)).astype(F32)
        self.bias = rng.uniform(-s, s, (nout,)).astype(F32)
        self.gradWeight = np.zeros_like(self.weight)
        self.gradBias = np.zeros_like(self.bias)

    def updateOutput(self, x):
        n = x.shape[0]
        cols, ho, wo = _im2col(x, self.kh, self.kw, self.padh, self.padw, self.dh, self.dw, out=getattr(self, 'finput', None))
        self.finput = cols
        wm = self.weight.reshape(self.nout, -1)
        y = np.matmul(wm, cols) + self.bias[None, :, None]
        self.output = y.reshape(n, self.nout, ho, wo)
        return self.output

    def updateGradInput(self, x, gy):
        # full correlation with the 180-degree rotated kernel == conv of gy with
        # W^T flipped, same padding arithmetic for odd k / same-pad / stride 1
        n, c, h, w = x.shape
        if self.dw != 1 or self.dh != 1:
            # strided: correlate the zero-inserted output gradient (THNN does col2im of W^T gy; same sums)
            gz = np.zeros((n, self.nout, h, w), gy.dtype)
            gz[:, :, 0:gy.shape[2] * self.dh:self.dh, 0:gy.shape[3] * self.dw:self.dw] = gy
            gy = gz
        wf = self.weight[:, :, ::-1, ::-1].transpose(1, 0, 2, 3)  # [I][O][kh][kw]
        cols, ho, wo = _im2col(gy, self.kh, self.kw, self.kh - 1 - self.padh, self.kw - 1 - self.padw,
                               out=getattr(self, '_gcols', None) if getattr(self, 'keep_buffers', False) else None)
        if getattr(self, 'keep_buffers', False):
            self._gcols = cols
        gx = np.matmul(np.ascontiguousarray(wf).reshape(c, -1), cols)
        self.gradInput = gx.reshape(n, c, ho, wo)
        assert ho == h and wo == w
        return self.gradInput

    def accGradParameters(self, x, gy, scale=1.0):
        n = x.shape[0]
        cols = getattr(self, 'finput', None)
        if cols is None or cols.shape[0] != n:
            cols, _, _ = _im2col(x, self.kh, self.kw, self.padh, self.padw, self.dh, self.dw)
        g = gy.reshape(n, self.nout, -1)
        gw = np.matmul(g, cols.transpose(0, 2, 1)).sum(0)
        self.gradWeight += (scale * gw).reshape(self.weight.shape).astype(self.weight.dtype)
        self.gradBias += (scale * g.sum(axis=(0, 2))).astype(self.bias.dtype)


class SpatialConvolutionUpsample(SpatialConvolution):
    """layers/cudnnSpatialConvolutionUpsample.lua:4-58: conv nIn -> nOut*f^2, k odd,
    stride 1, pad (k-1)/2, then a FLAT re-view of contiguous NCHW memory to
    [N][nOut][h*f][w*f] (NOT depth-to-space).  The reference always passes f=1."""

    def __init__(self, nin, nout, kw, kh, factor=2, rng=None):
        assert kw % 2 == 1 and kh % 2 == 1, "kernel must be odd (cudnnSpatialConvolutionUpsample.lua:6-8)"
        super().__init__(nin, nout * factor * factor, kw, kh, 1, 1, (kw - 1) // 2, (kh - 1) // 2, rng=rng)
        self.factor = factor
        self.nOutputPlaneU = nout

    def updateOutput(self, x):
        y = super().updateOutput(x)
        n, _, h, w = y.shape
        f = self.factor
        self.output = y.reshape(n, self.nOutputPlaneU, h * f, w * f)
        return self.output

    def _unview(self, x, gy):
        n, _, h, w = x.shape
        return gy.reshape(n, self.nout, h, w)

    def updateGradInput(self, x, gy):
        return super().updateGradInput(x, self._unview(x, gy))

    def accGradParameters(self, x, gy, scale=1.0):
        return super().accGradParameters(x, self._unview(x, gy), scale)


class SpatialBatchNormalization(Module):
    """nn.SpatialBatchNormalization(nF) (models.lua:65, 70): eps 1e-5, momentum 0.1,
    affine.  Train: biased batch var for normalisation; running_var updated with the
    UNBIASED var (THNN BatchNormalization.c; version-sensitive, evaluate-mode only)."""

    def __init__(self, nf, eps=1e-5, momentum=0.1, rng=None):
        super().__init__()
        rng = rng or np.random.default_rng(0)
        self.nf, self.eps, self.momentum = nf, eps, momentum
        self.weight = rng.uniform(0, 1, (nf,)).astype(F32)  # reset(): gamma ~ U(0,1)
        self.bias = np.zeros((nf,), F32)
        self.gradWeight = np.zeros_like(self.weight)
        self.gradBias = np.zeros_like(self.bias)
        self.running_mean = np.zeros((nf,), F32)
        self.running_var = np.ones((nf,), F32)

    def updateOutput(self, x):
        dt = x.dtype
        if self.train:
            x64 = x.astype(np.float64)
            n = x.shape[0] * x.shape[2] * x.shape[3]
            mean = x64.mean(axis=(0, 2, 3))
            var = ((x64 - mean[None, :, None, None]) ** 2).mean(axis=(0, 2, 3))
            self.save_mean = mean.astype(dt)
            self.save_invstd = (1.0 / np.sqrt(var + self.eps)).astype(dt)
            m = self.momentum
            unb = var * n / max(n - 1, 1)
            self.running_mean = ((1 - m) * self.running_mean + m * mean).astype(self.running_mean.dtype)
            self.running_var = ((1 - m) * self.running_var + m * unb).astype(self.running_var.dtype)
            mu, inv = self.save_mean, self.save_invstd
        else:
            mu = self.running_mean.astype(dt)
            inv = (1.0 / np.sqrt(self.running_var.astype(np.float64) + self.eps)).astype(dt)
        self.xhat = (x - mu[None, :, None, None]) * inv[None, :, None, None]
        self.output = self.xhat * self.weight.astype(dt)[None, :, None, None] + self.bias.astype(dt)[None, :, None, None]
        return self.output

    def updateGradInput(self, x, gy):
        dt = x.dtype
        g = self.weight.astype(dt)[None, :, None, None]
        if self.train:
            inv = self.save_invstd[None, :, None, None]
            n = x.shape[0] * x.shape[2] * x.shape[3]
            gy64 = gy.astype(np.float64)
            xh64 = self.xhat.astype(np.float64)
            m1 = (gy64.sum(axis=(0, 2, 3)) / n).astype(dt)[None, :, None, None]
            m2 = ((gy64 * xh64).sum(axis=(0, 2, 3)) / n).astype(dt)[None, :, None, None]
            self.gradInput = g * inv * (gy - m1 - self.xhat * m2)
        else:
            inv = (1.0 / np.sqrt(self.running_var.astype(np.float64) + self.eps)).astype(dt)
            self.gradInput = g * inv[None, :, None, None] * gy
        return self.gradInput

    def accGradParameters(self, x, gy, scale=1.0):
        gy64 = gy.astype(np.float64)
        self.gradWeight += (scale * (gy64 * self.xhat.astype(np.float64)).sum(axis=(0, 2, 3))).astype(self.weight.dtype)
        self.gradBias += (scale * gy64.sum(axis=(0, 2, 3))).astype(self.bias.dtype)


class SpatialAveragePooling(Module):
    """nn.SpatialAveragePooling(2,2,2,2) (models.lua:388-403)."""

    def __init__(self, kw=2, kh=2, dw=2, dh=2):
        super().__init__()
        assert (kw, kh, dw, dh) == (2, 2, 2, 2)

    def updateOutput(self, x):
        n, c, h, w = x.shape
        self.output = x.reshape(n, c, h // 2, 2, w // 2, 2).mean(axis=(3, 5)).astype(x.dtype)
        return self.output

    def updateGradInput(self, x, gy):
        self.gradInput = (np.repeat(np.repeat(gy, 2, 2), 2, 3) / x.dtype.type(4)).astype(x.dtype)
        return self.gradInput


class SpatialMaxPooling(Module):
    """nn.SpatialMaxPooling(2,2) (models_c2f.lua:251, 256): ties -> first max in scan order."""

    def __init__(self, kw=2, kh=2):
        super().__init__()
        assert (kw, kh) == (2, 2)

    def updateOutput(self, x):
        n, c, h, w = x.shape
        blk = x.reshape(n, c, h // 2, 2, w // 2, 2).transpose(0, 1, 2, 4, 3, 5).reshape(n, c, h // 2, w // 2, 4)
        # test hook (see PReLU.pos_override): the argmax of a near-tie block decided elsewhere
        self.indices = blk.argmax(axis=-1) if getattr(self, 'indices_override', None) is None else self.indices_override
        self.output = np.take_along_axis(blk, self.indices[..., None], -1)[..., 0]
        return self.output

    def updateGradInput(self, x, gy):
        n, c, h, w = x.shape
        g = np.zeros((n, c, h // 2, w // 2, 4), x.dtype)
        np.put_along_axis(g, self.indices[..., None], gy[..., None], -1)
        self.gradInput = g.reshape(n, c, h // 2, w // 2, 2, 2).transpose(0, 1, 2, 4, 3, 5).reshape(n, c, h, w)
        return self.gradInput


class SpatialDropout(Module):
    """nn.SpatialDropout(p) (models.lua:387-402, p=0.2): train: per-(n,c) Bernoulli(1-p)
    mask, NO 1/(1-p) rescale; eval: y = (1-p) x.  Mask may be injected (`noise`)."""

    def __init__(self, p=0.5, rng=None):
        super().__init__()
        self.p = p
        self.rng = rng or np.random.default_rng(0)
        self.noise = None
        self.injected = False

    def set_mask(self, mask):
        self.noise = np.asarray(mask)
        self.injected = True

    def updateOutput(self, x):
        if self.train:
            if not self.injected:
                self.noise = (self.rng.random(x.shape[:2]) < (1 - self.p))
            m = self.noise.reshape(x.shape[0], x.shape[1], 1, 1).astype(x.dtype)
            self.output = x * m
        else:
            self.output = x * x.dtype.type(1 - self.p)
        return self.output

    def updateGradInput(self, x, gy):
        assert self.train, "SpatialDropout backward is an error in evaluate mode upstream"
        m = self.noise.reshape(x.shape[0], x.shape[1], 1, 1).astype(x.dtype)
        self.gradInput = gy * m
        return self.gradInput


class Dropout(Module):
    """nn.Dropout() p=0.5 v2 (models.lua:408, 411): train: mask ~ Bernoulli(1-p)/(1-p);
    eval: identity."""

    def __init__(self, p=0.5, rng=None):
        super().__init__()
        self.p = p
        self.rng = rng or np.random.default_rng(0)
        self.noise = None
        self.injected = False

    def set_mask(self, mask):
        """mask: 0/1 keep mask (scaled by 1/(1-p) internally)."""
        self.noise = np.asarray(mask)
        self.injected = True

    def updateOutput(self, x):
        if self.train:
            if not self.injected:
                self.noise = (self.rng.random(x.shape) < (1 - self.p))
            self._m = self.noise.reshape(x.shape).astype(x.dtype) / x.dtype.type(1 - self.p)
            self.output = x * self._m
        else:
            self.output = x
        return self.output

    def updateGradInput(self, x, gy):
        self.gradInput = gy * self._m if self.train else gy
        return self.gradInput


class Sigmoid(Module):
    """nn.Sigmoid (models.lua:74, 413)."""

    def updateOutput(self, x):
        self.output = (1.0 / (1.0 + np.exp(-x))).astype(x.dtype)
        return self.output

    def updateGradInput(self, x, gy):
        y = self.output
        self.gradInput = gy * y * (1 - y)
        return self.gradInput


class JoinTable(Module):
    """nn.JoinTable(2,2) (models_c2f.lua:116): concat 4-D inputs along channels."""

    def updateOutput(self, xs):
        self.output = np.concatenate(xs, axis=1)
        return self.output

    def updateGradInput(self, xs, gy):
        splits = np.cumsum([x.shape[1] for x in xs])[:-1]
        self.gradInput = np.split(gy, splits, axis=1)
        return self.gradInput


class CAddTable(Module):
    """nn.CAddTable (models_c2f.lua:240)."""

    def updateOutput(self, xs):
        self.output = sum(xs[1:], xs[0].copy())
        return self.output

    def updateGradInput(self, xs, gy):
        self.gradInput = [gy for _ in xs]
        return self.gradInput


class ConcatTable(Module):
    """nn.ConcatTable (models.lua:307-309): every branch sees the same input, output = table of branch outputs;
    gradInput = sum of the branch gradInputs.  Parameters: branch order."""

    def __init__(self, *branches):
        super().__init__()
        self.modules = list(branches)

    def add(self, m):
        self.modules.append(m)
        return self

    def updateOutput(self, x):
        self.output = [m.forward(x) for m in self.modules]
        return self.output

    def backward(self, x, gys, scale=1.0):
        gs = [m.backward(x, gy, scale) for m, gy in zip(self.modules, gys)]
        self.gradInput = sum(gs[1:], gs[0].copy())
        return self.gradInput

    def updateGradInput(self, x, gys):
        gs = [m.updateGradInput(x, gy) for m, gy in zip(self.modules, gys)]
        self.gradInput = sum(gs[1:], gs[0].copy())
        return self.gradInput

    def parameters(self):
        out = []
        for m in self.modules:
            out.extend(m.parameters())
        return out

    def training(self):
        self.train = True
        for m in self.modules:
            m.training()

    def evaluate(self):
        self.train = False
        for m in self.modules:
            m.evaluate()

    def astype(self, dtype):
        self.dtype = dtype
        for m in self.modules:
            m.astype(dtype)
        return self


class BCECriterion:
    """nn.BCECriterion() (train.lua:148): eps = 1e-12, sizeAverage.
    f = -(1/n) sum[t log(x+eps) + (1-t) log(1-x+eps)];
    gx = -(1/n) (t-x)/((1-x+eps)(x+eps))."""
    EPS = 1e-12

    def forward(self, x, t):
        x64 = x.reshape(-1).astype(np.float64)
        t64 = np.asarray(t).reshape(-1).astype(np.float64)
        n = x64.size
        # upstream computes log(x+eps) in the tensor dtype; fp32(x)+1e-12 rounds to x for
        # x >= ~6e-5, which the float64 evaluation reproduces to < 1e-7 relative.
        x32 = x.reshape(-1).astype(x.dtype)
        eps = x.dtype.type(self.EPS)
        la = np.log(x32 + eps).astype(np.float64)
        lb = np.log((x.dtype.type(1) - x32) + eps).astype(np.float64)
        self.output = float(-(t64 * la + (1 - t64) * lb).sum() / n)
        return self.output

    def backward(self, x, t):
        dt = x.dtype.type
        xv = x.reshape(-1)
        tv = np.asarray(t).reshape(-1).astype(x.dtype)
        n = xv.size
        eps = dt(self.EPS)
        g = -(tv - xv) / ((dt(1) - xv + eps) * (xv + eps)) / dt(n)
        self.gradInput = g.reshape(x.shape).astype(x.dtype)
        return self.gradInput


# ----------------------------------------------------------------------------------
# optimizers: interruptable_optimizers.lua (in-tree, authoritative)
# ----------------------------------------------------------------------------------
def interruptable_adam(opfunc, x, config, state=None):
    """interruptable_optimizers.lua:49-94.  eps added BEFORE bias correction."""
    state = config if state is None else state
    lr = config.get('learningRate', 0.001)
    b1 = config.get('beta1', 0.9)
    b2 = config.get('beta2', 0.999)
    eps = config.get('epsilon', 1e-8)
    fx, dfdx = opfunc(x)
    if fx is False:
        return False
    state['t'] = state.get('t', 0)
    if 'm' not in state:
        state['m'] = np.zeros_like(dfdx)
        state['v'] = np.zeros_like(dfdx)
        state['denom'] = np.zeros_like(dfdx)
    state['t'] += 1
    dt = x.dtype.type
    m, v = state['m'], state['v']
    m *= dt(b1); m += dt(1 - b1) * dfdx
    v *= dt(b2); v += dt(1 - b2) * dfdx * dfdx
    state['denom'][...] = np.sqrt(v) + dt(eps)
    bc1 = 1 - b1 ** state['t']
    bc2 = 1 - b2 ** state['t']
    step = lr * np.sqrt(bc2) / bc1
    x += dt(-step) * m / state['denom']
    return x, [fx]


def interruptable_sgd(opfunc, x, config, state=None):
    """interruptable_optimizers.lua:97-167 (optim.sgd with the false-gate)."""
    state = config if state is None else state
    lr = config.get('learningRate', 1e-3)
    lrd = config.get('learningRateDecay', 0)
    wd = config.get('weightDecay', 0)
    mom = config.get('momentum', 0)
    damp = config.get('dampening', mom)
    nesterov = config.get('nesterov', False)
    state['evalCounter'] = state.get('evalCounter', 0)
    nevals = state['evalCounter']
    fx, dfdx = opfunc(x)
    if fx is False:
        return False
    dt = x.dtype.type
    if wd != 0:
        dfdx = dfdx + dt(wd) * x
    if mom != 0:
        if 'dfdx' not in state:
            state['dfdx'] = dfdx.copy()
        else:
            state['dfdx'] *= dt(mom); state['dfdx'] += dt(1 - damp) * dfdx
        if nesterov:
            dfdx = dfdx + dt(mom) * state['dfdx']
        else:
            dfdx = state['dfdx']
    clr = lr / (1 + nevals * lrd)
    x += dt(-clr) * dfdx
    state['evalCounter'] += 1
    return x, [fx]


def interruptable_adagrad(opfunc, x, config, state=None):
    """interruptable_optimizers.lua:7-46."""
    state = config if state is None else state
    lr = config.get('learningRate', 1e-3)
    lrd = config.get('learningRateDecay', 0)
    state['evalCounter'] = state.get('evalCounter', 0)
    nevals = state['evalCounter']
    fx, dfdx = opfunc(x)
    if fx is False:
        return False
    dt = x.dtype.type
    clr = lr / (1 + nevals * lrd)
    if 'paramVariance' not in state:
        state['paramVariance'] = np.zeros_like(dfdx)
    state['paramVariance'] += dfdx * dfdx
    std = np.sqrt(state['paramVariance']) + dt(1e-10)
    x += dt(-clr) * dfdx / std
    state['evalCounter'] += 1
    return x, [fx]


# ----------------------------------------------------------------------------------
# models.lua (32x32 path)
# ----------------------------------------------------------------------------------
def weight_init(net, arg='heuristic', rng=None):
    """weight-init.lua:41-76 `require('weight-init')(model, 'heuristic')`: TOP-LEVEL modules only (no recursion, :52); modules
    typed nn.SpatialConvolution / nn.SpatialConvolutionMM / nn.Linear get `m:reset(method(fan_in, fan_out))` -- Torch7's reset(stdv)
    draws weight AND bias from U(-stdv*sqrt(3), stdv*sqrt(3)) -- and then every top-level module with a bias has it zeroed (:71-73).
    cudnn.SpatialConvolution (`cudnn=True` on the oracle's module: G's convolutions, models.lua:63-73) is not in the list: only its
    bias is zeroed.  With 'heuristic' = sqrt(1 / (3 fan_in)) the Linear's new range is 1/sqrt(fan_in), the default one."""
    rng = rng or np.random.default_rng(5)
    method = {'heuristic': lambda fi, fo: np.sqrt(1.0 / (3.0 * fi)), 'xavier': lambda fi, fo: np.sqrt(2.0 / (fi + fo)),
              'xavier_caffe': lambda fi, fo: np.sqrt(1.0 / fi), 'kaiming': lambda fi, fo: np.sqrt(4.0 / (fi + fo))}[arg]
    for m in getattr(net, 'modules', []):
        std = None
        if type(m) is SpatialConvolution and not getattr(m, 'cudnn', False):
            o, i, kh, kw = m.weight.shape
            std = method(i * kh * kw, o * kh * kw)
        elif type(m) is Linear:
            std = method(m.weight.shape[1], m.weight.shape[0])
        if std is not None:
            s = std * np.sqrt(3.0)
            m.weight[...] = rng.uniform(-s, s, m.weight.shape).astype(m.weight.dtype)
            m.bias[...] = rng.uniform(-s, s, m.bias.shape).astype(m.bias.dtype)
        if getattr(m, 'bias', None) is not None:
            m.bias[...] = 0
    return net


def _cudnn(conv):
    conv.cudnn = True        # typed cudnn.SpatialConvolution in the reference (models.lua:63, 68, 73): weight_init skips its reset
    return conv


def create_G32(dimensions, noise_dim, rng=None, weight_init_=True):
    """models.lua:57-81 create_G_decoder_upsampling32, including the weight-init 'heuristic' call of :78 (train.lua:137-138
    overrides it with NN_UTILS.initializeWeights -- SURVEY F9; a caller of MODELS.create_G alone sees it).  `weight_init_=False`
    leaves Torch7's default reset() state (non-zero biases): what the parity tests feed both sides."""
    rng = rng or np.random.default_rng(1)
    c = dimensions[0]
    net = _create_G_decoder(8, c, noise_dim, rng)
    return weight_init(net, 'heuristic', rng) if weight_init_ else net


def create_G16(dimensions, noise_dim, rng=None, weight_init_=True):
    """models.lua:27-51 create_G_decoder_upsampling16: the 32-px decoder started from a 4x4 map (weight-init at :48)."""
    rng = rng or np.random.default_rng(1)
    net = _create_G_decoder(4, dimensions[0], noise_dim, rng)
    return weight_init(net, 'heuristic', rng) if weight_init_ else net


def _create_G_decoder(s0, c, noise_dim, rng):
    return Sequential(
        Linear(noise_dim, 128 * s0 * s0, rng), View(128, s0, s0), PReLU(),
        SpatialUpSamplingNearest(2), _cudnn(SpatialConvolution(128, 256, 5, 5, 1, 1, 2, 2, rng)),
        SpatialBatchNormalization(256, rng=rng), PReLU(),
        SpatialUpSamplingNearest(2), _cudnn(SpatialConvolution(256, 128, 5, 5, 1, 1, 2, 2, rng)),
        SpatialBatchNormalization(128, rng=rng), PReLU(),
        _cudnn(SpatialConvolution(128, c, 3, 3, 1, 1, 1, 1, rng)), Sigmoid())


def create_D16_d(dimensions, rng=None):
    """models.lua:279-316 create_D16_d: ConcatTable{conv branch (two stride-2 convs), dense branch} -> JoinTable(2) ->
    Linear(1152, 1) -> Sigmoid."""
    rng = rng or np.random.default_rng(3)
    c, h, w = dimensions
    insz = c * h * w
    fine_sz = int(1024 * 0.25 * 0.25 * 0.25 * h * w)
    fine = Sequential(
        SpatialConvolution(c, 128, 3, 3, 1, 1, 1, None, rng), PReLU(),
        SpatialConvolution(128, 128, 3, 3, 1, 1, 1, None, rng), PReLU(),
        SpatialAveragePooling(2, 2, 2, 2),
        SpatialConvolution(128, 512, 3, 3, 2, 2, 1, None, rng), PReLU(),
        SpatialConvolution(512, 1024, 3, 3, 2, 2, 1, None, rng), PReLU(),
        SpatialDropout(0.5, rng), View(fine_sz), Linear(fine_sz, 1024, rng), PReLU())
    dense = Sequential(View(insz), Linear(insz, 128, rng), PReLU(), Dropout(0.5, rng), Linear(128, 128, rng), PReLU())
    return Sequential(ConcatTable(fine, dense), JoinTable(), Linear(1024 + 128, 1, rng), Sigmoid())


def create_D32b(dimensions, rng=None):
    """models.lua:382-416 create_D32b."""
    rng = rng or np.random.default_rng(2)
    c, h, w = dimensions
    nfeat = int(512 * 0.25 ** 4 * h * w)
    m = Sequential()
    for (i, o) in ((c, 64), (64, 128), (128, 256), (256, 512)):
        m.add(SpatialConvolution(i, o, 3, 3, 1, 1, 1, None, rng)).add(PReLU())
        m.add(SpatialDropout(0.2, rng)).add(SpatialAveragePooling(2, 2, 2, 2))
    m.add(View(nfeat)).add(Linear(nfeat, 512, rng)).add(PReLU()).add(Dropout(0.5, rng))
    m.add(Linear(512, 512, rng)).add(PReLU()).add(Dropout(0.5, rng))
    m.add(Linear(512, 1, rng)).add(Sigmoid())
    return m


def initialize_weights(model, rw=0.005, rb=0.001, rng=None):
    """nn_utils.lua:8-29: top-level modules only; weight <- randn*rw, bias <- randn*rb
    (incl. BN gamma/beta and the PReLU slope)."""
    rng = rng or np.random.default_rng(3)
    for m in model.modules:
        if getattr(m, 'weight', None) is not None:
            m.weight[...] = (rng.standard_normal(m.weight.shape) * rw).astype(m.weight.dtype)
        if getattr(m, 'bias', None) is not None:
            m.bias[...] = (rng.standard_normal(m.bias.shape) * rb).astype(m.bias.dtype)


# ----------------------------------------------------------------------------------
# adversarial.lua: one D-step and one G-step from explicit inputs
# ----------------------------------------------------------------------------------
class GanState:
    """Holds what train.lua:134-191 sets up: nets, criterion, flat params, OPTSTATE."""

    def __init__(self, G, D, opt=None):
        self.G, self.D = G, D
        self.crit = BCECriterion()
        self.pG, self.gG = G.getParameters()
        self.pD, self.gD = D.getParameters()
        self.opt = dict(D_L1=0.0, D_L2=1e-4, G_L1=0.0, G_L2=0.0, D_clamp=1.0, G_clamp=5.0)
        if opt:
            self.opt.update(opt)
        self.adamD, self.adamG = {}, {}


def set_dropout_masks(net, masks):
    """masks: list (in module order) for every SpatialDropout/Dropout in `net`."""
    it = iter(masks)
    for m in walk_modules(net):
        if isinstance(m, (SpatialDropout, Dropout)):
            m.set_mask(next(it))


def walk_modules(net):
    """Leaf modules in execution / parameter order, descending into Sequential / ConcatTable containers."""
    out = []
    for m in net.modules:
        if isinstance(m, (Sequential, ConcatTable)):
            out.extend(walk_modules(m))
        else:
            out.append(m)
    return out


def feval_D(st, inputs, targets):
    """adversarial.lua:83-179 without the accuracy gate bookkeeping.
    -> f, grad (flat), outputs[B], (confusion counts tp/.. as 2x2 [pred][target])."""
    o = st.opt
    st.gD[...] = 0
    out = st.D.forward(inputs)
    f = st.crit.forward(out, targets)
    st.last_f_bce = f
    df = st.crit.backward(out, targets)
    st.D.backward(inputs, df)
    if o['D_L1'] != 0 or o['D_L2'] != 0:
        p64 = st.pD.astype(np.float64)
        f += o['D_L1'] * np.abs(p64).sum()
        f += o['D_L2'] * (p64 ** 2).sum() / 2
        dt = st.pD.dtype.type
        st.gD += np.sign(st.pD) * dt(o['D_L1']) + st.pD * dt(o['D_L2'])
    conf = np.zeros((2, 2), np.int64)
    for i in range(out.shape[0]):
        c = 1 if out[i, 0] > 0.5 else 0
        conf[c, int(targets[i])] += 1
    if o['D_clamp'] != 0:
        np.clip(st.gD, -o['D_clamp'], o['D_clamp'], out=st.gD)
    return f, st.gD, out, conf


def feval_G_on_D(st, noise, targets):
    """adversarial.lua:187-231.  Note :223 uses G_L2 as the L1 multiplier (quirk C4)."""
    o = st.opt
    st.gG[...] = 0
    samples = st.G.forward(noise)
    out = st.D.forward(samples)
    f = st.crit.forward(out, targets)
    st.last_f_bce = f
    df = st.crit.backward(out, targets)
    st.D.backward(samples, df)
    df_do = st.D.modules[0].gradInput
    st.G.backward(noise, df_do)
    if o['G_L1'] != 0 or o['G_L2'] != 0:
        p64 = st.pG.astype(np.float64)
        f += o['G_L1'] * np.abs(p64).sum()
        f += o['G_L2'] * (p64 ** 2).sum() / 2
        dt = st.pG.dtype.type
        st.gG += np.sign(st.pG) * dt(o['G_L2']) + st.pG * dt(o['G_L2'])
    if o['G_clamp'] != 0:
        np.clip(st.gG, -o['G_clamp'], o['G_clamp'], out=st.gG)
    return f, st.gG, samples, out


def step_D(st, real, noise_half, masks=None, gate=None):
    """adversarial.lua:240-268: B/2 real (target 1) || B/2 fake from G in TRAIN mode
    (target 0) -> fevalD -> Adam on D.  gate(tV) -> bool is the maxAccuracyD interrupt of adversarial.lua:124-178:
    when it says no, fevalD returns false,false and interruptableAdam skips the update (no `t` increment)."""
    fake = st.G.forward(noise_half).copy()
    inputs = np.concatenate([real, fake], 0)
    targets = np.concatenate([np.ones(real.shape[0]), np.zeros(fake.shape[0])]).astype(real.dtype)
    if masks is not None:
        set_dropout_masks(st.D, masks)
    res = {}

    def op(x):
        f, g, out, conf = feval_D(st, inputs, targets)
        res.update(f=f, f_bce=st.last_f_bce, out=out.copy(), conf=conf, grad=g.copy(), inputs=inputs, trained=True)
        if gate is not None:
            tV = float(conf[0, 0] + conf[1, 1]) / float(conf.sum())     # confusionBatchD.totalValid (:125-126)
            if not gate(tV):
                res['trained'] = False
                return False, False
        return f, g
    optimizer_for(st, 'D')(op, st.pD, optstate_for(st, 'D'))
    return res


def optimizer_for(st, which):
    """adversarial.lua:259-267 / 279-287: --{D,G}_optmethod selects the interruptable optimizer."""
    return dict(adam=interruptable_adam, sgd=interruptable_sgd, adagrad=interruptable_adagrad)[
        st.opt.get(which + '_optmethod', 'adam')]


def optstate_for(st, which):
    m = st.opt.get(which + '_optmethod', 'adam')
    if m == 'adam':
        return st.adamD if which == 'D' else st.adamG
    if not hasattr(st, 'optstate'):
        st.optstate = {}
    o = st.opt
    if m == 'sgd':       # train.lua:184-187
        init = dict(learningRate=o.get(which + '_SGD_lr', 0.02), momentum=o.get(which + '_SGD_momentum', 0))
    else:
        init = {}
    return st.optstate.setdefault((m, which), init)


def step_G(st, noise, masks=None):
    """adversarial.lua:275-288: targets all 1 -> fevalG_on_D -> Adam on G."""
    targets = np.ones(noise.shape[0], noise.dtype)
    if masks is not None:
        set_dropout_masks(st.D, masks)
    res = {}

    def op(x):
        f, g, samples, out = feval_G_on_D(st, noise, targets)
        res.update(f=f, f_bce=st.last_f_bce, out=out.copy(), grad=g.copy(), samples=samples.copy())
        return f, g
    optimizer_for(st, 'G')(op, st.pG, optstate_for(st, 'G'))
    return res


def lua_mean(t):
    """adversarial.mean (adversarial.lua:15-27)."""
    return sum(t) / len(t)


def train_epoch(st, dataset, opt, max_accuracy_d, accs_interval, accs, pick, draw_noise, draw_masks, before_step=None):
    """adversarial.train(dataset, maxAccuracyD, accsInterval) -- adversarial.lua:30-335, one epoch.

    opt: batchSize, N_epoch, D_iterations, G_iterations, noiseDim.  `accs` is the module-level adversarial.accs list
    (:13) and persists across epochs.  The three RNG streams of the reference are callbacks so that a test can feed
    both this loop and the device loop the same draws: pick(n) = math.random(n) - 1 (:245), draw_noise(n) =
    NN_UTILS.createNoiseInputs(n) (nn_utils.lua:35-39), draw_masks(B) = the dropout masks of one D forward.
    before_step(kind, k) is a test hook called right before step number k ('D' or 'G').

    Odd thisBatchSize (N_epoch odd): Lua's `for i = 1, thisBatchSize / 2` fills floor(thisBatchSize / 2) real and as
    many fake rows and leaves the last row of `inputs` / `targets` uninitialised (torch.Tensor(n) does not clear);
    that garbage row is not restated -- the even part is used."""
    n_epoch = opt['N_epoch'] if opt['N_epoch'] > 0 else len(dataset)
    B = opt['batchSize']
    data_bs = B // 2
    log = dict(iters=[], trained=0, not_trained=0, conf=np.zeros((2, 2), np.int64), skipped_at=None)

    def gate(tV):
        accs.append(tV)                                   # :156-159
        if len(accs) > accs_interval:
            accs.pop(0)
        do_train = lua_mean(accs) < max_accuracy_d        # :162-167
        if do_train:
            log['trained'] += 1
        else:
            log['not_trained'] += 1
        return do_train
    k = 0
    for t in range(1, n_epoch + 1, data_bs):
        this = min(B, n_epoch - t + 1)
        if this < 4:                                      # :73-76
            log['skipped_at'] = t
            break
        half = this // 2
        this = 2 * half
        it = dict(t=t, batch=this, D=[], G=[])
        for _ in range(opt.get('D_iterations', 1)):
            idx = [pick(len(dataset)) for _ in range(half)]
            real = np.stack([np.asarray(dataset[i], F32) for i in idx])
            nz = draw_noise(half)
            if before_step:
                before_step('D', k)
            r = step_D(st, real, nz, draw_masks(this), gate=gate)
            log['conf'] += r['conf']                      # CONFUSION:add (:115) happens whether or not D trains
            r['idx'] = idx
            it['D'].append(r)
            k += 1
        for _ in range(opt.get('G_iterations', 1)):
            nz = draw_noise(this)
            if before_step:
                before_step('G', k)
            it['G'].append(step_G(st, nz, draw_masks(this)))
            k += 1
        log['iters'].append(it)
    c = log['conf']
    log['totalValid'] = float(c[0, 0] + c[1, 1]) / max(1.0, float(c.sum()))
    return log


# ----------------------------------------------------------------------------------
# coarse-to-fine nets (models_c2f.lua) and step (adversarial_c2f.lua)
# ----------------------------------------------------------------------------------
class TableNet(Module):
    """{first table module, inner Sequential}: models_c2f.lua:113-145 (JoinTable + inner) and :237-278 (CAddTable + inner).
    The nn.Copy modules of the cuda variant are host<->device moves and carry no arithmetic."""

    def __init__(self, first, inner):
        super().__init__()
        self.first, self.inner = first, inner
        self.modules = inner.modules          # parameters live in the inner Sequential only

    def forward(self, xs):
        self._xs = xs
        self._joined = self.first.updateOutput(xs)
        self.output = self.inner.forward(self._joined)
        return self.output

    def backward(self, xs, gy):
        g = self.inner.backward(self._joined, gy)
        self.gradInput = self.first.updateGradInput(xs, g)
        return self.gradInput

    def getParameters(self):
        return self.inner.getParameters()

    def parameters(self):
        return self.inner.parameters()

    def training(self):
        self.inner.training()

    def evaluate(self):
        self.inner.evaluate()

    def astype(self, dtype):
        self.inner.astype(dtype)
        self.dtype = dtype
        return self


def create_G_d(dimensions, rng=None):
    """models_c2f.lua:113-145 create_G_d: JoinTable(2,2){noise[B,1,S,S], cond[B,C,S,S]} -> 5 'same' convs (factor 1)."""
    rng = rng or np.random.default_rng(11)
    c, h, w = dimensions
    inner = Sequential(
        SpatialConvolutionUpsample(c + 1, 64, 3, 3, 1, rng), PReLU(),
        SpatialConvolutionUpsample(64, 64, 3, 3, 1, rng), PReLU(),
        SpatialConvolutionUpsample(64, 128, 5, 5, 1, rng), PReLU(),
        SpatialConvolutionUpsample(128, 256, 5, 5, 1, rng), PReLU(),
        SpatialConvolutionUpsample(256, c, 7, 7, 1, rng), View(c, h, w))
    return TableNet(JoinTable(), inner)


def create_D_c(dimensions, rng=None):
    """models_c2f.lua:237-278 create_D_c: CAddTable{x, cond} -> 4 convs, 2 maxpools, Dropout, 2 Linears."""
    rng = rng or np.random.default_rng(12)
    c, h, w = dimensions
    nfeat = int(256 * 0.25 * 0.25 * h * w)
    inner = Sequential(
        SpatialConvolution(c, 64, 3, 3, 1, 1, 1, None, rng), PReLU(),
        SpatialConvolution(64, 64, 3, 3, 1, 1, 1, None, rng), PReLU(), SpatialMaxPooling(2, 2),
        SpatialConvolution(64, 128, 3, 3, 1, 1, 1, None, rng), PReLU(),
        SpatialConvolution(128, 256, 3, 3, 1, 1, 1, None, rng), PReLU(), SpatialMaxPooling(2, 2),
        Dropout(0.5, rng), View(nfeat), Linear(nfeat, 512, rng), PReLU(), Dropout(0.5, rng),
        Linear(512, 1, rng), Sigmoid())
    return TableNet(CAddTable(), inner)


C2F_OPT = dict(D_L1=1e-7, D_L2=0.0, G_L1=0.0, G_L2=0.0, D_clamp=1.0, G_clamp=5.0)   # train_c2f.lua:27-34


def step_D_c2f(st, diff_real, cond_real, noise_half, cond_fake, masks=None):
    """adversarial_c2f.lua:133-187: B/2 real {diff, coarse} (target 1) || B/2 {G(noise, coarse'), coarse'} (target 0)."""
    fake = st.G.forward([noise_half, cond_fake]).copy()
    inputs = np.concatenate([diff_real, fake], 0)
    cond = np.concatenate([cond_real, cond_fake], 0)
    targets = np.concatenate([np.ones(diff_real.shape[0]), np.zeros(fake.shape[0])]).astype(diff_real.dtype)
    if masks is not None:
        set_dropout_masks(st.D, masks)
    res = {}

    def op(x):
        f, g, out, conf = feval_D(st, [inputs, cond], targets)
        res.update(f=f, f_bce=st.last_f_bce, out=out.copy(), conf=conf, grad=g.copy(), inputs=inputs, cond=cond)
        return f, g
    interruptable_adam(op, st.pD, st.adamD)
    return res


def step_G_c2f(st, noise, cond, masks=None):
    """adversarial_c2f.lua:166-187 + fevalG_on_D (:83-119): df_do = MODEL_D.gradInput[1]."""
    o = st.opt
    targets = np.ones(noise.shape[0], noise.dtype)
    if masks is not None:
        set_dropout_masks(st.D, masks)
    res = {}

    def op(x):
        st.gG[...] = 0
        samples = st.G.forward([noise, cond])
        out = st.D.forward([samples, cond])
        f = st.crit.forward(out, targets)
        f_bce = f
        st.D.backward([samples, cond], st.crit.backward(out, targets))
        st.G.backward([noise, cond], st.D.gradInput[0])
        if o['G_L1'] != 0 or o['G_L2'] != 0:
            p64 = st.pG.astype(np.float64)
            f += o['G_L1'] * np.abs(p64).sum() + o['G_L2'] * (p64 ** 2).sum() / 2
            dt = st.pG.dtype.type
            st.gG += np.sign(st.pG) * dt(o['G_L2']) + st.pG * dt(o['G_L2'])      # quirk C4 (adversarial_c2f.lua:108)
        if o['G_clamp'] != 0:
            np.clip(st.gG, -o['G_clamp'], o['G_clamp'], out=st.gG)
        res.update(f=f, f_bce=f_bce, out=out.copy(), grad=st.gG.copy(), samples=samples.copy())
        return f, st.gG
    interruptable_adam(op, st.pG, st.adamG)
    return res


def train_epoch_c2f(st, train_data, opt, pick, draw_noise, draw_masks, before_step=None):
    """adversarial.train(trainData) of the coarse-to-fine trainer -- adversarial_c2f.lua:10-223, one epoch.
    train_data[i] = dict(diff=, coarse=, fine=) (CHW).  Pick order per D-iteration: B/2 real examples (diff + coarse of the
    SAME example, :125-133), then B/2 fresh examples whose coarse conditions the fake half (:137-142, quirk C13);
    per G-iteration: B fresh conditions (:170-174).  optim.adam == interruptableAdam without the gate."""
    n_epoch = opt['N_epoch'] if opt['N_epoch'] > 0 else len(train_data)
    B = opt['batchSize']
    log = dict(iters=[], conf=np.zeros((2, 2), np.int64), skipped_at=None)
    k = 0
    for t in range(1, n_epoch + 1, B // 2):
        this = min(B, n_epoch - t + 1)
        if this < 4:
            log['skipped_at'] = t
            break
        half = this // 2
        this = 2 * half
        it = dict(t=t, batch=this, D=[], G=[])
        for _ in range(opt.get('D_iterations', 1)):
            idx = [pick(len(train_data)) for _ in range(half)]
            diff = np.stack([np.asarray(train_data[i]['diff'], F32) for i in idx])
            cond_r = np.stack([np.asarray(train_data[i]['coarse'], F32) for i in idx])
            idx_f = [pick(len(train_data)) for _ in range(half)]
            cond_f = np.stack([np.asarray(train_data[i]['coarse'], F32) for i in idx_f])
            nz = draw_noise(half)
            if before_step:
                before_step('D', k)
            r = step_D_c2f(st, diff, cond_r, nz, cond_f, draw_masks(this))
            log['conf'] += r['conf']
            it['D'].append(r)
            k += 1
        for _ in range(opt.get('G_iterations', 1)):
            idx = [pick(len(train_data)) for _ in range(this)]
            cond = np.stack([np.asarray(train_data[i]['coarse'], F32) for i in idx])
            nz = draw_noise(this)
            if before_step:
                before_step('G', k)
            it['G'].append(step_G_c2f(st, nz, cond, draw_masks(this)))
            k += 1
        log['iters'].append(it)
    c = log['conf']
    log['totalValid'] = float(c[0, 0] + c[1, 1]) / max(1.0, float(c.sum()))
    return log


def approx_parzen(G, ds, nsamples, nneighbors, pick, draw_noise):
    """adversarial.approxParzen (adversarial_c2f.lua:305-344) without the checkpoint side effect: for each of `nsamples`
    random examples, the L2 distance (torch.dist: accumulate in double, THTensor_(dist) with accreal) from the ground-truth
    fine image to the nearest of `nneighbors` generations G({noise, coarse}) + coarse.  G runs in its current mode."""
    distances = np.zeros(nsamples, F32)
    for n in range(nsamples):
        ex = ds[pick(len(ds))]
        cond = np.repeat(np.asarray(ex['coarse'], F32)[None], nneighbors, 0)
        noise = draw_noise(nneighbors)
        neighbors = G.forward([noise, cond]) + cond
        fine = np.asarray(ex['fine'], F32)
        d = 1e10
        for i in range(nneighbors):
            diff = neighbors[i].astype(np.float64) - fine.astype(np.float64)
            d = min(float(np.sqrt((diff * diff).sum())), d)
        distances[n] = d
    return distances
