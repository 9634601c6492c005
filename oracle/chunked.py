"""Test infrastructure (like the rest of oracle/): the closures of adversarial_c2f.lua on a WHOLE batch, evaluated by walking
the batch in chunks.

`create_G_d` / `create_D_c` (models_c2f.lua:113-145, 237-278) contain no BatchNorm, so every sample's forward and backward is
independent of the rest of the batch; only three things couple the samples and they are all sums:
  * BCECriterion's sizeAverage (train_c2f.lua:95; the 1/B of loss and gradient),
  * accGradParameters (weight / bias / PReLU-slope gradients are sums over the batch),
  * the confusion counts.
Walking the batch in chunks of 8 and accumulating those sums (in the dtype of the nets: the parity tests run this on nets
converted to float64, so the sums over up to 524 288 pixels are float64 sums) gives exactly fevalD / fevalG_on_D of
adversarial_c2f.lua:40-119 at BASELINE configs[3] / [4]'s own batch sizes (128, and 64 with D_iterations = 2) without the
oracle's im2col buffers growing with B.  Nothing here is imported by the product path."""
import numpy as np

from . import torch7_nn as O


def _mods(net):
    return getattr(net, "inner", net).modules


def collect_branches(net):
    """The branch decisions `adopt_device_branches` left on the modules (whole-batch arrays), as {module index: array}."""
    br = {}
    for i, m in enumerate(_mods(net)):
        if isinstance(m, O.PReLU) and m.pos_override is not None:
            br[i] = ("pos", m.pos_override)
        if isinstance(m, O.SpatialMaxPooling) and getattr(m, "indices_override", None) is not None:
            br[i] = ("idx", m.indices_override)
    return br


def _apply_branches(net, br, sl):
    mods = _mods(net)
    for i, (kind, full) in (br or {}).items():
        if kind == "pos":
            mods[i].pos_override = full[sl]
        else:
            mods[i].indices_override = full[sl]


def _restore_branches(net, br):
    mods = _mods(net)
    for i, (kind, full) in (br or {}).items():
        if kind == "pos":
            mods[i].pos_override = full
        else:
            mods[i].indices_override = full


class ChunkedC2F:
    """fevalD / fevalG_on_D of adversarial_c2f.lua over a whole batch, `chunk` samples at a time.  `st` is a GanState whose
    nets are TableNets without BatchNorm."""

    def __init__(self, st, chunk=8):
        self.st, self.chunk = st, chunk
        for net in (st.G, st.D):
            for m in O.walk_modules(getattr(net, "inner", net)):
                assert not isinstance(m, O.SpatialBatchNormalization), "a BatchNorm couples the samples: no chunking"
                m.keep_buffers = True
        self.flips = {"G": 0, "D": 0}
        self.units = {"G": 0, "D": 0}

    # -- bookkeeping -----------------------------------------------------------------------------------------------------
    def _count(self, which, net):
        """PReLU units of the last chunk the oracle itself would have decided differently from the adopted decisions."""
        inner = getattr(net, "inner", net)
        for m, x in zip(inner.modules, inner._inputs):
            if isinstance(m, O.PReLU):
                self.units[which] += x.size
                if m.pos_override is not None:
                    self.flips[which] += int(((x > 0) != m.pos_override.reshape(x.shape)).sum())

    def _begin(self, nets):
        self.flips = {"G": 0, "D": 0}
        self.units = {"G": 0, "D": 0}
        self._cond2 = {}
        for net in nets:
            for i, m in enumerate(_mods(net)):
                if isinstance(m, O.PReLU):
                    self._cond2[id(m)] = 0.0

    def _after_backward(self, net):
        for m in _mods(net):
            if isinstance(m, O.PReLU) and id(m) in self._cond2:
                self._cond2[id(m)] += getattr(m, "gw_cond", 0.0) ** 2

    def _end(self, nets):
        for net in nets:
            for m in _mods(net):
                if isinstance(m, O.PReLU) and id(m) in self._cond2:
                    m.gw_cond = float(np.sqrt(self._cond2[id(m)]))     # condition scale of the whole-batch slope sum

    def _slices(self, B):
        return [slice(i, min(i + self.chunk, B)) for i in range(0, B, self.chunk)]

    # -- the closures ----------------------------------------------------------------------------------------------------
    def generate(self, noise, cond):
        """MODEL_G:forward({noise, cond}) in train mode (adversarial_c2f.lua:148): no branch decisions matter (no backward)."""
        st = self.st
        out = None
        for sl in self._slices(noise.shape[0]):
            y = st.G.forward([noise[sl], cond[sl]])
            if out is None:
                out = np.empty((noise.shape[0],) + y.shape[1:], y.dtype)
            out[sl] = y
        return out

    def feval_D(self, inputs, cond, targets, masks, brD=None):
        """adversarial_c2f.lua:40-80: zero grads, D forward, BCE, D backward, L1/L2 penalty, clamp.
        -> dict(f, f_bce, out, conf, grad (a copy of the flat gradient after penalty + clamp))."""
        st, o = self.st, self.st.opt
        B = inputs.shape[0]
        dt = st.pD.dtype.type
        st.gD[...] = 0
        self._begin([st.D])
        out = np.empty((B, 1), st.pD.dtype)
        f = 0.0
        for sl in self._slices(B):
            n = sl.stop - sl.start
            O.set_dropout_masks(st.D, [m[sl] for m in masks])
            _apply_branches(st.D, brD, sl)
            oc = st.D.forward([inputs[sl], cond[sl]])
            self._count("D", st.D)
            f += st.crit.forward(oc, targets[sl]) * n / B
            df = st.crit.backward(oc, targets[sl]) * dt(n / B)          # sizeAverage over the WHOLE batch
            st.D.backward([inputs[sl], cond[sl]], df)
            self._after_backward(st.D)
            out[sl] = oc
        _restore_branches(st.D, brD)
        self._end([st.D])
        f_bce = f
        if o['D_L1'] != 0 or o['D_L2'] != 0:                             # adversarial_c2f.lua:60-70
            p64 = st.pD.astype(np.float64)
            f += o['D_L1'] * np.abs(p64).sum() + o['D_L2'] * (p64 ** 2).sum() / 2
            st.gD += np.sign(st.pD) * dt(o['D_L1']) + st.pD * dt(o['D_L2'])
        conf = np.zeros((2, 2), np.int64)
        for i in range(B):
            conf[1 if out[i, 0] > 0.5 else 0, int(targets[i])] += 1
        if o['D_clamp'] != 0:
            np.clip(st.gD, -o['D_clamp'], o['D_clamp'], out=st.gD)
        return dict(f=f, f_bce=f_bce, out=out, conf=conf, grad=st.gD.copy())

    def feval_G(self, noise, cond, masks, brD=None, brG=None):
        """adversarial_c2f.lua:83-119: zero G grads, G forward, D forward, BCE vs ones, D backward (gradInput[1]), G backward,
        penalty (quirk C4: :108 multiplies sign(p) by G_L2), clamp."""
        st, o = self.st, self.st.opt
        B = noise.shape[0]
        dt = st.pG.dtype.type
        targets = np.ones(B, st.pG.dtype)
        st.gG[...] = 0
        self._begin([st.G, st.D])
        out = np.empty((B, 1), st.pG.dtype)
        samples = None
        f = 0.0
        for sl in self._slices(B):
            n = sl.stop - sl.start
            O.set_dropout_masks(st.D, [m[sl] for m in masks])
            _apply_branches(st.D, brD, sl)
            _apply_branches(st.G, brG, sl)
            s = st.G.forward([noise[sl], cond[sl]])
            self._count("G", st.G)
            oc = st.D.forward([s, cond[sl]])
            self._count("D", st.D)
            f += st.crit.forward(oc, targets[sl]) * n / B
            st.D.backward([s, cond[sl]], st.crit.backward(oc, targets[sl]) * dt(n / B))
            st.G.backward([noise[sl], cond[sl]], st.D.gradInput[0])
            self._after_backward(st.G)
            if samples is None:
                samples = np.empty((B,) + s.shape[1:], s.dtype)
            samples[sl] = s
            out[sl] = oc
        _restore_branches(st.D, brD)
        _restore_branches(st.G, brG)
        self._end([st.G])
        f_bce = f
        if o['G_L1'] != 0 or o['G_L2'] != 0:
            p64 = st.pG.astype(np.float64)
            f += o['G_L1'] * np.abs(p64).sum() + o['G_L2'] * (p64 ** 2).sum() / 2
            st.gG += np.sign(st.pG) * dt(o['G_L2']) + st.pG * dt(o['G_L2'])
        if o['G_clamp'] != 0:
            np.clip(st.gG, -o['G_clamp'], o['G_clamp'], out=st.gG)
        return dict(f=f, f_bce=f_bce, out=out, samples=samples, grad=st.gG.copy())

    # -- the steps (closure + optim.adam, adversarial_c2f.lua:123-187) ----------------------------------------------------
    def step_D(self, diff_real, cond_real, noise_half, cond_fake, masks, brD=None):
        st = self.st
        fake = self.generate(noise_half, cond_fake)
        inputs = np.concatenate([diff_real, fake], 0)
        cond = np.concatenate([cond_real, cond_fake], 0)
        targets = np.concatenate([np.ones(diff_real.shape[0]), np.zeros(fake.shape[0])]).astype(diff_real.dtype)
        res = {}

        def op(x):
            res.update(self.feval_D(inputs, cond, targets, masks, brD))
            return res["f"], st.gD
        O.interruptable_adam(op, st.pD, st.adamD)
        return res

    def step_G(self, noise, cond, masks, brD=None, brG=None):
        st = self.st
        res = {}

        def op(x):
            res.update(self.feval_G(noise, cond, masks, brD, brG))
            return res["f"], st.gG
        O.interruptable_adam(op, st.pG, st.adamG)
        return res
