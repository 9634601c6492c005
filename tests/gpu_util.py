import numpy as np
import torch


def nhwc(a, dev):
    """numpy NCHW -> device NHWC tensor (test plumbing only)."""
    t = torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device=dev)
    return t.permute(0, 2, 3, 1).contiguous() if t.dim() == 4 else t


def nchw(t):
    """device NHWC tensor -> numpy NCHW."""
    if t.dim() == 4:
        t = t.permute(0, 3, 1, 2)
    return t.contiguous().cpu().numpy()


def dev(a, device):
    return torch.tensor(np.ascontiguousarray(a), dtype=torch.float32, device=device)


def close(a, b, atol, rtol=0.0, what=""):
    a = np.asarray(a, np.float64); b = np.asarray(b, np.float64)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    err = np.abs(a - b)
    tol = atol + rtol * np.abs(b)
    bad = err > tol
    if bad.any():
        i = np.unravel_index(np.argmax(err - tol), err.shape)
        raise AssertionError("%s: %d/%d elements differ; worst at %s: got %.8g want %.8g (|err| %.3g, max|ref| %.3g)"
                             % (what, bad.sum(), bad.size, i, a[i], b[i], err[i], np.abs(b).max()))


def close_after_first_adam_step(p_dev, p_ref, g_dev, g_ref, what, lr=1e-3, beta2=0.999, eps=1e-8, atol=2e-6):
    """Parameters after the FIRST Adam step from zero state (interruptable_optimizers.lua:72-90): dp = -lr * g / (|g| +
    eps / sqrt(1 - beta2)), i.e. essentially -lr * sign(g).  Where |g| is of the order of eps the step is a steep function
    of g, so two fp32 computations of the same gradient (different summation orders) may legitimately land 1e-6..1e-3 apart
    on that parameter.  Bar: 2e-6 plus exactly the spread the two gradients themselves imply -- this still pins the
    optimizer arithmetic, without demanding bit-equal near-zero gradients."""
    g_dev = np.asarray(g_dev, np.float64); g_ref = np.asarray(g_ref, np.float64)
    e = eps / np.sqrt(1.0 - beta2)
    f = lambda g: g / (np.abs(g) + e)
    tol = atol + 1.01 * lr * np.abs(f(g_dev) - f(g_ref))
    err = np.abs(np.asarray(p_dev, np.float64) - np.asarray(p_ref, np.float64))
    bad = err > tol
    if bad.any():
        i = int(np.argmax(err - tol))
        raise AssertionError("%s: %d/%d parameters differ beyond the Adam-sensitivity bound; worst at %d: |err| %.3g tol %.3g "
                             "(g_dev %.3e g_ref %.3e)" % (what, bad.sum(), bad.size, i, err[i], tol[i], g_dev[i], g_ref[i]))


def adopt_device_branches(ctx, dn, onet, clear=False, params=None, also=()):
    """Copy the DEVICE's branch decisions of its last forward into the oracle net: PReLU (pre-activation > 0) and the 2x2
    max-pool argmax (PReLU.pos_override / SpatialMaxPooling.indices_override in oracle/torch7_nn.py).  PReLU is not
    differentiable at 0 and a max-pool not at a tie: with 10^7 units per step at the BASELINE sizes a few always sit within
    fp32 rounding of the kink, where the oracle's GEMM and the device's legitimately land on different sides; copying the
    decisions compares the gradients on identical branches at the plain SURVEY 8(c) bars, with no flip allowance.

    The pre-activation is read from the device plan: the stage output in front of the PReLU (conv / Linear[+View] / fused
    PReLU in front of a max-pool); for the fused BatchNorm+PReLU stage z = gamma * xhat + beta is re-evaluated from the
    conv output and the batch statistics the forward saved (fg_net_bn_saved_stats), in float32 with every operation rounded
    separately -- the expression bn_z() of pointwise.hip evaluates in forward and backward alike.
    `params`: the device's flat parameter vector as it was DURING that forward (the optimizer step that follows a backward
    moves gamma / beta); `also`: further oracle nets (e.g. the float64 twin) that receive the same decisions."""
    import ctypes
    from oracle import torch7_nn as O
    mods = getattr(onet, "inner", onet).modules
    twins = [getattr(o, "inner", o).modules for o in also]
    lib = ctx.lib
    P = dn.params if params is None else params
    for i, m in enumerate(mods):
        if isinstance(m, O.PReLU):
            if clear:
                for mm in [m] + [t[i] for t in twins]:
                    mm.pos_override = None
                continue
            prev = mods[i - 1]
            if isinstance(prev, O.SpatialBatchNormalization):
                # z = ((x - mean) * invstd) * gamma + beta, every operation rounded separately (bn_z in pointwise.hip), from the
                # statistics the forward SAVED -- the same expression the device's backward takes its branch from
                x = nchw(dn.layer_output(i - 2))
                mo, io, cc = ctypes.c_longlong(), ctypes.c_longlong(), ctypes.c_int()
                ctx.check(lib.fg_net_bn_saved_stats(dn.h, i - 1, ctypes.byref(mo), ctypes.byref(io), ctypes.byref(cc)))
                C = cc.value
                mu = dn.ws[mo.value: mo.value + C].cpu().numpy().reshape(1, C, 1, 1)
                istd = dn.ws[io.value: io.value + C].cpu().numpy().reshape(1, C, 1, 1)
                wo, wn, bo, bn = dn.param_offsets(i - 1)
                gam = P[wo:wo + wn].cpu().numpy().reshape(1, C, 1, 1)
                bet = P[bo:bo + bn].cpu().numpy().reshape(1, C, 1, 1)
                zz = (((x - mu).astype(np.float32) * istd).astype(np.float32) * gam).astype(np.float32) + bet
                pos = zz.astype(np.float32) > 0
                for mm in [m] + [t[i] for t in twins]:
                    mm.pos_override = pos
                continue
            else:
                z = dn.layer_output(i - 1)
            pos = nchw(z) > 0
            for mm in [m] + [t[i] for t in twins]:
                mm.pos_override = pos
        elif isinstance(m, O.SpatialMaxPooling):
            if clear:
                for mm in [m] + [t[i] for t in twins]:
                    mm.indices_override = None
                continue
            x = nchw(dn.layer_output(i - 1))
            n, c, h, w = x.shape
            blk = x.reshape(n, c, h // 2, 2, w // 2, 2).transpose(0, 1, 2, 4, 3, 5).reshape(n, c, h // 2, w // 2, 4)
            idx = blk.argmax(axis=-1)                           # first max in scan order, like the kernel
            for mm in [m] + [t[i] for t in twins]:
                mm.indices_override = idx


def count_branch_flips(onet):
    """After an oracle forward with adopted branches: how many PReLU units would the oracle itself have decided otherwise."""
    from oracle import torch7_nn as O
    net = getattr(onet, "inner", onet)
    k = 0
    for m, x in zip(net.modules, net._inputs):
        if isinstance(m, O.PReLU) and m.pos_override is not None:
            k += int(((x > 0) != m.pos_override.reshape(x.shape)).sum())
    return k
