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


from oracle.device_branches import adopt_device_branches, count_branch_flips, count_branch_units   # noqa: E402,F401  (shared with smoke())


def oracle_backward_on_device_branches(ctx, dn, net, x, gy, gflat=None, max_flip_frac=1e-5):
    """Small-batch gradient tests (round 5): the device runs first, then the oracle re-runs its forward on the DEVICE's PReLU /
    max-pool branch decisions and takes its backward from there -- the procedure of the BASELINE-size tests.  Before round 5 these
    tests re-drew their inputs until no oracle pre-activation sat within 1e-7 (relative) of a kink; a Winograd contraction rounds
    ~1.7 x more than the direct convolution (tests/test_gpu_wino.py), which that margin does not cover, and a margin that would
    cover it is never met by a million units.  The number of adopted decisions that differ from the oracle's own is asserted
    (<= max(1, 1e-5 of the units)): the hook cannot hide a wrong kernel.  BatchNorm running statistics are restored (the re-run is
    a second train-mode forward).  Returns the oracle's input gradient."""
    from oracle import torch7_nn as O
    bns = [m for m in O.walk_modules(net) if isinstance(m, O.SpatialBatchNormalization)]
    saved = [(m.running_mean.copy(), m.running_var.copy()) for m in bns]
    adopt_device_branches(ctx, dn, net)
    try:
        net.forward(x)
        for m, (rm, rv) in zip(bns, saved):
            m.running_mean, m.running_var = rm, rv
        if gflat is not None:
            gflat[...] = 0
        gin = net.backward(x, gy)
        flips, units = count_branch_flips(net), count_branch_units(net)
    finally:
        adopt_device_branches(ctx, dn, net, clear=True)
    assert flips <= max(1, max_flip_frac * units), "%d of %d PReLU decisions adopted from the device differ from the oracle's" % (flips, units)
    return gin
