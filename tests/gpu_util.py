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
