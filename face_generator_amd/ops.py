"""Module-level operator wrappers (NHWC device tensors) over the C ABI -- the nn.Module-protocol entries
(updateOutput / updateGradInput / accGradParameters) of include/facegen_hip.h.  Weights are passed in REFERENCE
layout ([O][I][kH][kW] / [out][in]); packing happens inside the library."""
import torch

from .runtime import get_context


def _ws(ctx, nbytes):
    return torch.empty((nbytes + 3) // 4, dtype=torch.float32, device=ctx.device)


def conv2d_forward(x, w, b, pad=None, upsample2x=False, ctx=None):
    ctx = ctx or get_context()
    B, H, W, Cin = x.shape
    Cout, _, k, _ = w.shape
    pad = (k - 1) // 2 if pad is None else pad
    f = 2 if upsample2x else 1
    y = ctx.empty(B, H * f, W * f, Cout)
    nb = ctx.lib.fg_conv2d_workspace_bytes(B, H, W, Cin, Cout, k, int(upsample2x))
    ws = _ws(ctx, nb)
    ctx.check(ctx.lib.fg_conv2d_forward(ctx.h, x.data_ptr(), w.data_ptr(), b.data_ptr() if b is not None else None,
                                        y.data_ptr(), B, H, W, Cin, Cout, k, pad, int(upsample2x), ws.data_ptr(),
                                        ws.numel() * 4))
    return y


def conv2d_backward_data(gy, w, in_hw, pad=None, upsample2x=False, ctx=None):
    ctx = ctx or get_context()
    B = gy.shape[0]
    H, W = in_hw
    Cout, Cin, k, _ = w.shape
    pad = (k - 1) // 2 if pad is None else pad
    gx = ctx.empty(B, H, W, Cin)
    nb = ctx.lib.fg_conv2d_workspace_bytes(B, H, W, Cin, Cout, k, int(upsample2x))
    ws = _ws(ctx, nb)
    ctx.check(ctx.lib.fg_conv2d_backward_data(ctx.h, gy.data_ptr(), w.data_ptr(), gx.data_ptr(), B, H, W, Cin, Cout, k,
                                              pad, int(upsample2x), ws.data_ptr(), ws.numel() * 4))
    return gx


def conv2d_backward_weight(x, gy, k, pad=None, upsample2x=False, gw=None, gb=None, beta=0.0, ctx=None):
    ctx = ctx or get_context()
    B, H, W, Cin = x.shape
    Cout = gy.shape[-1]
    pad = (k - 1) // 2 if pad is None else pad
    gw = ctx.zeros(Cout, Cin, k, k) if gw is None else gw
    gb = ctx.zeros(Cout) if gb is None else gb
    nb = ctx.lib.fg_conv2d_workspace_bytes(B, H, W, Cin, Cout, k, int(upsample2x))
    ws = _ws(ctx, nb)
    ctx.check(ctx.lib.fg_conv2d_backward_weight(ctx.h, x.data_ptr(), gy.data_ptr(), gw.data_ptr(), gb.data_ptr(), beta,
                                                B, H, W, Cin, Cout, k, pad, int(upsample2x), ws.data_ptr(),
                                                ws.numel() * 4))
    return gw, gb


def linear_forward(x, w, b, ctx=None):
    ctx = ctx or get_context()
    B, K = x.shape
    N = w.shape[0]
    y = ctx.empty(B, N)
    ws = _ws(ctx, ctx.lib.fg_linear_workspace_bytes(B, K, N))
    ctx.check(ctx.lib.fg_linear_forward(ctx.h, x.data_ptr(), w.data_ptr(), b.data_ptr() if b is not None else None,
                                        y.data_ptr(), B, K, N, ws.data_ptr(), ws.numel() * 4))
    return y


def linear_backward_data(gy, w, ctx=None):
    ctx = ctx or get_context()
    B, N = gy.shape
    K = w.shape[1]
    gx = ctx.empty(B, K)
    ws = _ws(ctx, ctx.lib.fg_linear_workspace_bytes(B, K, N))
    ctx.check(ctx.lib.fg_linear_backward_data(ctx.h, gy.data_ptr(), w.data_ptr(), gx.data_ptr(), B, K, N,
                                              ws.data_ptr(), ws.numel() * 4))
    return gx


def linear_backward_weight(x, gy, gw=None, gb=None, beta=0.0, ctx=None):
    ctx = ctx or get_context()
    B, K = x.shape
    N = gy.shape[1]
    gw = ctx.zeros(N, K) if gw is None else gw
    gb = ctx.zeros(N) if gb is None else gb
    ws = _ws(ctx, ctx.lib.fg_linear_workspace_bytes(B, K, N))
    ctx.check(ctx.lib.fg_linear_backward_weight(ctx.h, x.data_ptr(), gy.data_ptr(), gw.data_ptr(), gb.data_ptr(), beta,
                                                B, K, N, ws.data_ptr(), ws.numel() * 4))
    return gw, gb


def batchnorm_forward(x, gamma, beta, slope=None, running_mean=None, running_var=None, eps=1e-5, momentum=0.1,
                      train=True, ctx=None):
    ctx = ctx or get_context()
    C = x.shape[-1]
    rows = x.numel() // C
    y = torch.empty_like(x)
    mean, invstd = ctx.empty(C), ctx.empty(C)
    scratch = ctx.empty(ctx.lib.fg_bn_scratch_floats(C))
    ctx.check(ctx.lib.fg_batchnorm_forward(
        ctx.h, x.data_ptr(), y.data_ptr(), rows, C, gamma.data_ptr(), beta.data_ptr(),
        slope.data_ptr() if slope is not None else None, mean.data_ptr(), invstd.data_ptr(),
        running_mean.data_ptr() if running_mean is not None else None,
        running_var.data_ptr() if running_var is not None else None, eps, momentum, int(train), scratch.data_ptr()))
    return y, mean, invstd


def batchnorm_backward(x, gy, gamma, beta, mean, invstd, slope=None, ctx=None):
    ctx = ctx or get_context()
    C = x.shape[-1]
    rows = x.numel() // C
    gx = torch.empty_like(x)
    gg, gb, gs = ctx.zeros(C), ctx.zeros(C), ctx.zeros(1)
    scratch = ctx.empty(ctx.lib.fg_bn_scratch_floats(C))
    ctx.check(ctx.lib.fg_batchnorm_backward(
        ctx.h, x.data_ptr(), gy.data_ptr(), gx.data_ptr(), rows, C, gamma.data_ptr(), beta.data_ptr(),
        slope.data_ptr() if slope is not None else None, mean.data_ptr(), invstd.data_ptr(), gg.data_ptr(),
        gb.data_ptr(), gs.data_ptr() if slope is not None else None, 0.0, scratch.data_ptr()))
    return gx, gg, gb, gs


def prelu_forward(x, slope, mask=None, mscale=1.0, ctx=None):
    ctx = ctx or get_context()
    y = torch.empty_like(x)
    ctx.check(ctx.lib.fg_prelu_forward(ctx.h, x.data_ptr(), slope.data_ptr(),
                                       mask.data_ptr() if mask is not None else None, mscale, y.data_ptr(), x.numel()))
    return y


def prelu_backward(x, gy, slope, mask=None, mscale=1.0, ctx=None):
    ctx = ctx or get_context()
    gx = torch.empty_like(x)
    gs = ctx.zeros(1)
    scratch = ctx.empty(1024)
    ctx.check(ctx.lib.fg_prelu_backward(ctx.h, x.data_ptr(), gy.data_ptr(), slope.data_ptr(),
                                        mask.data_ptr() if mask is not None else None, mscale, gx.data_ptr(),
                                        gs.data_ptr(), 0.0, x.numel(), scratch.data_ptr()))
    return gx, gs


def actpool_forward(x, slope, mask, mscale=1.0, ctx=None):
    ctx = ctx or get_context()
    B, H, W, C = x.shape
    y = ctx.empty(B, H // 2, W // 2, C)
    ctx.check(ctx.lib.fg_actpool_forward(ctx.h, x.data_ptr(), slope.data_ptr(),
                                         mask.data_ptr() if mask is not None else None, mscale, y.data_ptr(), B, H, W, C))
    return y


def actpool_backward(x, gy, slope, mask, mscale=1.0, ctx=None):
    ctx = ctx or get_context()
    B, H, W, C = x.shape
    gx = torch.empty_like(x)
    gs = ctx.zeros(1)
    scratch = ctx.empty(1024)
    ctx.check(ctx.lib.fg_actpool_backward(ctx.h, x.data_ptr(), gy.data_ptr(), slope.data_ptr(),
                                          mask.data_ptr() if mask is not None else None, mscale, gx.data_ptr(),
                                          gs.data_ptr(), 0.0, B, H, W, C, scratch.data_ptr()))
    return gx, gs


def adam_step(ctx, p, g, m, v, t, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, gscale=1.0, l1_mul=0.0, l2=0.0, clamp=0.0,
              g_out=None):
    ctx.check(ctx.lib.fg_adam_fused(ctx.h, p.data_ptr(), g.data_ptr(), m.data_ptr(), v.data_ptr(), p.numel(), gscale,
                                    l1_mul, l2, clamp, lr, beta1, beta2, eps, t,
                                    g_out.data_ptr() if g_out is not None else None))


def scale_bilinear(x, width, height, layout="nhwc", ctx=None):
    """image.scale(x, width, height) (Torch7 `image`, bilinear; dataset_c2f.lua:54-55) on a device batch: x [N][H][W][C] (layout
    "nhwc") or [N][C][H][W] ("nchw").  include/facegen_hip.h fg_scale_bilinear."""
    ctx = ctx or get_context()
    nchw = {"nhwc": 0, "nchw": 1}[layout]
    x = x.contiguous()
    if nchw:
        N, C, H, W = x.shape
        y = ctx.empty(N, C, height, width)
    else:
        N, H, W, C = x.shape
        y = ctx.empty(N, height, width, C)
    ctx.check(ctx.lib.fg_scale_bilinear(ctx.h, x.data_ptr(), y.data_ptr(), N, C, H, W, height, width, nchw))
    return y


def c2f_coarse_diff(fine, coarse_scale, layout="nhwc", ctx=None):
    """dataset._toResult's arithmetic (dataset_c2f.lua:53-61) on a device batch of square images: -> (coarse, diff) with
    coarse = scale(scale(fine, cs, cs), s, s) and diff = fine - coarse.  include/facegen_hip.h fg_c2f_coarse_diff."""
    ctx = ctx or get_context()
    nchw = {"nhwc": 0, "nchw": 1}[layout]
    fine = fine.contiguous()
    if nchw:
        N, C, S, S2 = fine.shape
    else:
        N, S, S2, C = fine.shape
    if S != S2:
        raise ValueError("c2f_coarse_diff: square images expected, got %dx%d" % (S, S2))
    coarse, diff = torch.empty_like(fine), torch.empty_like(fine)
    tmp = ctx.empty(N * C * coarse_scale * coarse_scale)
    ctx.check(ctx.lib.fg_c2f_coarse_diff(ctx.h, fine.data_ptr(), coarse.data_ptr(), diff.data_ptr(), tmp.data_ptr(), N, C, S,
                                         coarse_scale, nchw))
    return coarse, diff
