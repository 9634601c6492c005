import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
from oracle import torch7_nn as O
from gpu_util import nhwc, nchw, dev
from face_generator_amd import ops
from face_generator_amd.runtime import get_context
ctx = get_context(0); d = ctx.device
rng = np.random.default_rng(0)
for (B, H, W, Cin, Cout) in [(6, 32, 32, 128, 1), (2, 32, 32, 128, 1), (6, 16, 16, 128, 1), (6, 32, 32, 128, 3), (6, 32, 32, 64, 1)]:
    conv = O.SpatialConvolution(Cin, Cout, 3, 3, 1, 1, 1, 1, rng)
    x = rng.standard_normal((B, Cin, H, W)).astype(np.float32)
    y = conv.forward(x); gy = rng.standard_normal(y.shape).astype(np.float32)
    gx = conv.backward(x, gy)
    gxd = nchw(ops.conv2d_backward_data(nhwc(gy, d), dev(conv.weight, d), (H, W)))
    err = np.abs(gxd - gx)
    perch = err.max(axis=(0, 2, 3))
    print((B, H, W, Cin, Cout), "dgrad max err", err.max(), "bad channels", np.where(perch > 1e-4)[0][:10], "scale", np.abs(gx).max())
    gw, gb = ops.conv2d_backward_weight(nhwc(x, d), nhwc(gy, d), 3)
    print("   wgrad err", np.abs(gw.cpu().numpy() - conv.gradWeight).max(), np.abs(gb.cpu().numpy() - conv.gradBias).max())
# BN backward at M = 6144, C = 128
for (B, H, W, C) in [(6, 32, 32, 128), (6, 16, 16, 256)]:
    bn = O.SpatialBatchNormalization(C, rng=rng); pr = O.PReLU()
    x = rng.standard_normal((B, C, H, W)).astype(np.float32)
    z = bn.forward(x); y = pr.forward(z); gy = rng.standard_normal(y.shape).astype(np.float32)
    gx = bn.backward(x, pr.backward(z, gy))
    rm, rv = ctx.zeros(C), torch.ones(C, device=d)
    yd, mean, invstd = ops.batchnorm_forward(nhwc(x, d), dev(bn.weight, d), dev(bn.bias, d), dev(pr.weight, d), rm, rv)
    gxd, gg, gb, gs = ops.batchnorm_backward(nhwc(x, d), nhwc(gy, d), dev(bn.weight, d), dev(bn.bias, d), mean, invstd, dev(pr.weight, d))
    err = np.abs(nchw(gxd) - gx).max(axis=(0, 2, 3))
    print("bn", (B, H, W, C), "gx err", err.max(), np.where(err > 1e-4)[0][:10], "gg", np.abs(gg.cpu().numpy() - bn.gradWeight).max())
