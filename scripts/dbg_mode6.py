import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from oracle import torch7_nn as O
from gpu_util import nhwc, nchw, dev
from test_gpu_net import build, draw_kink_safe
from face_generator_amd.runtime import get_context
ctx = get_context(0); d = ctx.device
B, C = 4, 3
st, Gd, Dd, rng = build(ctx, C, B, seed=700)
noise, img = draw_kink_safe(rng, lambda: rng.uniform(-1, 1, (B, 100)).astype(np.float32), st.G.forward, [st.G])
gy = rng.standard_normal(img.shape).astype(np.float32)
st.gG[...] = 0; st.G.backward(noise, gy)
res = {}
for mode in (0, 6, 0, 6):
    ctx.set_math(mode)
    Gd.zeroGradParameters()
    y = Gd.forward_modules(dev(noise, d))
    Gd.backward_modules(nhwc(gy, d))
    pG, gG = Gd.getParameters()
    g = gG.cpu().numpy().copy()
    print("mode", mode, "max|g - oracle|", np.abs(g - st.gG).max())
    if mode in res:
        print("   repeat diff", np.abs(g - res[mode]).max())
    res[mode] = g
dd = np.abs(res[0] - res[6]); idx = np.nonzero(dd > 0)[0]
print("entries differing between modes:", len(idx), "first/last", idx[:3], idx[-3:])
off = 0
for i, m in enumerate(st.G.modules):
    for name in ("weight", "bias"):
        t = getattr(m, name, None)
        if t is None: continue
        n = t.size; seg = dd[off:off + n]
        print("module %2d %-28s %-6s off %8d n %8d  max mode diff %.3e  (max|ref| %.3e)  err0 %.3e err6 %.3e" % (i, type(m).__name__, name, off, n, seg.max(), np.abs(st.gG[off:off+n]).max(),
              np.abs(res[0][off:off+n]-st.gG[off:off+n]).max(), np.abs(res[6][off:off+n]-st.gG[off:off+n]).max()))
        off += n
