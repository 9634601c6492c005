import sys, numpy as np, torch
sys.path.insert(0, '.'); sys.path.insert(0, 'tests')
import test_gpu_net as T
from gpu_util import nhwc, nchw, dev
from face_generator_amd.runtime import get_context
ctx = get_context(0)
for (C, B) in [(1, 4), (1, 6), (1, 18), (3, 6), (3, 8), (1, 8), (3, 2), (1, 2)]:
    st, Gd, Dd, rng = T.build(ctx, C, B, seed=100 + C + B)
    noise = rng.uniform(-1, 1, (B, 100)).astype(np.float32)
    img = st.G.forward(noise)
    gy = rng.standard_normal(img.shape).astype(np.float32)
    st.gG[...] = 0
    st.G.backward(noise, gy)
    dn = Gd.device_net
    y = dn.forward(dev(noise, ctx.device))
    dn.backward(nhwc(gy, ctx.device), param_grads=True, input_grad=False)
    try:
        T.check_flat_grads(dn.grads.cpu().numpy(), st.G, "G")
        print((C, B), "OK")
    except AssertionError as e:
        print((C, B), "FAIL\n", str(e)[:1500])
    # which BN10 channel?
    g = dn.grads.cpu().numpy()
    off = 0
    for i, m in enumerate(st.G.modules):
        for (mm, pn, gn) in m.parameters():
            ref = getattr(mm, gn).reshape(-1)
            if i == 9:
                err = np.abs(g[off:off + ref.size] - ref)
                print("   BN10", pn, "worst channel", err.argmax(), err.max())
            off += ref.size
