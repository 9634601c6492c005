import sys, os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__)))); sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import torch, numpy as np
from face_generator_amd.runtime import get_context
from test_gpu_step_abi import make32, masks32
ctx = get_context(0)
B = 8
outs = {}
for flags in (15, 7, 15):
    ctx.set_fusion(flags)
    tr, G, D = make32(ctx, B, dict(D_L1=1e-5, D_L2=1e-4, G_L2=1e-5), True, seed=11)
    real = ctx.uniform((B // 2, 32, 32, 3), 0.0, 1.0, seed=9)
    for it in range(1):
        tr.step_D(real, ctx.uniform((B // 2, 100), -1.0, 1.0, seed=10 + it), masks32(ctx, B, 20 + it))
        tr.step_G(ctx.uniform((B, 100), -1.0, 1.0, seed=40 + it), masks32(ctx, B, 50 + it))
    tr.finish_pending()
    outs.setdefault(flags, []).append(dict(pG=G.getParameters()[0].clone().cpu().numpy(), pD=D.getParameters()[0].clone().cpu().numpy(),
                       gG=G.getParameters()[1].clone().cpu().numpy(), gD=D.getParameters()[1].clone().cpu().numpy()))
    nets = (G, D)
def layout(net):
    off = 0; rows = []
    for m in net.modules if hasattr(net, "modules") else []:
        for (mm, pn, gn) in m.parameters():
            n = getattr(mm, pn).size
            rows.append((off, off + n, type(mm).__name__ + "." + pn)); off += n
    return rows
for k, net in (("pG", nets[0]), ("pD", nets[1]), ("gG", nets[0]), ("gD", nets[1])):
    a, b, a2 = outs[15][0][k], outs[7][0][k], outs[15][1][k]
    d = np.nonzero(a != b)[0]; d2 = np.nonzero(a != a2)[0]
    print(k, "size", a.size, "fused vs unfused mismatches", d.size, "| fused vs fused again", d2.size)
    if d.size:
        print("   first", d[:8], "last", d[-4:], "max abs diff", np.abs(a - b).max(), "max rel", (np.abs(a - b) / (np.abs(b) + 1e-30)).max())
        for (lo, hi, name) in layout(net):
            c = ((d >= lo) & (d < hi)).sum()
            if c: print("   %-40s [%d, %d): %d of %d differ" % (name, lo, hi, c, hi - lo))
