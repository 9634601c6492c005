"""Diagnose a smoke() mismatch: which flat-parameter entries differ after the Adam step, and what their gradients were."""
import os, sys
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import torch7_nn as O
from face_generator_amd import models, adversarial
from face_generator_amd.runtime import get_context
ctx = get_context(0)
B, C = 4, 3
rng = np.random.default_rng(1)
G = O.create_G32((C, 32, 32), 100, rng, weight_init_=False); D = O.create_D32b((C, 32, 32), rng)
st = O.GanState(G, D)
Gd = models.create_G((C, 32, 32), 100).cuda(ctx, max_batch=B)
Dd = models.create_D((C, 32, 32)).cuda(ctx, max_batch=B)
Gd.getParameters()[0].copy_(torch.tensor(st.pG)); Dd.getParameters()[0].copy_(torch.tensor(st.pD))
Gd.device_net.params_changed(); Dd.device_net.params_changed()
tr = adversarial.Trainer(ctx, Gd, Dd, dict(batchSize=B, noiseDim=100))
real = rng.uniform(0, 1, (B // 2, C, 32, 32)).astype(np.float32)
nz = rng.uniform(-1, 1, (B // 2, 100)).astype(np.float32)
masks = [(rng.random((B, c)) < 0.8).astype(np.float32) for c in (64, 128, 256, 512)] + \
        [(rng.random((B, 512)) < 0.5).astype(np.float32) for _ in range(2)]
dm = [torch.tensor(m.reshape(-1), device=ctx.device) for m in masks]
def report(name, got_g, ref_g, p_dev, p_ref, net):
    d = np.abs(p_dev - p_ref); idx = np.argsort(-d)[:8]
    print("%s: params max diff %.3e; grad max diff %.3e (grad scale %.3e); #entries > 1e-5: %d" %
          (name, d.max(), np.abs(got_g - ref_g).max(), np.abs(ref_g).max(), int((d > 1e-5).sum())))
    offs = [net.param_offset(i) for i in range(net.num_layers)] if hasattr(net, "num_layers") else None
    for i in idx:
        print("   idx %8d  dp %.3e  g_dev % .4e  g_ref % .4e" % (i, d[i], got_g[i], ref_g[i]))
ref = O.step_D(st, real, nz, masks)
got = tr.step_D(torch.tensor(real, device=ctx.device).permute(0, 2, 3, 1).contiguous(), torch.tensor(nz, device=ctx.device), dm, keep_grad=True)
report("D-step D", got["grad"].cpu().numpy(), ref["grad"], Dd.getParameters()[0].cpu().numpy(), st.pD, Dd.device_net)
nz2 = rng.uniform(-1, 1, (B, 100)).astype(np.float32)
ref = O.step_G(st, nz2, masks)
got = tr.step_G(torch.tensor(nz2, device=ctx.device), dm, keep_grad=True)
print("samples max diff %.3e" % np.abs(got["samples"].permute(0, 3, 1, 2).cpu().numpy() - ref["samples"]).max())
report("G-step G", got["grad"].cpu().numpy(), ref["grad"], Gd.getParameters()[0].cpu().numpy(), st.pG, Gd.device_net)
