"""A/B correctness check of an experimental igemm variant selected by env: run once per setting, compares against the
tensor file left by the previous run."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from face_generator_amd import ops
from face_generator_amd.runtime import get_context
ctx = get_context(0); d = ctx.device
g = torch.Generator(device='cpu').manual_seed(0)
outs = []
for (B, H, W, Cin, Cout, k, up) in [(128, 16, 16, 256, 128, 5, 1), (128, 8, 8, 128, 256, 5, 1), (128, 16, 16, 64, 128, 3, 0)]:
    f = 2 if up else 1
    x = torch.randn(B, H, W, Cin, generator=g).to(d)
    w = (torch.randn(Cout, Cin, k, k, generator=g) * 0.05).to(d)
    b = torch.randn(Cout, generator=g).to(d)
    gy = torch.randn(B, H * f, W * f, Cout, generator=g).to(d)
    outs.append(ops.conv2d_forward(x, w, b, upsample2x=bool(up)).cpu())
    outs.append(ops.conv2d_backward_data(gy, w, (H, W), upsample2x=bool(up)).cpu())
path = "/tmp/ws_check.pt"
if os.path.exists(path):
    ref = torch.load(path)
    for i, (a, r) in enumerate(zip(outs, ref)):
        print("out %d max|diff| %.3e  (scale %.3e)" % (i, (a - r).abs().max().item(), r.abs().max().item()))
else:
    torch.save(outs, path); print("saved reference")
