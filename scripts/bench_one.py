"""Run one conv shape repeatedly (for rocprofv3 --pmc).
usage: bench_one.py <fwd|dgrad|wgrad> [reps] [math] [B H W Cin Cout k up]      (default shape: G's dominant layer, g9)"""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from face_generator_amd import ops
from face_generator_amd.runtime import get_context
ctx = get_context(0); d = ctx.device
which = sys.argv[1]; reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
if len(sys.argv) > 3 and int(sys.argv[3]): ctx.set_math(int(sys.argv[3]))
B, H, W, Cin, Cout, k, up = 128, 16, 16, 256, 128, 5, 1
if len(sys.argv) > 10:
    B, H, W, Cin, Cout, k, up = (int(v) for v in sys.argv[4:11])
f = 2 if up else 1
g = torch.Generator().manual_seed(0)
x = torch.randn(B, H, W, Cin, generator=g).to(d); w = (torch.randn(Cout, Cin, k, k, generator=g) * 0.05).to(d)
b = torch.randn(Cout, generator=g).to(d); gy = torch.randn(B, f * H, f * W, Cout, generator=g).to(d)
for _ in range(reps):
    if which == "fwd": ops.conv2d_forward(x, w, b, upsample2x=bool(up))
    elif which == "dgrad": ops.conv2d_backward_data(gy, w, (H, W), upsample2x=bool(up))
    else: ops.conv2d_backward_weight(x, gy, k, upsample2x=bool(up))
torch.cuda.synchronize()
