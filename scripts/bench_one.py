"""Run one conv shape repeatedly (for rocprofv3 --pmc).  usage: bench_one.py <fwd|dgrad|wgrad> [reps] [math]"""
import sys, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from face_generator_amd import ops
from face_generator_amd.runtime import get_context
ctx = get_context(0); d = ctx.device
which = sys.argv[1]; reps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
if len(sys.argv) > 3: ctx.set_math(int(sys.argv[3]))
B, H, W, Cin, Cout, k = 128, 16, 16, 256, 128, 5
g = torch.Generator().manual_seed(0)
x = torch.randn(B, H, W, Cin, generator=g).to(d); w = (torch.randn(Cout, Cin, k, k, generator=g) * 0.05).to(d)
b = torch.randn(Cout, generator=g).to(d); gy = torch.randn(B, 2 * H, 2 * W, Cout, generator=g).to(d)
for _ in range(reps):
    if which == "fwd": ops.conv2d_forward(x, w, b, upsample2x=True)
    elif which == "dgrad": ops.conv2d_backward_data(gy, w, (H, W), upsample2x=True)
    else: ops.conv2d_backward_weight(x, gy, k, upsample2x=True)
torch.cuda.synchronize()
