"""Thin 3x3 convolutions of the cfg2 path, repeated (run under rocprofv3 --kernel-trace --stats for per-kernel durations).
usage: bench_thin.py [reps]"""
import sys, os, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from face_generator_amd import ops
from face_generator_amd.runtime import get_context
ctx = get_context(0); d = ctx.device
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
g = torch.Generator().manual_seed(0)
def t(*shape): return torch.randn(*shape, generator=g).to(d)
B = 128
x3 = t(B, 32, 32, 3); w_in64 = t(64, 3, 3, 3) * 0.1; b64 = t(64)              # d1 forward: 3 -> 64
x128 = t(B, 32, 32, 128); w_out = t(3, 128, 3, 3) * 0.1; b3 = t(3)            # g12 forward: 128 -> 3 (+ sigmoid in the net)
g3 = t(B, 32, 32, 3)                                                          # g12 data gradient: 3 -> 128
g64 = t(B, 32, 32, 64)                                                        # d1 data gradient: 64 -> 3
for _ in range(reps):
    ops.conv2d_forward(x3, w_in64, b64)
    ops.conv2d_forward(x128, w_out, b3)
    ops.conv2d_backward_data(g3, w_out, (32, 32))
    ops.conv2d_backward_data(g64, w_in64, (32, 32))
torch.cuda.synchronize()
