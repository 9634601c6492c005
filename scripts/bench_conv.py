"""Micro-benchmark of the contraction kernels on the hot-path layer shapes (HIP-event timing through the library's
profiler).  usage: python scripts/bench_conv.py [iters]"""
import ctypes, sys
import numpy as np, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from face_generator_amd import ops
from face_generator_amd.runtime import get_context

ctx = get_context(0); d = ctx.device
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 10
C2F = [  # the coarse-to-fine nets at 64x64 (models_c2f.lua:113-145, 237-278), B = 128
    ("G2  3x3 64->64 @64", 128, 64, 64, 64, 64, 3, 0),
    ("G3  5x5 64->128 @64", 128, 64, 64, 64, 128, 5, 0),
    ("G4  5x5 128->256 @64", 128, 64, 64, 128, 256, 5, 0),
    ("D2  3x3 64->64 @64", 128, 64, 64, 64, 64, 3, 0),
    ("D3  3x3 64->128 @32", 128, 32, 32, 64, 128, 3, 0),
    ("D4  3x3 128->256 @32", 128, 32, 32, 128, 256, 3, 0),
]
SHAPES = [  # name, B, H, W, Cin, Cout, k, up
    ("g9  up5x5 256->128 @16->32", 128, 16, 16, 256, 128, 5, 1),
    ("g5  up5x5 128->256 @8->16", 128, 8, 8, 128, 256, 5, 1),
    ("d5  3x3 64->128 @16", 128, 16, 16, 64, 128, 3, 0),
    ("d9  3x3 128->256 @8", 128, 8, 8, 128, 256, 3, 0),
    ("d13 3x3 256->512 @4", 128, 4, 4, 256, 512, 3, 0),
]
if len(sys.argv) > 2 and sys.argv[2] == "c2f":
    SHAPES = C2F
g = torch.Generator(device='cpu').manual_seed(0)
for (name, B, H, W, Cin, Cout, k, up) in SHAPES:
    f = 2 if up else 1
    x = torch.randn(B, H, W, Cin, generator=g).to(d)
    w = (torch.randn(Cout, Cin, k, k, generator=g) * 0.05).to(d)
    b = torch.randn(Cout, generator=g).to(d)
    gy = torch.randn(B, H * f, W * f, Cout, generator=g).to(d)
    for _ in range(2):
        ops.conv2d_forward(x, w, b, upsample2x=bool(up)); ops.conv2d_backward_data(gy, w, (H, W), upsample2x=bool(up))
        ops.conv2d_backward_weight(x, gy, k, upsample2x=bool(up))
    ctx.check(ctx.lib.fg_prof_enable(ctx.h, 1))
    for _ in range(iters):
        ops.conv2d_forward(x, w, b, upsample2x=bool(up)); ops.conv2d_backward_data(gy, w, (H, W), upsample2x=bool(up))
        ops.conv2d_backward_weight(x, gy, k, upsample2x=bool(up))
    buf = ctypes.create_string_buffer(1 << 16)
    ctx.check(ctx.lib.fg_prof_report(ctx.h, buf, len(buf), 1))
    ctx.check(ctx.lib.fg_prof_enable(ctx.h, 0))
    print(name)
    for line in buf.value.decode().strip().splitlines():
        n, calls, ms, alg, exe, _ = line.split()
        calls = int(calls); ms = float(ms); alg = float(alg); exe = float(exe)
        print("   %-40s %7.1f us  exec %6.1f TF  alg %6.1f TF" % (n, 1000 * ms / calls, exe / ms / 1e9, alg / ms / 1e9))
