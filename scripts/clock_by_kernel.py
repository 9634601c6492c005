"""The clock each contraction kernel is granted: loop ONE op of ONE layer shape for ~40 ms beside the one-wave clock probe
(fg_prof_clock_start / _read) and report the HIP-event rate of the kernel, the granted clock, and the rate re-priced at 2.4 GHz.
usage: clock_by_kernel.py [cfg2|c2f]"""
import sys, ctypes, torch
import os; sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from face_generator_amd import ops
from face_generator_amd.runtime import get_context
ctx = get_context(0); d = ctx.device
C2F = [("G2  3x3 64->64 @64", 128, 64, 64, 64, 64, 3, 0), ("G3  5x5 64->128 @64", 128, 64, 64, 64, 128, 5, 0),
       ("G4  5x5 128->256 @64", 128, 64, 64, 128, 256, 5, 0), ("D3  3x3 64->128 @32", 128, 32, 32, 64, 128, 3, 0),
       ("D4  3x3 128->256 @32", 128, 32, 32, 128, 256, 3, 0)]
CFG2 = [("g9  up5x5 256->128 @16->32", 128, 16, 16, 256, 128, 5, 1), ("g5  up5x5 128->256 @8->16", 128, 8, 8, 128, 256, 5, 1),
        ("d5  3x3 64->128 @16", 128, 16, 16, 64, 128, 3, 0), ("d9  3x3 128->256 @8", 128, 8, 8, 128, 256, 3, 0)]
SHAPES = C2F if (len(sys.argv) > 1 and sys.argv[1] == "c2f") else CFG2
g = torch.Generator(device='cpu').manual_seed(0)
def prof(fn, reps):
    ctx.check(ctx.lib.fg_prof_enable(ctx.h, 1))
    for _ in range(reps): fn()
    buf = ctypes.create_string_buffer(1 << 16)
    ctx.check(ctx.lib.fg_prof_report(ctx.h, buf, len(buf), 1))
    ctx.check(ctx.lib.fg_prof_enable(ctx.h, 0))
    best = None
    for line in buf.value.decode().strip().splitlines():
        n, calls, ms, alg, exe, _ = line.split()
        if float(exe) > 0 and (best is None or float(ms) > best[2]): best = (n, int(calls), float(ms), float(exe))
    return best
print("%-28s %-8s %-34s %9s %8s %7s %12s" % ("layer", "op", "kernel", "us", "TFLOP/s", "GHz", "TF @2.4GHz"))
for (name, B, H, W, Cin, Cout, k, up) in SHAPES:
    f = 2 if up else 1
    x = torch.randn(B, H, W, Cin, generator=g).to(d); w = (torch.randn(Cout, Cin, k, k, generator=g) * 0.05).to(d)
    b = torch.randn(Cout, generator=g).to(d); gy = torch.randn(B, H * f, W * f, Cout, generator=g).to(d)
    fns = {"fwd": lambda: ops.conv2d_forward(x, w, b, upsample2x=bool(up)),
           "dgrad": lambda: ops.conv2d_backward_data(gy, w, (H, W), upsample2x=bool(up)),
           "wgrad": lambda: ops.conv2d_backward_weight(x, gy, k, upsample2x=bool(up))}
    for op, fn in fns.items():
        for _ in range(3): fn()
        torch.cuda.synchronize()
        n0, c0, ms0, _ = prof(fn, 4)
        reps = max(8, int(60.0 / (ms0 / c0 * (c0 / 4))))          # ~60 ms of back-to-back launches
        for _ in range(reps // 3): fn()                            # the clock settles over the first ~10 ms of a body
        ctx.check(ctx.lib.fg_prof_clock_start(ctx.h, ctypes.c_double(40.0)))
        n, calls, ms, exe = prof(fn, reps)
        ghz, cov = ctypes.c_double(0), ctypes.c_double(0)
        ctx.check(ctx.lib.fg_prof_clock_read(ctx.h, ctypes.byref(ghz), ctypes.byref(cov)))
        tf = exe / ms / 1e9
        print("%-28s %-8s %-34s %9.1f %8.1f %7.3f %12.1f" % (name, op, n.split("/")[0][:34], 1000 * ms / calls, tf, ghz.value, tf * 2.4 / max(ghz.value, 1e-9)))
