"""A/B check of the contraction math modes on the hot-path layer shapes: mode 0 (fp32 MFMA) vs mode 6 (bf16x6) against an
fp64 torch reference (first images for forward / data-gradient, full batch for the weight gradient), plus timings."""
import ctypes, os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from face_generator_amd import ops
from face_generator_amd.runtime import get_context
ctx = get_context(0); d = ctx.device
g = torch.Generator(device='cpu').manual_seed(0)
import torch.nn.functional as F
torch.set_num_threads(64)
SHAPES = [(128, 16, 16, 64, 128, 3, 0), (128, 32, 32, 64, 64, 3, 0), (128, 16, 16, 256, 128, 5, 1), (128, 8, 8, 128, 256, 5, 1), (128, 16, 16, 64, 128, 3, 0), (128, 8, 8, 128, 256, 3, 0),
          (128, 4, 4, 256, 512, 3, 0)]
if len(sys.argv) > 1: SHAPES = SHAPES[:int(sys.argv[1])]
for (B, H, W, Cin, Cout, k, up) in SHAPES:
    f = 2 if up else 1
    x = torch.randn(B, H, W, Cin, generator=g); w = torch.randn(Cout, Cin, k, k, generator=g) * 0.05
    b = torch.randn(Cout, generator=g); gy = torch.randn(B, H * f, W * f, Cout, generator=g)
    nb = 4      # fp64 reference on the first nb images
    xr = x[:nb].permute(0, 3, 1, 2).double()
    if up: xr = F.interpolate(xr, scale_factor=2, mode="nearest")
    yref = F.conv2d(xr, w.double(), b.double(), padding=k // 2).permute(0, 2, 3, 1)
    gyr = gy[:nb].permute(0, 3, 1, 2).double()
    gxr = F.conv_transpose2d(gyr, w.double(), padding=k // 2)
    if up: gxr = gxr.reshape(nb, Cin, H, 2, W, 2).sum((3, 5))
    gxr = gxr.permute(0, 2, 3, 1)
    xa = x.permute(0, 3, 1, 2).double()
    if up: xa = F.interpolate(xa, scale_factor=2, mode="nearest")
    gwr = torch.nn.grad.conv2d_weight(xa, (Cout, Cin, k, k), gy.permute(0, 3, 1, 2).double(), padding=k // 2)
    xd, wd, bd, gyd = x.to(d), w.to(d), b.to(d), gy.to(d)
    for mode in (0, 6):
        ctx.set_math(mode)
        y = ops.conv2d_forward(xd, wd, bd, upsample2x=bool(up)); gx = ops.conv2d_backward_data(gyd, wd, (H, W), upsample2x=bool(up))
        gw = ops.conv2d_backward_weight(xd, gyd, k, upsample2x=bool(up))
        gw = gw[0] if isinstance(gw, (tuple, list)) else gw
        ey = (y[:nb].cpu().double() - yref); egx = (gx[:nb].cpu().double() - gxr); egw = gw.cpu().double() - gwr
        ctx.check(ctx.lib.fg_prof_enable(ctx.h, 1))
        for _ in range(10):
            ops.conv2d_forward(xd, wd, bd, upsample2x=bool(up)); ops.conv2d_backward_data(gyd, wd, (H, W), upsample2x=bool(up))
            ops.conv2d_backward_weight(xd, gyd, k, upsample2x=bool(up))
        buf = ctypes.create_string_buffer(1 << 16)
        ctx.check(ctx.lib.fg_prof_report(ctx.h, buf, len(buf), 1)); ctx.check(ctx.lib.fg_prof_enable(ctx.h, 0))
        print("shape %s mode %d: rel rms err  fwd %.3e  dgrad %.3e  wgrad %.3e   (max/rms: %.2e %.2e %.2e)" %
              ((B, H, W, Cin, Cout, k, up), mode, ey.pow(2).mean().sqrt() / yref.pow(2).mean().sqrt(),
               egx.pow(2).mean().sqrt() / gxr.pow(2).mean().sqrt(), egw.pow(2).mean().sqrt() / gwr.pow(2).mean().sqrt(),
               ey.abs().max() / yref.pow(2).mean().sqrt(), egx.abs().max() / gxr.pow(2).mean().sqrt(), egw.abs().max() / gwr.pow(2).mean().sqrt()))
        for line in buf.value.decode().strip().splitlines():
            n, calls, ms = line.split()[:3]
            print("      %-48s %8.1f us x %d" % (n, 1000 * float(ms) / int(calls), int(calls) // 10))
ctx.set_math(0)
