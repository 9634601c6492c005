"""What the part's WRITE path sustains (thin_in_mfma_kernel<3,0> writes 33.6 MB in 15.7 us = 2.1 TB/s = 27 % of the
8 TB/s HBM figure -- is that the kernel or the path?).  Pure stores (fg_fill: one float4 store per thread, nothing read), pure loads
(a sum reduction), and a copy (one load + one store per element), each at the size of d1's output (33.6 MB) and at 268 MB, HIP-event
timed over 20 launches after 5 warm-ups.  usage: python scripts/bench_write_bw.py"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from face_generator_amd.runtime import get_context
ctx = get_context(0)


def timed(fn, reps=20):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e3      # us per launch (back-to-back launches: the per-launch floor is amortised)


for mb in (33.6, 268.4):
    n = int(mb * 1e6 / 4) // 1024 * 1024
    a, b = ctx.empty(n), ctx.empty(n)
    t_fill = timed(lambda: ctx.check(ctx.lib.fg_fill(ctx.h, a.data_ptr(), 1.0, n)))
    t_tfill = timed(lambda: a.fill_(2.0))
    t_copy = timed(lambda: ctx.check(ctx.lib.fg_d2d(ctx.h, b.data_ptr(), a.data_ptr(), n * 4)))
    t_tcopy = timed(lambda: b.copy_(a))
    t_sum = timed(lambda: a.sum())
    gb = n * 4 / 1e9
    print("%6.1f MB: fg_fill %6.1f us = %5.2f TB/s written | torch fill_ %6.1f us = %5.2f | fg_d2d %6.1f us = %5.2f TB/s moved (r+w) | torch copy_ %6.1f us = %5.2f | "
          "torch sum %6.1f us = %5.2f TB/s read" % (mb, t_fill, gb / t_fill * 1e3, t_tfill, gb / t_tfill * 1e3, t_copy, 2 * gb / t_copy * 1e3, t_tcopy, 2 * gb / t_tcopy * 1e3, t_sum, gb / t_sum * 1e3))
