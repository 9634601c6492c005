"""Upper bound for the horizontal fusion of dgrad(d_k) and wgrad(d_k): both depend only on dY, today they are
two single-round launches on one stream.  A fused grid can at best give what the hardware gives two INDEPENDENT kernels that are
resident at the same time -- so measure exactly that, without any event in the timed path: stream A runs `reps` data gradients of
one layer back to back, stream B `reps` weight gradients of the same layer back to back, (a) one stream after the other, (b) both at
once.  Wall time by host clock around a full device sync, reps large enough (about 10 ms per leg) that launch latency and the clock
ramp are amortised.  Two fg_ctx objects (one per stream) in one process.

usage: python scripts/bench_hfuse.py [reps]"""
import ctypes, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from face_generator_amd.runtime import Context

reps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
cA, cB = Context(0), Context(0)
sA, sB = torch.cuda.Stream(), torch.cuda.Stream()
cA.bind_stream(sA); cB.bind_stream(sB)
d = cA.device
SHAPES = [("d5  3x3 64->128 @16", 128, 16, 16, 64, 128, 3), ("d9  3x3 128->256 @8", 128, 8, 8, 128, 256, 3),
          ("d13 3x3 256->512 @4", 128, 4, 4, 256, 512, 3)]
g = torch.Generator(device="cpu").manual_seed(0)
lib = cA.lib


def wall(fn):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) * 1e3


tot = dict(serial=0.0, both=0.0)
for (name, B, H, W, Cin, Cout, k) in SHAPES:
    x = torch.randn(B, H, W, Cin, generator=g).to(d)
    w = (torch.randn(Cout, Cin, k, k, generator=g) * 0.05).to(d)
    gy = torch.randn(B, H, W, Cout, generator=g).to(d)
    gx, gw, gb = torch.empty_like(x), torch.zeros_like(w), torch.zeros(Cout, device=d)
    nb = lib.fg_conv2d_workspace_bytes(B, H, W, Cin, Cout, k, 0)
    wsA, wsB = torch.empty((nb + 3) // 4, device=d), torch.empty((nb + 3) // 4, device=d)
    pad = (k - 1) // 2

    def dgrads():
        for _ in range(reps):
            cA.check(lib.fg_conv2d_backward_data(cA.h, gy.data_ptr(), w.data_ptr(), gx.data_ptr(), B, H, W, Cin, Cout, k, pad, 0,
                                                 wsA.data_ptr(), wsA.numel() * 4))

    def wgrads():
        for _ in range(reps):
            cB.check(lib.fg_conv2d_backward_weight(cB.h, x.data_ptr(), gy.data_ptr(), gw.data_ptr(), gb.data_ptr(), ctypes.c_float(0.0),
                                                   B, H, W, Cin, Cout, k, pad, 0, wsB.data_ptr(), wsB.numel() * 4))

    for _ in range(2):           # warm-up (clock, packs, attribute calls)
        wall(lambda: (dgrads(), wgrads()))
    res = []
    for rep in range(3):
        tA = wall(dgrads); tB = wall(wgrads); tAB = wall(lambda: (dgrads(), wgrads()))
        res.append((tA, tB, tAB))
    tA, tB, tAB = [min(r[i] for r in res) for i in range(3)]
    tot["serial"] += tA + tB; tot["both"] += tAB
    print("%-22s dgrad x%d %7.2f ms (%5.1f us each)  wgrad %7.2f ms (%5.1f us each)  one after the other %7.2f ms  both streams at once %7.2f ms  "
          "-> %+5.1f %%" % (name, reps, tA, 1e3 * tA / reps, tB, 1e3 * tB / reps, tA + tB, tAB, 100.0 * (tAB - tA - tB) / (tA + tB)))
print("all three layers: %.2f ms one after the other, %.2f ms co-resident: %+.1f %%  (per training iteration these six launches are %.0f us of ~4 250)"
      % (tot["serial"], tot["both"], 100.0 * (tot["both"] - tot["serial"]) / tot["serial"], 1e3 * tot["serial"] / reps))
