"""Summarise an FG_WS_TRACE file (s_memtime rows of igemm_ws_trace_kernel): where inside a tile does the time go?
usage: ws_trace_report.py <trace file>"""
import sys
import numpy as np
launch = None
rows = []
def report():
    if not rows:
        return
    kt = rows[0][2]
    T = np.array([r[4:4 + kt + 3] for r in rows if r[2] == kt and len(r) >= kt + 7], dtype=np.float64)
    xcc = np.array([r[1] for r in rows if r[2] == kt and len(r) >= kt + 7])
    t_start = T[:, 0].min()
    entry, first, steps, epi = T[:, 0] - t_start, T[:, 1] - T[:, 0], np.diff(T[:, 1:kt + 2], axis=1), T[:, kt + 2] - T[:, kt + 1]
    total = T[:, kt + 2] - T[:, 0]
    print(launch.strip())
    print("  blocks %d, K-steps per tile %d; kernel span %.0f cycles" % (len(T), kt, T[:, kt + 2].max() - t_start))
    print("  per block (cycles): entry->first barrier  median %.0f  p90 %.0f   | K loop  median %.0f (per step median %.0f, mean %.0f, max-step median %.0f)"
          " | epilogue median %.0f p90 %.0f | whole block median %.0f" % (np.median(first), np.percentile(first, 90), np.median(steps.sum(1)),
          np.median(steps), steps.mean(), np.median(steps.max(1)), np.median(epi), np.percentile(epi, 90), np.median(total)))
    q = np.percentile(steps, [10, 50, 90, 99])
    print("  step time distribution: p10 %.0f p50 %.0f p90 %.0f p99 %.0f; mean over step index (first 12): %s ... last 4: %s"
          % (q[0], q[1], q[2], q[3], " ".join("%.0f" % v for v in steps.mean(0)[:12]), " ".join("%.0f" % v for v in steps.mean(0)[-4:])))
    order = np.argsort(entry)
    print("  block entry times (cycles after the first block): p25 %.0f p50 %.0f p75 %.0f max %.0f" % tuple(np.percentile(entry, [25, 50, 75, 100])))
    # gaps: time between a block's end and the entry of the next block that starts after it (same dispatch slot unknown: report global)
    ends = np.sort(T[:, kt + 2] - t_start); starts = np.sort(entry)
    n1 = int((entry < 1000).sum())
    print("  blocks that start within 1000 cycles of the first: %d; ideal MFMA cycles per step (alone on its SIMD): %d" % (n1, 64 * (64 if 'BN=128' in launch else 32)))
for line in open(sys.argv[1]):
    if line.startswith("#"):
        report(); rows = []; launch = line
    else:
        rows.append([int(v) for v in line.split()])
report()
