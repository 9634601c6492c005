"""Summarise an FG_WS_TRACE file (s_memtime rows of igemm_ws_trace_kernel, one row per block: entry, first barrier, every K-step
barrier, exit; HW_ID of the CU; the launch's wall time).  s_memtime ticks are shader cycles (calibration lines at the top of
the file: one fp32 32x32x2 MFMA = 64 ticks, `s_nop 15` = 16), so
  * inside a block: cycles per K-step against the MFMA cycles of the step, prologue and epilogue shares;
  * per CU (blocks chained by HW_ID): gaps between consecutive blocks, the CU's span, the share of the span its MFMA pipe is busy;
  * span / wall time = the clock the chip granted this launch.
usage: ws_trace_report.py <trace file(.gz)>"""
import sys, gzip, re
import numpy as np
op = gzip.open if sys.argv[1].endswith(".gz") else open
launch, rows = None, []
def report():
    if not rows:
        return
    kt = rows[0][2]
    n = min(kt, 120)                         # recorded K-step slots (KT > 120: steps 0..118 and the last one)
    R = [r for r in rows if r[2] == kt and len(r) >= n + 7]
    if not R:
        return
    bn128 = 'BN=128' in launch
    hw = np.array([r[3] for r in R]); xcc = np.array([r[1] for r in R]); T = np.array([r[4:4 + n + 3] for r in R], dtype=np.float64)
    first, kloop, epi, total = T[:, 1] - T[:, 0], T[:, n + 1] - T[:, 1], T[:, n + 2] - T[:, n + 1], T[:, n + 2] - T[:, 0]
    steps = np.diff(T[:, 1:(n + 2 if kt <= 120 else n + 1)], axis=1)
    mfma_step = 64 * (64 if bn128 else 32)   # MFMA cycles of one K-step of one wave (BN=64: two blocks share the SIMD)
    print(launch.strip())
    print("  blocks %d, K-steps per tile %d, MFMA cycles per K-step of a wave %d" % (len(T), kt, mfma_step))
    print("  per block (cycles): entry->first barrier median %.0f p90 %.0f | K loop median %.0f = %.0f per step (recorded steps: median %.0f, p90 %.0f)"
          " | epilogue median %.0f p90 %.0f | whole block median %.0f (K loop %.1f %%)"
          % (np.median(first), np.percentile(first, 90), np.median(kloop), np.median(kloop) / kt, np.median(steps), np.percentile(steps, 90),
             np.median(epi), np.percentile(epi, 90), np.median(total), 100 * np.median(kloop) / np.median(total)))
    cu = ((hw >> 8) & 0xf) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5) | (xcc.astype(np.int64) << 8)     # cu_id | sh | se | xcc
    gaps, spans, nb = [], [], []
    for c in np.unique(cu):
        t = T[cu == c]; t = t[np.argsort(t[:, 0])]
        ends = []                            # co-resident blocks interleave: place each block in the first slot that is free
        for row in t:
            for k, e in enumerate(ends):
                if row[0] >= e:
                    gaps.append(row[0] - e); ends[k] = row[n + 2]; break
            else:
                ends.append(row[n + 2])
        spans.append(t[:, n + 2].max() - t[:, 0].min()); nb.append(len(t))
    gaps = np.array(gaps if gaps else [0.0]); spans = np.array(spans)
    busy = np.array(nb) * kt * mfma_step / spans
    print("  per CU: %d CUs, %s blocks each; span (first entry -> last exit) median %.0f cycles (min %.0f max %.0f); gap between consecutive"
          " blocks of a slot median %.0f p90 %.0f (%.1f %% of the span in total)"
          % (len(spans), sorted(set(nb)), np.median(spans), spans.min(), spans.max(), np.median(gaps), np.percentile(gaps, 90),
             100 * gaps.sum() / spans.sum()))
    print("  MFMA-pipe cycles / span per CU: median %.3f" % np.median(busy))
    m = re.search(r"wall_us=([0-9.]+)", launch)
    if m:
        w = float(m.group(1)); ghz = np.median(spans) / w / 1e3
        print("  wall %.1f us -> granted clock %.3f GHz (nominal 2.4): fraction of the 157.3 TFLOP/s peak = %.3f busy x %.3f clock = %.3f"
              % (w, ghz, np.median(busy), ghz / 2.4, np.median(busy) * ghz / 2.4))
for line in op(sys.argv[1], "rt"):
    if line.startswith("# calib"):
        print(line.strip())
    elif line.startswith("#"):
        report(); rows = []; launch = line
    else:
        rows.append([int(v) for v in line.split()])
report()
