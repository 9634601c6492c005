"""FG_WS_TRACE file -> per-CU chains of blocks: gaps between consecutive blocks of one CU, busy cycles per CU.
usage: ws_trace_gaps.py <trace file(.gz)> [wall_us per launch ...]"""
import sys, gzip
import numpy as np
op = gzip.open if sys.argv[1].endswith(".gz") else open
walls = [float(v) for v in sys.argv[2:]]
launch, rows, li = None, [], 0
def rep():
    global li
    if not rows: return
    kt = rows[0][2]
    R = [r for r in rows if r[2] == kt and len(r) >= kt + 7]
    if not R: return
    hw = np.array([r[3] for r in R]); T = np.array([r[4:4 + kt + 3] for r in R], dtype=np.float64); xcc = np.array([r[1] for r in R])
    cu = ((hw >> 8) & 0xf) | (((hw >> 12) & 1) << 4) | (((hw >> 13) & 7) << 5) | (xcc.astype(np.int64) << 8)     # cu_id | sh | se | xcc
    gaps, busy, spans, nb = [], [], [], []
    for c in np.unique(cu):
        t = T[cu == c]; t = t[np.argsort(t[:, 0])]
        # two co-resident blocks interleave: walk greedily over "slots"
        ends = []
        for row in t:
            placed = False
            for k, e in enumerate(ends):
                if row[0] >= e:
                    gaps.append(row[0] - e); ends[k] = row[kt + 2]; placed = True; break
            if not placed: ends.append(row[kt + 2])
        spans.append(t[:, kt + 2].max() - t[:, 0].min()); busy.append((t[:, kt + 2] - t[:, 0]).sum()); nb.append(len(t))
    gaps = np.array(gaps); spans = np.array(spans)
    print(launch.strip()[:70])
    print("  CUs seen %d, blocks per CU %s, co-resident slots inferred %d" % (len(spans), sorted(set(nb)), 1 if 'BN=128' in launch else 2))
    print("  span per CU (first entry -> last exit), cycles: median %.0f  min %.0f  max %.0f" % (np.median(spans), spans.min(), spans.max()))
    print("  gap between consecutive blocks of a slot, cycles: n %d  median %.0f  p90 %.0f  max %.0f  (sum per CU median %.0f)"
          % (len(gaps), np.median(gaps), np.percentile(gaps, 90), gaps.max(), gaps.sum() / len(spans)))
    import re
    m = re.search(r"wall_us=([0-9.]+)", launch)
    if m:
        w = float(m.group(1)); kt_steps = kt
        print("  wall %.1f us -> counter rate implied by the median span: %.3f GHz (2.4 = nominal)" % (w, np.median(spans) / w / 1e3))
    li += 1
for line in op(sys.argv[1], "rt"):
    if line.startswith("# calib"):
        print(line.strip())
    elif line.startswith("#"):
        rep(); rows = []; launch = line
    else:
        rows.append([int(v) for v in line.split()])
rep()
