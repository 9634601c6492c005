"""FG_WINO_TRACE=1 FG_WINO_DBG=100: s_memtime every 8 MFMA slots (one position pair = 8 MFMAs = 512 pipe cycles) of the first 14 K chunks
of every block -> median cycles of each of the 8 slot groups of a steady-state chunk.  usage: wino_trace2_report.py <file(.gz)>"""
import sys, gzip
import numpy as np
op = gzip.open if sys.argv[1].endswith(".gz") else open
rows, launch = [], None
def report():
    if not rows:
        return
    R = np.array([r[4:4 + 126] for r in rows], dtype=np.float64)
    kt = rows[0][2]
    n = min(kt, 14)
    print(launch.strip())
    for ci in range(2, n - 1):                   # steady-state chunks (not the first two, not the tail)
        t = R[:, 8 + ci * 8: 8 + ci * 8 + 8]
        nxt = R[:, 8 + (ci + 1) * 8]
        seg = np.concatenate([np.diff(t, axis=1), (nxt - t[:, 7])[:, None]], axis=1)
        print("  chunk %2d: cycles per 8-MFMA group (ideal 512): %s   sum %.0f" % (ci, " ".join("%5.0f" % v for v in np.median(seg, axis=0)), np.median(seg.sum(axis=1))))
for line in op(sys.argv[1], "rt"):
    if line.startswith("# calib"):
        continue
    if line.startswith("#"):
        report(); rows = []; launch = line
    else:
        rows.append([int(v) for v in line.split()])
report()
