import sys, gzip, numpy as np
op = gzip.open if sys.argv[1].endswith(".gz") else open
rows, launch = [], None
def report():
    if not rows: return
    R = np.array([r[4:4 + 126] for r in rows], dtype=np.float64)
    kt = rows[0][2]; n = min(kt, 14)
    print(launch.strip()[:100])
    print("  entry->barrier", np.median(R[:,1]-R[:,0]))
    for ci in range(0, n - 1):
        t = R[:, 8 + ci * 8: 8 + ci * 8 + 8]; nxt = R[:, 8 + (ci + 1) * 8]
        seg = np.concatenate([np.diff(t, axis=1), (nxt - t[:, 7])[:, None]], axis=1)
        print("  chunk %2d: %s   sum %.0f" % (ci, " ".join("%5.0f" % v for v in np.median(seg, axis=0)), np.median(seg.sum(axis=1))))
for line in op(sys.argv[1], "rt"):
    if line.startswith("# calib"): continue
    if line.startswith("#"): report(); rows = []; launch = line
    else: rows.append([int(v) for v in line.split()])
report()
