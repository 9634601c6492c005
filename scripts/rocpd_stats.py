"""Summarise a rocprofv3 rocpd SQLite database (kernel trace) per kernel name: calls, total/avg ms, % of GPU time.
usage: python scripts/rocpd_stats.py <results.db> [iters | auto]   -> markdown table on stdout
(auto: iterations = rng_multi_kernel launches / 2 -- every closure of the training loop starts with one Philox launch -- and every
dispatch BEFORE the first such launch is dropped and listed apart: net construction (parameter uploads: `__amd_rocclr_copyBuffer`
blits of up to 134 MB, fills, the first weight re-pack) is not per-iteration cost)"""
import re
import sqlite3
import sys


def short(name):
    name = re.sub(r"\(.*$", "", name)
    name = name.replace("void ", "")
    return name[:90]


def main():
    db = sqlite3.connect(sys.argv[1])
    iters = sys.argv[2] if len(sys.argv) > 2 else None      # a number, or "auto": one Philox launch per closure = 2 / iteration
    cur = db.cursor()
    cols = [r[1] for r in cur.execute("pragma table_info(kernels)")]
    namecol = "name" if "name" in cols else [c for c in cols if "name" in c][0]
    gridcols = [c for c in cols if c.lower() in ("grid_size_x", "grid_x", "grid_size")][:1]
    rows = sorted(cur.execute("select %s, start, end%s from kernels" % (namecol, (", " + gridcols[0]) if gridcols else "")), key=lambda r: r[1])
    shapes = {}                                   # (kernel, grid) -> [calls, ms]: launches of one kernel with different grids, apart
    if gridcols:
        for r in rows:
            a = shapes.setdefault((short(r[0]), r[3]), [0, 0.0])
            a[0] += 1
            a[1] += (r[2] - r[1]) / 1e6
        rows = [r[:3] for r in rows]
    setup = {}
    if iters == "auto":
        first = next((i for i, r in enumerate(rows) if short(r[0]) == "rng_multi_kernel"), 0)
        for name, s, e in rows[:first]:
            a = setup.setdefault(short(name), [0, 0.0])
            a[0] += 1
            a[1] += (e - s) / 1e6
        rows = rows[first:]
    agg = {}
    for name, s, e in rows:
        a = agg.setdefault(short(name), [0, 0.0])
        a[0] += 1
        a[1] += (e - s) / 1e6
    if iters == "auto":
        iters = agg.get("rng_multi_kernel", [0])[0] / 2.0 or None
    elif iters is not None:
        iters = float(iters)
    total = sum(v[1] for v in agg.values())
    span = (max(r[2] for r in rows) - min(r[1] for r in rows)) / 1e6
    print("| kernel | calls | total ms | avg us | % of kernel time |" + (" ms/iter |" if iters else ""))
    print("|---|---|---|---|---|" + ("---|" if iters else ""))
    for k, (n, ms) in sorted(agg.items(), key=lambda kv: -kv[1][1]):
        print("| %s | %d | %.3f | %.1f | %.1f |" % (k, n, ms, 1000 * ms / n, 100 * ms / total)
              + (" %.3f |" % (ms / iters) if iters else ""))
    print("\ntotal kernel time %.3f ms over %d dispatches; first-to-last span %.3f ms" % (total, len(rows), span))
    if iters:
        print("%.1f iterations: %.3f ms of kernel time and %.1f dispatches per iteration" % (iters, total / iters, len(rows) / iters))
    multi = sorted(set(k for (k, g) in shapes if sum(1 for (k2, g2) in shapes if k2 == k) > 1))
    if multi and "--by-grid" in sys.argv:
        print("\nper launch shape (kernels launched with more than one grid; grid = work-items along x):")
        print("| kernel | grid x | calls | avg us |\n|---|---|---|---|")
        for k in sorted(multi, key=lambda k: -agg.get(k, [0, 0])[1])[:14]:
            for (k2, g), (n, ms) in sorted(shapes.items(), key=lambda kv: -kv[1][1]):
                if k2 == k and n >= 3:
                    print("| %s | %s | %d | %.1f |" % (k, g, n, 1000 * ms / n))
    if setup:
        print("\nbefore the first training closure (net construction; NOT in the table): %d dispatches, %.3f ms -- %s"
              % (sum(v[0] for v in setup.values()), sum(v[1] for v in setup.values()),
                 ", ".join("%s x%d %.3f ms" % (k, n, ms) for k, (n, ms) in sorted(setup.items(), key=lambda kv: -kv[1][1])[:6])))


if __name__ == "__main__":
    main()
