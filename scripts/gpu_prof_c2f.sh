#!/bin/bash
set -u
OUT=gpurun_out; TAG=${1:-r3aa}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p $OUT
rm -rf $OUT/${TAG}_prof
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/${TAG}_prof -o c2f -- python bench.py --workload c2f --steps 10 --warmup 2 --no-cpu-baseline --no-alt-math --no-roofline > $OUT/${TAG}_prof_bench.json 2>$OUT/${TAG}_prof.err
f=$(find $OUT/${TAG}_prof -name "*kernel_stats.csv" | head -1)
python - "$f" <<'P'
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
for r in rows:
    n = r["Name"]
    if any(k in n for k in ("pack_jobs", "adam", "wgrad_finish", "multi_final")):
        print("%-60s calls %5s avg %9.1f us total %9.3f ms" % (n[:60], r["Calls"], float(r["AverageNs"]) / 1e3, float(r["TotalDurationNs"]) / 1e6))
P
rm -rf $OUT/${TAG}_prof
