#!/bin/bash
# Bounded GPU call: the whole -m gpu suite, bench lines of both workloads, per-iteration kernel table.
set -u
OUT=gpurun_out
TAG=${1:-chk}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $OUT
T0=$(date +%s)
timeout 900 python -m pytest tests -m gpu -x -q --durations=5 > $OUT/${TAG}_tests.log 2>&1
echo "tests rc=$? t=$(( $(date +%s) - T0 ))" | tee $OUT/${TAG}_summary.txt
for i in 1 2 3; do
timeout 200 python bench.py --no-cpu-baseline --no-alt-math --no-roofline > $OUT/${TAG}_bench_cfg2_$i.json 2>/dev/null
done
timeout 200 python bench.py --workload c2f --steps 10 --warmup 3 --no-cpu-baseline --no-alt-math --no-roofline > $OUT/${TAG}_bench_c2f.json 2>/dev/null
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/p1 -o run -- python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-alt-math --no-roofline > /dev/null 2>&1
python scripts/rocpd_stats.py $OUT/p1/run_results.db auto > $OUT/${TAG}_cfg2_kernel_stats.md
rm -rf $OUT/p1
echo "all done t=$(( $(date +%s) - T0 ))" | tee -a $OUT/${TAG}_summary.txt
tail -9 $OUT/${TAG}_tests.log
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/${TAG}_bench*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d["value"],1), round(d["ms_per_step"],4))
    except Exception as e: print(f, "ERR", e)
PY
tail -2 $OUT/${TAG}_cfg2_kernel_stats.md
