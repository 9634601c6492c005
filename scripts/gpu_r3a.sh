#!/bin/bash
# round 3, first GPU call: the new parity tests (verbose), the whole suite, the default bench line, kernel traces of both workloads
set -u
OUT=gpurun_out
TAG=${1:-r3a}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $OUT
T0=$(date +%s)
timeout 1500 python -m pytest tests/test_gpu_baseline_sizes.py tests/test_gpu_c_host.py -m gpu -q -s --durations=10 > $OUT/${TAG}_newtests.log 2>&1
echo "new tests rc=$? t=$(( $(date +%s) - T0 ))" | tee $OUT/${TAG}_summary.txt
timeout 900 python -m pytest tests -m gpu -x -q --durations=8 --deselect tests/test_gpu_baseline_sizes.py --deselect tests/test_gpu_c_host.py > $OUT/${TAG}_tests.log 2>&1
echo "other tests rc=$? t=$(( $(date +%s) - T0 ))" | tee -a $OUT/${TAG}_summary.txt
timeout 600 python bench.py > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
echo "bench rc=$? t=$(( $(date +%s) - T0 ))" | tee -a $OUT/${TAG}_summary.txt
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/p1 -o run -- python bench.py --workload cfg2 --steps 50 --warmup 10 --no-cpu-baseline --no-alt-math --no-roofline > /dev/null 2>&1
python scripts/rocpd_stats.py $OUT/p1/run_results.db auto > $OUT/${TAG}_cfg2_kernel_stats.md
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/p2 -o run -- python bench.py --workload c2f --steps 6 --warmup 2 --no-cpu-baseline --no-alt-math --no-roofline > /dev/null 2>&1
python scripts/rocpd_stats.py $OUT/p2/run_results.db auto > $OUT/${TAG}_c2f_kernel_stats.md
rm -rf $OUT/p1 $OUT/p2
echo "all done t=$(( $(date +%s) - T0 ))" | tee -a $OUT/${TAG}_summary.txt
tail -5 $OUT/${TAG}_newtests.log
tail -5 $OUT/${TAG}_tests.log
python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/${TAG}_bench.json").read().strip().splitlines()[-1])
    print("cfg2", round(d["value"],1), round(d["ms_per_step"],4), d["roofline"]["frac"], d["roofline"].get("traffic"), d["roofline"].get("traffic_freshness"))
    c=d.get("c2f",{}); print("c2f", c.get("value"), c.get("ms_per_step"), c.get("error"), c.get("roofline",{}).get("frac"))
except Exception as e: print("ERR", e)
PY
