#!/bin/bash
# round 3, second GPU call: the tests that changed (verbose), then the whole suite, one bench line
set -u
OUT=gpurun_out
TAG=${1:-r3b}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $OUT
T0=$(date +%s)
timeout 1500 python -m pytest tests/test_gpu_baseline_sizes.py tests/test_gpu_conv_upsample.py tests/test_gpu_c_host.py -m gpu -q -s --durations=10 > $OUT/${TAG}_newtests.log 2>&1
echo "new tests rc=$? t=$(( $(date +%s) - T0 ))" | tee $OUT/${TAG}_summary.txt
timeout 900 python -m pytest tests -m gpu -x -q --durations=8 --deselect tests/test_gpu_baseline_sizes.py --deselect tests/test_gpu_c_host.py --deselect tests/test_gpu_conv_upsample.py > $OUT/${TAG}_tests.log 2>&1
echo "other tests rc=$? t=$(( $(date +%s) - T0 ))" | tee -a $OUT/${TAG}_summary.txt
timeout 300 python bench.py --workload cfg2 --no-cpu-baseline --no-alt-math --no-live-traffic > $OUT/${TAG}_bench.json 2> $OUT/${TAG}_bench.err
echo "bench rc=$? t=$(( $(date +%s) - T0 ))" | tee -a $OUT/${TAG}_summary.txt
grep -E "passed|failed" $OUT/${TAG}_newtests.log | tail -3
tail -5 $OUT/${TAG}_tests.log
python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/${TAG}_bench.json").read().strip().splitlines()[-1])
    print("cfg2", round(d["value"],1), round(d["ms_per_step"],4), d["roofline"]["frac"])
except Exception as e: print("ERR", e)
PY
