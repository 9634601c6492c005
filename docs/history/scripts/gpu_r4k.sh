#!/bin/bash
# round 4, call k: Linear(K -> 1) + Sigmoid + BCE in one launch (FG_GEMV_BCE=0 switches back); the whole suite under FG_MATH=6
set -u
OUT=gpurun_out; TAG=${1:-r04k}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_step_abi.py tests/test_gpu_c_host.py tests/test_gpu_fusion.py tests/test_golden.py tests/test_gpu_train_epoch.py tests/test_gpu_dist.py -m gpu -q > $OUT/${TAG}_tests.log 2>&1
echo "tests rc=$?" | tee $OUT/${TAG}_summary.txt; tail -3 $OUT/${TAG}_tests.log
for rep in 1 2; do
 for gb in 1 0; do
    FG_GEMV_BCE=$gb timeout 300 python bench.py --workload cfg2 --steps 50 --warmup 10 --no-cpu-baseline --no-alt-math --no-live-traffic --no-clock-probe 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('gemv_bce=$gb cfg2 %.1f img/s %.4f ms' % (d['value'], d['ms_per_step']))"
 done
done 2>&1 | tee $OUT/${TAG}_bench.txt
FG_MATH=6 timeout 1500 python -m pytest tests -m gpu -q 2>&1 | grep -v "^  [0-9 ]" | tail -25 > $OUT/${TAG}_math6_tests.log
echo "math6 rc=$?" | tee -a $OUT/${TAG}_summary.txt; tail -6 $OUT/${TAG}_math6_tests.log
