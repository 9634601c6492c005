#!/bin/bash
# round 4, call j: thin weight-gradient sums written in the reference layout, one-launch permuted bias gradient -- parity, both lines
set -u
OUT=gpurun_out; TAG=${1:-r04j}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p $OUT
timeout 1200 python -m pytest tests/test_gpu_net.py tests/test_gpu_ops.py tests/test_golden.py tests/test_gpu_c2f.py tests/test_gpu_fusion.py tests/test_gpu_step_abi.py tests/test_gpu_fullsize.py tests/test_gpu_c_host.py "tests/test_gpu_baseline_sizes.py::test_cfg2_full_step_at_batch_128" -m gpu -q > $OUT/${TAG}_tests.log 2>&1
echo "tests rc=$?" | tee $OUT/${TAG}_summary.txt; tail -4 $OUT/${TAG}_tests.log
for rep in 1 2; do
  for wl in cfg2 c2f; do
    timeout 300 python bench.py --workload $wl --steps $([ $wl = c2f ] && echo 10 || echo 50) --warmup $([ $wl = c2f ] && echo 3 || echo 10) --no-cpu-baseline --no-alt-math --no-live-traffic --no-clock-probe > $OUT/${TAG}_b.json 2>/dev/null
    python - $wl $TAG <<'P'
import json,sys
d=json.loads(open("gpurun_out/%s_b.json" % sys.argv[2]).read().strip().splitlines()[-1])
print("%s %.1f img/s %.4f ms exec %.4f" % (sys.argv[1], d["value"], d["ms_per_step"], d["step_roofline"].get("executed_frac",0)))
P
  done
done 2>&1 | tee $OUT/${TAG}_bench.txt
rocprofv3 --kernel-trace --stats -d $OUT/p1 -o run -- python bench.py --workload cfg2 --steps 50 --warmup 10 --no-cpu-baseline --no-alt-math --no-roofline > /dev/null 2>&1
python scripts/rocpd_stats.py $OUT/p1/run_results.db auto > $OUT/${TAG}_bench_kernel_stats.md; rm -rf $OUT/p1
tail -5 $OUT/${TAG}_bench_kernel_stats.md | head -3
