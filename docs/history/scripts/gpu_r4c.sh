#!/bin/bash
# round 4, third call: whole -m gpu suite on the current tree, the write-path microbench, both bench lines twice, kernel tables
set -u
OUT=gpurun_out; TAG=${1:-r04c}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p $OUT
T0=$(date +%s)
timeout 1200 python -m pytest tests -m gpu -q -x --durations=4 > $OUT/${TAG}_gpu_tests.log 2>&1
echo "suite rc=$? t=$(( $(date +%s) - T0 ))" | tee $OUT/${TAG}_summary.txt; tail -8 $OUT/${TAG}_gpu_tests.log
python scripts/bench_write_bw.py 2>/dev/null | tee $OUT/${TAG}_write_bw.txt
for rep in 1 2; do
  for wl in cfg2 c2f; do
    timeout 300 python bench.py --workload $wl --steps $([ $wl = c2f ] && echo 10 || echo 50) --warmup $([ $wl = c2f ] && echo 3 || echo 10) --no-cpu-baseline --no-alt-math --no-live-traffic > $OUT/${TAG}_b.json 2>/dev/null
    python - $wl <<'P'
import json,sys
d=json.loads(open("gpurun_out/r04c_b.json").read().strip().splitlines()[-1])
print("%s %.1f img/s %.4f ms exec %.4f dom %.4f clock %s" % (sys.argv[1], d["value"], d["ms_per_step"], d["step_roofline"].get("executed_frac",0), d["roofline"]["frac"], d["step_roofline"].get("granted_clock_ghz")))
P
  done
done 2>&1 | tee $OUT/${TAG}_bench.txt
rocprofv3 --kernel-trace --stats -d $OUT/p1 -o run -- python bench.py --workload cfg2 --steps 50 --warmup 10 --no-cpu-baseline --no-alt-math --no-roofline > /dev/null 2>&1
python scripts/rocpd_stats.py $OUT/p1/run_results.db auto > $OUT/${TAG}_bench_kernel_stats.md
rocprofv3 --kernel-trace --stats -d $OUT/p2 -o run -- python bench.py --workload c2f --steps 6 --warmup 2 --no-cpu-baseline --no-alt-math --no-roofline > /dev/null 2>&1
python scripts/rocpd_stats.py $OUT/p2/run_results.db auto > $OUT/${TAG}_c2f_kernel_stats.md
rm -rf $OUT/p1 $OUT/p2
tail -4 $OUT/${TAG}_bench_kernel_stats.md; tail -4 $OUT/${TAG}_c2f_kernel_stats.md
