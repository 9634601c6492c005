#!/bin/bash
# Bounded GPU call: whole -m gpu suite, A/B of the in-kernel split-K reduction, bench lines, kernel traces.
set -u
OUT=gpurun_out
TAG=${1:-b}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $OUT
T0=$(date +%s)
timeout 900 python -m pytest tests -m gpu -x -q --durations=8 > $OUT/${TAG}_tests.log 2>&1
echo "tests rc=$? t=$(( $(date +%s) - T0 ))" | tee $OUT/${TAG}_summary.txt
for i in 1 2; do
FG_SPLITK_FUSED=0 timeout 200 python bench.py --no-cpu-baseline --no-alt-math --no-roofline > $OUT/${TAG}_bench_cfg2_sk0_$i.json 2>/dev/null
timeout 200 python bench.py --no-cpu-baseline --no-alt-math --no-roofline > $OUT/${TAG}_bench_cfg2_sk1_$i.json 2>/dev/null
done
timeout 200 python bench.py --no-cpu-baseline --no-alt-math > $OUT/${TAG}_bench_cfg2.json 2> $OUT/${TAG}_bench_cfg2.err
timeout 200 python bench.py --workload c2f --steps 10 --warmup 3 --no-cpu-baseline --no-alt-math > $OUT/${TAG}_bench_c2f.json 2> $OUT/${TAG}_bench_c2f.err
echo "benches done t=$(( $(date +%s) - T0 ))" | tee -a $OUT/${TAG}_summary.txt
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/p1 -o run -- python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-alt-math --no-roofline > /dev/null 2>&1
python scripts/rocpd_stats.py $OUT/p1/run_results.db 117 > $OUT/${TAG}_cfg2_kernel_stats.md
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/p2 -o run -- python bench.py --workload c2f --steps 6 --warmup 2 --no-cpu-baseline --no-alt-math --no-roofline > /dev/null 2>&1
python scripts/rocpd_stats.py $OUT/p2/run_results.db 12 > $OUT/${TAG}_c2f_kernel_stats.md
rm -rf $OUT/p1 $OUT/p2
echo "all done t=$(( $(date +%s) - T0 ))" | tee -a $OUT/${TAG}_summary.txt
tail -12 $OUT/${TAG}_tests.log
python - <<PY
import json,glob
for f in sorted(glob.glob("gpurun_out/${TAG}_bench*.json")):
    try:
        d=json.loads(open(f).read().strip().splitlines()[-1]); print(f, round(d["value"],1), round(d["ms_per_step"],4))
    except Exception as e: print(f, "ERR", e)
PY
