#!/bin/bash
# One bounded GPU call: parity of the PReLU-in-epilogue paths (both ways), c2f / cfg2 bench lines, kernel traces.
set -u
OUT=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $OUT
T0=$(date +%s)
timeout 600 python -m pytest tests/test_gpu_c2f.py tests/test_gpu_baseline_sizes.py tests/test_gpu_net.py tests/test_gpu_modules.py -m gpu -x -q --durations=12 > $OUT/a_tests_fused.log 2>&1
echo "tests fused rc=$? t=$(( $(date +%s) - T0 ))" | tee -a $OUT/a_summary.txt
FG_FUSE_PRELU=0 timeout 300 python -m pytest tests/test_gpu_c2f.py -m gpu -x -q > $OUT/a_tests_unfused.log 2>&1
echo "tests unfused rc=$? t=$(( $(date +%s) - T0 ))" | tee -a $OUT/a_summary.txt
timeout 200 python bench.py --workload c2f --steps 10 --warmup 3 --no-cpu-baseline --no-alt-math > $OUT/a_bench_c2f_fused.json 2> $OUT/a_bench_c2f_fused.err
FG_FUSE_PRELU=0 timeout 200 python bench.py --workload c2f --steps 10 --warmup 3 --no-cpu-baseline --no-alt-math > $OUT/a_bench_c2f_unfused.json 2> $OUT/a_bench_c2f_unfused.err
timeout 200 python bench.py --no-cpu-baseline --no-alt-math > $OUT/a_bench_cfg2.json 2> $OUT/a_bench_cfg2.err
echo "benches done t=$(( $(date +%s) - T0 ))" | tee -a $OUT/a_summary.txt
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/p1 -o run -- python bench.py --steps 50 --warmup 10 --no-cpu-baseline --no-alt-math --no-roofline > /dev/null 2>&1
python scripts/rocpd_stats.py $OUT/p1/run_results.db 64 > $OUT/a_cfg2_kernel_stats.md
timeout 200 rocprofv3 --kernel-trace --stats -d $OUT/p2 -o run -- python bench.py --workload c2f --steps 6 --warmup 2 --no-cpu-baseline --no-alt-math --no-roofline > /dev/null 2>&1
python scripts/rocpd_stats.py $OUT/p2/run_results.db 12 > $OUT/a_c2f_kernel_stats.md
rm -rf $OUT/p1 $OUT/p2
echo "all done t=$(( $(date +%s) - T0 ))" | tee -a $OUT/a_summary.txt
tail -4 $OUT/a_tests_fused.log; tail -3 $OUT/a_tests_unfused.log
python - <<'PY'
import json
for f in ("a_bench_c2f_fused","a_bench_c2f_unfused","a_bench_cfg2"):
    try:
        d=json.loads(open("gpurun_out/%s.json"%f).read().strip().splitlines()[-1]); print(f, d["value"], d["ms_per_step"])
    except Exception as e: print(f, "ERR", e)
PY
