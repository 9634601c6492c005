set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
T0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_wino.py tests/test_gpu_ops.py tests/test_gpu_net.py tests/test_gpu_fusion.py tests/test_gpu_step_abi.py tests/test_gpu_modules.py tests/test_gpu_c2f.py tests/test_gpu_conv_upsample.py -m gpu -q -x > gpurun_out/g5_tests.log 2>&1; echo "tests rc=$? t=$(( $(date +%s) - T0 ))"; tail -5 gpurun_out/g5_tests.log
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --no-alt-math --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2', round(d['ms_per_step'],4), round(d['value'],1))"; done
timeout 300 python bench.py --workload c2f --steps 10 --warmup 3 --no-cpu-baseline --no-alt-math --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2f', round(d['ms_per_step'],4), round(d['value'],1))"
bash scripts/gpu.sh kstats g5 > gpurun_out/g5_kstats.txt 2>&1; head -12 gpurun_out/g5_bench_kernel_stats.md; grep -A22 "per launch shape" gpurun_out/g5_bench_kernel_stats.md | grep "finish\|pack"
echo "done t=$(( $(date +%s) - T0 ))"
