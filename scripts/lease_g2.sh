set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
T0=$(date +%s)
timeout 900 python -m pytest tests/test_gpu_t7_fixture.py tests/test_gpu_wino.py tests/test_gpu_ops.py tests/test_gpu_net.py tests/test_gpu_fusion.py tests/test_gpu_step_abi.py tests/test_gpu_baseline_sizes.py -m gpu -q -x > gpurun_out/g2_tests.log 2>&1; echo "tests rc=$? t=$(( $(date +%s) - T0 ))"; tail -15 gpurun_out/g2_tests.log
for i in 1 2 3; do timeout 300 python bench.py --no-cpu-baseline --no-alt-math --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2', round(d['ms_per_step'],4), round(d['value'],1))"; done
bash scripts/gpu.sh kstats g2 > gpurun_out/g2_kstats.txt 2>&1; head -50 gpurun_out/g2_bench_kernel_stats.md
echo "done t=$(( $(date +%s) - T0 ))"
