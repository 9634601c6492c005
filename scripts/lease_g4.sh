set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
T0=$(date +%s)
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --no-alt-math --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2', round(d['ms_per_step'],4), round(d['value'],1))"; done
bash scripts/gpu.sh kstats g4 > gpurun_out/g4_kstats.txt 2>&1; head -12 gpurun_out/g4_bench_kernel_stats.md; grep -A40 "per launch shape" gpurun_out/g4_bench_kernel_stats.md
sqlite3 -version 2>/dev/null | head -1
echo "done t=$(( $(date +%s) - T0 ))"
