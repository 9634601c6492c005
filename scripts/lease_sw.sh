set -u
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p gpurun_out
OUT=gpurun_out/r06_store_width.txt
scripts/ubench/store_width > $OUT 2>&1
for pmc in FETCH_SIZE WRITE_SIZE; do
  d=gpurun_out/sw_$pmc
  timeout 120 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $d -- scripts/ubench/store_width > /dev/null 2>&1
  echo "## $pmc (KB per launch)" >> $OUT
  python scripts/pmc_summary.py $d st >> $OUT 2>&1
  rm -rf $d
done
cat $OUT
# and the step with the 16-byte Adam
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_step_abi.py tests/test_gpu_fusion.py -m gpu -q -x 2>&1 | tail -3
for i in 1 2; do timeout 300 python bench.py --no-cpu-baseline --no-alt-math --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2', round(d['ms_per_step'],4), round(d['value'],1))"; done
bash scripts/gpu.sh kstats g6 > gpurun_out/g6_kstats.txt 2>&1; grep "adam\|pack_jobs\|finish" gpurun_out/g6_bench_kernel_stats.md | head
