#!/bin/bash
# round 4, call d: (1) what thin_in_mfma_kernel<3,0> is made of (stores / gathers dropped), (2) dgrad || wgrad co-residency bound,
# (3) tests that changed
set -u
OUT=gpurun_out; TAG=${1:-r04d}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_step_abi.py tests/test_gpu_syncbn.py tests/test_gpu_train_epoch.py tests/test_gpu_fusion.py tests/test_gpu_ops.py tests/test_gpu_net.py -m gpu -q > $OUT/${TAG}_tests.log 2>&1
echo "tests rc=$?" | tee $OUT/${TAG}_summary.txt; tail -3 $OUT/${TAG}_tests.log
for env in "FG_X=0" "FG_THIN_DBG=1" "FG_THIN_DBG=2" "FG_THIN_DBG=3" "FG_THIN_TPW=1" "FG_THIN_TPW=2" "FG_THIN_TPW=8" "FG_THIN_TPW=1 FG_THIN_DBG=1"; do
  rm -rf $OUT/pt
  env $env rocprofv3 --kernel-trace --stats -d $OUT/pt -o run -- python scripts/bench_thin.py 40 > /dev/null 2>&1
  echo "== $env"; python scripts/rocpd_stats.py $OUT/pt/run_results.db | grep -E "thin_in_mfma_kernel|thin_out_slab"
done 2>&1 | tee $OUT/${TAG}_thin_in_parts.txt
rm -rf $OUT/pt
python scripts/bench_hfuse.py 200 2>/dev/null | tee $OUT/${TAG}_hfuse.txt
