#!/bin/bash
# HBM traffic (FETCH_SIZE / WRITE_SIZE, separate passes) and durations of the thin 3x3 kernels of the cfg2 path
set -u
OUT=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $OUT
rm -f $OUT/thin_pmc.txt
for pmc in FETCH_SIZE WRITE_SIZE; do
  d=$OUT/pmc_thin_$pmc
  timeout 120 rocprofv3 --pmc $pmc --kernel-trace --output-format csv -d $d -- python scripts/bench_thin.py 4 > /dev/null 2>&1
  echo "## $pmc" >> $OUT/thin_pmc.txt
  python scripts/pmc_summary.py $d thin_ >> $OUT/thin_pmc.txt 2>&1
  rm -rf $d
done
timeout 120 rocprofv3 --kernel-trace --stats -d $OUT/pt -o run -- python scripts/bench_thin.py 20 > /dev/null 2>&1
echo "## durations (20 repetitions)" >> $OUT/thin_pmc.txt
python scripts/rocpd_stats.py $OUT/pt/run_results.db 20 | grep "thin_" >> $OUT/thin_pmc.txt
rm -rf $OUT/pt
cat $OUT/thin_pmc.txt
