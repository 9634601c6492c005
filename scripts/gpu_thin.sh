#!/bin/bash
set -u
OUT=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $OUT
run() {  # label, env...
  L=$1; shift
  env "$@" timeout 120 rocprofv3 --kernel-trace --stats -d $OUT/pt -o run -- python scripts/bench_thin.py 20 > /dev/null 2>&1
  echo "== $L" >> $OUT/thin_variants.txt
  python scripts/rocpd_stats.py $OUT/pt/run_results.db 20 | grep "thin_out" >> $OUT/thin_variants.txt
  rm -rf $OUT/pt
}
rm -f $OUT/thin_variants.txt
run "R6 lds90 (1 block/CU)" FG_THIN_SLAB_R=6 FG_THIN_SLAB_LDS=90
run "R6 lds60 (2 blocks/CU)" FG_THIN_SLAB_R=6 FG_THIN_SLAB_LDS=60
run "R16" FG_THIN_SLAB_R=16
run "R14 lds90" FG_THIN_SLAB_R=14 FG_THIN_SLAB_LDS=90
run "R8" FG_THIN_SLAB_R=8
run "R32" FG_THIN_SLAB_R=32
cat $OUT/thin_variants.txt
