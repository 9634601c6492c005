#!/bin/bash
set -u
OUT=gpurun_out; TAG=${1:-r3at}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p $OUT
rm -rf $OUT/pp
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/pp -o run -- python bench.py --workload c2f --steps 10 --warmup 2 --no-cpu-baseline --no-alt-math --no-roofline > $OUT/${TAG}_prof_bench.json 2>/dev/null
python scripts/rocpd_stats.py $OUT/pp/run_results.db auto > $OUT/${TAG}_c2f_kernel_stats.md
grep -E "thin_|pack_jobs|adam|wgrad_finish|iterations|total kernel" $OUT/${TAG}_c2f_kernel_stats.md
rm -rf $OUT/pp
