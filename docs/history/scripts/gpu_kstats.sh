#!/bin/bash
# rocprofv3 --kernel-trace --stats tables of both workloads.     usage: gpu_kstats.sh [tag]
set -u
TAG=${1:-ks}; OUT=gpurun_out
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p $OUT
rocprofv3 --kernel-trace --stats -d $OUT/p1 -o run -- python bench.py --workload cfg2 --steps 50 --warmup 10 --no-cpu-baseline --no-alt-math --no-roofline > $OUT/${TAG}_bench_under_rocprof.json 2>/dev/null
python scripts/rocpd_stats.py $OUT/p1/run_results.db auto > $OUT/${TAG}_bench_kernel_stats.md
rocprofv3 --kernel-trace --stats -d $OUT/p2 -o run -- python bench.py --workload c2f --steps 6 --warmup 2 --no-cpu-baseline --no-alt-math --no-roofline > /dev/null 2>&1
python scripts/rocpd_stats.py $OUT/p2/run_results.db auto > $OUT/${TAG}_c2f_kernel_stats.md
rm -rf $OUT/p1 $OUT/p2
head -45 $OUT/${TAG}_bench_kernel_stats.md; head -40 $OUT/${TAG}_c2f_kernel_stats.md
