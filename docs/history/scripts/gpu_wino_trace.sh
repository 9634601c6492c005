#!/bin/bash
# s_memtime trace of wino_kernel (FG_WINO_TRACE=1) on the 3x3 layer shapes of both workloads + (optionally) the whole -m gpu suite
# usage: gpu_wino_trace.sh [tag] [full]
set -u
OUT=gpurun_out; TAG=${1:-wtr}; FULL=${2:-}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p $OUT
T0=$(date +%s)
rm -f $OUT/${TAG}_trace.txt
for shape in "128 16 16 64 128 3 0" "128 8 8 128 256 3 0" "128 4 4 256 512 3 0" "128 64 64 64 64 3 0" "128 32 32 64 128 3 0" "128 32 32 128 256 3 0"; do
  FG_WINO_TRACE=1 FG_WS_TRACE_FILE=$OUT/${TAG}_trace.txt timeout 120 python scripts/bench_one.py fwd 3 0 $shape > /dev/null 2>&1
done
python scripts/ws_trace_report.py $OUT/${TAG}_trace.txt > $OUT/${TAG}_trace_report.txt 2>&1
gzip -f $OUT/${TAG}_trace.txt
cat $OUT/${TAG}_trace_report.txt
echo "trace t=$(( $(date +%s) - T0 ))"
if [ -n "$FULL" ]; then
  timeout 1500 python -m pytest tests -m gpu -x -q --durations=5 > $OUT/${TAG}_tests.log 2>&1
  echo "tests rc=$? t=$(( $(date +%s) - T0 ))" | tee $OUT/${TAG}_summary.txt
  tail -12 $OUT/${TAG}_tests.log
fi
