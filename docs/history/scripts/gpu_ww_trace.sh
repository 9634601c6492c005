#!/bin/bash
# s_memtime traces of wino_wgrad_kernel (FG_WINO_WGRAD_TRACE=1: per chunk, =2: every 8 MFMA slots) on G's two up-convolutions and two
# coarse-to-fine shapes.     usage: gpu_ww_trace.sh [tag]
set -u
OUT=gpurun_out; TAG=${1:-wwtr}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p $OUT
rm -f $OUT/${TAG}_t1.txt $OUT/${TAG}_t2.txt
for shape in "128 16 16 256 128 5 1" "128 8 8 128 256 5 1" "128 64 64 64 128 5 0" "128 32 32 128 256 3 0"; do
  FG_WINO_WGRAD_TRACE=1 FG_WS_TRACE_FILE=$OUT/${TAG}_t1.txt timeout 120 python scripts/bench_one.py wgrad 2 0 $shape > /dev/null 2>&1
  FG_WINO_WGRAD_TRACE=2 FG_WS_TRACE_FILE=$OUT/${TAG}_t2.txt timeout 120 python scripts/bench_one.py wgrad 2 0 $shape > /dev/null 2>&1
done
python scripts/ws_trace_report.py $OUT/${TAG}_t1.txt 2>&1 | grep "launch\|per block\|MFMA-pipe\|wall" | cut -c1-330 | tee $OUT/${TAG}_t1_report.txt
python scripts/wino_trace2_report.py $OUT/${TAG}_t2.txt 2>&1 | tee $OUT/${TAG}_t2_report.txt | head -60
gzip -f $OUT/${TAG}_t1.txt $OUT/${TAG}_t2.txt
