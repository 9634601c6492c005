#!/bin/bash
set -u
OUT=gpurun_out; TAG=${1:-r3u}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p $OUT
rm -f $OUT/${TAG}_trace.txt
# shapes: G2-like K=576 (BN=64), D4-like K=1152, the dominant cfg2 launch K=3200
for shape in "128 64 64 64 64 3 0" "128 32 32 128 256 3 0" "128 64 64 128 256 5 0"; do
  FG_WS_TRACE=1 FG_WS_TRACE_FILE=$OUT/${TAG}_trace.txt timeout 120 python scripts/bench_one.py fwd 3 0 $shape > /dev/null 2>&1
done
python scripts/ws_trace_gaps.py $OUT/${TAG}_trace.txt | tee $OUT/${TAG}_trace_gaps.txt
python scripts/ws_trace_report.py $OUT/${TAG}_trace.txt > $OUT/${TAG}_trace_report.txt 2>&1
gzip -f $OUT/${TAG}_trace.txt
