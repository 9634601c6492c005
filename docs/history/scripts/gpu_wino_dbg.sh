#!/bin/bash
# what each class of side work costs the K loop of wino_kernel: s_memtime traces of the DBG variants (wino.hip) on one shape,
# then (optionally) the whole -m gpu suite without -x
# usage: gpu_wino_dbg.sh [tag] [full]
set -u
OUT=gpurun_out; TAG=${1:-wdbg}; FULL=${2:-}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p $OUT
T0=$(date +%s)
rm -f $OUT/${TAG}_trace.txt
for dbg in 0 16 1 2 4 6 7 15; do
  FG_WINO_DBG=$dbg FG_WINO_TRACE=1 FG_WS_TRACE_FILE=$OUT/${TAG}_trace.txt timeout 120 python scripts/bench_one.py fwd 2 0 128 32 32 128 256 3 0 > /dev/null 2>&1
done
python scripts/ws_trace_report.py $OUT/${TAG}_trace.txt 2>&1 | grep "launch\|per block" | cut -c1-330 > $OUT/${TAG}_trace_report.txt
gzip -f $OUT/${TAG}_trace.txt
cat $OUT/${TAG}_trace_report.txt
echo "trace t=$(( $(date +%s) - T0 ))"
if [ -n "$FULL" ]; then
  timeout 1500 python -m pytest tests -m gpu -q --durations=5 > $OUT/${TAG}_tests.log 2>&1
  echo "tests rc=$? t=$(( $(date +%s) - T0 ))" | tee $OUT/${TAG}_summary.txt
  grep -n "^FAILED\|^ERROR\|passed\|failed" $OUT/${TAG}_tests.log | tail -30
fi
