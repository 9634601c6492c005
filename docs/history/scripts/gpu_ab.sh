#!/bin/bash
# A/B of one environment switch: a parity subset with the default setting, the s_memtime trace of the Winograd kernel, then both
# workloads' step time with VAR=1 / VAR=0, interleaved.     usage: gpu_ab.sh VAR [tag] [pytest targets...]
set -u
OUT=gpurun_out; VAR=${1:-FG_WINO_SHARE}; TAG=${2:-ab}; shift 2 || true
TESTS=${@:-tests/test_gpu_wino.py tests/test_gpu_ops.py tests/test_gpu_net.py}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p $OUT
T0=$(date +%s)
timeout 900 python -m pytest $TESTS -m gpu -q > $OUT/${TAG}_tests.log 2>&1
echo "tests rc=$? t=$(( $(date +%s) - T0 ))" | tee $OUT/${TAG}_summary.txt
grep -n "^FAILED\|^ERROR\|passed\|failed\|Error" $OUT/${TAG}_tests.log | tail -12
rm -f $OUT/${TAG}_trace.txt
for shape in "128 16 16 64 128 3 0" "128 32 32 128 256 3 0" "128 16 16 256 128 5 1" "128 64 64 128 256 5 0"; do
  FG_WINO_TRACE=1 FG_WS_TRACE_FILE=$OUT/${TAG}_trace.txt timeout 120 python scripts/bench_one.py fwd 2 0 $shape > /dev/null 2>&1
done
python scripts/ws_trace_report.py $OUT/${TAG}_trace.txt 2>&1 | grep "launch\|per block\|MFMA-pipe\|wall" | cut -c1-330 > $OUT/${TAG}_trace_report.txt
gzip -f $OUT/${TAG}_trace.txt
awk 'NR%8<5' $OUT/${TAG}_trace_report.txt | head -40
for i in 1 2; do for v in 1 0; do
  env $VAR=$v timeout 200 python bench.py --workload cfg2 --no-cpu-baseline --no-alt-math --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2 $VAR=$v', round(d['ms_per_step'],4), round(d['value'],1))" | tee -a $OUT/${TAG}_summary.txt
done; done
for v in 1 0; do
  env $VAR=$v timeout 300 python bench.py --workload c2f --steps 10 --warmup 3 --no-cpu-baseline --no-alt-math --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2f $VAR=$v', round(d['ms_per_step'],3), round(d['value'],1))" | tee -a $OUT/${TAG}_summary.txt
done
echo "done t=$(( $(date +%s) - T0 ))" | tee -a $OUT/${TAG}_summary.txt
