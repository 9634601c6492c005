#!/bin/bash
# Winograd A/B: parity tests, then the contraction micro-benchmark and both workloads with the Winograd bits on / off (FG_WINO /
# FG_WINO_UP / FG_WINO_5X5 clear fg_set_fusion bits at context creation).  usage: gpu_wino.sh [tag] [full]   (full: the whole suite)
set -u
OUT=gpurun_out; TAG=${1:-wino}; FULL=${2:-}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p $OUT
T0=$(date +%s)
if [ -n "$FULL" ]; then
  timeout 1500 python -m pytest tests -m gpu -q --durations=5 > $OUT/${TAG}_tests.log 2>&1
else
  timeout 600 python -m pytest tests/test_gpu_wino.py tests/test_gpu_ops.py tests/test_gpu_net.py tests/test_gpu_c2f.py -m gpu -x -q > $OUT/${TAG}_tests.log 2>&1
fi
echo "tests rc=$? t=$(( $(date +%s) - T0 ))" | tee $OUT/${TAG}_summary.txt
grep -n "^FAILED\|^ERROR\|passed\|failed\|Error" $OUT/${TAG}_tests.log | tail -20
rm -f $OUT/${TAG}_conv.txt
for w in 1 0; do
  echo "== wino=$w cfg2 shapes" >> $OUT/${TAG}_conv.txt
  FG_WINO=$w FG_WINO_UP=$w FG_WINO_5X5=$w timeout 200 python scripts/bench_conv.py 20 2>&1 | grep -v wgrad | grep "^[gd][0-9]\|igemm\|wino" >> $OUT/${TAG}_conv.txt
  echo "== wino=$w c2f shapes" >> $OUT/${TAG}_conv.txt
  FG_WINO=$w FG_WINO_UP=$w FG_WINO_5X5=$w timeout 300 python scripts/bench_conv.py 10 c2f 2>&1 | grep -v wgrad | grep "^[GD][0-9]\|igemm\|wino" >> $OUT/${TAG}_conv.txt
done
cat $OUT/${TAG}_conv.txt
for i in 1 2; do for w in "1 1 1" "1 0 1" "0 0 0"; do set -- $w
  FG_WINO=$1 FG_WINO_UP=$2 FG_WINO_5X5=$3 timeout 200 python bench.py --workload cfg2 --no-cpu-baseline --no-alt-math --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2 wino/up/5x5=$1$2$3', d['ms_per_step'], d['value'])" | tee -a $OUT/${TAG}_summary.txt
done; done
for w in "1 1 1" "1 1 0" "0 0 0"; do set -- $w
  FG_WINO=$1 FG_WINO_UP=$2 FG_WINO_5X5=$3 timeout 300 python bench.py --workload c2f --steps 10 --warmup 3 --no-cpu-baseline --no-alt-math --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2f wino/up/5x5=$1$2$3', d['ms_per_step'], d['value'])" | tee -a $OUT/${TAG}_summary.txt
done
echo "done t=$(( $(date +%s) - T0 ))" | tee -a $OUT/${TAG}_summary.txt
