#!/bin/bash
# Winograd weight gradient A/B (FG_WINO_WGRAD clears fg_set_fusion bit 256 at context creation): parity tests, the per-layer
# micro-benchmark's wgrad rows, both workloads' step time on / off.     usage: gpu_ww.sh [tag] [pytest targets...]
set -u
OUT=gpurun_out; TAG=${1:-ww}; shift 1 || true
TESTS=${@:-tests/test_gpu_wino.py tests/test_gpu_ops.py tests/test_gpu_net.py tests/test_gpu_fusion.py tests/test_gpu_c2f.py}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p $OUT
T0=$(date +%s)
timeout 900 python -m pytest $TESTS -m gpu -q > $OUT/${TAG}_tests.log 2>&1
echo "tests rc=$? t=$(( $(date +%s) - T0 ))" | tee $OUT/${TAG}_summary.txt
grep -n "^FAILED\|^ERROR\|passed\|failed\|Error" $OUT/${TAG}_tests.log | tail -25
rm -f $OUT/${TAG}_conv.txt
for w in 1 0; do
  echo "== FG_WINO_WGRAD=$w cfg2 shapes" >> $OUT/${TAG}_conv.txt
  FG_WINO_WGRAD=$w timeout 200 python scripts/bench_conv.py 20 2>&1 | grep "wgrad" >> $OUT/${TAG}_conv.txt
  echo "== FG_WINO_WGRAD=$w c2f shapes" >> $OUT/${TAG}_conv.txt
  FG_WINO_WGRAD=$w timeout 300 python scripts/bench_conv.py 10 c2f 2>&1 | grep "wgrad" >> $OUT/${TAG}_conv.txt
done
cat $OUT/${TAG}_conv.txt
for i in 1 2; do for v in 1 0; do
  FG_WINO_WGRAD=$v timeout 200 python bench.py --workload cfg2 --no-cpu-baseline --no-alt-math --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2 FG_WINO_WGRAD=$v', round(d['ms_per_step'],4), round(d['value'],1))" | tee -a $OUT/${TAG}_summary.txt
done; done
for v in 1 0; do
  FG_WINO_WGRAD=$v timeout 300 python bench.py --workload c2f --steps 10 --warmup 3 --no-cpu-baseline --no-alt-math --no-roofline 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('c2f FG_WINO_WGRAD=$v', round(d['ms_per_step'],3), round(d['value'],1))" | tee -a $OUT/${TAG}_summary.txt
done
echo "done t=$(( $(date +%s) - T0 ))" | tee -a $OUT/${TAG}_summary.txt
