#!/bin/bash
# micro-bench of the contraction kernels on the hot-path layer shapes under one environment switch + parity subset with it on
# usage: gpu_conv.sh VAR
set -u
OUT=gpurun_out
VAR=${1:-FG_IGEMM_BK64}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
mkdir -p $OUT
rm -f $OUT/conv_variants.txt
env $VAR=1 timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_net.py -m gpu -x -q 2>&1 | tail -2 >> $OUT/conv_variants.txt
run() { L=$1; shift; echo "== $L" >> $OUT/conv_variants.txt; env "$@" timeout 120 python scripts/bench_conv.py 10 2>&1 | grep "^[gd][0-9]\|igemm\|wgrad" >> $OUT/conv_variants.txt; }
run "$VAR=1" $VAR=1
run "$VAR=0" $VAR=0
for i in 1 2; do
env $VAR=0 timeout 200 python bench.py --no-cpu-baseline --no-alt-math --no-roofline | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2 off', d['ms_per_step'])" >> $OUT/conv_variants.txt
env $VAR=1 timeout 200 python bench.py --no-cpu-baseline --no-alt-math --no-roofline | python -c "import sys,json; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('cfg2 on', d['ms_per_step'])" >> $OUT/conv_variants.txt
done
cat $OUT/conv_variants.txt
