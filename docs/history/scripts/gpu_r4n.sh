#!/bin/bash
# round 4, call n: the wave-specialised weight gradient for d13 (M = 2 048 pixels): FG_WGRAD_WS_MINM=2048 against the default 4096
set -u
OUT=gpurun_out; TAG=${1:-r04n}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p $OUT
for rep in 1 2; do
 for mm in 2048 4096; do
    FG_WGRAD_WS_MINM=$mm timeout 300 python bench.py --workload cfg2 --steps 50 --warmup 10 --no-cpu-baseline --no-alt-math --no-live-traffic --no-clock-probe 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
w=' '.join('%s=%.1fus/%.0fTF' % (n, 1e3*v['ms_per_iter']/v['calls_per_iter'], v['executed_tflops']) for n,v in k.items() if 'conv_wgrad' in n)
print('minm=$mm cfg2 %.1f img/s %.4f ms | %s' % (d['value'], d['ms_per_step'], w))"
 done
done 2>&1 | tee $OUT/${TAG}_bench.txt
FG_WGRAD_WS_MINM=2048 timeout 600 python -m pytest "tests/test_gpu_baseline_sizes.py::test_cfg2_full_step_at_batch_128" tests/test_gpu_net.py -m gpu -q 2>&1 | grep -E "passed|failed" | tee -a $OUT/${TAG}_bench.txt
