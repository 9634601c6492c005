#!/bin/bash
# round 4, call p: Linear(65536 -> 512) forward on 128 x 128 tiles split 64-fold (FG_LINEAR_FWD128=0 switches back): parity + A/B
set -u
OUT=gpurun_out; TAG=${1:-r04p}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_c2f.py tests/test_golden.py "tests/test_gpu_baseline_sizes.py::test_c2f_S64_full_steps" "tests/test_gpu_baseline_sizes.py::test_c2f_S64_forward_backward" -m gpu -q 2>&1 | grep -E "passed|failed" | tee $OUT/${TAG}_bench.txt
for rep in 1 2; do
 for v in 1 0; do
    FG_LINEAR_FWD128=$v timeout 300 python bench.py --workload c2f --steps 10 --warmup 3 --no-cpu-baseline --no-alt-math --no-live-traffic --no-clock-probe 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); k=d['kernels']
w=' '.join('%s=%.1fus/%.0fTF' % (n, 1e3*v['ms_per_iter']/v['calls_per_iter'], v['executed_tflops']) for n,v in k.items() if 'linear' in n)
print('fwd128=$v c2f %.1f img/s %.4f ms | %s' % (d['value'], d['ms_per_step'], w))"
 done
done 2>&1 | tee -a $OUT/${TAG}_bench.txt
