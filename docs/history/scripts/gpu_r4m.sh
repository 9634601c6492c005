#!/bin/bash
# round 4, call m: fragment prefetch fenced in the 64 x 64-tile contraction kernel -- parity subset, both bench lines with per-kernel rates
set -u
OUT=gpurun_out; TAG=${1:-r04m}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_net.py tests/test_golden.py tests/test_gpu_c2f.py tests/test_gpu_fullsize.py -m gpu -q > $OUT/${TAG}_tests.log 2>&1
echo "tests rc=$?" | tee $OUT/${TAG}_summary.txt; tail -3 $OUT/${TAG}_tests.log
for rep in 1 2; do
  for wl in cfg2 c2f; do
    timeout 300 python bench.py --workload $wl --steps $([ $wl = c2f ] && echo 10 || echo 50) --warmup $([ $wl = c2f ] && echo 3 || echo 10) --no-cpu-baseline --no-alt-math --no-live-traffic --no-clock-probe > $OUT/${TAG}_b.json 2>/dev/null
    python - $wl $TAG <<'P'
import json,sys
d=json.loads(open("gpurun_out/%s_b.json" % sys.argv[2]).read().strip().splitlines()[-1])
k=d["kernels"]
s=" ".join("%s=%.1fus/%.0fTF" % (n.replace("igemm_kernel<64,64,64>/","ig64/").replace("igemm_kernel<128,128,32>/","ig128/"), 1e3*v["ms_per_iter"]/v["calls_per_iter"], v["executed_tflops"]) for n,v in k.items() if n.startswith("igemm_kernel") or n.startswith("igemm_act_kernel"))
print("%s %.1f img/s %.4f ms exec %.4f | %s" % (sys.argv[1], d["value"], d["ms_per_step"], d["step_roofline"].get("executed_frac",0), s))
P
  done
done 2>&1 | tee $OUT/${TAG}_bench.txt
