#!/bin/bash
set -u
OUT=gpurun_out; TAG=${1:-r3au}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_ops.py tests/test_gpu_c2f.py -x -q -m gpu 2>&1 | tail -3 | tee $OUT/${TAG}_tests.log
for env in "FG_WGRAD_ROUNDS=1" "FG_WGRAD_ROUNDS=0" "FG_WGRAD_ROUNDS=1" "FG_WGRAD_ROUNDS=0"; do
  env $env timeout 300 python bench.py --workload c2f --steps 10 --warmup 3 --no-cpu-baseline --no-alt-math --no-clock-probe > $OUT/${TAG}_b.json 2>/dev/null
  python - <<P
import json
d=json.loads(open("$OUT/${TAG}_b.json").read().strip().splitlines()[-1])
print("[$env] c2f %.1f img/s %.3f ms" % (d["value"], d["ms_per_step"]), " | ".join("%s %.3f ms %.1f TF" % (n.split("/")[0][-20:], v["ms_per_iter"], v["executed_tflops"]) for n,v in d["kernels"].items() if "wgrad_ws" in n))
P
done 2>&1 | tee $OUT/${TAG}_ab.txt
