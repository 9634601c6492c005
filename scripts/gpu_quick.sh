#!/bin/bash
set -u
OUT=gpurun_out; TAG=${1:-r3ap}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_ops.py tests/test_gpu_net.py tests/test_gpu_c2f.py tests/test_gpu_modules.py tests/test_gpu_fusion.py -x -q -m gpu 2>&1 | tail -6 | tee $OUT/${TAG}_tests.log
for env in "" "FG_WGRAD_WS64=0" "" "FG_WGRAD_WS64=0"; do
  env $env timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-alt-math --no-clock-probe --no-live-traffic --c2f-steps 10 > $OUT/${TAG}_b.json 2>/dev/null
  python - <<P
import json
d=json.loads(open("$OUT/${TAG}_b.json").read().strip().splitlines()[-1])
c=d.get("c2f",{})
print("[$env] cfg2 %.0f img/s %.4f ms | c2f %.1f img/s %.3f ms" % (d["value"], d["ms_per_step"], c.get("value",0), c.get("ms_per_step",0)))
for n,v in c.get("kernels",{}).items():
    if "wgrad" in n: print("   ", n, "%.3f ms %.1f TF" % (v["ms_per_iter"], v["executed_tflops"]))
for n,v in d.get("kernels",{}).items():
    if "wgrad" in n and "conv" in n: print("   cfg2", n, "%.3f ms %.1f TF" % (v["ms_per_iter"], v["executed_tflops"]))
P
done 2>&1 | tee $OUT/${TAG}_ab.txt
