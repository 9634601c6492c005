#!/bin/bash
set -u
OUT=gpurun_out; TAG=${1:-r3z}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_fusion.py tests/test_gpu_step_abi.py tests/test_gpu_net.py tests/test_gpu_c2f.py -x -q -m gpu 2>&1 | tail -25 | tee $OUT/${TAG}_tests.log
for env in "" "FG_DEFER_WFINISH=0" "FG_ADAM_PACK=0" "FG_DEFER_WFINISH=0 FG_ADAM_PACK=0" ""; do
  env $env timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-alt-math --no-live-traffic --no-clock-probe --c2f-steps 6 > $OUT/${TAG}_b.json 2>$OUT/${TAG}_b.err
  python - <<P
import json
d=json.loads(open("$OUT/${TAG}_b.json").read().strip().splitlines()[-1])
c=d.get("c2f",{})
print("[$env] cfg2 %.0f img/s %.4f ms | c2f %s img/s %s ms" % (d["value"], d["ms_per_step"], c.get("value"), c.get("ms_per_step")))
P
done 2>&1 | tee $OUT/${TAG}_ab.txt
