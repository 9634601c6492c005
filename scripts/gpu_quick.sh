#!/bin/bash
set -u
OUT=gpurun_out; TAG=${1:-r3am}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p $OUT
timeout 600 python -m pytest tests/test_gpu_fusion.py tests/test_gpu_net.py tests/test_gpu_c2f.py -x -q -m gpu 2>&1 | tail -8
for v in 1 2; do
  timeout 300 python bench.py --steps 30 --warmup 5 --no-cpu-baseline --no-alt-math --no-clock-probe --no-roofline --no-live-traffic --c2f-steps 10 > $OUT/${TAG}_b.json 2>/dev/null
  python - <<P
import json
d=json.loads(open("$OUT/${TAG}_b.json").read().strip().splitlines()[-1])
c=d.get("c2f",{})
print("cfg2 %.0f img/s %.4f ms | c2f %.1f img/s %.3f ms" % (d["value"], d["ms_per_step"], c.get("value",0), c.get("ms_per_step",0)))
P
done
