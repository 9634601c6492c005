#!/bin/bash
set -u
OUT=gpurun_out; TAG=${1:-r3ac}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p $OUT
for ns in 0 1 0 1; do
  FG_DEBUG_NOSTORE=$ns timeout 300 python bench.py --workload c2f --steps 8 --warmup 2 --no-cpu-baseline --no-alt-math --no-clock-probe > $OUT/${TAG}_ns.json 2>/dev/null
  python - <<P
import json
d=json.loads(open("$OUT/${TAG}_ns.json").read().strip().splitlines()[-1])
k=d.get("kernels",{})
print("nostore=$ns c2f %.2f ms" % d["ms_per_step"], " | ".join("%s %.3f ms %.1f TF" % (n.split("/")[0][-22:]+"/"+n.split("/")[1], v["ms_per_iter"], v["executed_tflops"]) for n,v in list(k.items())[:6]))
P
done
