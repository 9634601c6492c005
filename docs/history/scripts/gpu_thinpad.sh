#!/bin/bash
set -u
OUT=gpurun_out; TAG=${1:-r3al}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p $OUT
for v in 1 0 1 0; do
  FG_THIN_PADDED=$v timeout 300 python bench.py --workload c2f --steps 10 --warmup 3 --no-cpu-baseline --no-alt-math --no-clock-probe --no-roofline > $OUT/${TAG}_b.json 2>/dev/null
  python - <<P
import json
d=json.loads(open("$OUT/${TAG}_b.json").read().strip().splitlines()[-1])
print("padded=$v c2f %.1f img/s %.3f ms" % (d["value"], d["ms_per_step"]))
P
done
