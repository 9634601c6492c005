#!/bin/bash
set -u
OUT=gpurun_out; TAG=${1:-r3ag}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p $OUT
timeout 900 python -m pytest tests/test_gpu_c2f.py -x -q -m gpu 2>&1 | tail -6 | tee $OUT/${TAG}_tests.log
for v in 1 0 1 0; do
  FG_THIN_WGRAD_PADDED=$v timeout 300 python bench.py --workload c2f --steps 8 --warmup 2 --no-cpu-baseline --no-alt-math --no-clock-probe --no-roofline > $OUT/${TAG}_b.json 2>/dev/null
  python - <<P
import json
d=json.loads(open("$OUT/${TAG}_b.json").read().strip().splitlines()[-1])
print("padded=$v c2f %.1f img/s %.3f ms" % (d["value"], d["ms_per_step"]))
P
done
