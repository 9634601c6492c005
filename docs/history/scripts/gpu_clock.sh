#!/bin/bash
set -u
OUT=gpurun_out; TAG=${1:-r3w}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"; mkdir -p $OUT
timeout 200 python scripts/clock_by_kernel.py cfg2 2>&1 | tee $OUT/${TAG}_clock_cfg2.txt
timeout 200 python scripts/clock_by_kernel.py c2f 2>&1 | tee $OUT/${TAG}_clock_c2f.txt
for v in probe noprobe probe noprobe; do
  flag=""; [ $v = noprobe ] && flag="--no-clock-probe"
  timeout 300 python bench.py --workload cfg2 --steps 20 --warmup 5 --no-cpu-baseline --no-alt-math --no-live-traffic $flag > $OUT/${TAG}_bench_$v.json 2>$OUT/${TAG}_bench.err
  python - <<P
import json
d=json.loads(open("$OUT/${TAG}_bench_$v.json").read().strip().splitlines()[-1])
print("$v cfg2", d["value"], d["ms_per_step"], d["roofline"]["frac"], d["roofline"].get("granted_clock_ghz"), d["roofline"].get("frac_at_granted_clock"))
P
done
